// k_hnsw_build.hip — HNSW construction on the GPU: hnsw.go:713-984 (insert / insertNode), :986-1106
// (selectNeighbors*, applyHeuristic, fillUpNeighbors), :455-555 (addConnection*), :885-900
// (updateEntryPoint), with the ids and levels of ApplyInsert / ApplyBatchInsert (:629-684: ids are the
// row numbers, level = layerForApplyInsert(id) :2103-2116).
//
// Nodes are inserted in id order in batches; a batch is what ApplyBatchInsert's goroutines are to each
// other: every node of the batch searches the graph as it stood when the batch began, then the batch's
// links are applied in id order.  batch = clamp(inserted / growth_div, 1, max_batch); max_batch = 1 is the
// reference's sequential Insert loop.  Per batch:
//   1. build_search_kernel   one wavefront per new node: greedy descent above its level, then
//                            searchLayerUnfiltered(ef) on every level it owns (vg_hnsw_layer.hpp); the
//                            results heap is popped into a best-first candidate list per (node, level).
//   2. build_select_kernel   one workgroup per (node, level): selectNeighborsHeuristic over the list; the
//                            node's own row is written, and one back-link record per chosen neighbour.
//   3. build_count / offsets / fill kernels group the back-link records by target row;
//      build_link_kernel     one workgroup (4 waves) per target row applies its records in id order: append while
//                            the row has room (addConnectionSimple), else addConnectionPrune; the four waves share
//                            the 64 pair distances of a link, wave 0 keeps the row.
//
// What makes addConnectionPrune affordable: a row keeps, next to the ids and the cached distances the
// reference keeps (node.go Neighbor{ID, Dist}), a 64 x 64 BIT matrix: bit j of bits[i] = "d(c_i, c_j) <
// d(s, c_i)", the only thing applyHeuristic ever asks about a pair of candidates.  The reference recomputes
// ~2000 pair distances of 768 floats on every back link of a full row (64 back links per insert); here a
// back link costs the 64 distances between the new node and the row's members, the heuristic itself is a
// replay over bit masks.  Every distance is the reference's kernel in its summation order (vg_exact.hpp) and
// every candidate order is the order the reference's 4-ary heap would pop (rank by distance; on equal
// distances the heap is replayed in LDS), so the graph equals what a letter-by-letter CPU run of hnsw.go builds
// (the test oracle) — bit for bit, ties included.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <vector>

#include "vg_device.hpp"
#include "vg_exact.hpp"
#include "vg_heap.hpp"
#include "vg_hnsw_layer.hpp"
#include "vg_internal.hpp"

namespace vg {

constexpr int kBuildMaxEf = 1024;   // candidates per (node, level); heaps of the insert search live in LDS
constexpr int kSelThreads = 256;

// The graph under construction.  Rows: layer 0 row of node i = i (m0 slots); the row of node i on level
// l >= 1 = n + level_off[l-1] + slots[(l-1)*n + i] (m slots).  Slot arrays are laid out as one array with
// the n*m0 layer-0 slots first: that prefix IS the l0 table of vg_index_set_hnsw_graph, the rest its adj.
struct BuildGraph {
    const float *base;
    int64_t n;
    int dim, metric, m0, m;
    uint32_t *ids;    // 0xFFFFFFFF beyond cnt
    float *dist;      // d(row's node, member) as the insert search computed it
    uint64_t *bits;   // bit j of bits[slot i]: d(member i, member j) < dist[i]
    int32_t *cnt;     // per row
    uint8_t *good;    // per row: full, and every member was CHOSEN by the heuristic (none filled up): see build_link_kernel
    const uint32_t *slots;
    const int64_t *level_off;
};

__device__ __forceinline__ int64_t bg_row(const BuildGraph &g, uint32_t node, int level)
{
    if (level == 0) return node;
    return g.n + g.level_off[level - 1] + g.slots[static_cast<int64_t>(level - 1) * g.n + node];
}
__device__ __forceinline__ int64_t bg_off(const BuildGraph &g, int64_t row)
{
    return row < g.n ? row * g.m0 : g.n * g.m0 + (row - g.n) * g.m;
}
__device__ __forceinline__ int bg_deg(const BuildGraph &g, int64_t row) { return row < g.n ? g.m0 : g.m; }

// h.distanceFunc (newDistanceFunc hnsw.go:2218-2238) of two rows, all lanes of the 16-lane group
__device__ __forceinline__ float bg_pair(const BuildGraph &g, uint32_t a, uint32_t b, Sub16 sub)
{
    return hnsw_node_dist(g.base, g.dim, g.metric, g.base + static_cast<int64_t>(b) * g.dim, a, sub);
}

// ---- 1. insert search -----------------------------------------------------------------------------
// pair_base[t] = index of node t's first (node, level) pair counted from node 0 (levels min(level,top)..0
// → pair index + level); the batch's lists are stored relative to pair_base[t0].
// UK: the metric is not Dot, distances are >= +0 and the heaps compare bit patterns (heap_sift_down_uk, vg_heap.hpp)
template <bool UK>
__global__ __launch_bounds__(64) void build_search_kernel(BuildGraph g, int64_t t0, uint32_t entry, int cur_top,
                                                          const int32_t *__restrict__ levels,
                                                          const int64_t *__restrict__ pair_base, int ef,
                                                          uint32_t *__restrict__ visited_ws, int64_t vis_words,
                                                          uint32_t *__restrict__ cand_ids,
                                                          float *__restrict__ cand_d, int32_t *__restrict__ cand_n)
{
    extern __shared__ __attribute__((aligned(8))) unsigned char smem[];
    float *nb_pair = reinterpret_cast<float *>(smem);
    float *nb_bnd = nb_pair + 64;
    HItem *cand = reinterpret_cast<HItem *>(nb_bnd + 64);
    HItem *res = cand + 2 * ef;
    const int lane = threadIdx.x;
    const int64_t t = t0 + blockIdx.x;
    F32Scorer sc;
    sc.base = g.base;
    sc.qv = g.base + t * g.dim;
    sc.dim = g.dim;
    sc.metric = g.metric;
    sc.sub = Sub16::make(lane);
    uint32_t *vis = visited_ws + static_cast<int64_t>(blockIdx.x) * vis_words;
    const int lt = levels[t];
    const int64_t pair0 = pair_base[t] - pair_base[t0];

    uint32_t cur = entry;
    float cur_d = sc.one(cur);
    for (int level = cur_top; level > lt; level--) {  // hnsw.go:918-934
        auto row_of = [&](uint32_t node) -> const uint32_t * { return g.ids + bg_off(g, bg_row(g, node, level)); };
        greedy_layer(sc, lane, row_of, g.m, nb_pair, nb_bnd, cur, cur_d);
    }
    const int first = lt < cur_top ? lt : cur_top;
    LayerStats st;
    for (int level = first; level >= 0; level--) {  // hnsw.go:940-957
        if (level != first) {  // initializeSearch: Visited.Reset()
            for (int64_t w = lane; w < vis_words; w += 64) vis[w] = 0;
            __threadfence();
            __syncthreads();
        }
        const int deg = level == 0 ? g.m0 : g.m;
        auto row_of = [&](uint32_t node) -> const uint32_t * { return g.ids + bg_off(g, bg_row(g, node, level)); };
        int res_len = 0;
        search_layer<UK>(sc, g.metric == kMetricL2, lane, row_of, deg, cur, cur_d, ef, cand, res, nb_pair, nb_bnd, vis,
                         res_len, st);
        // candidates.MinItem() (queue.go:46-57): the first minimum in heap-array order
        uint64_t best = kKeyMax;
        for (int i = lane; i < res_len; i += 64) {
            const uint64_t key = (static_cast<uint64_t>(f32_ordered(res[i].dist)) << 32) | static_cast<uint32_t>(i);
            best = key < best ? key : best;
        }
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) {
            const uint32_t lo = __shfl_xor(static_cast<uint32_t>(best), s);
            const uint32_t hi = __shfl_xor(static_cast<uint32_t>(best >> 32), s);
            const uint64_t o = (static_cast<uint64_t>(hi) << 32) | lo;
            best = o < best ? o : best;
        }
        const HItem bi = res[static_cast<uint32_t>(best)];
        cur = bi.node;
        cur_d = bi.dist;
        // extractSortedCandidates (hnsw.go:1026-1046) / selectNeighborsSimple: pop everything, nearest first
        const int64_t p = pair0 + level;
        const int nres = res_len;
        bool sorted = false;
        if constexpr (UK) {  // no two results tie: the pops' order is the ascending order (vg_hnsw_layer.hpp)
            uint64_t *keys = reinterpret_cast<uint64_t *>(cand);
            sorted = results_sorted_lds(res, nres, keys, nres, lane);
            if (sorted) {
                for (int i = lane; i < nres; i += 64) {
                    cand_ids[p * ef + i] = static_cast<uint32_t>(keys[i]);
                    cand_d[p * ef + i] = __uint_as_float(static_cast<uint32_t>(keys[i] >> 32));
                }
            }
        }
        for (int i = nres - 1; i >= 0 && !sorted; i--) {
            const HItem it = heap_pop<true, UK>(res, res_len);
            if (lane == 0) {
                cand_ids[p * ef + i] = it.node;
                cand_d[p * ef + i] = it.dist;
            }
        }
        if (lane == 0) cand_n[p] = nres;
        __syncthreads();
    }
}

// ---- 2. the new node's own neighbours ----------------------------------------------------------------
// pair_node / pair_level: the (node, level) of every pair of the batch
struct SelShared {
    float dm[64][65];     // pair distances between the members of the final list
    float dtmp[64];       // distances of the candidate under test to the selected ones
    uint32_t fid[64];     // final list: node ids
    float fd[64];         //             their distance to the new node
    int fsel[64];         //             1 = chosen by the heuristic, 0 = filled up
    int flag, nsel, nfinal;
};

__global__ __launch_bounds__(kSelThreads) void build_select_kernel(BuildGraph g, const uint32_t *__restrict__ pair_node,
                                                                   const int32_t *__restrict__ pair_level, int ef,
                                                                   const uint32_t *__restrict__ cand_ids,
                                                                   const float *__restrict__ cand_d,
                                                                   const int32_t *__restrict__ cand_n, int rec_stride,
                                                                   uint32_t *__restrict__ rec_row,
                                                                   uint32_t *__restrict__ rec_t, float *__restrict__ rec_d)
{
    __shared__ SelShared sh;
    const int64_t p = blockIdx.x;
    const int tid = threadIdx.x, grp = tid >> 4;
    const Sub16 sub = Sub16::make(tid);
    const uint32_t t = pair_node[p];
    const int level = pair_level[p];
    const int64_t row = bg_row(g, t, level);
    const int64_t off = bg_off(g, row);
    const int m = bg_deg(g, row);
    const int nc = cand_n[p];
    const uint32_t *cid = cand_ids + p * ef;
    const float *cdist = cand_d + p * ef;

    if (nc <= m) {  // selectNeighborsSimple (hnsw.go:993-1009): everything, nearest first
        for (int i = tid; i < nc; i += kSelThreads) {
            sh.fid[i] = cid[i];
            sh.fd[i] = cdist[i];
            sh.fsel[i] = 0;
        }
        if (tid == 0) {
            sh.nsel = 0;
            sh.nfinal = nc;
        }
        __syncthreads();
    } else {
        if (tid == 0) sh.nsel = 0;
        __syncthreads();
        for (int i = 0; i < nc; i++) {  // applyHeuristic (hnsw.go:1048-1085)
            const int nk = sh.nsel;
            if (nk >= m) break;
            const uint32_t id = cid[i];
            const float cd = cdist[i];
            if (tid == 0) sh.flag = 0;
            __syncthreads();
            for (int s0 = 0; s0 < nk; s0 += kSelThreads / 16) {
                const int s = s0 + grp;
                if (s < nk) {
                    const float d = bg_pair(g, id, sh.fid[s], sub);
                    if ((tid & 15) == 0) {
                        sh.dtmp[s] = d;
                        if (d < cd) sh.flag = 1;
                    }
                }
            }
            __syncthreads();
            if (!sh.flag) {
                for (int s = tid; s < nk; s += kSelThreads) {
                    sh.dm[nk][s] = sh.dtmp[s];
                    sh.dm[s][nk] = sh.dtmp[s];
                }
                if (tid == 0) {
                    sh.fid[nk] = id;
                    sh.fd[nk] = cd;
                    sh.fsel[nk] = 1;
                    sh.nsel = nk + 1;
                }
            }
            __syncthreads();
        }
        if (tid == 0) {  // fillUpNeighbors (hnsw.go:1087-1106)
            int nk = sh.nsel;
            const int nsel = nk;
            for (int i = 0; i < nc && nk < m; i++) {
                bool found = false;
                for (int s = 0; s < nsel; s++) found |= sh.fid[s] == cid[i];
                if (!found) {
                    sh.fid[nk] = cid[i];
                    sh.fd[nk] = cdist[i];
                    sh.fsel[nk] = 0;
                    nk++;
                }
            }
            sh.nfinal = nk;
        }
        __syncthreads();
    }
    // pair distances the heuristic did not need: every pair with a filled-up member
    const int nf = sh.nfinal, nsel = sh.nsel;
    const int nfill = nf - nsel;
    for (int e0 = 0; e0 < nfill * nf; e0 += kSelThreads / 16) {
        const int e = e0 + grp;
        if (e < nfill * nf) {
            const int a = nsel + e / nf, b = e % nf;
            if (b < a) {  // b selected, or an earlier fill-up
                const float d = bg_pair(g, sh.fid[a], sh.fid[b], sub);
                if ((tid & 15) == 0) {
                    sh.dm[a][b] = d;
                    sh.dm[b][a] = d;
                }
            }
        }
    }
    __syncthreads();
    // the row: setConnections (hnsw.go:445-453), plus the bit matrix, plus one back-link record per member
    for (int i = tid; i < m; i += kSelThreads) {
        uint64_t bits = 0;
        if (i < nf) {
            const float di = sh.fd[i];
            for (int j = 0; j < nf; j++)
                if (j != i && sh.dm[i][j] < di) bits |= 1ull << j;
        }
        g.ids[off + i] = i < nf ? sh.fid[i] : VG_INVALID_ID;
        g.dist[off + i] = i < nf ? sh.fd[i] : 0.0f;
        g.bits[off + i] = bits;
        if (i < rec_stride) {
            rec_row[p * rec_stride + i] = i < nf ? static_cast<uint32_t>(bg_row(g, sh.fid[i], level)) : VG_INVALID_ID;
            rec_t[p * rec_stride + i] = t;
            rec_d[p * rec_stride + i] = i < nf ? sh.fd[i] : 0.0f;
        }
    }
    for (int i = m + tid; i < rec_stride; i += kSelThreads) rec_row[p * rec_stride + i] = VG_INVALID_ID;
    if (tid == 0) {
        g.cnt[row] = nf;
        g.good[row] = nf == m && nsel == m ? 1 : 0;
    }
}

// ---- 3. back links ---------------------------------------------------------------------------------
struct LinkCounters {
    unsigned int nwork, total;
};
struct LinkTotals {  // over the whole build (VG_BUILD_DEBUG prints them)
    unsigned long long records, skipped, appended, pruned, good_rows_seen, longest_chain;
    // rows with >= 1000 records in a batch (hubs), 100 MHz ticks: where one workgroup's time goes
    unsigned long long hub_rows, hub_records, hub_applied, hub_sort, hub_scan, hub_gather, hub_replay, hub_total, hub_ties, max_wg;
};

// Both kernels append to ONE counter: a per-thread atomicAdd on it would serialise ~400 k atomics per batch on one
// L2 line (measured: most of the back-link stage).  The lanes of a wave are counted with a ballot / summed with a
// shuffle scan and the wave does one atomicAdd.
__global__ void build_count_kernel(const uint32_t *__restrict__ rec_row, int64_t nrec, int32_t *__restrict__ rcnt,
                                   uint32_t *__restrict__ work, LinkCounters *__restrict__ ctr)
{
    const int64_t r = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    uint32_t row = VG_INVALID_ID;
    if (r < nrec) row = rec_row[r];
    const bool first = row != VG_INVALID_ID && atomicAdd(&rcnt[row], 1) == 0;  // the row's first record this batch
    const uint64_t m = __ballot(first);
    if (m == 0) return;
    unsigned int base = 0;
    if (lane == __builtin_ctzll(m)) base = atomicAdd(&ctr->nwork, static_cast<unsigned int>(__popcll(m)));
    base = __shfl(base, __builtin_ctzll(m));
    if (first) work[base + __popcll(m & ((1ull << lane) - 1))] = row;
}

__global__ void build_offsets_kernel(const uint32_t *__restrict__ work, const int32_t *__restrict__ rcnt,
                                     uint32_t *__restrict__ roff, LinkCounters *__restrict__ ctr)
{
    const unsigned int w = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool live = w < ctr->nwork;
    const uint32_t row = live ? work[w] : 0;
    const unsigned int mine = live ? static_cast<unsigned int>(rcnt[row]) : 0u;
    unsigned int incl = mine;  // inclusive scan over the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned int up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
    }
    const unsigned int total = __shfl(incl, 63);
    if (total == 0) return;
    unsigned int base = 0;
    if (lane == 63) base = atomicAdd(&ctr->total, total);
    base = __shfl(base, 63);
    if (live) roff[row] = base + incl - mine;
}

__global__ void build_fill_kernel(const uint32_t *__restrict__ rec_row, const uint32_t *__restrict__ rec_t,
                                  const float *__restrict__ rec_d, int64_t nrec, const uint32_t *__restrict__ roff,
                                  int32_t *__restrict__ rfill, uint32_t *__restrict__ srt_t, float *__restrict__ srt_d)
{
    const int64_t r = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (r >= nrec) return;
    const uint32_t row = rec_row[r];
    if (row == VG_INVALID_ID) return;
    const uint32_t pos = roff[row] + static_cast<uint32_t>(atomicAdd(&rfill[row], 1));
    srt_t[pos] = rec_t[r];
    srt_d[pos] = rec_d[r];
}

struct LinkShared {
    HItem heap[65];
    float pd[64];
    float cd[65];          // candidate distances (slot 0..deg-1, then the new node)
    uint64_t cbits[65];    // bits over the old slots
    uint32_t ctbit[65];    // bit against the new node
    uint32_t cid[65];
    int seq[65];           // candidates in the order extractSortedCandidates would give (nearest first)
    int newpos[65];        // their slot after the prune, -1 = dropped
    uint32_t mid[64];      // the row's member ids, for the scoring waves
    uint32_t t;            // the record being applied
    int action;            // 0 = nothing to score for this record, 1 = score the new node against the members
    int cnt;
};

constexpr int kLinkWaves = 4;
constexpr int kLinkThreads = kLinkWaves * 64;

// LDS hand-over inside ONE wave (the bookkeeping wave below): its LDS instructions execute in order, so all it takes
// is that the compiler does not move them across this point
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// One workgroup of four waves per target row: the row's records in ascending id of the new node (the order a
// sequential pass over the batch applies them), each one = addConnection (hnsw.go:455-499).  Wave 0 keeps the row
// (lane i = slot i: id, cached distance, bit row) and does the bookkeeping; for a record that needs them, the 64
// distances between the new node and the row's members are scored by all four waves at once (16 members each, 4
// rows in flight per wave).  A row's records are a serial chain — a hub row (near to very many nodes, as
// high-dimensional data has them) can receive thousands per batch — so what counts is the latency of one link.
__global__ __launch_bounds__(kLinkThreads) void build_link_kernel(BuildGraph g, const uint32_t *__restrict__ work,
                                                                  const LinkCounters *__restrict__ ctr,
                                                                  int32_t *__restrict__ rcnt, int32_t *__restrict__ rfill,
                                                                  const uint32_t *__restrict__ roff,
                                                                  const uint32_t *__restrict__ srt_t,
                                                                  const float *__restrict__ srt_d,
                                                                  uint32_t *__restrict__ ord, float *__restrict__ ord_d,
                                                                  LinkTotals *__restrict__ totals)
{
    __shared__ LinkShared sh;
    if (blockIdx.x >= ctr->nwork) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const Sub16 sub = Sub16::make(lane);
    const int64_t row = work[blockIdx.x];
    const int64_t off = bg_off(g, row);
    const int deg = bg_deg(g, row);
    const int nadd = rcnt[row];
    const uint32_t *at = srt_t + roff[row];
    const float *ad = srt_d + roff[row];
    const unsigned long long tk_all = totals ? wall_clock64() : 0;
    const bool timed = totals != nullptr && nadd >= 1000;
    unsigned long long tk_sort = 0, tk_scan = 0, tk_gather = 0, tk_replay = 0;
    const unsigned long long tk_begin = timed ? wall_clock64() : 0;
    uint32_t *st = ord + roff[row];   // the row's records in ascending t (distinct per row): new node ...
    float *sd = ord_d + roff[row];    // ... and its distance to the row's node
    for (int i = tid; i < nadd; i += kLinkThreads) {  // rank sort by the whole workgroup, once
        const uint32_t ti = at[i];
        int rank = 0;
        for (int j = 0; j < nadd; j++) rank += at[j] < ti ? 1 : 0;
        st[rank] = ti;
        sd[rank] = ad[i];
    }
    __threadfence_block();
    __syncthreads();
    if (timed) tk_sort = wall_clock64() - tk_begin;

    // row state: wave 0 only
    int cnt = 0;
    uint32_t id = VG_INVALID_ID;
    float dist = 0.0f;
    uint64_t bits = 0;
    bool good = false, ties = false;
    bool stable = false;  // a good row WITH ties that a far record was seen to leave exactly as it is (see below)
    unsigned int n_skip = 0, n_app = 0, n_prune = 0, n_tie = 0;
    auto row_has_ties = [&]() {
        const float nxt = __shfl_down(dist, 1);
        return __ballot(lane + 1 < cnt && dist == nxt) != 0;
    };
    if (wave == 0) {
        cnt = g.cnt[row];
        id = lane < deg ? g.ids[off + lane] : VG_INVALID_ID;
        dist = lane < deg ? g.dist[off + lane] : 0.0f;
        bits = lane < deg ? g.bits[off + lane] : 0;
        // The cheap way out.  `good` = the row is full and applyHeuristic CHOSE every member (nothing was filled up),
        // so the row is in ascending distance order.  A new node farther than the farthest member then sorts last, the
        // heuristic re-chooses the 64 members, stops (len(result) >= m, hnsw.go:1054) and never looks at the newcomer:
        // the row is left exactly as it is — unless two members are equally far, in which case the reference's heap
        // may hand them back in another order, so a row with ties takes the long way ONCE: a newcomer farther than
        // every member compares the same way against everything whatever its distance, so if one such record leaves the
        // row exactly as it was (every member re-chosen, in the same slots), every later one will too, until the row
        // changes (`stable`).  Hub rows — thousands of back links per batch, and two bit-equal cached distances among
        // their 64 almost always — would otherwise replay the heap for every record.  75 % of the back links of the
        // 1M x 768 build end here.
        good = g.good[row] != 0;
        ties = row_has_ties();
        sh.mid[lane] = id;
    }

    // Wave 0 walks the records 64 at a time (one coalesced load per chunk, lane i = record chunk*64 + i) and tests
    // the cheap way out for a whole chunk at once: the row's state only changes when a record is actually applied,
    // so everything up to the first record that needs work is skipped without a barrier or a memory access.  The
    // other waves only see the records that need their 64 distances.
    int a = 0, chunk = -1;
    uint32_t ct = 0;
    float cdist = 0.0f;
    for (;;) {
        uint32_t t = 0;
        float dt = 0.0f;
        const unsigned long long tk0 = timed ? wall_clock64() : 0;
        if (wave == 0) {
            int action = 2;  // 2 = no record left
            while (a < nadd) {
                if ((a >> 6) != chunk) {
                    chunk = a >> 6;
                    const int i = chunk * 64 + lane;
                    ct = i < nadd ? st[i] : 0u;
                    cdist = i < nadd ? sd[i] : 0.0f;
                }
                const int i = chunk * 64 + lane;
                const bool pending = i >= a && i < nadd;
                const bool far = cnt == deg && good && (!ties || stable) && cdist > __shfl(dist, deg - 1);
                const uint64_t work = __ballot(pending && !far);  // records of this chunk that need a closer look
                const int rest = nadd - chunk * 64 < 64 ? nadd - chunk * 64 : 64;  // records in this chunk
                const int f = work ? __builtin_ctzll(work) : rest;                  // lane of the first of them
                n_skip += static_cast<unsigned int>(f - (a & 63));
                a = chunk * 64 + f;
                if (f == rest) continue;  // the chunk is done: next chunk (or the end)
                t = __shfl(ct, f);
                dt = __shfl(cdist, f);
                a++;
                if (__ballot(lane < cnt && id == t)) continue;  // already connected (hnsw.go:477-486)
                action = 1;
                break;
            }
            if (lane == 0) {
                sh.t = t;
                sh.action = action;
                sh.cnt = cnt;
            }
        }
        __syncthreads();
        const unsigned long long tk1 = timed ? wall_clock64() : 0;
        tk_scan += tk1 - tk0;
        if (sh.action == 2) break;
        {   // distances between the new node and the row's members: wave w scores members 16w .. 16w+15
            const uint32_t tt = sh.t;
            const int c = sh.cnt;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int j = wave * 16 + r * 4 + (lane >> 4);
                if (j < c) {
                    const float d = bg_pair(g, sh.mid[j], tt, sub);
                    if ((lane & 15) == 0) sh.pd[j] = d;
                }
            }
        }
        __syncthreads();
        const unsigned long long tk2 = timed ? wall_clock64() : 0;
        tk_gather += tk2 - tk1;
        if (wave == 0) {
            const float pd = lane < cnt ? sh.pd[lane] : 0.0f;
            const uint64_t tbits = __ballot(lane < cnt && pd < dt);       // row of the new node
            const uint32_t colbit = (lane < cnt && pd < dist) ? 1u : 0u;  // bit (member, new node)
            if (cnt < deg) {  // addConnectionSimple (hnsw.go:501-518)
                if (lane < cnt) bits |= static_cast<uint64_t>(colbit) << cnt;
                if (lane == cnt) {
                    id = t;
                    dist = dt;
                    bits = tbits;
                }
                cnt++;
                n_app++;
                stable = false;
                good = false;  // appended, not chosen
                ties = row_has_ties();
            } else {
                // addConnectionPrune (hnsw.go:520-555): members in list order, then the new node, through the max-heap
                // and back out nearest first.  Distinct distances: the order is the sort by distance.  Equal
                // distances: whatever the heap does, so the heap is replayed.
                n_prune++;
                const bool was_far = good && dt > __shfl(dist, deg - 1);  // farther than every member of a fully chosen row
                const int nc = deg + 1;
                if (lane < deg) {
                    sh.cd[lane] = dist;
                    sh.cbits[lane] = bits;
                    sh.ctbit[lane] = colbit;
                    sh.cid[lane] = id;
                }
                if (lane == 0) {
                    sh.cd[deg] = dt;
                    sh.cbits[deg] = tbits;
                    sh.ctbit[deg] = 0;
                    sh.cid[deg] = t;
                }
                wave_sync();
                int rank = 0;
                bool tie = false;
                if (lane < deg) {
                    for (int j = 0; j < nc; j++) {
                        const float dj = sh.cd[j];
                        rank += dj < dist ? 1 : 0;
                        tie |= (j != lane) && dj == dist;
                    }
                }
                int rank_t = 0;
                for (int j = 0; j < deg; j++) rank_t += sh.cd[j] < dt ? 1 : 0;
                if (__ballot(tie)) {
                    n_tie++;
                    int hl = 0;
                    for (int j = 0; j < nc; j++) heap_push<true>(sh.heap, hl, HItem{static_cast<uint32_t>(j), sh.cd[j]});
                    for (int j = nc - 1; j >= 0; j--) {
                        const HItem it = heap_pop<true>(sh.heap, hl);
                        if (lane == 0) sh.seq[j] = static_cast<int>(it.node);
                    }
                } else {
                    if (lane < deg) sh.seq[rank] = lane;
                    if (lane == 0) sh.seq[rank_t] = deg;
                }
                wave_sync();
                // applyHeuristic + fillUpNeighbors over the bit rows.  The greedy pass is sequential by nature; what
                // it reads is brought into registers first — lane p: the candidate at sorted position p, its bit row and
                // its bit against the new node (position 64, the last of 65, is kept as scalars) — so that a step is a
                // few readlanes instead of a chain of dependent LDS reads (this loop was half of a back link's time).
                const int my_ci = sh.seq[lane < nc ? lane : 0];
                const uint64_t my_cb = sh.cbits[my_ci];
                const uint32_t my_ctb = sh.ctbit[my_ci];
                const int last_ci = sh.seq[nc - 1];  // only meaningful when nc == 65
                const uint64_t last_cb = sh.cbits[last_ci];
                const uint32_t last_ctb = sh.ctbit[last_ci];
                int my_newpos = -1;  // lane j < deg: new slot of member j
                int newpos_t = -1;   // new slot of the new node
                uint64_t selmask = 0;
                bool sel_t = false;
                int nsel = 0;
                for (int p = 0; p < nc && nsel < deg; p++) {
                    int ci;
                    uint64_t cb;
                    uint32_t ctb;
                    if (p < 64) {
                        ci = __builtin_amdgcn_readlane(my_ci, p);
                        cb = readlane_u64(my_cb, p);
                        ctb = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(my_ctb), p));
                    } else {
                        ci = last_ci;
                        cb = last_cb;
                        ctb = last_ctb;
                    }
                    const bool bad = (cb & selmask) != 0 || (ctb != 0 && sel_t);
                    if (!bad) {
                        if (ci == deg) {
                            sel_t = true;
                            newpos_t = nsel;
                        } else {
                            selmask |= 1ull << ci;
                            if (lane == ci) my_newpos = nsel;
                        }
                        nsel++;
                    }
                }
                const int nchosen = nsel;
                for (int p = 0; p < nc && nsel < deg; p++) {
                    const int ci = p < 64 ? __builtin_amdgcn_readlane(my_ci, p) : last_ci;
                    const bool chosen = ci == deg ? sel_t : ((selmask >> ci) & 1) != 0;
                    if (!chosen) {
                        if (ci == deg)
                            newpos_t = nsel;
                        else if (lane == ci)
                            my_newpos = nsel;
                        nsel++;
                    }
                }
                // the row in its new order (slot -> candidate through an LDS scatter); the columns of the bit matrix
                // move with their members
                sh.newpos[lane] = -1;  // reused as slot -> candidate
                wave_sync();
                if (lane < deg && my_newpos >= 0) sh.newpos[my_newpos] = lane;
                if (lane == 0 && newpos_t >= 0) sh.newpos[newpos_t] = deg;
                wave_sync();
                const int src = sh.newpos[lane];
                uint64_t nb = 0;
                if (src >= 0) {
                    const uint64_t ob = sh.cbits[src];
                    for (int j = 0; j < deg; j++) {
                        const int np = __builtin_amdgcn_readlane(my_newpos, j);
                        if (np >= 0) nb |= ((ob >> j) & 1ull) << np;
                    }
                    if (newpos_t >= 0) nb |= static_cast<uint64_t>(sh.ctbit[src]) << newpos_t;
                    id = sh.cid[src];
                    dist = sh.cd[src];
                } else {
                    id = VG_INVALID_ID;
                    dist = 0.0f;
                }
                const bool unchanged = newpos_t < 0 && __ballot(lane < deg && src != lane) == 0;
                bits = nb;
                cnt = nsel;
                good = nchosen == deg;
                ties = row_has_ties();
                if (!unchanged)
                    stable = false;
                else if (was_far && good)
                    stable = true;
            }
            wave_sync();
            sh.mid[lane] = id;
        }
        __syncthreads();
        if (timed) tk_replay += wall_clock64() - tk2;
    }
    if (wave == 0) {
        if (lane < deg) {
            g.ids[off + lane] = lane < cnt ? id : VG_INVALID_ID;
            g.dist[off + lane] = lane < cnt ? dist : 0.0f;
            g.bits[off + lane] = lane < cnt ? bits : 0;
        }
        if (lane == 0) {
            if (totals) {
                atomicAdd(&totals->records, static_cast<unsigned long long>(nadd));
                atomicAdd(&totals->skipped, static_cast<unsigned long long>(n_skip));
                atomicAdd(&totals->appended, static_cast<unsigned long long>(n_app));
                atomicAdd(&totals->pruned, static_cast<unsigned long long>(n_prune));
                atomicAdd(&totals->good_rows_seen, static_cast<unsigned long long>(g.good[row] ? 1 : 0));
                atomicMax(&totals->longest_chain, static_cast<unsigned long long>(nadd));
                if (timed) {
                    atomicAdd(&totals->hub_rows, 1ull);
                    atomicAdd(&totals->hub_records, static_cast<unsigned long long>(nadd));
                    atomicAdd(&totals->hub_applied, static_cast<unsigned long long>(n_app + n_prune));
                    atomicAdd(&totals->hub_sort, tk_sort);
                    atomicAdd(&totals->hub_scan, tk_scan);
                    atomicAdd(&totals->hub_gather, tk_gather);
                    atomicAdd(&totals->hub_replay, tk_replay);
                    atomicAdd(&totals->hub_total, wall_clock64() - tk_begin);
                    atomicAdd(&totals->hub_ties, static_cast<unsigned long long>(n_tie));
                }
                {
                    const unsigned long long tk = wall_clock64() - tk_all;
                    atomicMax(&totals->max_wg, (tk << 40) | (static_cast<unsigned long long>(nadd & 0xFFFF) << 24) |
                                                   (static_cast<unsigned long long>(n_prune & 0xFFFF) << 8) | (n_tie > 255 ? 255 : n_tie));
                }
            }
            g.cnt[row] = cnt;
            g.good[row] = good ? 1 : 0;
            rcnt[row] = 0;
            rfill[row] = 0;
        }
    }
}

__global__ void fill_u32_kernel(uint32_t *p, int64_t n, uint32_t v)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// layerForApplyInsert (hnsw.go:2103-2116), layerMultiplier = 1 / ln(M) (hnsw.go:218)
static int32_t level_for_id(uint64_t id, double mult)
{
    uint64_t x = id + 0x9e3779b97f4a7c15ull;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
    x ^= x >> 31;
    const double inv = 1.0 / 9007199254740992.0;
    double r = static_cast<double>(x >> 11) * inv;
    if (r == 0) r = inv;
    const int32_t lv = static_cast<int32_t>(std::floor(-std::log(r) * mult));
    return lv > 62 ? 62 : lv;
}

template <typename T>
struct DevBuf {
    T *p = nullptr;
    ~DevBuf()
    {
        if (p) (void)hipFree(p);
    }
    int32_t alloc(size_t count)
    {
        VG_HIP(hipMalloc(reinterpret_cast<void **>(&p), std::max<size_t>(count, 1) * sizeof(T)));
        return VG_OK;
    }
    T *release()
    {
        T *r = p;
        p = nullptr;
        return r;
    }
};

}  // namespace vg

VG_API int32_t vg_hnsw_level_for_id(uint64_t id, int32_t m)
{
    return vg::level_for_id(id, 1.0 / std::log(static_cast<double>(m < 2 ? 2 : m)));
}

VG_API int32_t vg_hnsw_build(vg_index *idx, int32_t m, int32_t ef_construction, int32_t max_batch,
                             int32_t growth_div, void *stream)
{
    VG_CHECK(idx, VG_ERR_INVALID_ARG, "vg_hnsw_build: NULL index");
    VG_CHECK(idx->d_vectors && idx->n > 0, VG_ERR_NOT_READY, "vg_hnsw_build: index has no fp32 vectors");
    VG_CHECK(idx->metric != VG_METRIC_HAMMING, VG_ERR_UNSUPPORTED, "unsupported metric for float32: Hamming");
    VG_CHECK(m >= 2 && m <= 32, VG_ERR_UNSUPPORTED, "vg_hnsw_build: M=%d must be in 2..32 (M0 = 2M <= 64)", m);
    VG_CHECK(ef_construction >= 1 && ef_construction <= vg::kBuildMaxEf, VG_ERR_UNSUPPORTED,
             "vg_hnsw_build: ef_construction=%d must be in 1..%d", ef_construction, vg::kBuildMaxEf);
    VG_CHECK(max_batch >= 1 && growth_div >= 1, VG_ERR_INVALID_ARG, "vg_hnsw_build: max_batch and growth_div must be >= 1");
    VG_CHECK(idx->n < (int64_t(1) << 31), VG_ERR_UNSUPPORTED, "vg_hnsw_build: at most 2^31 rows");
    VG_HIP(hipSetDevice(idx->ctx->device));
    hipStream_t st = vg::pick_stream(idx->ctx, stream);
    const int64_t n = idx->n;
    const int m0 = 2 * m;  // mmax0Multiplier hnsw.go:28
    const int ef = ef_construction;

    // levels, slots, and what every node's insert will see as the top level (all known up front: ids and
    // levels are deterministic and so is the batch schedule)
    const double mult = 1.0 / std::log(static_cast<double>(m));
    std::vector<int32_t> levels(static_cast<size_t>(n));
    int top = 0;
    for (int64_t i = 0; i < n; i++) {
        levels[i] = vg::level_for_id(static_cast<uint64_t>(i), mult);
        top = std::max(top, levels[i]);
    }
    std::vector<int64_t> level_rows(static_cast<size_t>(top), 0), level_off(static_cast<size_t>(top) + 1, 0);
    std::vector<uint32_t> slots(static_cast<size_t>(top) * n);
    for (int l = 0; l < top; l++) {
        uint32_t next = 0;
        for (int64_t i = 0; i < n; i++) slots[static_cast<size_t>(l) * n + i] = levels[i] >= l + 1 ? next++ : VG_INVALID_ID;
        level_rows[l] = next;
        level_off[l + 1] = level_off[l] + next;
    }
    const int64_t upper_rows = level_off[top];
    const int64_t total_rows = n + upper_rows;
    VG_CHECK(total_rows < (int64_t(1) << 32) - 1, VG_ERR_UNSUPPORTED, "vg_hnsw_build: too many rows");
    const int64_t total_slots = n * m0 + upper_rows * m;

    struct Batch {
        int64_t t0, size, npairs;
        uint32_t entry;
        int cur_top;
    };
    std::vector<Batch> batches;
    std::vector<int64_t> pair_base(static_cast<size_t>(n) + 1, 0);
    std::vector<uint32_t> pair_node;
    std::vector<int32_t> pair_level;
    {
        uint32_t entry = 0;
        int cur_top = levels[0];
        int64_t done = 1;
        pair_base[1] = 0;
        int64_t max_pairs = 0;
        while (done < n) {
            int64_t b = done / growth_div;
            b = std::max<int64_t>(1, std::min<int64_t>(b, max_batch));
            b = std::min(b, n - done);
            Batch bt{done, b, 0, entry, cur_top};
            for (int64_t t = done; t < done + b; t++) {
                const int np = std::min(levels[t], cur_top) + 1;
                pair_base[t + 1] = pair_base[t] + np;
                bt.npairs += np;
            }
            for (int64_t t = done; t < done + b; t++)  // updateEntryPoint hnsw.go:885-900
                if (levels[t] > cur_top) {
                    cur_top = levels[t];
                    entry = static_cast<uint32_t>(t);
                }
            batches.push_back(bt);
            max_pairs = std::max(max_pairs, bt.npairs);
            done += b;
        }
        pair_node.resize(static_cast<size_t>(pair_base[n]));
        pair_level.resize(static_cast<size_t>(pair_base[n]));
        for (const Batch &bt : batches)
            for (int64_t t = bt.t0; t < bt.t0 + bt.size; t++)
                for (int l = 0; l <= std::min(levels[t], bt.cur_top); l++) {
                    pair_node[static_cast<size_t>(pair_base[t] + l)] = static_cast<uint32_t>(t);
                    pair_level[static_cast<size_t>(pair_base[t] + l)] = l;
                }
    }
    int64_t max_pairs = 1, max_b = 1;
    for (const auto &bt : batches) {
        max_pairs = std::max(max_pairs, bt.npairs);
        max_b = std::max(max_b, bt.size);
    }
    const int64_t vis_words_max = (n + 31) / 32;
    // visited bitmaps: one per node of a batch, at most 4 GiB — larger batches are not worth more
    VG_CHECK(max_b * vis_words_max * 4 <= (int64_t(1) << 32), VG_ERR_UNSUPPORTED,
             "vg_hnsw_build: max_batch=%d needs more than 4 GiB of visited bitmaps at %lld rows", max_batch,
             static_cast<long long>(n));

    vg::DevBuf<uint32_t> d_ids, d_slots, d_vis, d_cand_ids, d_rec_row, d_rec_t, d_work, d_roff, d_srt_t, d_pair_node, d_ord;
    vg::DevBuf<float> d_dist, d_cand_d, d_rec_d, d_srt_d, d_ordd;
    vg::DevBuf<uint64_t> d_bits;
    vg::DevBuf<int32_t> d_cnt, d_levels, d_cand_n, d_rcnt, d_rfill, d_pair_level;
    vg::DevBuf<uint8_t> d_good;
    vg::DevBuf<int64_t> d_level_off, d_pair_base;
    vg::DevBuf<vg::LinkCounters> d_ctr;
    vg::DevBuf<vg::LinkTotals> d_totals;
    const bool debug = vg::hook(vg::kHookBuildDebug);
    VG_TRY(d_ids.alloc(static_cast<size_t>(total_slots)));
    VG_TRY(d_dist.alloc(static_cast<size_t>(total_slots)));
    VG_TRY(d_bits.alloc(static_cast<size_t>(total_slots)));
    VG_TRY(d_cnt.alloc(static_cast<size_t>(total_rows)));
    VG_TRY(d_good.alloc(static_cast<size_t>(total_rows)));
    VG_TRY(d_slots.alloc(slots.size()));
    VG_TRY(d_level_off.alloc(level_off.size()));
    VG_TRY(d_levels.alloc(static_cast<size_t>(n)));
    VG_TRY(d_pair_base.alloc(pair_base.size()));
    VG_TRY(d_pair_node.alloc(pair_node.size()));
    VG_TRY(d_pair_level.alloc(pair_level.size()));
    VG_TRY(d_vis.alloc(static_cast<size_t>(max_b * vis_words_max)));
    VG_TRY(d_cand_ids.alloc(static_cast<size_t>(max_pairs) * ef));
    VG_TRY(d_cand_d.alloc(static_cast<size_t>(max_pairs) * ef));
    VG_TRY(d_cand_n.alloc(static_cast<size_t>(max_pairs)));
    const int64_t max_rec = max_pairs * m0;
    VG_TRY(d_rec_row.alloc(static_cast<size_t>(max_rec)));
    VG_TRY(d_rec_t.alloc(static_cast<size_t>(max_rec)));
    VG_TRY(d_rec_d.alloc(static_cast<size_t>(max_rec)));
    VG_TRY(d_srt_t.alloc(static_cast<size_t>(max_rec)));
    VG_TRY(d_srt_d.alloc(static_cast<size_t>(max_rec)));
    VG_TRY(d_ord.alloc(static_cast<size_t>(max_rec)));
    VG_TRY(d_ordd.alloc(static_cast<size_t>(max_rec)));
    VG_TRY(d_work.alloc(static_cast<size_t>(std::min(max_rec, total_rows))));
    VG_TRY(d_roff.alloc(static_cast<size_t>(total_rows)));
    VG_TRY(d_rcnt.alloc(static_cast<size_t>(total_rows)));
    VG_TRY(d_rfill.alloc(static_cast<size_t>(total_rows)));
    VG_TRY(d_ctr.alloc(1));
    VG_TRY(d_totals.alloc(1));
    VG_HIP(hipMemsetAsync(d_totals.p, 0, sizeof(vg::LinkTotals), st));
    VG_HIP(hipMemsetAsync(d_ids.p, 0xFF, static_cast<size_t>(total_slots) * 4, st));
    VG_HIP(hipMemsetAsync(d_dist.p, 0, static_cast<size_t>(total_slots) * 4, st));
    VG_HIP(hipMemsetAsync(d_bits.p, 0, static_cast<size_t>(total_slots) * 8, st));
    VG_HIP(hipMemsetAsync(d_cnt.p, 0, static_cast<size_t>(total_rows) * 4, st));
    VG_HIP(hipMemsetAsync(d_good.p, 0, static_cast<size_t>(total_rows), st));
    VG_HIP(hipMemsetAsync(d_rcnt.p, 0, static_cast<size_t>(total_rows) * 4, st));
    VG_HIP(hipMemsetAsync(d_rfill.p, 0, static_cast<size_t>(total_rows) * 4, st));
    VG_HIP(hipMemcpyAsync(d_slots.p, slots.data(), slots.size() * 4, hipMemcpyHostToDevice, st));
    VG_HIP(hipMemcpyAsync(d_level_off.p, level_off.data(), level_off.size() * 8, hipMemcpyHostToDevice, st));
    VG_HIP(hipMemcpyAsync(d_levels.p, levels.data(), levels.size() * 4, hipMemcpyHostToDevice, st));
    VG_HIP(hipMemcpyAsync(d_pair_base.p, pair_base.data(), pair_base.size() * 8, hipMemcpyHostToDevice, st));
    VG_HIP(hipMemcpyAsync(d_pair_node.p, pair_node.data(), pair_node.size() * 4, hipMemcpyHostToDevice, st));
    VG_HIP(hipMemcpyAsync(d_pair_level.p, pair_level.data(), pair_level.size() * 4, hipMemcpyHostToDevice, st));
    VG_HIP(hipStreamSynchronize(st));  // the host vectors above go out of use only at return, but be explicit

    vg::BuildGraph g{idx->d_vectors, n, idx->dim, idx->metric, m0, m, d_ids.p, d_dist.p, d_bits.p, d_cnt.p, d_good.p,
                     d_slots.p, d_level_off.p};
    const size_t lds = static_cast<size_t>(3 * ef + 4) * sizeof(vg::HItem) + 128 * sizeof(float);
    auto search_kern = idx->metric != VG_METRIC_DOT ? vg::build_search_kernel<true> : vg::build_search_kernel<false>;
    VG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(search_kern),
                               hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    auto dbg_t0 = std::chrono::steady_clock::now();
    unsigned long long dbg_max_chain = 0;
    for (const Batch &bt : batches) {
        const int64_t vis_words = (bt.t0 + 31) / 32;  // only nodes below t0 are reachable
        const int64_t nrec = bt.npairs * m0;
        VG_HIP(hipMemsetAsync(d_vis.p, 0, static_cast<size_t>(bt.size * vis_words) * 4, st));
        VG_HIP(hipMemsetAsync(d_ctr.p, 0, sizeof(vg::LinkCounters), st));
        {
            vg::ProfScope prof(idx->ctx, "hnsw_build_search", st);
            VG_LAUNCH(search_kern, dim3(static_cast<unsigned>(bt.size)), dim3(64), lds, st, g, bt.t0,
                      bt.entry, bt.cur_top, d_levels.p, d_pair_base.p, ef, d_vis.p, vis_words, d_cand_ids.p,
                      d_cand_d.p, d_cand_n.p);
        }
        const int64_t pb = pair_base[bt.t0];
        {
            vg::ProfScope prof(idx->ctx, "hnsw_build_select", st);
            VG_LAUNCH(vg::build_select_kernel, dim3(static_cast<unsigned>(bt.npairs)), dim3(vg::kSelThreads), 0, st, g,
                      d_pair_node.p + pb, d_pair_level.p + pb, ef, d_cand_ids.p, d_cand_d.p, d_cand_n.p, m0,
                      d_rec_row.p, d_rec_t.p, d_rec_d.p);
        }
        vg::ProfScope prof(idx->ctx, "hnsw_build_link", st);
        const unsigned gb = static_cast<unsigned>((nrec + 255) / 256);
        VG_LAUNCH(vg::build_count_kernel, dim3(gb), dim3(256), 0, st, d_rec_row.p, nrec, d_rcnt.p, d_work.p, d_ctr.p);
        const int64_t max_work = std::min(nrec, total_rows);
        VG_LAUNCH(vg::build_offsets_kernel, dim3(static_cast<unsigned>((max_work + 255) / 256)), dim3(256), 0, st,
                  d_work.p, d_rcnt.p, d_roff.p, d_ctr.p);
        VG_LAUNCH(vg::build_fill_kernel, dim3(gb), dim3(256), 0, st, d_rec_row.p, d_rec_t.p, d_rec_d.p, nrec,
                  d_roff.p, d_rfill.p, d_srt_t.p, d_srt_d.p);
        VG_LAUNCH(vg::build_link_kernel, dim3(static_cast<unsigned>(max_work)), dim3(vg::kLinkThreads), 0, st, g, d_work.p,
                  d_ctr.p, d_rcnt.p, d_rfill.p, d_roff.p, d_srt_t.p, d_srt_d.p, d_ord.p, d_ordd.p, debug ? d_totals.p : nullptr);
        if (debug) {  // per-batch wall time and the batch's longest per-row chain (the chain is a running maximum: reset it)
            VG_HIP(hipStreamSynchronize(st));
            const auto now = std::chrono::steady_clock::now();
            vg::LinkTotals t{};
            VG_HIP(hipMemcpy(&t, d_totals.p, sizeof(t), hipMemcpyDeviceToHost));
            const size_t bi = static_cast<size_t>(&bt - batches.data());
            if (bi % 16 == 0 || bi + 1 == batches.size())
                fprintf(stderr, "vg_hnsw_build: batch %zu: %lld nodes, %.2f ms, longest chain %llu; longest link workgroup %.2f ms "
                                "(%llu records, %llu pruned, %llu heap replays)\n", bi, static_cast<long long>(bt.size),
                        std::chrono::duration<double, std::milli>(now - dbg_t0).count(), t.longest_chain, (t.max_wg >> 40) / 1e5,
                        (t.max_wg >> 24) & 0xFFFF, (t.max_wg >> 8) & 0xFFFF, t.max_wg & 0xFF);
            dbg_t0 = now;
            dbg_max_chain = std::max(dbg_max_chain, t.longest_chain);
            const unsigned long long zero = 0;
            VG_HIP(hipMemcpy(&d_totals.p->longest_chain, &zero, sizeof(zero), hipMemcpyHostToDevice));
            VG_HIP(hipMemcpy(&d_totals.p->max_wg, &zero, sizeof(zero), hipMemcpyHostToDevice));
        }
    }
    VG_HIP(hipStreamSynchronize(st));
    if (debug) {
        vg::LinkTotals t{};
        VG_HIP(hipMemcpy(&t, d_totals.p, sizeof(t), hipMemcpyDeviceToHost));
        fprintf(stderr, "vg_hnsw_build: back links %llu = %llu skipped (farther than a fully chosen row's last member) + %llu appended + "
                        "%llu pruned; target-row visits that found the row fully chosen %llu; longest per-row chain in one batch %llu\n",
                t.records, t.skipped, t.appended, t.pruned, t.good_rows_seen, std::max(dbg_max_chain, t.longest_chain));
        fprintf(stderr, "vg_hnsw_build: rows with >= 1000 back links in a batch: %llu visits, %llu records of which %llu applied; per visit "
                        "%.1f us in all = sort %.1f + scan %.1f + gather %.1f + replay %.1f (us); heap replays (ties) %llu\n", t.hub_rows, t.hub_records, t.hub_applied,
                t.hub_rows ? t.hub_total / 100.0 / t.hub_rows : 0.0, t.hub_rows ? t.hub_sort / 100.0 / t.hub_rows : 0.0,
                t.hub_rows ? t.hub_scan / 100.0 / t.hub_rows : 0.0, t.hub_rows ? t.hub_gather / 100.0 / t.hub_rows : 0.0,
                t.hub_rows ? t.hub_replay / 100.0 / t.hub_rows : 0.0, t.hub_ties);
    }

    // hand the graph to the index in vg_index_set_hnsw_graph's layout
    uint32_t entry = 0;
    int cur_top = levels[0];
    for (int64_t t = 1; t < n; t++)
        if (levels[t] > cur_top) {
            cur_top = levels[t];
            entry = static_cast<uint32_t>(t);
        }
    // the new arrays are allocated and filled BEFORE the index lets go of its previous graph: an allocation that
    // fails here (the build's own scratch is still held) leaves the previous graph searchable, metadata and all
    vg::DevBuf<uint32_t> l0, adj;
    VG_TRY(l0.alloc(static_cast<size_t>(n) * m0));
    VG_TRY(adj.alloc(static_cast<size_t>(upper_rows) * m));
    VG_HIP(hipMemcpyAsync(l0.p, d_ids.p, static_cast<size_t>(n) * m0 * 4, hipMemcpyDeviceToDevice, st));
    if (upper_rows)
        VG_HIP(hipMemcpyAsync(adj.p, d_ids.p + n * m0, static_cast<size_t>(upper_rows) * m * 4,
                              hipMemcpyDeviceToDevice, st));
    VG_HIP(hipStreamSynchronize(st));
    for (uint32_t **slot : {&idx->d_hnsw_l0, &idx->d_hnsw_slot, &idx->d_hnsw_adj})
        if (*slot) {
            (void)hipFree(*slot);
            *slot = nullptr;
        }
    if (idx->d_hnsw_level_off) {
        (void)hipFree(idx->d_hnsw_level_off);
        idx->d_hnsw_level_off = nullptr;
    }
    idx->d_hnsw_l0 = l0.release();
    if (idx->d_hnsw_l0_dist) {  // the old graph's edge distances
        (void)hipFree(idx->d_hnsw_l0_dist);
        idx->d_hnsw_l0_dist = nullptr;
    }
    idx->d_hnsw_adj = adj.release();
    idx->d_hnsw_slot = d_slots.release();
    idx->d_hnsw_level_off = d_level_off.release();
    idx->hnsw_m0 = m0;
    idx->hnsw_m = m;
    idx->hnsw_max_level = cur_top;
    idx->hnsw_entry = entry;
    return VG_OK;
}

VG_API int32_t vg_index_get_hnsw_graph(const vg_index *idx, int32_t *m0, int32_t *m, int32_t *max_level,
                                       uint32_t *entry_point, int64_t *level_rows, uint32_t *l0,
                                       uint32_t *upper_slot, uint32_t *upper_adj, void *stream)
{
    VG_CHECK(idx, VG_ERR_INVALID_ARG, "vg_index_get_hnsw_graph: NULL index");
    VG_CHECK(idx->d_hnsw_l0, VG_ERR_NOT_READY, "vg_index_get_hnsw_graph: index has no HNSW graph");
    VG_HIP(hipSetDevice(idx->ctx->device));
    hipStream_t st = vg::pick_stream(idx->ctx, stream);
    const int L = idx->hnsw_max_level;
    std::vector<int64_t> off(static_cast<size_t>(L) + 1, 0);
    if (L > 0)
        VG_HIP(hipMemcpyAsync(off.data(), idx->d_hnsw_level_off, off.size() * 8, hipMemcpyDeviceToHost, st));
    VG_HIP(hipStreamSynchronize(st));
    if (m0) *m0 = idx->hnsw_m0;
    if (m) *m = idx->hnsw_m;
    if (max_level) *max_level = L;
    if (entry_point) *entry_point = idx->hnsw_entry;
    if (level_rows)
        for (int l = 0; l < L; l++) level_rows[l] = off[l + 1] - off[l];
    if (l0)
        VG_HIP(hipMemcpyAsync(l0, idx->d_hnsw_l0, static_cast<size_t>(idx->n) * idx->hnsw_m0 * 4, hipMemcpyDefault, st));
    if (upper_slot && L > 0)
        VG_HIP(hipMemcpyAsync(upper_slot, idx->d_hnsw_slot, static_cast<size_t>(L) * idx->n * 4, hipMemcpyDefault, st));
    if (upper_adj && L > 0)
        VG_HIP(hipMemcpyAsync(upper_adj, idx->d_hnsw_adj, static_cast<size_t>(off[L]) * idx->hnsw_m * 4,
                              hipMemcpyDefault, st));
    VG_HIP(hipStreamSynchronize(st));
    return VG_OK;
}
