// k_probe.hip — flat.Segment.Search over a partitioned (IVF) flat segment
// (internal/segment/flat/segment.go:727-749): with more than one partition the reference scans only
// the row ranges of the nprobes centroids closest to the query (kmeans.FindClosestCentroids,
// kmeans.go:217-280; nprobes <= 0 means 1), every range into the same bounded heap.  The heap order
// is total — (score, row id) — so the result is the k best keys of the union of the probed ranges:
//   1. per query: distances to the P centroids in SquaredL2Batch / DotBatch order, the nprobes best
//   2. per (query, probe[, slice]): scan of that partition's rows with the segment's scan type
//      (fp32 pair order here; PQ table lookups k_adc.hip; SQ8 k_sq8.hip), a k-list each
//   3. one merge of the lists per query
#include <algorithm>

#include "vg_device.hpp"
#include "vg_exact.hpp"
#include "vg_flat_gemm.hpp"
#include "vg_internal.hpp"
#include "vg_cand_replay.hpp"

namespace vg {

int32_t launch_topk_merge(const uint64_t *partial, int64_t nq, int lists, int k, bool descending,
                          uint32_t *ids, float *scores, hipStream_t st, const int *only_if = nullptr,
                          const int *always = nullptr);
int32_t launch_pq_build_table(const vg_pq *pq, const float *d_queries, int64_t nq, float *d_tables,
                              bool scan_layout, hipStream_t st);
int32_t launch_probe_scan_adc(const vg_index *idx, const float *tables, const uint32_t *probes, const uint32_t *part_off,
                              int64_t nq, int np, int split, int k, uint64_t *partial, const uint64_t *min_keys, bool desc,
                              const uint8_t *mask, int64_t mask_stride, hipStream_t st);
int32_t launch_probe_scan_sq8(const vg_index *idx, const float *queries, const uint32_t *probes, const uint32_t *part_off,
                              int64_t nq, int np, int sub, int k, uint64_t *partial, const uint64_t *min_keys,
                              const uint8_t *mask, int64_t mask_stride, hipStream_t st);
int32_t launch_probe_scan_sq8_grouped(const vg_index *idx, const float *queries, const uint32_t *part_off,
                                      const uint32_t *pair_of, const ProbeGroup *groups, const uint32_t *ngroups, unsigned gmax,
                                      int np, int sub, int k, uint64_t *partial, const uint64_t *min_keys, const uint8_t *mask,
                                      int64_t mask_stride, hipStream_t st);
int32_t flat_search_masked(vg_index *idx, const float *queries, int64_t nq, int32_t k, const uint8_t *mask, int64_t mask_stride,
                           uint32_t *ids, float *scores, void *stream, bool l2_scores = false, bool cand_replay = true);
int32_t sq8_nan_replay(vg_index *idx, const float *d_queries, int64_t nq, int k, const uint8_t *d_mask, int64_t mask_stride,
                       const uint32_t *d_probes, int np, const uint32_t *d_part_off, uint32_t *d_ids, float *d_scores, hipStream_t st);
int32_t pq_nan_replay(vg_index *idx, const float *d_queries, int64_t nq, int k, bool desc, const uint8_t *d_mask, int64_t mask_stride,
                      const uint32_t *d_probes, int np, const uint32_t *d_part_off, uint32_t *d_ids, float *d_scores, hipStream_t st);
int32_t pq_adc_search_masked(vg_index *idx, const float *queries, int64_t nq, int32_t k, const uint8_t *mask, int64_t mask_stride,
                             bool desc, uint32_t *ids, float *scores, void *stream);
size_t flat_probe_gemm_scratch_bytes(int64_t pairs, int64_t ns_max, int k, int bf16_dim);
int32_t launch_sq8_verify(vg_index *idx, const float *queries, int64_t nq, const ProbeNominated &nom, int k, uint32_t *ids, float *scores,
                          int *fail, hipStream_t st);
int32_t flat_probe_gemm(vg_index *idx, const float *pair_queries, int64_t pairs, const GemmGroup *groups, const int64_t *const first_block[4],
                        int ngroups, const int64_t grid[4], int sample_stride, int64_t ns_max, int k, uint32_t *pair_ids,
                        float *pair_scores, int *fail, char *scratch, const uint8_t *mask, const int64_t *mask_off, hipStream_t st,
                        const uint16_t *rows_bf16, const float *rows_norms, ProbeNominated *nominated);
bool sq8_nomination_applies(const vg_index *idx, const float *d_queries, int64_t nq, int k);
int32_t sq8_nominated_pass(vg_index *idx, const float *q, int64_t nq, int k, const uint8_t *mask, int64_t mask_stride, uint32_t *oid,
                           float *osc, hipStream_t st, std::vector<int> &failed);
int32_t launch_page_patch(int64_t nq, int k, int off, int kk, bool descending, const int *always_one,
                          const uint32_t *fids, const float *fscores, uint32_t *ids, float *scores, uint64_t *min_keys,
                          hipStream_t st);

// ---- 1. the nprobes closest centroids (kmeans.go:217-280) -----------------------------------------
// 16 lanes per centroid, batch-kernel order; for Dot / Cosine the reference sorts -dot ascending,
// i.e. the largest dot products first, which is the DOT key order.
// Equal distances (duplicated centroids): the reference's full sort (kmeans.go:272, pdqsort) leaves their order unpinned — by
// centroid id here, as in the oracle — but its SELECTION loop (n <= k/4 && n < 16, kmeans.go:255-269) is deterministic and not
// by id: each step takes the first minimum by POSITION and swaps it with the element at position i, which moves that element
// behind others of its own distance.  `emulate` (the host sets it when the selection loop is the reference's path): np + 1 keys
// are kept, and when two neighbours among them are equal the loop itself is replayed on all the distances in LDS (dynamic:
// parts floats + parts positions); with no tie among them every step's minimum is unique and the key order IS the loop's.
template <bool DOT>
__global__ __launch_bounds__(256) void probe_select_kernel(const float *__restrict__ queries, int dim,
                                                           const float *__restrict__ centroids, int parts, int np,
                                                           uint32_t *__restrict__ probes, int emulate)
{
    extern __shared__ float sel_dist[];
    __shared__ uint64_t lists[4 * 64];
    __shared__ int valid[4];
    __shared__ uint64_t best[64];
    __shared__ float red_d[4];
    __shared__ uint32_t red_p[4], red_c[4];
    const int64_t q = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const Sub16 sub = Sub16::make(tid);
    const float *qv = queries + q * dim;
    const int want = emulate ? np + 1 : np;
    __shared__ int any_nan;
    if (tid == 0) any_nan = 0;
    __syncthreads();
    WaveTopK tk;
    tk.init(want);
    bool nan_seen = false;
    for (int c0 = wave * 4; c0 < parts; c0 += 16) {
        const int c = c0 + (lane >> 4);
        uint64_t key = kKeyMax;
        if (c < parts) {
            const float v = exact_pair16<DOT, kBatch>(centroids + static_cast<int64_t>(c) * dim, qv, dim, sub);
            if ((lane & 15) == 0) key = make_key(v, static_cast<uint32_t>(c), DOT);
            nan_seen = nan_seen || v != v;
        }
        tk.offer(key, lane);
    }
    if (nan_seen) any_nan = 1;
    wg_rank_merge<4>(tk, lists, valid, wave, lane, tid, want, best);
    __syncthreads();
    // The full sort (kmeans.go:272, slices.SortFunc with cmpCentroidDistByDist: a NaN compares EQUAL to everything) of up to 12
    // entries is Go's insertionSortCmpFunc (sort/zsortanyfunc.go: pdqsort's small-slice case) — stable, so without a NaN it is the
    // key order; with one it is replayed: an entry moves left while it is `<` its left neighbour, and nothing moves past a NaN.
    // (More than 12 partitions: pdqsort proper, whose order with NaN distances is not restated — they sort last here.)
    constexpr int kInsMax = 12;
    __shared__ float ins_d[kInsMax];
    if (!emulate && any_nan != 0 && parts <= kInsMax) {
        for (int c0 = wave * 4; c0 < parts; c0 += 16) {
            const int c = c0 + (lane >> 4);
            if (c < parts) {
                const float v = exact_pair16<DOT, kBatch>(centroids + static_cast<int64_t>(c) * dim, qv, dim, sub);
                if ((lane & 15) == 0) ins_d[c] = DOT ? -v : v;
            }
        }
        __syncthreads();
        if (tid == 0) {
            int order[kInsMax];
            for (int i = 0; i < parts; i++) order[i] = i;
            for (int i = 1; i < parts; i++)
                for (int j = i; j > 0 && ins_d[order[j]] < ins_d[order[j - 1]]; j--) {
                    const int t = order[j];
                    order[j] = order[j - 1];
                    order[j - 1] = t;
                }
            for (int i = 0; i < np; i++) probes[q * np + i] = static_cast<uint32_t>(order[i]);
        }
        return;
    }
    // (a NaN distance anywhere: the selection loop takes a NaN standing at position i — the keys never would — so it is replayed)
    bool tie = emulate && any_nan != 0;
    if (emulate)
        for (int i = 0; i + 1 < want; i++)
            tie = tie || (best[i + 1] != kKeyMax && key_score(best[i], DOT) == key_score(best[i + 1], DOT));
    if (!tie) {
        if (tid < np) probes[q * np + tid] = key_row(best[tid]);
        return;
    }
    constexpr uint32_t kNone = 0xFFFFFFFFu;
    uint32_t *sel_pos = reinterpret_cast<uint32_t *>(sel_dist + parts);  // where each centroid's entry stands; kNone: taken
    for (int c0 = wave * 4; c0 < parts; c0 += 16) {
        const int c = c0 + (lane >> 4);
        if (c < parts) {
            const float v = exact_pair16<DOT, kBatch>(centroids + static_cast<int64_t>(c) * dim, qv, dim, sub);
            if ((lane & 15) == 0) {
                sel_dist[c] = DOT ? -v : v;  // kmeans.go:240: the dot product negated
                sel_pos[c] = static_cast<uint32_t>(c);
            }
        }
    }
    __syncthreads();
    for (int i = 0; i < np; i++) {
        float bd = 0.0f;
        uint32_t bp = kNone, bc = kNone;  // the first minimum by position among the entries not taken yet
        for (int c = tid; c < parts; c += 256) {
            const uint32_t p = sel_pos[c];
            if (p == kNone) continue;
            float d = sel_dist[c];
            // a NaN distance: `dists[j].dist < dists[minIdx].dist` is false either way, so the loop keeps minIdx = i when the
            // entry AT position i is NaN (it is taken: -Inf here, and position i wins every tie) and never moves to a NaN
            // behind it (skipped)
            if (d != d) {
                if (p != static_cast<uint32_t>(i)) continue;
                d = -INFINITY;
            }
            if (bc == kNone || d < bd || (d == bd && p < bp)) {
                bd = d;
                bp = p;
                bc = static_cast<uint32_t>(c);
            }
        }
        for (int off = 32; off > 0; off >>= 1) {
            const float od = __shfl_xor(bd, off);
            const uint32_t op = __shfl_xor(bp, off), oc = __shfl_xor(bc, off);
            if (oc != kNone && (bc == kNone || od < bd || (od == bd && op < bp))) {
                bd = od;
                bp = op;
                bc = oc;
            }
        }
        if (lane == 0) {
            red_d[wave] = bd;
            red_p[wave] = bp;
            red_c[wave] = bc;
        }
        __syncthreads();
        bd = red_d[0];
        bp = red_p[0];
        bc = red_c[0];
        for (int w = 1; w < 4; w++) {
            const float od = red_d[w];
            const uint32_t op = red_p[w], oc = red_c[w];
            if (oc != kNone && (bc == kNone || od < bd || (od == bd && op < bp))) {
                bd = od;
                bp = op;
                bc = oc;
            }
        }
        __syncthreads();
        // dists[i], dists[minIdx] = dists[minIdx], dists[i]: the entry standing at position i goes where the minimum stood
        for (int c = tid; c < parts; c += 256)
            if (sel_pos[c] == static_cast<uint32_t>(i) && static_cast<uint32_t>(c) != bc) sel_pos[c] = bp;
        if (tid == 0) {
            sel_pos[bc] = kNone;
            probes[q * np + i] = bc;
        }
        __syncthreads();
    }
}

// ---- 2. fp32 scan of one probed partition (segment.go:691-701) ------------------------------------
// MASKED (filter.Matches, segment.go:631-635: a row that fails is skipped unscored): every 16-lane group walks its own
// stream of rows (r0 + 4 * wave + group, step 16) and jumps over the rows whose bit is clear, so both the bytes and the
// arithmetic follow the filter's selectivity; which group scores a row does not matter, the key order is total.
template <bool DOT, bool MASKED>
__global__ __launch_bounds__(256) void probe_scan_f32_kernel(const float *__restrict__ base, int dim,
                                                             const float *__restrict__ queries,
                                                             const uint32_t *__restrict__ probes,
                                                             const uint32_t *__restrict__ part_off, int np, int sub_n,
                                                             int k, uint64_t *__restrict__ partial,
                                                             const uint64_t *__restrict__ min_keys,
                                                             const uint8_t *__restrict__ mask, int64_t mask_stride,
                                                             const int *__restrict__ only_if = nullptr)
{
    __shared__ uint64_t lists[4 * 64];
    __shared__ int valid[4];
    const int s = blockIdx.x, j = blockIdx.y;
    const int64_t q = blockIdx.z;
    if (only_if && !only_if[q]) return;  // the matrix-core path's fallback: only the flagged queries
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const Sub16 sub = Sub16::make(tid);
    const uint32_t p = probes[q * np + j];
    const int64_t R0 = part_off[p], R1 = part_off[p + 1];
    const int64_t r0 = R0 + (R1 - R0) * s / sub_n, r1 = R0 + (R1 - R0) * (s + 1) / sub_n;
    const float *qv = queries + q * dim;
    const uint8_t *mq = mask ? mask + q * mask_stride : nullptr;  // filter.Matches (segment.go:631-635): skipped unscored
    WaveTopK tk;
    tk.init(k);
    int64_t cur = r0 + wave * 4 + (lane >> 4);
    for (int64_t i0 = r0 + wave * 4; MASKED || i0 < r1; i0 += 16) {  // 4 rows per wave step, one per 16-lane group
        int64_t i = i0 + (lane >> 4);
        if (MASKED) {
            while (cur < r1 && !mask_bit(mq, cur)) cur += 16;
            i = cur;
            cur += 16;
            if (!__any(i < r1)) break;
        }
        uint64_t key = kKeyMax;
        if (i < r1) {
            const float v = exact_pair16<DOT, kPair>(base + i * dim, qv, dim, sub);
            if ((lane & 15) == 0) {
                key = make_key(v, static_cast<uint32_t>(i), DOT);
                if (min_keys && key <= min_keys[q]) key = kKeyMax;  // paged results (k > 64)
            }
        }
        tk.offer(key, lane);
    }
    wg_rank_merge<4>(tk, lists, valid, wave, lane, tid, k, partial + ((q * np + j) * sub_n + s) * k);
}

// ---- 2b. fp32 scan, queries grouped by partition ---------------------------------------------------
// With many (query, probe) pairs the same partition is probed by many queries: the pairs are bucketed
// by partition and cut into groups of up to kProbeQB; a workgroup then loads each row of its slice
// ONCE into registers and scores it against the group's queries held in LDS (the multi-query scan of
// k_flat.hip), instead of one pass over the partition per pair.

// one workgroup: bucket the pairs by partition, cut the buckets into groups.  counts[parts + 1] is
// zeroed by the caller; cursor / gstart are scratch of parts + 1 words; ngroups[0] receives the total.
__global__ __launch_bounds__(1024) void probe_group_kernel(const uint32_t *__restrict__ probes, int64_t pairs,
                                                           int parts, uint32_t *__restrict__ counts,
                                                           uint32_t *__restrict__ cursor, uint32_t *__restrict__ gstart,
                                                           uint32_t *__restrict__ pair_of,
                                                           ProbeGroup *__restrict__ groups, uint32_t *__restrict__ ngroups)
{
    __shared__ uint32_t seg_c[1024], seg_g[1024];
    const int tid = threadIdx.x;
    for (int64_t i = tid; i < pairs; i += 1024) atomicAdd(&counts[probes[i]], 1u);
    __syncthreads();
    const int per = (parts + 1023) / 1024;
    const int pb = tid * per < parts ? tid * per : parts, pe = pb + per < parts ? pb + per : parts;
    uint32_t mc = 0, mg = 0;
    for (int p = pb; p < pe; p++) {
        mc += counts[p];
        mg += (counts[p] + kProbeQB - 1) / kProbeQB;
    }
    seg_c[tid] = mc;
    seg_g[tid] = mg;
    __syncthreads();
    if (tid == 0) {
        uint32_t rc = 0, rg = 0;
        for (int t = 0; t < 1024; t++) {
            const uint32_t c = seg_c[t], g = seg_g[t];
            seg_c[t] = rc;
            seg_g[t] = rg;
            rc += c;
            rg += g;
        }
        ngroups[0] = rg;
    }
    __syncthreads();
    uint32_t rc = seg_c[tid], rg = seg_g[tid];
    for (int p = pb; p < pe; p++) {
        const uint32_t c = counts[p];
        cursor[p] = rc;
        gstart[p] = rg;
        const uint32_t ng = (c + kProbeQB - 1) / kProbeQB;
        for (uint32_t gi = 0; gi < ng; gi++) {
            ProbeGroup g;
            g.part = static_cast<uint32_t>(p);
            g.first = rc + gi * kProbeQB;
            g.count = c - gi * kProbeQB < kProbeQB ? c - gi * kProbeQB : kProbeQB;
            groups[rg + gi] = g;
        }
        rc += c;
        rg += ng;
    }
    __syncthreads();
    for (int64_t i = tid; i < pairs; i += 1024) pair_of[atomicAdd(&cursor[probes[i]], 1u)] = static_cast<uint32_t>(i);
}

template <bool DOT, bool MASKED>
__global__ __launch_bounds__(256) void probe_scan_f32_mq_kernel(const float *__restrict__ base, int dim,
                                                                const float *__restrict__ queries,
                                                                const uint32_t *__restrict__ part_off,
                                                                const uint32_t *__restrict__ pair_of,
                                                                const ProbeGroup *__restrict__ groups,
                                                                const uint32_t *__restrict__ ngroups, int np,
                                                                int sub_n, int k, uint64_t *__restrict__ partial,
                                                                const uint64_t *__restrict__ min_keys,
                                                                const uint8_t *__restrict__ mask, int64_t mask_stride)
{
    extern __shared__ float qlds[];  // kProbeQB * dim floats, then the merge scratch
    uint64_t *lists = reinterpret_cast<uint64_t *>(qlds + static_cast<size_t>(kProbeQB) * dim);
    int *valid = reinterpret_cast<int *>(lists + 4 * 64);
    __shared__ uint32_t pair[kProbeQB];
    __shared__ uint32_t qof[kProbeQB];  // the pairs' queries
    if (blockIdx.y >= ngroups[0]) return;
    const ProbeGroup g = groups[blockIdx.y];
    const int s = blockIdx.x;
    const int cnt = static_cast<int>(g.count);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < cnt) {
        pair[tid] = pair_of[g.first + tid];
        qof[tid] = pair[tid] / np;
    }
    __syncthreads();
    for (int qi = 0; qi < cnt; qi++) {
        const float *src = queries + static_cast<int64_t>(pair[qi] / np) * dim;
        for (int t = tid; t < dim; t += 256) qlds[qi * dim + t] = src[t];
    }
    __syncthreads();
    const Sub16 sub = Sub16::make(tid);
    const int nblk = dim >> 6;
    const int64_t R0 = part_off[g.part], R1 = part_off[g.part + 1];
    const int64_t r0 = R0 + (R1 - R0) * s / sub_n, r1 = R0 + (R1 - R0) * (s + 1) / sub_n;
    WaveTopK tk[kProbeQB];
#pragma unroll
    for (int qi = 0; qi < kProbeQB; qi++) tk[qi].init(k);
    // MASKED: a row some query of the group wants (its filter bit is set, segment.go:631-635) — the 16-lane groups each walk
    // their own stream of rows (r0 + 4 * wave + group, step 16) and jump over the others, so bytes and arithmetic follow the
    // selectivity; the per-query bit is checked again where the key is made
    auto wanted = [&](int64_t r) {
        if (mask_stride == 0) return mask_bit(mask, r);
        bool w = false;
        for (int qi = 0; qi < cnt; qi++) w = w || mask_bit(mask + static_cast<int64_t>(qof[qi]) * mask_stride, r);
        return w;
    };
    int64_t cur = r0 + wave * 4 + (lane >> 4);
    auto next_wanted = [&]() {
        while (cur < r1 && !wanted(cur)) cur += 16;
        const int64_t r = cur < r1 ? cur : r1;
        cur += 16;
        return r;
    };
    // a wave step = 8 rows, two per 16-lane group (rows i and i + 4): each LDS read of a query is used twice
    auto step = [&](const int64_t ia, const int64_t ib) {
        const bool livea = ia < r1, liveb = ib < r1;
        const float *rowa = base + (livea ? ia : r1 - 1) * dim;
        const float *rowb = base + (liveb ? ib : r1 - 1) * dim;
        float4 ra[16], rb[16];
        const float4 *a4 = reinterpret_cast<const float4 *>(rowa) + sub.f4;
        const float4 *b4 = reinterpret_cast<const float4 *>(rowb) + sub.f4;
#pragma unroll
        for (int e = 0; e < 16; e++)
            if (e < nblk) {
                ra[e] = a4[e * 16];
                rb[e] = b4[e * 16];
            }
        // lane 0 of a group carries row i, lane 1 row i + 4: one offer per 8 rows and query
        auto score = [&](int qi) {
            float va, vb;
            exact_rowregs16x2<DOT>(ra, rb, nblk, rowa, rowb, qlds + static_cast<size_t>(qi) * dim, dim, sub, va, vb);
            uint64_t key = kKeyMax;
            if ((lane & 15) == 0 && livea) key = make_key(va, static_cast<uint32_t>(ia), DOT);
            if ((lane & 15) == 1 && liveb) key = make_key(vb, static_cast<uint32_t>(ib), DOT);
            if (min_keys && key != kKeyMax && key <= min_keys[pair[qi] / np]) key = kKeyMax;  // paged results (k > 64)
            if (MASKED && mask_stride != 0 && key != kKeyMax &&  // each query its own filter
                !mask_bit(mask + static_cast<int64_t>(qof[qi]) * mask_stride, (lane & 15) == 0 ? ia : ib))
                key = kKeyMax;
            return key;
        };
        if (cnt == kProbeQB) {  // a full group: no per-query branches, the scores of all the queries first
            uint64_t keys[kProbeQB];
#pragma unroll
            for (int qi = 0; qi < kProbeQB; qi++) keys[qi] = score(qi);
#pragma unroll
            for (int qi = 0; qi < kProbeQB; qi++) tk[qi].offer(keys[qi], lane);
        } else {
#pragma unroll
            for (int qi = 0; qi < kProbeQB; qi++)
                if (qi < cnt) tk[qi].offer(score(qi), lane);
        }
    };
    if constexpr (MASKED) {
        for (;;) {
            const int64_t ia = next_wanted(), ib = next_wanted();
            if (!__any(ia < r1)) break;
            step(ia, ib);
        }
    } else {
        for (int64_t i0 = r0 + wave * 8; i0 < r1; i0 += 32) step(i0 + (lane >> 4), i0 + (lane >> 4) + 4);
    }
#pragma unroll
    for (int qi = 0; qi < kProbeQB; qi++) {
        if (qi < cnt) {
            wg_rank_merge<4>(tk[qi], lists, valid, wave, lane, tid, k,
                             partial + (static_cast<int64_t>(pair[qi]) * sub_n + s) * k);
            __syncthreads();
        }
    }
}

// ---- 2c. fp32 scan through the matrix cores: nomination + proof per (query, probe) pair ------------------------------------
// A batch whose partitions are each probed by many queries is a set of dense [queries of p] x [rows of p] products: the pairs
// are bucketed by partition, their query vectors gathered into one matrix, and vg_search_flat's machinery — threshold from a
// row sample, fp32 MFMA GEMM appending what falls below it, exact re-score of the 64 best, proof — runs over all partitions in
// one grouped launch per stage (flat_probe_gemm, k_flat.hip); a pair's k best rows are exact or its query is flagged, and the
// flagged queries are answered by the kernels above.  The per-query merge of the np lists is the usual one.
constexpr int kProbeSampleStride = 8;  // every 8th row tile of a partition sets its pairs' thresholds (16: the same times)

__global__ __launch_bounds__(256) void probe_bucket_count_kernel(const uint32_t *__restrict__ probes, int64_t pairs,
                                                                 uint32_t *__restrict__ counts)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < pairs) atomicAdd(&counts[probes[i]], 1u);
}

constexpr int64_t kProbeSmallMax = 512;  // pairs per partition up to which the 64-query tiles take the group
// one workgroup: the groups (pairs and rows of each partition) and each group's first workgroup in the two GEMM launches
__global__ __launch_bounds__(1024) void probe_bucket_scan_kernel(const uint32_t *__restrict__ counts, const uint32_t *__restrict__ part_off,
                                                                 int parts, uint32_t *__restrict__ cursor, GemmGroup *__restrict__ groups,
                                                                 int64_t *__restrict__ fb_sample, int64_t *__restrict__ fb_main,
                                                                 int64_t *__restrict__ fb_sample_small, int64_t *__restrict__ fb_main_small)
{
    // A group of up to kProbeSmallMax pairs takes 64-query tiles (flat_gemm_dma32_grouped_kernel<.., 2>: one workgroup per query tile
    // and row tile), a larger one the 128-query tiles; each group has workgroups in one of the two launches.  A full 64-query tile
    // costs half a 128-query one (6.2 against 16.6 us per 8192-row partition: the small tile is balanced between the matrix unit
    // and the rows' bytes) and a group's last tile is half empty on average whatever its size — 1024 queries x 8 probes over 122
    // partitions are groups of 67 +- 8 pairs: r05 sent those of more than 64 pairs (three in five) to one half-empty 128-query
    // tile each.
    __shared__ int64_t seg_c[1024], seg_s[1024], seg_m[1024], seg_ss[1024], seg_sm[1024];
    const int tid = threadIdx.x;
    const int per = (parts + 1023) / 1024;
    const int pb = tid * per < parts ? tid * per : parts, pe = pb + per < parts ? pb + per : parts;
    auto blocks = [&](int p, int64_t &bs, int64_t &bm, int64_t &ss, int64_t &sm) {
        const int64_t cnt = counts[p], rows = static_cast<int64_t>(part_off[p + 1]) - part_off[p];
        const int64_t mt = (cnt + kGemmBM - 1) / kGemmBM, nt = (rows + kGemmBN - 1) / kGemmBN;
        const int64_t nst = (nt + kProbeSampleStride - 1) / kProbeSampleStride;
        const bool any = cnt && rows, small = cnt <= kProbeSmallMax;
        const int64_t mt64 = (cnt + kProbeRB * kG32BM - 1) / (kProbeRB * kG32BM);
        bs = any && !small ? mt * ((nst + 7) / 8) * 8 : 0;
        bm = any && !small ? mt * ((nt + 7) / 8) * 8 : 0;
        ss = any && small ? mt64 * nst : 0;
        sm = any && small ? mt64 * nt : 0;
    };
    int64_t mc = 0, ms = 0, mm = 0, mss = 0, msm = 0;
    for (int p = pb; p < pe; p++) {
        int64_t bs, bm, ss, sm;
        blocks(p, bs, bm, ss, sm);
        mc += counts[p];
        ms += bs;
        mm += bm;
        mss += ss;
        msm += sm;
    }
    seg_c[tid] = mc;
    seg_s[tid] = ms;
    seg_m[tid] = mm;
    seg_ss[tid] = mss;
    seg_sm[tid] = msm;
    __syncthreads();
    if (tid == 0) {
        int64_t rc = 0, rs = 0, rm = 0, rss = 0, rsm = 0;
        for (int t = 0; t < 1024; t++) {
            const int64_t c = seg_c[t], s = seg_s[t], m = seg_m[t], a = seg_ss[t], b = seg_sm[t];
            seg_c[t] = rc;
            seg_s[t] = rs;
            seg_m[t] = rm;
            seg_ss[t] = rss;
            seg_sm[t] = rsm;
            rc += c;
            rs += s;
            rm += m;
            rss += a;
            rsm += b;
        }
        fb_sample[parts] = rs;
        fb_main[parts] = rm;
        fb_sample_small[parts] = rss;
        fb_main_small[parts] = rsm;
    }
    __syncthreads();
    int64_t rc = seg_c[tid], rs = seg_s[tid], rm = seg_m[tid], rss = seg_ss[tid], rsm = seg_sm[tid];
    for (int p = pb; p < pe; p++) {
        int64_t bs, bm, ss, sm;
        blocks(p, bs, bm, ss, sm);
        GemmGroup g;
        g.a_off = rc;
        g.b_off = part_off[p];
        g.a_cnt = static_cast<int32_t>(counts[p]);
        g.b_cnt = static_cast<int32_t>(part_off[p + 1] - part_off[p]);
        groups[p] = g;
        cursor[p] = static_cast<uint32_t>(rc);
        fb_sample[p] = rs;
        fb_main[p] = rm;
        fb_sample_small[p] = rss;
        fb_main_small[p] = rsm;
        rc += counts[p];
        rs += bs;
        rm += bm;
        rss += ss;
        rsm += sm;
    }
}

__global__ __launch_bounds__(256) void probe_bucket_fill_kernel(const uint32_t *__restrict__ probes, int64_t pairs,
                                                                uint32_t *__restrict__ cursor, uint32_t *__restrict__ pair_of)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < pairs) pair_of[atomicAdd(&cursor[probes[i]], 1u)] = static_cast<uint32_t>(i);
}

// row `pos` of the pair matrix = the query of the pos-th bucketed pair (dim % 4 == 0, 16-byte aligned)
__global__ __launch_bounds__(256) void probe_gather_queries_kernel(const float *__restrict__ queries, const uint32_t *__restrict__ pair_of,
                                                                   int np, int dim, float *__restrict__ out, int64_t mask_stride,
                                                                   int64_t *__restrict__ mask_off)
{
    const int64_t pos = blockIdx.x;
    if (threadIdx.x == 0) mask_off[pos] = static_cast<int64_t>(pair_of[pos] / np) * mask_stride;  // the pair's query's filter
    const float4 *src = reinterpret_cast<const float4 *>(queries + static_cast<int64_t>(pair_of[pos] / np) * dim);
    float4 *dst = reinterpret_cast<float4 *>(out + pos * dim);
    for (int t = threadIdx.x; t < dim / 4; t += 256) dst[t] = src[t];
}

// a pair's k results as keys in the slot of its (query, probe) in the per-query lists; a failed proof flags the query
__global__ __launch_bounds__(64) void probe_pack_kernel(const uint32_t *__restrict__ pair_of, const uint32_t *__restrict__ ids,
                                                        const float *__restrict__ scores, const int *__restrict__ fail, int k, int np,
                                                        bool desc, uint64_t *__restrict__ partial, int *__restrict__ qfail)
{
    const int64_t pos = blockIdx.x;
    const uint32_t pr = pair_of[pos];
    for (int i = threadIdx.x; i < k; i += 64) {
        const uint32_t id = ids[pos * k + i];
        partial[static_cast<int64_t>(pr) * k + i] = id == VG_INVALID_ID ? kKeyMax : make_key(scores[pos * k + i], id, desc);
    }
    if (threadIdx.x == 0 && fail[pos]) qfail[pr / np] = 1;
}

// A filtered search of an unpartitioned segment (segment.go:745-749 with `filter` set): ONE range, the whole segment.  The
// heap order is total, so the range is cut into `parts` equal pieces scanned like probed partitions — every query "probes"
// all of them — to spread the rows over the device.
__global__ void probe_whole_segment_kernel(int64_t n, int parts, int64_t pairs, uint32_t *__restrict__ part_off,
                                           uint32_t *__restrict__ probes)
{
    const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t <= parts) part_off[t] = static_cast<uint32_t>(n * t / parts);
    if (t < pairs) probes[t] = static_cast<uint32_t>(t % parts);
}

}  // namespace vg

VG_API int32_t vg_index_set_partitions(vg_index *idx, const float *centroids, const uint32_t *part_offsets,
                                       int32_t num_partitions, void *stream)
{
    VG_CHECK(idx, VG_ERR_INVALID_ARG, "vg_index_set_partitions: NULL index");
    VG_CHECK(num_partitions >= 0, VG_ERR_INVALID_ARG, "vg_index_set_partitions: negative partition count");
    VG_HIP(hipSetDevice(idx->ctx->device));
    hipStream_t st = vg::pick_stream(idx->ctx, stream);
    VG_HIP(hipStreamSynchronize(st));  // earlier searches may still read the old tables
    if (idx->d_centroids) VG_HIP(hipFree(idx->d_centroids));
    if (idx->d_part_off) VG_HIP(hipFree(idx->d_part_off));
    idx->d_centroids = nullptr;
    idx->d_part_off = nullptr;
    idx->num_partitions = 0;
    idx->h_part_off.clear();
    if (num_partitions == 0) return VG_OK;
    VG_CHECK(centroids && part_offsets, VG_ERR_INVALID_ARG, "vg_index_set_partitions: NULL buffer");
    // the offsets come from a file: check them once here instead of in every scan
    std::vector<uint32_t> off(static_cast<size_t>(num_partitions) + 1);
    VG_HIP(hipMemcpy(off.data(), part_offsets, off.size() * sizeof(uint32_t), hipMemcpyDefault));  // host or device
    for (size_t p = 0; p + 1 < off.size(); p++)
        VG_CHECK(off[p] <= off[p + 1], VG_ERR_INVALID_ARG, "vg_index_set_partitions: partition offsets decrease at %zu", p);
    VG_CHECK(static_cast<int64_t>(off.back()) <= idx->n, VG_ERR_INVALID_ARG,
             "vg_index_set_partitions: partition offsets end at %u, the index has %lld rows", off.back(),
             static_cast<long long>(idx->n));
    const size_t cbytes = sizeof(float) * static_cast<size_t>(num_partitions) * idx->dim;
    VG_HIP(hipMalloc(reinterpret_cast<void **>(&idx->d_centroids), cbytes));
    VG_HIP(hipMalloc(reinterpret_cast<void **>(&idx->d_part_off), off.size() * sizeof(uint32_t)));
    VG_HIP(hipMemcpyAsync(idx->d_centroids, centroids, cbytes, hipMemcpyDefault, st));
    VG_HIP(hipMemcpyAsync(idx->d_part_off, off.data(), off.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    VG_HIP(hipStreamSynchronize(st));  // `off` is a local; the caller's buffers are free again
    idx->num_partitions = num_partitions;
    idx->h_part_off = off;
    return VG_OK;
}

// probes_out: where the call leaves its probe lists (device, nq * np), or null: scratch
static int32_t flat_probed_impl(vg_index *idx, const float *queries, int64_t nq, int32_t k, int32_t nprobes, int32_t scan,
                                const uint8_t *mask, int64_t mask_stride, uint32_t *ids, float *scores, void *stream,
                                bool allow_nomination = true, uint32_t *probes_out = nullptr);

// kmeans.FindClosestCentroids for every query (device buffers): probes[q * np + j]
static int32_t launch_probe_select(const vg_index *idx, const float *d_queries, int64_t nq, int np, bool dot, uint32_t *d_probes, hipStream_t st)
{
    const size_t sel_lds = 8 * static_cast<size_t>(idx->num_partitions);
    const int emulate = np <= idx->num_partitions / 4 && np < 16 && sel_lds <= 156 * 1024;
    auto kern = dot ? vg::probe_select_kernel<true> : vg::probe_select_kernel<false>;
    if (emulate && sel_lds > 48 * 1024)
        VG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   static_cast<int>(sel_lds)));
    VG_LAUNCH(kern, dim3(static_cast<unsigned>(nq)), dim3(256), emulate ? sel_lds : 0, st, d_queries, idx->dim, idx->d_centroids,
              idx->num_partitions, np, d_probes, emulate);
    return VG_OK;
}

// The two entry points: flat_probed_impl on device buffers, then — include/vecgo_hip.h "NaN scores" — the queries whose scores may
// hold a NaN or an Inf once more through the reference's heap (vg_cand_replay.hpp): the rows a query's filter lets through, in
// the order the reference visits them (the whole range, or the probed partitions' ranges in FindClosestCentroids' order).
static int32_t flat_probed_entry(vg_index *idx, const float *queries, int64_t nq, int32_t k, int32_t nprobes, int32_t scan,
                                 const uint8_t *mask, int64_t mask_stride, uint32_t *ids, float *scores, void *stream)
{
    // (arguments flat_probed_impl refuses, and the unfiltered whole segment — vg_search_flat / _pq_adc / _sq8 replay for themselves)
    if (!idx || nq <= 0 || k <= 0 || !queries || !ids || !scores || (scan != VG_SCAN_F32 && scan != VG_SCAN_PQ && scan != VG_SCAN_SQ8) ||
        (idx->num_partitions <= 1 && mask == nullptr) || idx->n == 0 || vg::hook(vg::kHookNoCandReplay))
        return flat_probed_impl(idx, queries, nq, k, nprobes, scan, mask, mask_stride, ids, scores, stream);
    VG_HIP(hipSetDevice(idx->ctx->device));
    hipStream_t st = vg::pick_stream(idx->ctx, stream);
    vg::DevIn<float> q;
    vg::DevIn<uint8_t> mk;
    vg::DevOut<uint32_t> oid;
    vg::DevOut<float> osc;
    const int64_t mask_bytes = (idx->n + 7) / 8;
    VG_TRY(q.init(queries, static_cast<size_t>(nq) * idx->dim, st));
    VG_TRY(mk.init(mask, mask ? static_cast<size_t>(mask_stride ? (nq - 1) * mask_stride + mask_bytes : mask_bytes) : 0, st));
    VG_TRY(oid.init(ids, static_cast<size_t>(nq) * k, st));
    VG_TRY(osc.init(scores, static_cast<size_t>(nq) * k, st));
    const bool whole = idx->num_partitions <= 1, dot = idx->metric != VG_METRIC_L2;
    int np = nprobes <= 0 ? 1 : nprobes;
    if (np > idx->num_partitions) np = idx->num_partitions;
    vg::DevTmp<uint32_t> probes;  // the search leaves its probe lists here: the replay walks the same partitions in the same order
    if (!whole) VG_TRY(probes.init(static_cast<size_t>(nq) * np, st));
    VG_TRY(flat_probed_impl(idx, q.ptr, nq, k, nprobes, scan, mk.ptr, mask_stride, oid.ptr, osc.ptr, st, true, whole ? nullptr : probes.ptr));
    const uint32_t *pr = whole ? nullptr : probes.ptr, *po = whole ? nullptr : idx->d_part_off;
    if (scan == VG_SCAN_SQ8)
        VG_TRY(vg::sq8_nan_replay(idx, q.ptr, nq, k, mk.ptr, mask_stride, pr, np, po, oid.ptr, osc.ptr, st));
    else if (scan == VG_SCAN_PQ)
        VG_TRY(vg::pq_nan_replay(idx, q.ptr, nq, k, dot, mk.ptr, mask_stride, pr, np, po, oid.ptr, osc.ptr, st));
    else
        VG_TRY(vg::launch_cand_replay(vg::FlatF32Scorer{idx->d_vectors, idx->d_norm_max + 1, idx->dim, dot, 0}, q.ptr, idx->dim, idx->n, nq, k, dot,
                                      mk.ptr, mask_stride, oid.ptr, osc.ptr, st, nullptr, pr, np, po));
    VG_TRY(oid.finish());
    VG_TRY(osc.finish());
    if (oid.on_host() || osc.on_host()) VG_HIP(hipStreamSynchronize(st));
    return VG_OK;
}

VG_API int32_t vg_search_flat_probed(vg_index *idx, const float *queries, int64_t nq, int32_t k, int32_t nprobes,
                                     int32_t scan, uint32_t *ids, float *scores, void *stream)
{
    return flat_probed_entry(idx, queries, nq, k, nprobes, scan, nullptr, 0, ids, scores, stream);
}

VG_API int32_t vg_search_flat_filtered(vg_index *idx, const float *queries, int64_t nq, int32_t k, int32_t nprobes,
                                       int32_t scan, const uint8_t *mask, int64_t mask_stride, uint32_t *ids, float *scores,
                                       void *stream)
{
    VG_CHECK(idx, VG_ERR_INVALID_ARG, "vg_search_flat_filtered: NULL index");
    const int64_t mask_bytes = (idx->n + 7) / 8;
    VG_CHECK(mask == nullptr || mask_stride == 0 || mask_stride >= mask_bytes, VG_ERR_INVALID_ARG,
             "vg_search_flat_filtered: mask_stride %lld is shorter than a mask (%lld bytes)", static_cast<long long>(mask_stride),
             static_cast<long long>(mask_bytes));
    return flat_probed_entry(idx, queries, nq, k, nprobes, scan, mask, mask_stride, ids, scores, stream);
}

static int32_t flat_probed_impl(vg_index *idx, const float *queries, int64_t nq, int32_t k, int32_t nprobes, int32_t scan,
                                const uint8_t *mask, int64_t mask_stride, uint32_t *ids, float *scores, void *stream,
                                bool allow_nomination, uint32_t *probes_out)
{
    VG_CHECK(idx, VG_ERR_INVALID_ARG, "vg_search_flat_probed: NULL index");
    VG_CHECK(scan == VG_SCAN_F32 || scan == VG_SCAN_PQ || scan == VG_SCAN_SQ8, VG_ERR_INVALID_ARG,
             "vg_search_flat_probed: unknown scan type %d", scan);
    const bool whole = idx->num_partitions <= 1;  // segment.go:745-749: one range, the whole segment
    if (whole && mask == nullptr) {
        if (scan == VG_SCAN_PQ) return vg_search_pq_adc(idx, queries, nq, k, ids, scores, stream);
        if (scan == VG_SCAN_SQ8) return vg_search_sq8(idx, queries, nq, k, ids, scores, stream);
        return vg_search_flat(idx, queries, nq, k, ids, scores, stream);
    }
    VG_CHECK(nq >= 0 && k >= 0, VG_ERR_INVALID_ARG, "vg_search_flat_probed: negative nq or k");
    if (nq == 0 || k == 0) return VG_OK;
    VG_CHECK(queries && ids && scores, VG_ERR_INVALID_ARG, "vg_search_flat_probed: NULL buffer");
    VG_CHECK(k <= 512, VG_ERR_UNSUPPORTED, "vg_search_flat_probed: k=%d exceeds 512", k);
    VG_CHECK(idx->metric != VG_METRIC_HAMMING, VG_ERR_UNSUPPORTED, "unsupported metric for float32: Hamming");
    int np = nprobes <= 0 ? 1 : nprobes;  // segment.go:728-731
    if (np > idx->num_partitions) np = idx->num_partitions;  // kmeans.go:219-221
    // filtered and unpartitioned: the one range in up to 64 pieces of at least 4096 rows (probe_whole_segment_kernel)
    const int parts = whole ? static_cast<int>(std::min<int64_t>(64, std::max<int64_t>(1, idx->n / 4096))) : idx->num_partitions;
    if (whole) np = parts;
    VG_CHECK(np <= 64, VG_ERR_UNSUPPORTED, "vg_search_flat_probed: nprobes=%d exceeds 64", np);
    const bool dot = idx->metric != VG_METRIC_L2;
    if (scan == VG_SCAN_F32) {
        VG_CHECK(idx->d_vectors, VG_ERR_NOT_READY, "vg_search_flat_probed: index has no fp32 vectors");
    } else if (scan == VG_SCAN_PQ) {
        VG_CHECK(idx->pq && idx->d_pq_tiles, VG_ERR_NOT_READY, "vg_search_flat_probed: index has no PQ codes");
        VG_CHECK(idx->pq->k == 256, VG_ERR_UNSUPPORTED,
                 "vg_search_flat_probed: LUT scan needs numCentroids == 256 (got %d)", idx->pq->k);
    } else {
        VG_CHECK(idx->sq && idx->d_sq_tiles, VG_ERR_NOT_READY, "vg_search_flat_probed: index has no SQ8 codes");
    }
    VG_HIP(hipSetDevice(idx->ctx->device));
    hipStream_t st = vg::pick_stream(idx->ctx, stream);
    vg::DevIn<float> q;
    vg::DevOut<uint32_t> oid;
    vg::DevOut<float> osc;
    VG_TRY(q.init(queries, static_cast<size_t>(nq) * idx->dim, st));
    VG_TRY(oid.init(ids, static_cast<size_t>(nq) * k, st));
    VG_TRY(osc.init(scores, static_cast<size_t>(nq) * k, st));
    vg::DevIn<uint8_t> mk;
    const int64_t mask_bytes = (idx->n + 7) / 8;
    VG_TRY(mk.init(mask, mask ? static_cast<size_t>(mask_stride ? (nq - 1) * mask_stride + mask_bytes : mask_bytes) : 0, st));

    // A batch of filtered fp32 queries over the whole segment: the matrix-core nomination of vg_search_flat with the filter
    // applied where candidates are sampled and appended (k_flat.hip) — its cost does not depend on the selectivity, the
    // kernels below pay per wanted row: 8 queries up it wins or ties (tools/filtered_time.py).  Same results either way
    // (the test hook keeps the batch on the kernels below).
    if (whole && scan == VG_SCAN_F32 && mk.ptr && nq >= 8 && !vg::hook(vg::kHookProbeNoGroup)) {
        VG_TRY(vg::flat_search_masked(idx, q.ptr, nq, k, mk.ptr, mask_stride, oid.ptr, osc.ptr, stream));
        VG_TRY(oid.finish());
        VG_TRY(osc.finish());
        if (oid.on_host() || osc.on_host()) VG_HIP(hipStreamSynchronize(st));
        return VG_OK;
    }

    // A filtered SQ8 batch over the whole segment with vg_index_enable_sq8_nomination: the bf16 nomination with the filter in its
    // epilogue, the exact re-score from the codes, the proof (k_sq8.hip); queries whose proof fails take the kernels below
    if (whole && scan == VG_SCAN_SQ8 && mk.ptr && allow_nomination && vg::sq8_nomination_applies(idx, q.ptr, nq, k) &&
        !vg::hook(vg::kHookProbeNoGroup)) {
        std::vector<int> failed;
        VG_TRY(vg::sq8_nominated_pass(idx, q.ptr, nq, k, mk.ptr, mask_stride, oid.ptr, osc.ptr, st, failed));
        if (!failed.empty()) {
            const int64_t nf = static_cast<int64_t>(failed.size());
            vg::DevTmp<float> fq;
            vg::DevTmp<uint32_t> fid;
            vg::DevTmp<float> fsc;
            vg::DevTmp<uint8_t> fm;
            VG_TRY(fq.init(static_cast<size_t>(nf) * idx->dim, st));
            VG_TRY(fid.init(static_cast<size_t>(nf) * k, st));
            VG_TRY(fsc.init(static_cast<size_t>(nf) * k, st));
            VG_TRY(fm.init(mask_stride ? static_cast<size_t>(nf) * mask_bytes : 0, st));
            for (int64_t i = 0; i < nf; i++) {
                const int64_t src = failed[static_cast<size_t>(i)];
                VG_HIP(hipMemcpyAsync(fq.ptr + i * idx->dim, q.ptr + src * idx->dim, sizeof(float) * idx->dim, hipMemcpyDeviceToDevice, st));
                if (mask_stride)
                    VG_HIP(hipMemcpyAsync(fm.ptr + i * mask_bytes, mk.ptr + src * mask_stride, static_cast<size_t>(mask_bytes),
                                          hipMemcpyDeviceToDevice, st));
            }
            VG_TRY(flat_probed_impl(idx, fq.ptr, nf, k, nprobes, scan, mask_stride ? fm.ptr : mk.ptr, mask_stride ? mask_bytes : 0, fid.ptr,
                                    fsc.ptr, st, false));
            for (int64_t i = 0; i < nf; i++) {
                const int64_t at = static_cast<int64_t>(failed[static_cast<size_t>(i)]) * k;
                VG_HIP(hipMemcpyAsync(oid.ptr + at, fid.ptr + i * k, sizeof(uint32_t) * k, hipMemcpyDeviceToDevice, st));
                VG_HIP(hipMemcpyAsync(osc.ptr + at, fsc.ptr + i * k, sizeof(float) * k, hipMemcpyDeviceToDevice, st));
            }
        }
        VG_TRY(oid.finish());
        VG_TRY(osc.finish());
        if (oid.on_host() || osc.on_host()) VG_HIP(hipStreamSynchronize(st));
        return VG_OK;
    }

    // A filtered PQ scan of the whole segment, k <= 64 and a table that fits LDS: the pipelined scan of vg_search_pq_adc with
    // the filter where keys are made (the probe kernel below is the plain loop, twice its time)
    if (whole && scan == VG_SCAN_PQ && mk.ptr && k <= 64 && idx->pq->m <= 96 && !vg::hook(vg::kHookProbeNoGroup)) {
        VG_TRY(vg::pq_adc_search_masked(idx, q.ptr, nq, k, mk.ptr, mask_stride, dot, oid.ptr, osc.ptr, stream));
        VG_TRY(oid.finish());
        VG_TRY(osc.finish());
        if (oid.on_host() || osc.on_host()) VG_HIP(hipStreamSynchronize(st));
        return VG_OK;
    }

    // More (query, probe) pairs than one grouped nomination takes (65535): the batch in chunks of queries that do fit, each
    // chunk deciding for itself below — 8192 queries x 8 probes would otherwise run on the scan kernels, 3x the time per query.
    {
        const int64_t qc = 65535 / np;
        if (!whole && nq * np > 65535 && scan != VG_SCAN_PQ && k <= vg::kProbeGemmMaxK && qc * np >= 12 * static_cast<int64_t>(parts) &&
            !vg::hook(vg::kHookProbeNoGroup) && !vg::hook(vg::kHookProbeNoGemm)) {
            for (int64_t q0 = 0; q0 < nq; q0 += qc) {
                const int64_t cnt = std::min<int64_t>(qc, nq - q0);
                VG_TRY(flat_probed_impl(idx, q.ptr + q0 * idx->dim, cnt, k, nprobes, scan, mk.ptr ? mk.ptr + q0 * mask_stride : nullptr,
                                        mask_stride, oid.ptr + q0 * k, osc.ptr + q0 * k, st, allow_nomination,
                                        probes_out ? probes_out + q0 * np : nullptr));
            }
            VG_TRY(oid.finish());
            VG_TRY(osc.finish());
            if (oid.on_host() || osc.on_host()) VG_HIP(hipStreamSynchronize(st));
            return VG_OK;
        }
    }
    // enough workgroups to fill the device when there are few (query, probe) pairs
    const int64_t pairs = nq * np;
    // fp32: with enough pairs the queries are grouped by partition (rows read once per group); the
    // row-in-registers scan needs 16-byte aligned rows of at most 1024 floats
    // (a filtered whole segment is "probed" in every piece by every query: grouping pays from a handful of queries on)
    const bool group_ok = (whole ? nq >= 4 : pairs >= 16) && !vg::hook(vg::kHookProbeNoGroup);  // test hook: one pass per pair
    const bool grouped_f32 = scan == VG_SCAN_F32 && group_ok && idx->dim % 4 == 0 && idx->dim <= 1024 &&
                             (reinterpret_cast<uintptr_t>(idx->d_vectors) & 15) == 0;
    // SQ8: the group's queries (padded to whole 16-dimension groups) have to fit LDS next to the merge scratch
    const bool grouped_sq8 = scan == VG_SCAN_SQ8 && group_ok &&
                             sizeof(float) * vg::kProbeQB * static_cast<size_t>(idx->sq_groups) * 16 <= 128 * 1024;
    const bool grouped = grouped_f32 || grouped_sq8;
    const int want = 4 * idx->ctx->compute_units;
    int sub = static_cast<int>(std::min<int64_t>(32, std::max<int64_t>(1, (want + pairs - 1) / pairs)));
    int split = static_cast<int>(std::min<int64_t>(np, std::max<int64_t>(1, (idx->ctx->compute_units + nq - 1) / nq)));
    if (grouped) {  // workgroups are (group of up to kProbeQB pairs) x slice: size the slices for the groups
        const int64_t g_est = std::max<int64_t>(1, pairs / vg::kProbeQB) + std::min<int64_t>(parts, pairs) / 2;
        sub = static_cast<int>(std::min<int64_t>(32, std::max<int64_t>(1, (want + g_est - 1) / g_est)));
    }
    const int lists = scan == VG_SCAN_PQ ? split : np * sub;
    const int lut_words = scan == VG_SCAN_PQ ? (((idx->pq->m >> 4) + 1) >> 1) * 8192 + (idx->pq->m & 15) * 256 : 0;

    const int64_t qchunk = std::max<int64_t>(1, 65535 / np);
    const int64_t chunk_pairs = std::min<int64_t>(nq, qchunk) * np;
    const size_t gwords = grouped ? static_cast<size_t>(parts) + 1 : 0;

    vg::ArenaCall ar(idx->ctx, st);
    const int i_gcounts = ar.add(sizeof(uint32_t) * gwords);
    const int i_gcursor = ar.add(sizeof(uint32_t) * gwords);
    const int i_gstart = ar.add(sizeof(uint32_t) * gwords);
    const int i_pair_of = ar.add(grouped ? sizeof(uint32_t) * static_cast<size_t>(chunk_pairs) : 0);
    const int i_groups = ar.add(grouped ? sizeof(vg::ProbeGroup) * static_cast<size_t>(chunk_pairs) : 0);
    const int i_ngroups = ar.add(grouped ? 256 : 0);
    const bool paged = k > 64;
    const int pk = paged ? 64 : k;
    const int i_probes = ar.add(sizeof(uint32_t) * static_cast<size_t>(nq) * np);
    // (the grouped nomination leaves k keys per pair there, also when k is paged for the scan kernels)
    const size_t partial_keys = std::max(static_cast<size_t>(nq) * lists * pk, whole || k > vg::kProbeGemmMaxK || scan == VG_SCAN_PQ || pairs > 65535 ? size_t(0) : static_cast<size_t>(pairs) * k);
    const int i_partial = ar.add(sizeof(uint64_t) * partial_keys);
    const int i_pid = ar.add(paged ? sizeof(uint32_t) * static_cast<size_t>(nq) * pk : 0);
    const int i_psc = ar.add(paged ? sizeof(float) * static_cast<size_t>(nq) * pk : 0);
    const int i_floor = ar.add(paged ? sizeof(uint64_t) * static_cast<size_t>(nq) : 0);
    const int i_one = ar.add(paged ? 256 : 0);
    const int i_tables = ar.add(sizeof(float) * static_cast<size_t>(nq) * lut_words);
    const int i_whole = ar.add(whole ? sizeof(uint32_t) * (static_cast<size_t>(parts) + 1) : 0);
    // fp32, unfiltered, partitions probed by 12 or more queries each on average: nomination + proof on the matrix cores (2c)
    // (1M x 768 in 122 partitions, 1024 queries, ms per call, exact kernels -> this: nprobes 1 (8 per partition) 1.21 -> 1.18,
    // 2: 2.1 -> 1.4, 4: 3.75 -> 1.3, 8: 7.25 -> 2.3, 16: 11.6 -> 4.5, 32: 21.0 -> 7.0; tools/probe_gemm_time.py)
    // (a filtered batch too: a pair's filter is its query's — probe_gather_queries_kernel notes where each starts)
    // (k <= 48: the pairs' 64 best nominees re-scored; up to kProbeGemmMaxK: everything below a deeper threshold re-scored, and the
    // flagged queries — normally none — searched again as a subset, the scan kernels' k > 64 being paged)
    const bool gemm_shape = !whole && k <= vg::kProbeGemmMaxK && (k <= 64 || allow_nomination) && idx->dim % 4 == 0 && pairs <= 65535 &&
                            pairs >= 12 * static_cast<int64_t>(parts) &&
                            (reinterpret_cast<uintptr_t>(q.ptr) & 15) == 0 && static_cast<int>(idx->h_part_off.size()) == parts + 1 &&
                            !vg::hook(vg::kHookProbeNoGroup) && !vg::hook(vg::kHookProbeNoGemm);
    const bool gemm_f32 = scan == VG_SCAN_F32 && gemm_shape && (reinterpret_cast<uintptr_t>(idx->d_vectors) & 15) == 0;
    // SQ8 with vg_index_enable_sq8_nomination: the same grouped nomination on the bfloat16 image of the dequantised rows, the
    // pairs' 64 candidates re-scored from the codes and proven by sq8_verify_kernel (k_sq8.hip)
    const bool gemm_sq8 = scan == VG_SCAN_SQ8 && gemm_shape && allow_nomination && idx->d_sq_bf16 != nullptr;
    const bool gemm = gemm_f32 || gemm_sq8;
    int64_t grids[4] = {0, 0, 0, 0}, ns_max = 0;  // sample / main of the 128-query tiles, sample / main of the 64-query tiles
    if (gemm) {  // launch bounds from the partition sizes: one query tile per partition + the batch's further tiles on the largest
        int64_t sum_s = 0, sum_m = 0, max_s = 0, max_m = 0, max_nst = 0, max_nt = 0;
        for (int p = 0; p < parts; p++) {
            const int64_t rows = static_cast<int64_t>(idx->h_part_off[p + 1]) - idx->h_part_off[p];
            const int64_t nt = (rows + vg::kGemmBN - 1) / vg::kGemmBN, nst = (nt + vg::kProbeSampleStride - 1) / vg::kProbeSampleStride;
            const int64_t bs = ((nst + 7) / 8) * 8, bm = ((nt + 7) / 8) * 8;
            sum_s += bs;
            sum_m += bm;
            max_s = std::max(max_s, bs);
            max_m = std::max(max_m, bm);
            grids[2] += nst;
            grids[3] += nt;
            max_nst = std::max(max_nst, nst);
            max_nt = std::max(max_nt, nt);
            ns_max = std::max(ns_max, nst * vg::kGemmBN);
        }
        grids[0] = sum_s + (pairs / vg::kGemmBM) * max_s;
        grids[1] = sum_m + (pairs / vg::kGemmBM) * max_m;
        grids[2] += (pairs / (vg::kProbeRB * vg::kG32BM)) * max_nst;  // (64-query tiles: a partition's first + the batch's further ones on the largest)
        grids[3] += (pairs / (vg::kProbeRB * vg::kG32BM)) * max_nt;
    }
    const size_t gw = gemm ? static_cast<size_t>(parts) + 1 : 0;
    const int i_bcnt = ar.add(sizeof(uint32_t) * gw);
    const int i_bcur = ar.add(sizeof(uint32_t) * gw);
    const int i_bgrp = ar.add(sizeof(vg::GemmGroup) * gw);
    const int i_fbs = ar.add(sizeof(int64_t) * gw);
    const int i_fbm = ar.add(sizeof(int64_t) * gw);
    const int i_fbss = ar.add(sizeof(int64_t) * gw);
    const int i_fbsm = ar.add(sizeof(int64_t) * gw);
    const int i_bpair = ar.add(gemm ? sizeof(uint32_t) * static_cast<size_t>(pairs) : 0);
    const int i_pairq = ar.add(gemm ? sizeof(float) * static_cast<size_t>(pairs) * idx->dim : 0);
    const int i_pids = ar.add(gemm ? sizeof(uint32_t) * static_cast<size_t>(pairs) * k : 0);
    const int i_pscore = ar.add(gemm ? sizeof(float) * static_cast<size_t>(pairs) * k : 0);
    const int i_pfail = ar.add(gemm ? sizeof(int) * static_cast<size_t>(pairs) : 0);
    const int i_qfail = ar.add(gemm ? sizeof(int) * static_cast<size_t>(nq) : 0);
    const int i_moff = ar.add(gemm ? sizeof(int64_t) * static_cast<size_t>(pairs) : 0);
    // (fp32 rows with vg_index_enable_bf16_filter: the grouped nomination on that image too)
    const bool gemm_f32_bf16 = gemm_f32 && idx->d_vectors_bf16 != nullptr;
    const int i_gscr = ar.add(gemm ? vg::flat_probe_gemm_scratch_bytes(pairs, ns_max, k, gemm_sq8 ? idx->sq_bf16_dim : gemm_f32_bf16 ? idx->vectors_bf16_dim : 0) : 0);
    VG_TRY(ar.commit());
    uint32_t *probes = probes_out && !whole ? probes_out : ar.get<uint32_t>(i_probes);
    uint64_t *partial = ar.get<uint64_t>(i_partial);
    float *tables = ar.get<float>(i_tables);
    uint32_t *pid = ar.get<uint32_t>(i_pid);
    float *psc = ar.get<float>(i_psc);
    uint64_t *floor_keys = ar.get<uint64_t>(i_floor);
    int *one = ar.get<int>(i_one);
    uint32_t *gcounts = ar.get<uint32_t>(i_gcounts), *gcursor = ar.get<uint32_t>(i_gcursor);
    uint32_t *gstart = ar.get<uint32_t>(i_gstart), *pair_of = ar.get<uint32_t>(i_pair_of);
    uint32_t *ngroups = ar.get<uint32_t>(i_ngroups);
    vg::ProbeGroup *groups = ar.get<vg::ProbeGroup>(i_groups);

    const uint32_t *part_off = whole ? ar.get<uint32_t>(i_whole) : idx->d_part_off;
    if (whole) {
        const int64_t threads = std::max<int64_t>(pairs, parts + 1);
        VG_LAUNCH(vg::probe_whole_segment_kernel, dim3(static_cast<unsigned>((threads + 255) / 256)), dim3(256), 0, st, idx->n, parts,
                  pairs, ar.get<uint32_t>(i_whole), probes);
    } else {
        // the reference's selection loop (kmeans.go:255: n <= k/4 && n < 16) is replayed where centroid distances tie; its LDS
        // (8 bytes per partition, beside the kernel's 2.6 KB of static LDS) bounds that to 19 968 partitions — beyond, ties are
        // broken by centroid id
        VG_TRY(launch_probe_select(idx, q.ptr, nq, np, dot, probes, st));
    }
    // the heap direction follows the segment metric for EVERY scan (flat/segment.go:449): with Dot / Cosine a PQ
    // scan therefore keeps the k LARGEST table-lookup (squared-L2) distances — the reference as written
    const bool desc = dot;
    if (gemm) {
        uint32_t *bcnt = ar.get<uint32_t>(i_bcnt), *bcur = ar.get<uint32_t>(i_bcur), *bpair = ar.get<uint32_t>(i_bpair);
        vg::GemmGroup *bgrp = ar.get<vg::GemmGroup>(i_bgrp);
        int64_t *fbs = ar.get<int64_t>(i_fbs), *fbm = ar.get<int64_t>(i_fbm), *fbss = ar.get<int64_t>(i_fbss), *fbsm = ar.get<int64_t>(i_fbsm);
        float *pairq = ar.get<float>(i_pairq), *pair_sc = ar.get<float>(i_pscore);
        uint32_t *pair_ids = ar.get<uint32_t>(i_pids);
        int *pfail = ar.get<int>(i_pfail), *qfail = ar.get<int>(i_qfail);
        const unsigned pb = static_cast<unsigned>((pairs + 255) / 256);
        VG_HIP(hipMemsetAsync(bcnt, 0, sizeof(uint32_t) * (static_cast<size_t>(parts) + 1), st));
        VG_HIP(hipMemsetAsync(qfail, 0, sizeof(int) * static_cast<size_t>(nq), st));
        VG_LAUNCH(vg::probe_bucket_count_kernel, dim3(pb), dim3(256), 0, st, probes, pairs, bcnt);
        VG_LAUNCH(vg::probe_bucket_scan_kernel, dim3(1), dim3(1024), 0, st, bcnt, part_off, parts, bcur, bgrp, fbs, fbm, fbss, fbsm);
        VG_LAUNCH(vg::probe_bucket_fill_kernel, dim3(pb), dim3(256), 0, st, probes, pairs, bcur, bpair);
        int64_t *moff = ar.get<int64_t>(i_moff);
        VG_LAUNCH(vg::probe_gather_queries_kernel, dim3(static_cast<unsigned>(pairs)), dim3(256), 0, st, q.ptr, bpair, np, idx->dim, pairq,
                  mask_stride, moff);
        {
            vg::ProfScope prof(idx->ctx, "flat_probe", st);
            const int64_t *const fb[4] = {fbs, fbm, fbss, fbsm};
            vg::ProbeNominated nom{};
            VG_TRY(vg::flat_probe_gemm(idx, pairq, pairs, bgrp, fb, parts, grids, vg::kProbeSampleStride, ns_max, k, pair_ids, pair_sc,
                                       pfail, ar.get<char>(i_gscr), mk.ptr, moff, st,
                                       gemm_sq8 ? idx->d_sq_bf16 : gemm_f32_bf16 ? idx->d_vectors_bf16 : nullptr,
                                       gemm_sq8 ? idx->d_sq_norms : gemm_f32_bf16 ? idx->d_norms : nullptr, &nom));
            if (gemm_sq8) VG_TRY(vg::launch_sq8_verify(idx, pairq, pairs, nom, k, pair_ids, pair_sc, pfail, st));
        }
        // lists = np * sub in this configuration; the pairs' k results fill the first np lists' worth of `partial`
        VG_LAUNCH(vg::probe_pack_kernel, dim3(static_cast<unsigned>(pairs)), dim3(64), 0, st, bpair, pair_ids, pair_sc, pfail, k, np, desc,
                  partial, qfail);
        VG_TRY(vg::launch_topk_merge(partial, nq, np, k, desc, oid.ptr, osc.ptr, st));
        if (gemm_sq8 || k > 64) {  // the flagged queries (normally none) go through the scan kernels: read the flags, search that subset
            std::vector<int> hq(static_cast<size_t>(nq));
            VG_HIP(hipMemcpyAsync(hq.data(), qfail, sizeof(int) * static_cast<size_t>(nq), hipMemcpyDeviceToHost, st));
            VG_HIP(hipStreamSynchronize(st));
            std::vector<int> failed;
            for (int64_t i = 0; i < nq; i++)
                if (hq[static_cast<size_t>(i)]) failed.push_back(static_cast<int>(i));
            ar.lock.unlock();  // the subset's own search takes the arena
            if (!failed.empty()) {
                const int64_t nf = static_cast<int64_t>(failed.size());
                vg::DevTmp<float> fq;
                vg::DevTmp<uint32_t> fid;
                vg::DevTmp<float> fsc;
                vg::DevTmp<uint8_t> fm;
                VG_TRY(fq.init(static_cast<size_t>(nf) * idx->dim, st));
                VG_TRY(fid.init(static_cast<size_t>(nf) * k, st));
                VG_TRY(fsc.init(static_cast<size_t>(nf) * k, st));
                VG_TRY(fm.init(mk.ptr && mask_stride ? static_cast<size_t>(nf) * mask_bytes : 0, st));
                for (int64_t i = 0; i < nf; i++) {
                    const int64_t src = failed[static_cast<size_t>(i)];
                    VG_HIP(hipMemcpyAsync(fq.ptr + i * idx->dim, q.ptr + src * idx->dim, sizeof(float) * idx->dim, hipMemcpyDeviceToDevice, st));
                    if (mk.ptr && mask_stride)
                        VG_HIP(hipMemcpyAsync(fm.ptr + i * mask_bytes, mk.ptr + src * mask_stride, static_cast<size_t>(mask_bytes),
                                              hipMemcpyDeviceToDevice, st));
                }
                VG_TRY(flat_probed_impl(idx, fq.ptr, nf, k, nprobes, scan, mk.ptr ? (mask_stride ? fm.ptr : mk.ptr) : nullptr,
                                        mask_stride ? mask_bytes : 0, fid.ptr, fsc.ptr, st, false));
                for (int64_t i = 0; i < nf; i++) {
                    const int64_t at = static_cast<int64_t>(failed[static_cast<size_t>(i)]) * k;
                    VG_HIP(hipMemcpyAsync(oid.ptr + at, fid.ptr + i * k, sizeof(uint32_t) * k, hipMemcpyDeviceToDevice, st));
                    VG_HIP(hipMemcpyAsync(osc.ptr + at, fsc.ptr + i * k, sizeof(float) * k, hipMemcpyDeviceToDevice, st));
                }
            }
            VG_TRY(oid.finish());
            VG_TRY(osc.finish());
            if (oid.on_host() || osc.on_host()) VG_HIP(hipStreamSynchronize(st));
            return VG_OK;
        }
        // the queries with a failed proof (ties at the k-th score, more than 4096 rows below a threshold): the exact kernel, one
        // workgroup per (slice, probe, query) that leaves at once unless its query is flagged
        for (int64_t q0 = 0; q0 < nq; q0 += 65535) {
            const int64_t cnt = std::min<int64_t>(65535, nq - q0);
            const dim3 grid(static_cast<unsigned>(sub), static_cast<unsigned>(np), static_cast<unsigned>(cnt));
            auto kern = mk.ptr ? (dot ? vg::probe_scan_f32_kernel<true, true> : vg::probe_scan_f32_kernel<false, true>)
                               : (dot ? vg::probe_scan_f32_kernel<true, false> : vg::probe_scan_f32_kernel<false, false>);
            VG_LAUNCH(kern, grid, dim3(256), 0, st, idx->d_vectors, idx->dim, q.ptr + q0 * idx->dim, probes + q0 * np, part_off, np, sub, k,
                      partial + q0 * lists * k, nullptr, mk.ptr ? mk.ptr + q0 * mask_stride : nullptr, mask_stride, qfail + q0);
        }
        VG_TRY(vg::launch_topk_merge(partial, nq, lists, k, desc, oid.ptr, osc.ptr, st, qfail));
        VG_TRY(oid.finish());
        VG_TRY(osc.finish());
        if (oid.on_host() || osc.on_host()) VG_HIP(hipStreamSynchronize(st));
        return VG_OK;
    }
    const size_t mq_lds = sizeof(float) * vg::kProbeQB * static_cast<size_t>(idx->dim) + 4 * 64 * sizeof(uint64_t) + 64;
    auto mq_kern = mk.ptr ? (dot ? vg::probe_scan_f32_mq_kernel<true, true> : vg::probe_scan_f32_mq_kernel<false, true>)
                          : (dot ? vg::probe_scan_f32_mq_kernel<true, false> : vg::probe_scan_f32_mq_kernel<false, false>);
    if (grouped_f32)
        VG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(mq_kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   static_cast<int>(mq_lds)));
    if (scan == VG_SCAN_PQ) VG_TRY(vg::launch_pq_build_table(idx->pq, q.ptr, nq, tables, true, st));
    if (paged) VG_HIP(hipMemsetAsync(one, 1, sizeof(int), st));
    // a wave keeps 64 keys: k > 64 comes in pages of 64 results, each page a scan of the probed ranges for the
    // keys after the previous page's last one
    for (int off = 0; off < k; off += 64) {
        const int kk = paged ? std::min(64, k - off) : k;
        const uint64_t *floor = off ? floor_keys : nullptr;
        if (grouped) {
            for (int64_t q0 = 0; q0 < nq; q0 += qchunk) {
                const int64_t cnt = std::min<int64_t>(qchunk, nq - q0);
                const int64_t cpairs = cnt * np;
                const unsigned gmax = static_cast<unsigned>(std::min<int64_t>(cpairs, cpairs / vg::kProbeQB + parts));
                const uint8_t *m0 = mk.ptr ? mk.ptr + q0 * mask_stride : nullptr;
                VG_HIP(hipMemsetAsync(gcounts, 0, sizeof(uint32_t) * (static_cast<size_t>(parts) + 1), st));
                VG_LAUNCH(vg::probe_group_kernel, dim3(1), dim3(1024), 0, st, probes + q0 * np, cpairs, parts, gcounts, gcursor,
                          gstart, pair_of, groups, ngroups);
                if (grouped_sq8) {
                    VG_TRY(vg::launch_probe_scan_sq8_grouped(idx, q.ptr + q0 * idx->dim, part_off, pair_of, groups, ngroups, gmax, np,
                                                             sub, kk, partial + q0 * lists * kk, floor ? floor + q0 : nullptr, m0,
                                                             mask_stride, st));
                    continue;
                }
                vg::ProfScope prof(idx->ctx, "flat_probe", st);
                VG_LAUNCH(mq_kern, dim3(static_cast<unsigned>(sub), gmax), dim3(256), mq_lds, st, idx->d_vectors, idx->dim,
                          q.ptr + q0 * idx->dim, part_off, pair_of, groups, ngroups, np, sub, kk,
                          partial + q0 * lists * kk, floor ? floor + q0 : nullptr, m0, mask_stride);
            }
        } else if (scan == VG_SCAN_F32) {
            for (int64_t q0 = 0; q0 < nq; q0 += 65535) {
                const int64_t cnt = std::min<int64_t>(65535, nq - q0);
                const dim3 grid(static_cast<unsigned>(sub), static_cast<unsigned>(np), static_cast<unsigned>(cnt));
                vg::ProfScope prof(idx->ctx, "flat_probe", st);
                auto kern = mk.ptr ? (dot ? vg::probe_scan_f32_kernel<true, true> : vg::probe_scan_f32_kernel<false, true>)
                                   : (dot ? vg::probe_scan_f32_kernel<true, false> : vg::probe_scan_f32_kernel<false, false>);
                VG_LAUNCH(kern, grid, dim3(256), 0, st, idx->d_vectors, idx->dim, q.ptr + q0 * idx->dim, probes + q0 * np,
                          part_off, np, sub, kk, partial + q0 * lists * kk, floor ? floor + q0 : nullptr,
                          mk.ptr ? mk.ptr + q0 * mask_stride : nullptr, mask_stride, nullptr);
            }
        } else if (scan == VG_SCAN_PQ) {
            VG_TRY(vg::launch_probe_scan_adc(idx, tables, probes, part_off, nq, np, split, kk, partial, floor, desc, mk.ptr,
                                             mask_stride, st));
        } else {
            VG_TRY(vg::launch_probe_scan_sq8(idx, q.ptr, probes, part_off, nq, np, sub, kk, partial, floor, mk.ptr, mask_stride, st));
        }
        if (!paged) {
            VG_TRY(vg::launch_topk_merge(partial, nq, lists, k, desc, oid.ptr, osc.ptr, st));
        } else {
            VG_TRY(vg::launch_topk_merge(partial, nq, lists, kk, desc, pid, psc, st));
            VG_TRY(vg::launch_page_patch(nq, k, off, kk, desc, one, pid, psc, oid.ptr, osc.ptr, floor_keys, st));
        }
    }
    VG_TRY(oid.finish());
    VG_TRY(osc.finish());
    if (oid.on_host() || osc.on_host()) VG_HIP(hipStreamSynchronize(st));
    return VG_OK;
}
