// k_brute.hip — the two EXHAUSTIVE paths of the HNSW index with the heap each one is written with:
//   hnsw.BruteSearch + scanSegment   (internal/hnsw/hnsw.go:2021-2101)  PriorityQueue(max): len < k -> PushItem,
//                                     else `d < top.Distance` -> PopItem + PushItem; results popped into res[len-1..0]
//   hnsw.searchBitmap + extraction   (hnsw.go:2240-2263, :1732-1751)    s.Candidates.TryPushBounded(k); popped, reversed
// Which ids survive a tie at the k-th distance, and the order in which equal distances leave the heap, are decided
// by the heap's layout — the history of every accepted item — so the heap is REPLAYED, operation by operation
// (vg_heap.hpp = searcher/queue.go), not replaced by a sort on (distance, id).
//
// Two kernels per query chunk:
//   brute_dist_kernel    every (query, row) distance as the index wraps it (L2, -dot, 0.5*L2: hnsw.go:2218-2238),
//                        squaredL2Avx512 / dotProductAvx512 summation order, 16 lanes per pair -> dist[q][row]
//   brute_replay_kernel  one workgroup per query streams its distance row in id order, 4096 rows per step: lanes flag
//                        rows that beat the heap's top AS IT STOOD at the start of the step (the top only falls, so
//                        the flagged rows are a superset of the accepted ones, in order), an ordered compaction puts
//                        them in an LDS list, and wave 0 replays the list through the real heap with the reference's
//                        test repeated against the live top.  Few rows are flagged once the heap is full
//                        (~k ln(n/k) in total on unordered data); adversarial orders degrade to the serial loop the
//                        CPU runs, never to a wrong answer.
// Also here: vg_debug_heap_replay, a one-wave kernel that replays a script of PriorityQueue operations on the device
// heap — how the reference's own queue tests (searcher/queue_test.go) are run against vg_heap.hpp.
#include "vg_device.hpp"
#include "vg_exact.hpp"
#include "vg_heap.hpp"
#include "vg_internal.hpp"
#include "vg_cand_replay.hpp"

#include <algorithm>

namespace vg {

constexpr int kBruteThreads = 1024;              // replay: 16 waves stream, wave 0 replays
constexpr int kBruteStep = kBruteThreads * 4;    // rows per step (one float4 per lane)
constexpr int kBruteMaxK = 1024;
constexpr int kBruteDistThreads = 256;           // 16 pair-groups per workgroup
constexpr int kBruteRowsPerBlock = 1024;

// dist[q][i] for the rows of block y; blockIdx.x = query (consecutive workgroups share a slice of rows through L2)
// STREAM: one query — every row is read exactly once: nontemporal loads (0.535 -> 0.467 ms per 1M x 768 = 6.57 TB/s);
// several queries with their own masks share rows through L2 and keep the cached loads
template <int METRIC, bool STREAM>
__global__ __launch_bounds__(kBruteDistThreads) void brute_dist_kernel(const float *__restrict__ base, int64_t n, int dim,
                                                                       const float *__restrict__ queries,
                                                                       const uint8_t *__restrict__ mask, int64_t mask_stride,
                                                                       float *__restrict__ dist)
{
    const Sub16 sub = Sub16::make(threadIdx.x);
    const int64_t q = blockIdx.x;
    const float *qv = queries + q * dim;
    const uint8_t *mq = mask ? mask + q * mask_stride : nullptr;
    float *dq = dist + q * n;
    for (int64_t r0 = static_cast<int64_t>(blockIdx.y) * kBruteRowsPerBlock; r0 < n;
         r0 += static_cast<int64_t>(gridDim.y) * kBruteRowsPerBlock) {
        const int64_t r1 = r0 + kBruteRowsPerBlock < n ? r0 + kBruteRowsPerBlock : n;
        for (int64_t i = r0 + (threadIdx.x >> 4); i < r1; i += kBruteDistThreads / 16) {
            if (!mask_bit(mq, i)) continue;  // never read by the replay
            const float *row = base + i * dim;
            float d;
            if (METRIC == kMetricDot) {
                d = -exact_pair16<true, kPair, STREAM>(row, qv, dim, sub);
            } else {
                d = exact_pair16<false, kPair, STREAM>(row, qv, dim, sub);
                if (METRIC == kMetricCos) d = 0.5f * d;
            }
            if ((threadIdx.x & 15) == 0) dq[i] = d;
        }
    }
}

// The same distances for a BLOCK of queries: a 16-lane group keeps two rows in registers and scores them against up to
// kBruteQB queries held in LDS (exact_rowregs16x2: every LDS read of a query piece serves both rows) — a row is read
// from HBM once per query block instead of once per query through L2, with a quarter of the instructions per pair
// (tools/brute_time.py, 256 queries x 1M x 768: 44 ms with the kernel above).  dim % 4 == 0, dim <= 1024; `mask` is
// the batch's shared mask or none (per-query masks take the kernel above).
constexpr int kBruteQB = 16;
template <int METRIC>
__global__ __launch_bounds__(kBruteDistThreads) void brute_dist_mq_kernel(const float *__restrict__ base, int64_t n, int dim,
                                                                          const float *__restrict__ queries, int nq,
                                                                          const uint8_t *__restrict__ mask,
                                                                          float *__restrict__ dist, int qblocks, int slices)
{
    extern __shared__ float brute_q[];  // kBruteQB * dim
    // XCD-aware order (as the scans): workgroup b runs on XCD b % 8; the query blocks of ONE row slice are consecutive
    // workgroups of one XCD and walk the slice together, so a row crosses the fabric once per slice, not once per query
    // block (with blockIdx = (query block, row block) the 16 query blocks of a tile sat on all 8 XCDs)
    const int b = blockIdx.x;
    const int xcd = b & 7, o = b >> 3;
    const int qb = o % qblocks;
    const int sl = (o / qblocks) * 8 + xcd;
    if (sl >= slices) return;
    const int q0 = qb * kBruteQB;
    const int qn = nq - q0 < kBruteQB ? nq - q0 : kBruteQB;
    for (int t = threadIdx.x; t < qn * dim; t += kBruteDistThreads) brute_q[t] = queries[static_cast<int64_t>(q0) * dim + t];
    __syncthreads();
    const Sub16 sub = Sub16::make(threadIdx.x);
    const int nblk = dim >> 6;
    const int grp = threadIdx.x >> 4;  // 16 groups, two rows each
    constexpr bool DOT = METRIC == kMetricDot;
    const int64_t rs = (n * sl / slices) & ~int64_t(31), re = sl + 1 == slices ? n : ((n * (sl + 1) / slices) & ~int64_t(31));
    for (int64_t r0 = rs; r0 < re; r0 += 32) {
        const int64_t ia = r0 + 2 * grp, ib = ia + 1;
        const bool la = ia < re && mask_bit(mask, ia), lb = ib < re && mask_bit(mask, ib);
        if (!la && !lb) continue;  // (uniform over the 16 lanes of the group)
        const float *rowa = base + (la ? ia : (lb ? ib : 0)) * dim;
        const float *rowb = base + (lb ? ib : (la ? ia : 0)) * dim;
        float4 ra[16], rb[16];
        const float4 *a4 = reinterpret_cast<const float4 *>(rowa) + sub.f4;
        const float4 *b4 = reinterpret_cast<const float4 *>(rowb) + sub.f4;
#pragma unroll
        for (int e = 0; e < 16; e++)
            if (e < nblk) {
                ra[e] = load_stream(a4 + e * 16);
                rb[e] = load_stream(b4 + e * 16);
            }
        for (int qi = 0; qi < qn; qi++) {
            float va, vb;
            exact_rowregs16x2<DOT>(ra, rb, nblk, rowa, rowb, brute_q + static_cast<size_t>(qi) * dim, dim, sub, va, vb);
            if (DOT) {
                va = -va;
                vb = -vb;
            } else if (METRIC == kMetricCos) {
                va = 0.5f * va;
                vb = 0.5f * vb;
            }
            if ((threadIdx.x & 15) == 0) {
                float *dq = dist + static_cast<int64_t>(q0 + qi) * n;
                if (la) dq[ia] = va;
                if (lb) dq[ib] = vb;
            }
        }
    }
}

// one accepted-or-not decision of the reference's loop, uniform over wave 0
template <int MODE>
__device__ __forceinline__ void brute_offer(HItem *heap, int &len, int k, HItem it)
{
    if (len < k) {  // scanSegment hnsw.go:2089-2091 / TryPushBounded queue.go:192-196
        heap_push<true>(heap, len, it);
        return;
    }
    const float top = heap_get(heap, 0).dist;
    if (MODE == VG_BRUTE_SCAN) {
        if (it.dist < top) {  // hnsw.go:2093-2097: PopItem, then PushItem
            (void)heap_pop<true>(heap, len);
            heap_push<true>(heap, len, it);
        }
    } else {
        if (it.dist >= top) return;  // queue.go:199-203
        heap_sift_down_f32<true>(heap, len, 0, it);  // :211-213 replace the top, sift down
    }
}

template <int MODE>
__global__ __launch_bounds__(kBruteThreads) void brute_replay_kernel(const float *__restrict__ dist, int64_t n,
                                                                     const uint8_t *__restrict__ mask, int64_t mask_stride,
                                                                     int k, uint32_t *__restrict__ ids,
                                                                     float *__restrict__ scores)
{
    extern __shared__ uint64_t brute_lds[];
    HItem *heap = reinterpret_cast<HItem *>(brute_lds);           // k + 4 items
    HItem *list = heap + ((k + 4 + 3) & ~3);                        // kBruteStep items
    __shared__ int wave_cnt[2][kBruteThreads / 64];                 // (alternating: one barrier per step on the fast path)
    __shared__ int s_len;
    __shared__ float s_top;
    __shared__ uint32_t s_hit[3];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t q = blockIdx.x;
    const float *dq = dist + q * n;
    const uint8_t *mq = mask ? mask + q * mask_stride : nullptr;
    if (tid == 0) {
        s_len = 0;
        s_top = 0.0f;
        s_hit[0] = s_hit[1] = s_hit[2] = 0u;
    }
    __syncthreads();
    int len = 0;  // wave 0's copy is the live one
    int64_t step_no = 0;
    const bool vec = (n & 3) == 0;  // whole float4s, 16-byte aligned (each query's row starts at q * n floats)
    constexpr int kSub = 4;         // 4096-row sub-steps per super-step: one barrier per 16384 rows when nothing qualifies
    // a super-step's distances are requested one super-step ahead (a single workgroup streams its row: an exposed HBM
    // round trip per step was most of a one-query call); masked-out rows hold whatever the scratch held: never used
    auto fetch = [&](int64_t base, float (&d)[kSub][4], uint32_t &in) {
        in = 0;
        if (vec && base + static_cast<int64_t>(kSub) * kBruteStep <= n) {
            // a whole super-step inside the row (uniform): straight-line loads, nothing waits between them
            float4 v[kSub];
            uint32_t mb[kSub];
#pragma unroll
            for (int c = 0; c < kSub; c++) {
                const int64_t i0 = base + static_cast<int64_t>(c) * kBruteStep + static_cast<int64_t>(tid) * 4;
                v[c] = *reinterpret_cast<const float4 *>(dq + i0);
                mb[c] = mq ? mq[i0 >> 3] : 0xFFu;  // i0 is a multiple of 4: its four mask bits are one nibble of byte i0 / 8
            }
#pragma unroll
            for (int c = 0; c < kSub; c++) {
                const int sh = static_cast<int>((static_cast<int64_t>(c) * kBruteStep + static_cast<int64_t>(tid) * 4 + base) & 7);
                d[c][0] = v[c].x, d[c][1] = v[c].y, d[c][2] = v[c].z, d[c][3] = v[c].w;
                in |= ((mb[c] >> sh) & 15u) << (4 * c);
            }
            return;
        }
#pragma unroll
        for (int c = 0; c < kSub; c++) {  // the row's last super-step, or a row that is not whole float4s
            const int64_t i0 = base + static_cast<int64_t>(c) * kBruteStep + static_cast<int64_t>(tid) * 4;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const bool ok = i0 + e < n && mask_bit(mq, i0 + e);
                d[c][e] = ok ? dq[i0 + e] : 0.0f;
                in |= static_cast<uint32_t>(ok) << (4 * c + e);
            }
        }
    };
    // Two super-steps of distances in flight (64 KiB each), in two FIXED register sets that are consumed and refilled in
    // place (a rotating set costs register copies, and a copy of a set still in flight waits for it): one workgroup
    // streams its query's row, and with one request outstanding every super-step paid a whole memory round trip.
    constexpr int64_t kSuper = static_cast<int64_t>(kSub) * kBruteStep;
    float ra[kSub][4], rb[kSub][4];
    uint32_t ia = 0, ib = 0;
    fetch(0, ra, ia);
    if (kSuper < n) fetch(kSuper, rb, ib);
    int par = 0;
    // one super-step: `dd` / `in` are its distances and membership bits, `sbase` its first row
    auto super_step = [&](const float (&dd)[kSub][4], const uint32_t in, const int64_t sbase) {
        uint32_t hit;  // bit c: some row of sub-step c beats the top as it stands at the start of the super-step
        {
            const int cur_len = s_len;
            const float top = s_top;
            uint32_t mine = 0;
#pragma unroll
            for (int c = 0; c < kSub; c++)
#pragma unroll
                for (int e = 0; e < 4; e++)
                    if (((in >> (4 * c + e)) & 1u) && (cur_len < k || dd[c][e] < top)) mine |= 1u << c;
            // three alternating words: the one used two super-steps ago is cleared for the next one (every wave has
            // passed the previous barrier, so it has read that word; nobody adds to it before this super-step's barrier)
            const int slot = static_cast<int>(step_no % 3);
            if (tid == 0) s_hit[(slot + 1) % 3] = 0u;
            if (mine) atomicOr(&s_hit[slot], mine);
            __syncthreads();
            hit = s_hit[slot];
            step_no++;
            if (hit == 0) return;  // nothing of the 16384 rows can enter the heap: one barrier and on (the common case once it is full)
#if defined(VG_BRUTE_PROBE) && VG_BRUTE_PROBE == 1
            return;  // stage probe: streaming + the one barrier only
#endif
        }
#pragma unroll
        for (int c = 0; c < kSub; c++) {
            if (!((hit >> c) & 1u)) continue;  // uniform: no candidate in this sub-step (the top only falls meanwhile)
            par ^= 1;
            const int64_t base = sbase + static_cast<int64_t>(c) * kBruteStep;
            if (base >= n) break;  // uniform
            const int cur_len = s_len;
            const float top = s_top;
            const int64_t i0 = base + static_cast<int64_t>(tid) * 4;
            float d[4];
            bool f[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                d[e] = dd[c][e];
                f[e] = ((in >> (4 * c + e)) & 1u) && (cur_len < k || d[e] < top);
            }
            const int cnt = int(f[0]) + int(f[1]) + int(f[2]) + int(f[3]);
            // ordered exclusive prefix over the workgroup: row order = thread order
            int incl = cnt;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int y = __shfl_up(incl, off);
                if (lane >= off) incl += y;
            }
            if (lane == 63) wave_cnt[par][wave] = incl;
            __syncthreads();
            int before = 0, total = 0;
#pragma unroll
            for (int w = 0; w < kBruteThreads / 64; w++) {
                const int cw = wave_cnt[par][w];
                if (w < wave) before += cw;
                total += cw;
            }
            if (total == 0) continue;  // uniform: nothing in this sub-step can enter the heap (the other counter set is next)
            int pos = before + incl - cnt;
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (f[e]) heap_store(list + pos++, HItem{static_cast<uint32_t>(i0 + e), d[e]});
            __syncthreads();
#if defined(VG_BRUTE_PROBE) && VG_BRUTE_PROBE == 2
            if (false) {
#else
            if (wave == 0) {
#endif
                // the list 64 items at a time (one LDS read per lane), the reference's test against the LIVE top in a
                // register: only items that pass it reach the heap, in list order — a sub-step whose rows all beat a stale
                // top (the first one: the heap is still filling) costs one ballot per 64 rows, not a heap round trip per row
                float top_live = len >= k ? heap_get(heap, 0).dist : 0.0f;
                for (int j0 = 0; j0 < total; j0 += 64) {
                    const int j = j0 + lane;
                    const HItem mine = j < total ? heap_load(list + j) : HItem{0u, 0.0f};
                    uint64_t m = __ballot(j < total && (len < k || mine.dist < top_live));
                    while (m) {
                        const int b = __builtin_ctzll(m);
                        m &= m - 1;
                        const HItem x{heap_readlane(mine.node, b), __uint_as_float(heap_readlane(__float_as_uint(mine.dist), b))};
                        if (len < k) {  // scanSegment hnsw.go:2089-2091 / TryPushBounded queue.go:192-196
                            heap_push<true>(heap, len, x);
                            if (len == k) top_live = heap_get(heap, 0).dist;
                        } else if (x.dist < top_live) {  // hnsw.go:2093-2097 / queue.go:199-213 (`>=` rejects)
                            brute_offer<MODE>(heap, len, k, x);
                            top_live = heap_get(heap, 0).dist;
                        }
                    }
                }
                if (lane == 0) {
                    s_len = len;
                    s_top = len > 0 ? heap_get(heap, 0).dist : 0.0f;
                }
            }
            __syncthreads();
        }
    };
    for (int64_t sbase = 0; sbase < n; sbase += 2 * kSuper) {
        super_step(ra, ia, sbase);
        if (sbase + 2 * kSuper < n) fetch(sbase + 2 * kSuper, ra, ia);
        if (sbase + kSuper < n) {
            super_step(rb, ib, sbase + kSuper);
            if (sbase + 3 * kSuper < n) fetch(sbase + 3 * kSuper, rb, ib);
        }
    }
    if (wave == 0) {
        // BruteSearch hnsw.go:2067-2071: res[i] = PopItem() for i = len-1 .. 0 (extraction :1738-1751 pops, then reverses)
        const int nres = len;
        for (int i = nres - 1; i >= 0; i--) {
            const HItem it = heap_pop<true>(heap, len);
            if (lane == 0) {
                ids[q * k + i] = it.node;
                scores[q * k + i] = it.dist;
            }
        }
        for (int i = nres + lane; i < k; i += 64) {
            ids[q * k + i] = VG_INVALID_ID;
            scores[q * k + i] = INFINITY;
        }
    }
}

// ---- searcher.PriorityQueue script replay (test entry point) ---------------------------------------------------
// ops[i] = {op, node, dist bits, arg}; out[i] = {flag, node, dist bits}
template <bool MAX, bool UK>
__device__ void heap_script(HItem *heap, const int32_t *__restrict__ ops, int n_ops, int32_t *__restrict__ out, int &len)
{
    const int lane = heap_lane();
    for (int i = 0; i < n_ops; i++) {
        const int op = ops[4 * i], arg = ops[4 * i + 3];
        const HItem it{static_cast<uint32_t>(ops[4 * i + 1]), __int_as_float(ops[4 * i + 2])};
        int flag = 0;
        HItem res{0u, 0.0f};
        switch (op) {
        case VG_HEAP_PUSH:  // PushItem queue.go:59-62
            heap_push<MAX>(heap, len, it);
            flag = 1;
            break;
        case VG_HEAP_POP:  // PopItem :113-128
            if (len > 0) {
                res = heap_pop<MAX, UK>(heap, len);
                flag = 1;
            }
            break;
        case VG_HEAP_PUSH_BOUNDED:  // PushItemBounded :67-92
            if (len < arg) {
                heap_push<MAX>(heap, len, it);
                flag = 1;
            } else if (len > 0) {
                const float top = heap_get(heap, 0).dist;
                if (MAX ? (it.dist < top) : (it.dist > top)) {
                    heap_sift_down<MAX, UK>(heap, len, 0, it);
                    flag = 1;
                }
            }
            break;
        case VG_HEAP_TRY_PUSH_BOUNDED:  // TryPushBounded :190-215
            if (len < arg) {
                heap_push<MAX>(heap, len, it);
                flag = 1;
            } else if (len > 0) {
                const float top = heap_get(heap, 0).dist;
                if (!(MAX ? (it.dist >= top) : (it.dist <= top))) {
                    heap_sift_down<MAX, UK>(heap, len, 0, it);
                    flag = 1;
                }
            }
            break;
        case VG_HEAP_TOP:  // TopItem :37-42
            if (len > 0) {
                res = heap_get(heap, 0);
                flag = 1;
            }
            break;
        case VG_HEAP_MIN_ITEM:  // MinItem :46-57 (first strict minimum in array order)
            if (len > 0) {
                res = heap_get(heap, 0);
                for (int j = 1; j < len; j++) {
                    const HItem c = heap_get(heap, j);
                    if (c.dist < res.dist) res = c;
                }
                flag = 1;
            }
            break;
        case VG_HEAP_RESET:  // Reset :32-34
            len = 0;
            flag = 1;
            break;
        case VG_HEAP_LEN:
            flag = len;
            break;
        default:
            flag = -1;
        }
        if (lane == 0) {
            out[3 * i] = flag;
            out[3 * i + 1] = static_cast<int32_t>(res.node);
            out[3 * i + 2] = __float_as_int(res.dist);
        }
    }
}

__global__ __launch_bounds__(64) void heap_replay_kernel(int is_max, int uk, const int32_t *__restrict__ ops, int n_ops,
                                                         int32_t *__restrict__ out, int32_t *__restrict__ final_len,
                                                         uint64_t *__restrict__ final_items, int cap)
{
    extern __shared__ uint64_t replay_lds[];
    HItem *heap = reinterpret_cast<HItem *>(replay_lds);
    int len = 0;
    if (is_max) {
        if (uk)
            heap_script<true, true>(heap, ops, n_ops, out, len);
        else
            heap_script<true, false>(heap, ops, n_ops, out, len);
    } else {
        if (uk)
            heap_script<false, true>(heap, ops, n_ops, out, len);
        else
            heap_script<false, false>(heap, ops, n_ops, out, len);
    }
    const int lane = heap_lane();
    if (lane == 0) *final_len = len;
    for (int i = lane; i < len && i < cap; i += 64) final_items[i] = heap_load_u64(heap, i);
}

// set bits of a mask (bytes [0, total)), summed into *count: how selective a batch's masks are decides its path
__global__ __launch_bounds__(256) void mask_popcount_kernel(const uint8_t *__restrict__ mask, int64_t total,
                                                            unsigned long long *__restrict__ count)
{
    __shared__ unsigned long long part[4];
    unsigned long long mine = 0;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += static_cast<int64_t>(gridDim.x) * blockDim.x)
        mine += __popc(static_cast<unsigned>(mask[i]));
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(count, part[0] + part[1] + part[2] + part[3]);
}

// The unmasked scan from vg_search_flat's answer: the PriorityQueue's final SET is the k smallest distances and its pops
// leave in ascending distance — the heap's layout (the history of accepted rows) only decides between EQUAL distances.
// `fid` / `fsc` = the k + 1 best rows by (score, id) with their exact scores (the same squaredL2Avx512 /
// dotProductAvx512 values brute_dist_kernel computes).  If the k + 1 values are finite and pairwise different, rows
// 0 .. k-1 ARE the reference's answer (Dot: the index's distance is -dot); any tie or NaN sends the query to the replay.
// conv: 0 = L2 (the score), 1 = Dot (-score), 2 = Cosine (0.5 * the squared L2 the flat search was asked for: the halving is
// monotone, and exact except at the bottom of the range — so the ties are looked for among the HALVED values)
__global__ void brute_from_flat_kernel(const uint32_t *__restrict__ fid, const float *__restrict__ fsc, int64_t nq, int k, int conv,
                                       uint32_t *__restrict__ ids, float *__restrict__ scores, int32_t *__restrict__ redo)
{
    const int64_t q = blockIdx.x;
    const int lane = threadIdx.x;  // 64
    const bool dot = conv == 1;
    auto as_index = [&](float s) { return conv == 2 ? 0.5f * s : s; };
    bool bad = false;
    for (int i = lane; i <= k; i += 64) {
        const uint32_t id = fid[q * (k + 1) + i];
        const float s = as_index(fsc[q * (k + 1) + i]);
        bad |= id == VG_INVALID_ID || !(fabsf(s) <= 3.40282346638528859811704183484516925440e+38f);
        if (i > 0) {
            const float p = as_index(fsc[q * (k + 1) + i - 1]);
            bad |= __float_as_uint(p) == __float_as_uint(s) || p == s;
        }
    }
    const bool any_bad = __ballot(bad) != 0;
    if (lane == 0) redo[q] = any_bad ? 1 : 0;
    if (any_bad) return;
    for (int i = lane; i < k; i += 64) {
        ids[q * k + i] = fid[q * (k + 1) + i];
        const float s = as_index(fsc[q * (k + 1) + i]);
        scores[q * k + i] = dot ? -s : s;
    }
}

}  // namespace vg

namespace vg {
// vg_cand_replay.hpp's heap for the two exhaustive paths of the HNSW index: searcher.PriorityQueue(max) driven as scanSegment
// (hnsw.go:2089-2097) / TryPushBounded (queue.go:192-213) drive it — brute_offer above, whose comparisons are the reference's
// floats.  brute_replay_kernel flags rows against the top as it stood at the start of a step, which is sound while the top
// only falls; with a NaN inside the heap its order is broken and a later top can be LARGER than an earlier one, so queries
// whose distances may hold a NaN are decided row by row against the live top here.
template <int MODE>
struct BruteHeapPolicy {
    __device__ static bool accepts(const CItem x, int len, int k, const CItem root, bool)
    {
        return len < k || (MODE == VG_BRUTE_SCAN ? x.score < root.score : !(x.score >= root.score));
    }
    __device__ static void offer(CItem *h, int &len, int k, const CItem x, bool)
    {
        brute_offer<MODE>(reinterpret_cast<HItem *>(h), len, k, HItem{x.row, x.score});  // (both are {row, score bits} in 8 bytes)
    }
    __device__ static CItem pop(CItem *h, int &len, bool)
    {
        const HItem it = heap_pop<true>(reinterpret_cast<HItem *>(h), len);
        return CItem{it.dist, it.node};
    }
};
static int32_t brute_nan_replay(vg_index *idx, const float *d_queries, int64_t nq, int k, int mode, const uint8_t *d_mask, int64_t mask_stride,
                                uint32_t *d_ids, float *d_scores, hipStream_t st)
{
    if (idx->n == 0) return VG_OK;
    const bool dot = idx->metric == VG_METRIC_DOT;
    const FlatF32Scorer sc{idx->d_vectors, idx->d_norm_max + 1, idx->dim, dot, dot ? 1 : idx->metric == VG_METRIC_COSINE ? 2 : 0};
    if (mode == VG_BRUTE_SCAN)
        return launch_cand_replay<BruteHeapPolicy<VG_BRUTE_SCAN>>(sc, d_queries, idx->dim, idx->n, nq, k, false, d_mask, mask_stride, d_ids, d_scores, st);
    return launch_cand_replay<BruteHeapPolicy<VG_BRUTE_BITMAP>>(sc, d_queries, idx->dim, idx->n, nq, k, false, d_mask, mask_stride, d_ids, d_scores, st);
}
}  // namespace vg

static int32_t brute_impl(vg_index *idx, const float *queries, int64_t nq, int32_t k, int32_t mode, const uint8_t *mask,
                          int64_t mask_stride, uint32_t *ids, float *scores, void *stream);
namespace vg {
int32_t flat_search_masked(vg_index *idx, const float *queries, int64_t nq, int32_t k, const uint8_t *mask, int64_t mask_stride,
                           uint32_t *ids, float *scores, void *stream, bool l2_scores = false, bool cand_replay = true);
}

VG_API int32_t vg_search_hnsw_brute(vg_index *idx, const float *queries, int64_t nq, int32_t k, int32_t mode,
                                    const uint8_t *mask, int64_t mask_stride, uint32_t *ids, float *scores, void *stream)
{
    // Unmasked queries under L2 / Dot: every (query, row) distance is a 1M x dim product per query — the
    // flat search's MFMA nomination + exact re-score + proof answers "the k + 1 best by (score, id)" at 11 ms per 1024
    // queries where brute_dist_mq_kernel's vector-ALU pass takes 16 ms per 256; brute_from_flat_kernel turns that answer
    // into the reference's when no two of the k + 1 distances are equal, and the few queries with a tie are replayed
    // as before.
    // (one query too: its exact scan + merge is 0.48 ms at 1M x 768 where the distance pass + heap replay took 0.74)
    // With a mask (rows that take part): the same through the masked nomination (vg::flat_search_masked, k_flat.hip) for
    // batches — a query whose mask leaves fewer than k + 1 rows, or a tie, is replayed like the others.
    const int64_t mask_bytes_fast = idx ? (idx->n + 7) / 8 : 0;
    const bool mask_ok = mask == nullptr || (nq >= 8 && (mask_stride == 0 || mask_stride >= mask_bytes_fast));
    const bool fast = idx && mask_ok && nq >= 1 && k >= 1 && queries && ids && scores && idx->d_vectors &&
                      (idx->metric == VG_METRIC_L2 || idx->metric == VG_METRIC_DOT || idx->metric == VG_METRIC_COSINE) &&
                      static_cast<int64_t>(k) + 1 <= idx->n &&
                      k + 1 <= 512 /* vg_search_flat's kFlatMaxK */ && k <= vg::kBruteMaxK && (mode == VG_BRUTE_SCAN || mode == VG_BRUTE_BITMAP) &&
                      !vg::hook(vg::kHookBruteNoFlat);
    if (!fast) return brute_impl(idx, queries, nq, k, mode, mask, mask_stride, ids, scores, stream);
    VG_HIP(hipSetDevice(idx->ctx->device));
    hipStream_t st = vg::pick_stream(idx->ctx, stream);
    vg::DevIn<float> q;
    vg::DevOut<uint32_t> oid;
    vg::DevOut<float> osc;
    VG_TRY(q.init(queries, static_cast<size_t>(nq) * idx->dim, st));
    VG_TRY(oid.init(ids, static_cast<size_t>(nq) * k, st));
    VG_TRY(osc.init(scores, static_cast<size_t>(nq) * k, st));
    vg::DevTmp<uint32_t> fid;
    vg::DevTmp<float> fsc;
    vg::DevTmp<int32_t> redo;
    VG_TRY(fid.init(static_cast<size_t>(nq) * (k + 1), st));
    VG_TRY(fsc.init(static_cast<size_t>(nq) * (k + 1), st));
    VG_TRY(redo.init(static_cast<size_t>(nq), st));
    vg::DevIn<uint8_t> mk;
    const int64_t mask_total = mask ? (mask_stride ? (nq - 1) * mask_stride + mask_bytes_fast : mask_bytes_fast) : 0;
    VG_TRY(mk.init(mask, static_cast<size_t>(mask_total), st));
    if (mk.ptr) {
        // The replay's distance pass reads only the rows that take part: below ~2 % of the rows it is the faster form
        // (1M x 768, per 1024 queries: replay 12 ms at 1 %, 23 at 3 %, 68 at 50 %; this form 12 - 18 whatever the fraction)
        vg::DevTmp<unsigned long long> cnt;
        VG_TRY(cnt.init(1, st));
        VG_HIP(hipMemsetAsync(cnt.ptr, 0, sizeof(unsigned long long), st));
        VG_LAUNCH(vg::mask_popcount_kernel, dim3(static_cast<unsigned>(std::min<int64_t>(1024, (mask_total + 255) / 256))), dim3(256), 0, st,
                  mk.ptr, mask_total, cnt.ptr);
        unsigned long long set_bits = 0;
        VG_HIP(hipMemcpyAsync(&set_bits, cnt.ptr, sizeof(set_bits), hipMemcpyDeviceToHost, st));
        VG_HIP(hipStreamSynchronize(st));
        const double masks = mask_stride ? static_cast<double>(nq) : 1.0;
        if (static_cast<double>(set_bits) < 0.02 * masks * static_cast<double>(idx->n))
            return brute_impl(idx, q.ptr, nq, k, mode, mk.ptr, mask_stride, ids, scores, stream);
    }
    {
        vg::ProfScope prof(idx->ctx, "hnsw_brute_dist", st);  // (the distance work of this form)
        // (Cosine: the index's distance is 0.5 * squared L2 of the normalised rows — the flat search is asked for L2 scores)
        VG_TRY(vg::flat_search_masked(idx, q.ptr, nq, k + 1, mk.ptr, mask_stride, fid.ptr, fsc.ptr, st, idx->metric == VG_METRIC_COSINE, false));
    }
    VG_LAUNCH(vg::brute_from_flat_kernel, dim3(static_cast<unsigned>(nq)), dim3(64), 0, st, fid.ptr, fsc.ptr, nq, k,
              idx->metric == VG_METRIC_DOT ? 1 : idx->metric == VG_METRIC_COSINE ? 2 : 0, oid.ptr, osc.ptr, redo.ptr);
    std::vector<int32_t> h(static_cast<size_t>(nq));
    VG_HIP(hipMemcpyAsync(h.data(), redo.ptr, sizeof(int32_t) * static_cast<size_t>(nq), hipMemcpyDeviceToHost, st));
    VG_HIP(hipStreamSynchronize(st));
    int64_t nredo = 0;
    for (int64_t i = 0; i < nq; i++) nredo += h[static_cast<size_t>(i)] != 0;
    if (nredo * 4 > nq)  // (masks that leave fewer than k + 1 rows to most queries: one batched pass instead of nq small ones)
        return brute_impl(idx, q.ptr, nq, k, mode, mk.ptr, mask_stride, ids, scores, stream);
    for (int64_t i = 0; i < nq; i++)  // ties / NaN: the heap's history decides — replayed one query at a time (rare)
        if (h[static_cast<size_t>(i)])
            VG_TRY(brute_impl(idx, q.ptr + i * idx->dim, 1, k, mode, mk.ptr ? mk.ptr + i * mask_stride : nullptr, 0, oid.ptr + i * k,
                              osc.ptr + i * k, st));
    VG_TRY(vg::brute_nan_replay(idx, q.ptr, nq, k, mode, mk.ptr, mask_stride, oid.ptr, osc.ptr, st));  // queries whose distances may hold a NaN
    VG_TRY(oid.finish());
    VG_TRY(osc.finish());
    return VG_OK;
}

static int32_t brute_impl(vg_index *idx, const float *queries, int64_t nq, int32_t k, int32_t mode,
                                    const uint8_t *mask, int64_t mask_stride, uint32_t *ids, float *scores, void *stream)
{
    VG_CHECK(idx, VG_ERR_INVALID_ARG, "vg_search_hnsw_brute: NULL index");
    VG_CHECK(nq >= 0 && k >= 0, VG_ERR_INVALID_ARG, "vg_search_hnsw_brute: negative nq or k");
    VG_CHECK(mode == VG_BRUTE_SCAN || mode == VG_BRUTE_BITMAP, VG_ERR_INVALID_ARG, "vg_search_hnsw_brute: unknown mode %d", mode);
    if (nq == 0 || k == 0) return VG_OK;
    VG_CHECK(idx->metric != VG_METRIC_HAMMING, VG_ERR_UNSUPPORTED, "unsupported metric for float32: Hamming");
    VG_CHECK(k <= vg::kBruteMaxK, VG_ERR_UNSUPPORTED, "vg_search_hnsw_brute: k=%d exceeds %d", k, vg::kBruteMaxK);
    VG_CHECK(queries && ids && scores, VG_ERR_INVALID_ARG, "vg_search_hnsw_brute: NULL buffer");
    VG_CHECK(idx->n == 0 || idx->d_vectors, VG_ERR_NOT_READY, "vg_search_hnsw_brute: index has no fp32 vectors");
    const int64_t mask_bytes = (idx->n + 7) / 8;
    VG_CHECK(mask == nullptr || mask_stride == 0 || mask_stride >= mask_bytes, VG_ERR_INVALID_ARG,
             "vg_search_hnsw_brute: mask_stride %lld is shorter than a mask (%lld bytes)", static_cast<long long>(mask_stride),
             static_cast<long long>(mask_bytes));
    VG_HIP(hipSetDevice(idx->ctx->device));
    hipStream_t st = vg::pick_stream(idx->ctx, stream);
    vg::DevIn<float> q;
    vg::DevIn<uint8_t> mk;
    vg::DevOut<uint32_t> oid;
    vg::DevOut<float> osc;
    VG_TRY(q.init(queries, static_cast<size_t>(nq) * idx->dim, st));
    VG_TRY(mk.init(mask, mask ? static_cast<size_t>(mask_stride ? (nq - 1) * mask_stride + mask_bytes : mask_bytes) : 0, st));
    VG_TRY(oid.init(ids, static_cast<size_t>(nq) * k, st));
    VG_TRY(osc.init(scores, static_cast<size_t>(nq) * k, st));
    const int64_t n = idx->n;
    // dist[q][row] of one chunk of queries.  The query-blocked kernel gets its row reuse from kBruteQB = 16 queries per
    // workgroup, so a chunk of thousands of queries buys nothing: 2 GiB of distances (512 queries at 1M rows), but at
    // least one block of 16 queries while that stays within 1/16 of the device's memory.  The buffer comes from the
    // scratch cache (idle blocks go back to the driver when an allocation fails, and with the context), not from the
    // grow-only arena — r04 took up to 16 GiB from the arena and kept it until the context was destroyed.  When the
    // index leaves no room, the chunk is halved until the allocation succeeds.
    const int64_t row_bytes = std::max<int64_t>(n, 1) * 4;
    const int64_t cap16 = std::min<int64_t>(int64_t(16) << 30, std::max<int64_t>(int64_t(1) << 30, idx->ctx->hbm_bytes / 16));
    int64_t chunk = std::max<int64_t>(1, (int64_t(2) << 30) / row_bytes);
    if (chunk < vg::kBruteQB) chunk = std::max<int64_t>(1, std::min<int64_t>(vg::kBruteQB, cap16 / row_bytes));
    chunk = std::min<int64_t>(std::min(chunk, nq), 65535);
    vg::DevTmp<float> dist_buf;
    for (;;) {
        const int32_t rc = dist_buf.init(static_cast<size_t>(chunk) * static_cast<size_t>(std::max<int64_t>(n, 1)), st);
        if (rc == VG_OK) break;
        VG_CHECK(chunk > 1, rc, "vg_search_hnsw_brute: no room for one query's %lld distances: %s", static_cast<long long>(n),
                 vg_last_error());
        chunk = (chunk + 1) / 2;
    }
    float *dist = dist_buf.ptr;
    const size_t lds = (static_cast<size_t>((k + 4 + 3) & ~3) + vg::kBruteStep) * sizeof(vg::HItem);
    auto replay = mode == VG_BRUTE_SCAN ? vg::brute_replay_kernel<VG_BRUTE_SCAN> : vg::brute_replay_kernel<VG_BRUTE_BITMAP>;
    VG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(replay), hipFuncAttributeMaxDynamicSharedMemorySize,
                               static_cast<int>(lds)));
    const unsigned row_blocks = static_cast<unsigned>(std::min<int64_t>(std::max<int64_t>((n + vg::kBruteRowsPerBlock - 1) /
                                                                                          vg::kBruteRowsPerBlock, 1), 65535));
    for (int64_t q0 = 0; q0 < nq; q0 += chunk) {
        const int64_t cnt = std::min(chunk, nq - q0);
        const uint8_t *m0 = mk.ptr ? mk.ptr + q0 * mask_stride : nullptr;
        if (n > 0) {
            vg::ProfScope prof(idx->ctx, "hnsw_brute_dist", st);
            const bool mq = cnt >= 2 && idx->dim % 4 == 0 && idx->dim <= 1024 && (m0 == nullptr || mask_stride == 0);
            if (mq) {  // query-blocked: rows read once per 16 queries
                const int qblocks = static_cast<int>((cnt + vg::kBruteQB - 1) / vg::kBruteQB);
                // ~8 workgroups per CU in all; slices in whole groups of 8 (one per XCD), at least 32 rows each
                int64_t slices = (int64_t(8) * std::max(idx->ctx->compute_units, 1) + qblocks - 1) / qblocks;
                slices = std::min<int64_t>(((slices + 7) / 8) * 8, std::max<int64_t>(8, ((n + 31) / 32 / 8) * 8));
                const size_t qlds = static_cast<size_t>(vg::kBruteQB) * idx->dim * sizeof(float);
                const dim3 grid(static_cast<unsigned>(qblocks * slices)), block(vg::kBruteDistThreads);
                auto launch = [&](auto kern) -> int32_t {
                    VG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               static_cast<int>(qlds)));
                    VG_LAUNCH(kern, grid, block, qlds, st, idx->d_vectors, n, idx->dim, q.ptr + q0 * idx->dim,
                              static_cast<int>(cnt), m0, dist, qblocks, static_cast<int>(slices));
                    return VG_OK;
                };
                if (idx->metric == VG_METRIC_L2)
                    VG_TRY(launch(vg::brute_dist_mq_kernel<vg::kMetricL2>));
                else if (idx->metric == VG_METRIC_COSINE)
                    VG_TRY(launch(vg::brute_dist_mq_kernel<vg::kMetricCos>));
                else
                    VG_TRY(launch(vg::brute_dist_mq_kernel<vg::kMetricDot>));
            } else {
                const dim3 grid(static_cast<unsigned>(cnt), row_blocks), block(vg::kBruteDistThreads);
                auto launch1 = [&](auto kern) -> int32_t {
                    VG_LAUNCH(kern, grid, block, 0, st, idx->d_vectors, n, idx->dim, q.ptr + q0 * idx->dim, m0, mask_stride, dist);
                    return VG_OK;
                };
                const bool one = cnt == 1;
                if (idx->metric == VG_METRIC_L2)
                    VG_TRY(one ? launch1(vg::brute_dist_kernel<vg::kMetricL2, true>) : launch1(vg::brute_dist_kernel<vg::kMetricL2, false>));
                else if (idx->metric == VG_METRIC_COSINE)
                    VG_TRY(one ? launch1(vg::brute_dist_kernel<vg::kMetricCos, true>) : launch1(vg::brute_dist_kernel<vg::kMetricCos, false>));
                else
                    VG_TRY(one ? launch1(vg::brute_dist_kernel<vg::kMetricDot, true>) : launch1(vg::brute_dist_kernel<vg::kMetricDot, false>));
            }
        }
        vg::ProfScope prof(idx->ctx, "hnsw_brute_replay", st);
        VG_LAUNCH(replay, dim3(static_cast<unsigned>(cnt)), dim3(vg::kBruteThreads), lds, st, dist, n, m0, mask_stride, k,
                  oid.ptr + q0 * k, osc.ptr + q0 * k);
    }
    VG_TRY(vg::brute_nan_replay(idx, q.ptr, nq, k, mode, mk.ptr, mask_stride, oid.ptr, osc.ptr, st));  // queries whose distances may hold a NaN
    VG_TRY(oid.finish());
    VG_TRY(osc.finish());
    return VG_OK;
}

VG_API int32_t vg_debug_heap_replay(vg_ctx *ctx, int32_t is_max, int32_t unsigned_keys, const int32_t *ops, int32_t n_ops,
                                    int32_t *out, int32_t *final_len, uint64_t *final_items, int32_t cap, void *stream)
{
    VG_CHECK(ctx, VG_ERR_INVALID_ARG, "vg_debug_heap_replay: ctx is NULL");
    VG_CHECK(n_ops >= 0 && cap >= 0 && cap <= 8192, VG_ERR_INVALID_ARG, "vg_debug_heap_replay: n_ops < 0 or cap outside 0..8192");
    VG_CHECK(final_len && (n_ops == 0 || (ops && out)) && (cap == 0 || final_items), VG_ERR_INVALID_ARG,
             "vg_debug_heap_replay: NULL buffer");
    VG_HIP(hipSetDevice(ctx->device));
    hipStream_t st = vg::pick_stream(ctx, stream);
    // the script must never hold more than `cap` items at once (checked on the host side of the script)
    int len = 0;
    std::vector<int32_t> host_ops;
    const int32_t *h = ops;
    if (n_ops && vg::is_device_ptr(ops)) {
        host_ops.resize(static_cast<size_t>(n_ops) * 4);
        VG_HIP(hipMemcpy(host_ops.data(), ops, host_ops.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
        h = host_ops.data();
    }
    for (int i = 0; i < n_ops; i++) {
        const int op = h[4 * i], arg = h[4 * i + 3];
        if (op == VG_HEAP_PUSH) len++;
        else if (op == VG_HEAP_POP) len -= len > 0;
        else if ((op == VG_HEAP_PUSH_BOUNDED || op == VG_HEAP_TRY_PUSH_BOUNDED) && len < arg) len++;
        else if (op == VG_HEAP_RESET) len = 0;
        VG_CHECK(len <= cap, VG_ERR_INVALID_ARG, "vg_debug_heap_replay: the script holds %d items at op %d, cap is %d", len, i, cap);
    }
    vg::DevIn<int32_t> dops;
    vg::DevOut<int32_t> dout, dlen;
    vg::DevOut<uint64_t> ditems;
    VG_TRY(dops.init(ops, static_cast<size_t>(n_ops) * 4, st));
    VG_TRY(dout.init(out, static_cast<size_t>(n_ops) * 3, st));
    VG_TRY(dlen.init(final_len, 1, st));
    VG_TRY(ditems.init(final_items, static_cast<size_t>(cap), st));
    const size_t lds = static_cast<size_t>(cap + 8) * sizeof(vg::HItem);
    VG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(vg::heap_replay_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               static_cast<int>(lds)));
    VG_LAUNCH(vg::heap_replay_kernel, dim3(1), dim3(64), lds, st, is_max, unsigned_keys, dops.ptr, n_ops, dout.ptr, dlen.ptr,
              ditems.ptr, cap);
    VG_TRY(dout.finish());
    VG_TRY(dlen.finish());
    VG_TRY(ditems.finish());
    return VG_OK;
}
