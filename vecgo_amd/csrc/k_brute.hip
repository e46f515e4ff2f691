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

#include <algorithm>

namespace vg {

constexpr int kBruteThreads = 1024;              // replay: 16 waves stream, wave 0 replays
constexpr int kBruteStep = kBruteThreads * 4;    // rows per step (one float4 per lane)
constexpr int kBruteMaxK = 1024;
constexpr int kBruteDistThreads = 256;           // 16 pair-groups per workgroup
constexpr int kBruteRowsPerBlock = 1024;

__device__ __forceinline__ bool mask_bit(const uint8_t *__restrict__ mask, int64_t i)
{
    return mask == nullptr || ((mask[i >> 3] >> (i & 7)) & 1);
}

// dist[q][i] for the rows of block y; blockIdx.x = query (consecutive workgroups share a slice of rows through L2)
template <int METRIC>
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
                d = -exact_pair16<true, kPair>(row, qv, dim, sub);
            } else {
                d = exact_pair16<false, kPair>(row, qv, dim, sub);
                if (METRIC == kMetricCos) d = 0.5f * d;
            }
            if ((threadIdx.x & 15) == 0) dq[i] = d;
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
    __shared__ int wave_cnt[kBruteThreads / 64];
    __shared__ int s_len;
    __shared__ float s_top;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t q = blockIdx.x;
    const float *dq = dist + q * n;
    const uint8_t *mq = mask ? mask + q * mask_stride : nullptr;
    if (tid == 0) {
        s_len = 0;
        s_top = 0.0f;
    }
    __syncthreads();
    int len = 0;  // wave 0's copy is the live one
    for (int64_t base = 0; base < n; base += kBruteStep) {
        const int cur_len = s_len;
        const float top = s_top;
        const int64_t i0 = base + static_cast<int64_t>(tid) * 4;
        float d[4];
        bool f[4];
        if (i0 + 3 < n && mq == nullptr && (n & 3) == 0) {
            const float4 v = *reinterpret_cast<const float4 *>(dq + i0);  // n*4 bytes per query row: 16-byte aligned when n % 4 == 0
            d[0] = v.x, d[1] = v.y, d[2] = v.z, d[3] = v.w;
#pragma unroll
            for (int e = 0; e < 4; e++) f[e] = cur_len < k || d[e] < top;
        } else {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const bool in = i0 + e < n && mask_bit(mq, i0 + e);
                d[e] = in ? dq[i0 + e] : 0.0f;
                f[e] = in && (cur_len < k || d[e] < top);
            }
        }
        const int cnt = int(f[0]) + int(f[1]) + int(f[2]) + int(f[3]);
        // ordered exclusive prefix over the workgroup: row order = thread order
        int incl = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int y = __shfl_up(incl, off);
            if (lane >= off) incl += y;
        }
        if (lane == 63) wave_cnt[wave] = incl;
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < kBruteThreads / 64; w++) {
            const int c = wave_cnt[w];
            if (w < wave) before += c;
            total += c;
        }
        if (total == 0) {  // uniform: nothing in this step can enter the heap
            __syncthreads();  // wave_cnt is rewritten by the next step
            continue;
        }
        int pos = before + incl - cnt;
#pragma unroll
        for (int e = 0; e < 4; e++)
            if (f[e]) heap_store(list + pos++, HItem{static_cast<uint32_t>(i0 + e), d[e]});
        __syncthreads();
        if (wave == 0) {
            for (int j = 0; j < total; j++) brute_offer<MODE>(heap, len, k, heap_load(list + j));
            if (lane == 0) {
                s_len = len;
                s_top = len > 0 ? heap_get(heap, 0).dist : 0.0f;
            }
        }
        __syncthreads();
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

}  // namespace vg

VG_API int32_t vg_search_hnsw_brute(vg_index *idx, const float *queries, int64_t nq, int32_t k, int32_t mode,
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
    const int64_t cap = std::min<int64_t>(int64_t(16) << 30, std::max<int64_t>(int64_t(1) << 30, idx->ctx->hbm_bytes / 16));
    int64_t chunk = std::max<int64_t>(1, cap / std::max<int64_t>(n * 4, 1));
    chunk = std::min<int64_t>(std::min(chunk, nq), 65535);
    vg::ArenaCall ar(idx->ctx, st);
    const int i_dist = ar.add(sizeof(float) * static_cast<size_t>(chunk) * std::max<int64_t>(n, 1));
    VG_TRY(ar.commit());
    float *dist = ar.get<float>(i_dist);
    const size_t lds = (static_cast<size_t>((k + 4 + 3) & ~3) + vg::kBruteStep) * sizeof(vg::HItem);
    auto replay = mode == VG_BRUTE_SCAN ? vg::brute_replay_kernel<VG_BRUTE_SCAN> : vg::brute_replay_kernel<VG_BRUTE_BITMAP>;
    VG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(replay), hipFuncAttributeMaxDynamicSharedMemorySize,
                               static_cast<int>(lds)));
    const unsigned row_blocks = static_cast<unsigned>(std::min<int64_t>(std::max<int64_t>((n + vg::kBruteRowsPerBlock - 1) /
                                                                                          vg::kBruteRowsPerBlock, 1), 65535));
    for (int64_t q0 = 0; q0 < nq; q0 += chunk) {
        const int64_t cnt = std::min(chunk, nq - q0);
        const uint8_t *m0 = mk.ptr ? mk.ptr + q0 * mask_stride : nullptr;
        vg::ProfScope prof(idx->ctx, "hnsw_brute", st);
        if (n > 0) {
            const dim3 grid(static_cast<unsigned>(cnt), row_blocks), block(vg::kBruteDistThreads);
            if (idx->metric == VG_METRIC_L2)
                VG_LAUNCH(vg::brute_dist_kernel<vg::kMetricL2>, grid, block, 0, st, idx->d_vectors, n, idx->dim,
                          q.ptr + q0 * idx->dim, m0, mask_stride, dist);
            else if (idx->metric == VG_METRIC_COSINE)
                VG_LAUNCH(vg::brute_dist_kernel<vg::kMetricCos>, grid, block, 0, st, idx->d_vectors, n, idx->dim,
                          q.ptr + q0 * idx->dim, m0, mask_stride, dist);
            else
                VG_LAUNCH(vg::brute_dist_kernel<vg::kMetricDot>, grid, block, 0, st, idx->d_vectors, n, idx->dim,
                          q.ptr + q0 * idx->dim, m0, mask_stride, dist);
        }
        VG_LAUNCH(replay, dim3(static_cast<unsigned>(cnt)), dim3(vg::kBruteThreads), lds, st, dist, n, m0, mask_stride, k,
                  oid.ptr + q0 * k, osc.ptr + q0 * k);
    }
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
