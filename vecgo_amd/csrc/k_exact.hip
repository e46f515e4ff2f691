// k_exact.hip — exact fp32 scoring in the reference's summation order: simd batch kernels,
// Segment.Rerank, candidate scoring.  16 lanes per (query, row) pair, 4 pairs per wave.
#include "vg_device.hpp"
#include "vg_exact.hpp"
#include "vg_internal.hpp"

namespace vg {

constexpr int kExactThreads = 256;  // 4 waves = 16 pair-groups per workgroup

// simd.SquaredL2Batch / DotBatch: one query, n contiguous targets.
template <bool DOT>
__global__ __launch_bounds__(kExactThreads) void batch_kernel(const float *__restrict__ query,
                                                              const float *__restrict__ targets,
                                                              int dim, int64_t n,
                                                              float *__restrict__ out)
{
    const Sub16 sub = Sub16::make(threadIdx.x);
    const int64_t groups = static_cast<int64_t>(gridDim.x) * (kExactThreads / 16);
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * (kExactThreads / 16) + (threadIdx.x >> 4);
         i < n; i += groups) {
        const float v = exact_pair16<DOT, kBatch, true>(targets + i * dim, query, dim, sub);  // targets are read once: nontemporal
        if ((threadIdx.x & 15) == 0) out[i] = v;
    }
}

// scores[q][c] = exact distance(query q, row cand[q][c]); one workgroup per (query, chunk)
template <bool DOT>
__global__ __launch_bounds__(kExactThreads) void score_candidates_kernel(
    const float *__restrict__ base, int64_t n, int dim, const float *__restrict__ queries,
    const uint32_t *__restrict__ cand, int nc, float *__restrict__ scores)
{
    const Sub16 sub = Sub16::make(threadIdx.x);
    const int64_t q = blockIdx.y;
    const float *qv = queries + q * dim;
    const int stride = gridDim.x * (kExactThreads / 16);
    for (int c = blockIdx.x * (kExactThreads / 16) + (threadIdx.x >> 4); c < nc; c += stride) {
        const uint32_t id = cand[q * nc + c];
        float v = DOT ? -INFINITY : INFINITY;
        if (id != VG_INVALID_ID && id < n)
            v = exact_pair16<DOT, kPair>(base + static_cast<int64_t>(id) * dim, qv, dim, sub);
        if ((threadIdx.x & 15) == 0) scores[q * nc + c] = v;
    }
}

// Rerank: exact scores of nc candidates, then the k best by (score, id); one workgroup per query
template <bool DOT>
__global__ __launch_bounds__(kExactThreads) void rerank_kernel(
    const float *__restrict__ base, int64_t n, int dim, const float *__restrict__ queries,
    const uint32_t *__restrict__ cand, int nc, int k, uint32_t *__restrict__ ids,
    float *__restrict__ scores)
{
    extern __shared__ uint64_t keys[];  // nc keys
    const Sub16 sub = Sub16::make(threadIdx.x);
    const int64_t q = blockIdx.x;
    const float *qv = queries + q * dim;
    for (int c = threadIdx.x >> 4; c < nc; c += kExactThreads / 16) {
        const uint32_t id = cand[q * nc + c];
        uint64_t key = kKeyMax;
        if (id != VG_INVALID_ID && id < n) {
            const float v = exact_pair16<DOT, kPair>(base + static_cast<int64_t>(id) * dim, qv, dim, sub);
            key = make_key(v, id, DOT);
        }
        if ((threadIdx.x & 15) == 0) keys[c] = key;
    }
    __syncthreads();
    if (k > 64) {  // more than one wave's register list: sort the keys in LDS (n2 = the padded power of two)
        int n2 = 64;
        while (n2 < nc) n2 <<= 1;
        for (int c = nc + threadIdx.x; c < n2; c += kExactThreads) keys[c] = kKeyMax;
        __syncthreads();
        bitonic_sort_lds(keys, n2, threadIdx.x, kExactThreads);
        // a row listed twice is scored twice and reported twice, like the reference's loop (segment.go:757-779)
        for (int i = threadIdx.x; i < k; i += kExactThreads) {
            const uint64_t e = i < n2 ? keys[i] : kKeyMax;
            ids[q * k + i] = e == kKeyMax ? VG_INVALID_ID : key_row(e);
            scores[q * k + i] = e == kKeyMax ? (DOT ? -INFINITY : INFINITY) : key_score(e, DOT);
        }
        return;
    }
    if (threadIdx.x < 64) {  // wave 0 selects the k best (k <= 64)
        const int lane = threadIdx.x;
        WaveTopK tk;
        tk.init(k);
        for (int c0 = 0; c0 < nc; c0 += 64) {
            const int c = c0 + lane;
            uint64_t key = c < nc ? keys[c] : kKeyMax;
            // a row listed twice is scored twice and reported twice, like the reference's loop
            tk.offer(key, lane);
        }
        if (lane < k) {
            const uint64_t e = tk.list;
            ids[q * k + lane] = e == kKeyMax ? VG_INVALID_ID : key_row(e);
            scores[q * k + lane] = e == kKeyMax ? (DOT ? -INFINITY : INFINITY) : key_score(e, DOT);
        }
    }
}

__global__ void norm_max_kernel(const float *__restrict__ v, int64_t n, float *__restrict__ out)
{
    __shared__ float sm[1024];
    float m = 0.0f;
    for (int64_t i = threadIdx.x; i < n; i += 1024) m = fmaxf(m, v[i]);
    sm[threadIdx.x] = m;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (threadIdx.x < s) sm[threadIdx.x] = fmaxf(sm[threadIdx.x], sm[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sm[0];
}

// (+ max |x| over every element, NaN / Inf counted as +Inf, into *maxabs_bits: what vg_cand_replay.hpp's risk test reads)
__global__ void row_norms_kernel(const float *__restrict__ base, int64_t n, int dim,
                                 float *__restrict__ norms, int *__restrict__ maxabs_bits)
{
    // ||x||^2 for the GEMM-form candidate generation only (never reported): plain order
    const int64_t row = static_cast<int64_t>(blockIdx.x) * (blockDim.x / 64) + (threadIdx.x >> 6);
    if (row >= n) return;
    const int lane = threadIdx.x & 63;
    float s = 0.0f, mx = 0.0f;
    for (int j = lane; j < dim; j += 64) {
        const float v = base[row * dim + j];
        s = __builtin_fmaf(v, v, s);
        const float a = fabsf(v);
        mx = fmaxf(mx, a == a ? a : INFINITY);
    }
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_xor(s, off);
        mx = fmaxf(mx, __shfl_xor(mx, off));
    }
    if (lane == 0) {
        norms[row] = s;
        // (non-negative floats order like their bits; a million waves on one address: the atomic only where it would raise the value —
        // unconditional it made this pass 11 ms instead of 0.5)
        if (__float_as_int(mx) > *reinterpret_cast<volatile int *>(maxabs_bits)) atomicMax(maxabs_bits, __float_as_int(mx));
    }
}

}  // namespace vg

static bool metric_is_dot(int32_t metric) { return metric != VG_METRIC_L2; }

VG_API int32_t vg_index_set_vectors(vg_index *idx, const float *base, void *stream)
{
    VG_CHECK(idx, VG_ERR_INVALID_ARG, "vg_index_set_vectors: NULL index");
    VG_CHECK(idx->n == 0 || base, VG_ERR_INVALID_ARG, "vg_index_set_vectors: base is NULL");
    VG_CHECK(idx->metric != VG_METRIC_HAMMING, VG_ERR_UNSUPPORTED,
             "unsupported metric for float32: Hamming");
    VG_HIP(hipSetDevice(idx->ctx->device));
    hipStream_t st = vg::pick_stream(idx->ctx, stream);
    if (idx->d_vectors) {
        VG_HIP(hipStreamSynchronize(st));
        VG_HIP(hipFree(idx->d_vectors));
        VG_HIP(hipFree(idx->d_norms));
        VG_HIP(hipFree(idx->d_norm_max));
        VG_HIP(hipFree(idx->d_flat_stats));
        if (idx->d_vectors_bf16) VG_HIP(hipFree(idx->d_vectors_bf16));  // a copy of the OLD rows: enable again after this
        idx->d_vectors = idx->d_norms = idx->d_norm_max = nullptr;
        idx->d_flat_stats = nullptr;
        idx->d_vectors_bf16 = nullptr;
    }
    if (idx->n == 0) return VG_OK;
    // all four allocations or none: a search must never find rows without their norms / counters
    const int32_t status = [&]() -> int32_t {
        size_t count = static_cast<size_t>(idx->n) * idx->dim;
        VG_HIP(hipMalloc(reinterpret_cast<void **>(&idx->d_vectors), count * sizeof(float)));
        VG_HIP(hipMalloc(reinterpret_cast<void **>(&idx->d_norms), static_cast<size_t>(idx->n) * sizeof(float)));
        VG_HIP(hipMalloc(reinterpret_cast<void **>(&idx->d_norm_max), 2 * sizeof(float)));  // [1]: max |x|, non-finite -> +Inf
        VG_HIP(hipMemsetAsync(idx->d_norm_max, 0, 2 * sizeof(float), st));
        VG_HIP(hipMalloc(reinterpret_cast<void **>(&idx->d_flat_stats), 2 * sizeof(unsigned long long)));
        VG_HIP(hipMemcpyAsync(idx->d_vectors, base, count * sizeof(float), hipMemcpyDefault, st));
        VG_LAUNCH(vg::row_norms_kernel, dim3(static_cast<unsigned>((idx->n + 3) / 4)), dim3(256),
                           0, st, idx->d_vectors, idx->n, idx->dim, idx->d_norms, reinterpret_cast<int *>(idx->d_norm_max + 1));
        VG_LAUNCH(vg::norm_max_kernel, dim3(1), dim3(1024), 0, st, idx->d_norms, idx->n, idx->d_norm_max);
        VG_HIP(hipMemsetAsync(idx->d_flat_stats, 0, 2 * sizeof(unsigned long long), st));
        VG_HIP(hipStreamSynchronize(st));
        return VG_OK;
    }();
    if (status != VG_OK) {
        (void)hipStreamSynchronize(st);
        if (idx->d_vectors) (void)hipFree(idx->d_vectors);
        if (idx->d_norms) (void)hipFree(idx->d_norms);
        if (idx->d_norm_max) (void)hipFree(idx->d_norm_max);
        if (idx->d_flat_stats) (void)hipFree(idx->d_flat_stats);
        idx->d_vectors = idx->d_norms = idx->d_norm_max = nullptr;
        idx->d_flat_stats = nullptr;
    }
    return status;
}

static int32_t batch_impl(vg_ctx *ctx, bool dot, const float *query, const float *targets,
                          int64_t dim, int64_t n, float *out, void *stream)
{
    VG_CHECK(ctx, VG_ERR_INVALID_ARG, "batch kernel: ctx is NULL");
    // simd/kernels.go:255-257: the generic silently returns on bad dims; mirror that as success
    if (dim <= 0 || n <= 0) return VG_OK;
    VG_CHECK(query && targets && out, VG_ERR_INVALID_ARG, "batch kernel: NULL buffer");
    VG_CHECK(dim < (1 << 30), VG_ERR_INVALID_ARG, "batch kernel: dim too large");
    VG_HIP(hipSetDevice(ctx->device));
    hipStream_t st = vg::pick_stream(ctx, stream);
    vg::DevIn<float> q, t;
    vg::DevOut<float> o;
    VG_TRY(q.init(query, static_cast<size_t>(dim), st));
    VG_TRY(t.init(targets, static_cast<size_t>(n) * dim, st));
    VG_TRY(o.init(out, static_cast<size_t>(n), st));
    int64_t blocks = (n + 15) / 16;
    if (blocks > 4096) blocks = 4096;
    if (dot)
        VG_LAUNCH(vg::batch_kernel<true>, dim3(static_cast<unsigned>(blocks)),
                           dim3(vg::kExactThreads), 0, st, q.ptr, t.ptr, static_cast<int>(dim), n, o.ptr);
    else
        VG_LAUNCH(vg::batch_kernel<false>, dim3(static_cast<unsigned>(blocks)),
                           dim3(vg::kExactThreads), 0, st, q.ptr, t.ptr, static_cast<int>(dim), n, o.ptr);
    VG_TRY(o.finish());
    if (o.on_host()) VG_HIP(hipStreamSynchronize(st));
    return VG_OK;
}

VG_API int32_t vg_squared_l2_batch(vg_ctx *ctx, const float *query, const float *targets,
                                   int64_t dim, int64_t n, float *out, void *stream)
{
    return batch_impl(ctx, false, query, targets, dim, n, out, stream);
}

VG_API int32_t vg_dot_batch(vg_ctx *ctx, const float *query, const float *targets, int64_t dim,
                            int64_t n, float *out, void *stream)
{
    return batch_impl(ctx, true, query, targets, dim, n, out, stream);
}

VG_API int32_t vg_score_candidates(vg_index *idx, const float *queries, int64_t nq,
                                   const uint32_t *cand_ids, int32_t nc, float *scores,
                                   void *stream)
{
    VG_CHECK(idx, VG_ERR_INVALID_ARG, "vg_score_candidates: NULL index");
    VG_CHECK(nq >= 0 && nc >= 0, VG_ERR_INVALID_ARG, "vg_score_candidates: negative count");
    if (nq == 0 || nc == 0) return VG_OK;
    VG_CHECK(idx->d_vectors, VG_ERR_NOT_READY, "vg_score_candidates: index has no fp32 vectors");
    VG_CHECK(queries && cand_ids && scores, VG_ERR_INVALID_ARG, "vg_score_candidates: NULL buffer");
    VG_HIP(hipSetDevice(idx->ctx->device));
    hipStream_t st = vg::pick_stream(idx->ctx, stream);
    vg::DevIn<float> q;
    vg::DevIn<uint32_t> c;
    vg::DevOut<float> o;
    VG_TRY(q.init(queries, static_cast<size_t>(nq) * idx->dim, st));
    VG_TRY(c.init(cand_ids, static_cast<size_t>(nq) * nc, st));
    VG_TRY(o.init(scores, static_cast<size_t>(nq) * nc, st));
    const bool dot = metric_is_dot(idx->metric);
    const int64_t maxy = 65535;
    unsigned gx = static_cast<unsigned>((nc + 15) / 16);
    if (gx > 64) gx = 64;
    for (int64_t q0 = 0; q0 < nq; q0 += maxy) {
        int64_t cnt = nq - q0 < maxy ? nq - q0 : maxy;
        dim3 grid(gx, static_cast<unsigned>(cnt));
        if (dot)
            VG_LAUNCH(vg::score_candidates_kernel<true>, grid, dim3(vg::kExactThreads), 0, st,
                               idx->d_vectors, idx->n, idx->dim, q.ptr + q0 * idx->dim,
                               c.ptr + q0 * nc, nc, o.ptr + q0 * nc);
        else
            VG_LAUNCH(vg::score_candidates_kernel<false>, grid, dim3(vg::kExactThreads), 0, st,
                               idx->d_vectors, idx->n, idx->dim, q.ptr + q0 * idx->dim,
                               c.ptr + q0 * nc, nc, o.ptr + q0 * nc);
    }
    VG_TRY(o.finish());
    if (o.on_host()) VG_HIP(hipStreamSynchronize(st));
    return VG_OK;
}

VG_API int32_t vg_rerank(vg_index *idx, const float *queries, int64_t nq, const uint32_t *cand_ids,
                         int32_t nc, int32_t k, uint32_t *ids, float *scores, void *stream)
{
    VG_CHECK(idx, VG_ERR_INVALID_ARG, "vg_rerank: NULL index");
    VG_CHECK(nq >= 0 && nc >= 0 && k >= 0, VG_ERR_INVALID_ARG, "vg_rerank: negative count");
    if (nq == 0 || k == 0) return VG_OK;
    VG_CHECK(idx->d_vectors, VG_ERR_NOT_READY, "vg_rerank: index has no fp32 vectors");
    VG_CHECK(queries && ids && scores && (nc == 0 || cand_ids), VG_ERR_INVALID_ARG,
             "vg_rerank: NULL buffer");
    VG_CHECK(k <= 512, VG_ERR_UNSUPPORTED, "vg_rerank: k=%d exceeds 512", k);
    size_t key_slots = static_cast<size_t>(nc > 0 ? nc : 1);
    if (k > 64) {  // sorted in LDS: padded to a power of two
        key_slots = 64;
        while (key_slots < static_cast<size_t>(nc)) key_slots <<= 1;
    }
    VG_CHECK(key_slots * 8 <= 160 * 1024 - 1024, VG_ERR_UNSUPPORTED,
             "vg_rerank: nc=%d candidates per query exceed the LDS key buffer", nc);
    VG_HIP(hipSetDevice(idx->ctx->device));
    hipStream_t st = vg::pick_stream(idx->ctx, stream);
    vg::DevIn<float> q;
    vg::DevIn<uint32_t> c;
    vg::DevOut<uint32_t> oid;
    vg::DevOut<float> osc;
    VG_TRY(q.init(queries, static_cast<size_t>(nq) * idx->dim, st));
    VG_TRY(c.init(cand_ids, static_cast<size_t>(nq) * nc, st));
    VG_TRY(oid.init(ids, static_cast<size_t>(nq) * k, st));
    VG_TRY(osc.init(scores, static_cast<size_t>(nq) * k, st));
    const bool dot = metric_is_dot(idx->metric);
    size_t lds = key_slots * 8;
    auto kern = dot ? vg::rerank_kernel<true> : vg::rerank_kernel<false>;
    VG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                               hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    {
        vg::ProfScope prof(idx->ctx, "rerank", st);
        VG_LAUNCH(kern, dim3(static_cast<unsigned>(nq)), dim3(vg::kExactThreads), lds, st,
                  idx->d_vectors, idx->n, idx->dim, q.ptr, c.ptr, nc, k, oid.ptr, osc.ptr);
    }
    VG_TRY(oid.finish());
    VG_TRY(osc.finish());
    if (oid.on_host() || osc.on_host()) VG_HIP(hipStreamSynchronize(st));
    return VG_OK;
}
