// k_pq_train.hip — ProductQuantizer Train / Encode / Decode / ComputeAsymmetricDistance on the
// device (internal/quantization/pq.go:68-260, :275-433).
//
// Determinism: every fp32 operation the reference performs sequentially (k-means++ running
// sums, the per-cluster coordinate sums of updateCentroids, the per-term int8-dequant L2) is
// performed here in the same order, so for a given seed the device result equals the CPU
// oracle's bit for bit.  Parallelism comes from the independent units:
// sub-quantizers x points (assignment), sub-quantizers x clusters x coordinates (update).
#include "vg_device.hpp"
#include "vg_internal.hpp"

namespace vg {

// ---- counter-based RNG shared with the oracle (vgo_rng_u64) ---------------------------------
__host__ __device__ inline uint64_t splitmix64(uint64_t x)
{
    x += 0x9e3779b97f4a7c15ULL;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL;
    return x ^ (x >> 31);
}
__host__ __device__ inline uint64_t rng_u64(uint64_t seed, uint64_t a, uint64_t b, uint64_t c)
{
    uint64_t h = splitmix64(seed);
    h = splitmix64(h ^ a);
    h = splitmix64(h ^ b);
    h = splitmix64(h ^ c);
    return h;
}
__device__ inline float rng_f32(uint64_t r) { return static_cast<float>(r >> 40) * (1.0f / 16777216.0f); }

// squaredL2Avx512 (floats_avx512.c:69-129) of two short vectors by ONE thread: n < 64 is only the
// FMA-contracted scalar tail; longer sub-vectors emulate the 4x16 accumulators and the tree.
__device__ inline float l2_avx512_thread(const float *a, const float *b, int n)
{
    float total = 0.0f;
    int i = 0;
    if (n >= 64) {
        float acc[64];
        for (int l = 0; l < 64; l++) acc[l] = 0.0f;
        for (; i + 64 <= n; i += 64)
            for (int l = 0; l < 64; l++) {
                const float d = a[i + l] - b[i + l];
                acc[l] = __builtin_fmaf(d, d, acc[l]);
            }
        float s[16];
        for (int l = 0; l < 16; l++) s[l] = (acc[l] + acc[16 + l]) + (acc[32 + l] + acc[48 + l]);
        total = reduce16_regs(s);
    }
    for (; i < n; i++) {
        const float d = a[i] - b[i];
        total = __builtin_fmaf(d, d, total);
    }
    return total;
}

// squaredL2Int8DequantizedGeneric (internal/simd/kernels.go:354-362) against a pre-dequantised
// centroid (v = float32(code)*scale + offset, two separately rounded ops done once)
__device__ inline float l2_deq_thread(const float *q, const float *v, int n)
{
    float sum = 0.0f;
    for (int i = 0; i < n; i++) {
        const float d = q[i] - v[i];
        const float dd = d * d;
        sum = sum + dd;
    }
    return sum;
}

// ---- k-means++ (pq.go:281-338): one workgroup per sub-quantizer -----------------------------
constexpr int kPPThreads = 256;
constexpr int kPPChunk = 4096;

__global__ __launch_bounds__(kPPThreads) void pq_kmeanspp_kernel(
    const float *__restrict__ vectors, int64_t n, int dim, int sd, int k, uint64_t seed,
    float *__restrict__ mind_all, float *__restrict__ cent_all, int sub0)
{
    __shared__ float chunk[kPPChunk];
    __shared__ float cur[256];  // current centroid (sd <= 256)
    __shared__ float s_sum;
    __shared__ float s_cum;
    __shared__ long long s_chosen;
    // sub = the sub-quantizer (global index: column offset and RNG stream); ls = its slot in the
    // scratch arrays of this call, which may train only the range [sub0, sub0 + gridDim.x)
    const int ls = blockIdx.x, sub = sub0 + ls;
    const int tid = threadIdx.x;
    const float *base = vectors + static_cast<int64_t>(sub) * sd;
    float *mind = mind_all + static_cast<int64_t>(ls) * n;
    float *cent = cent_all + static_cast<int64_t>(ls) * k * sd;

    if (n < k) {  // pq.go:285-291
        for (int t = tid; t < k * sd; t += kPPThreads)
            cent[t] = base[static_cast<int64_t>((t / sd) % n) * dim + (t % sd)];
        return;
    }
    uint64_t ctr = 0;
    long long first = static_cast<long long>(rng_u64(seed, sub, 1, ctr++) % static_cast<uint64_t>(n));
    for (int t = tid; t < sd; t += kPPThreads) {
        cur[t] = base[first * dim + t];
        cent[t] = cur[t];
    }
    __syncthreads();
    for (int c = 0; c < k; c++) {
        // (c >= 1) choose the next centroid from the running sum; (c == 0) it is `first`
        if (c >= 1) {
            const float sum = s_sum;
            if (sum == 0.0f) {  // pq.go:306-310: random vector, mind/sum untouched
                if (tid == 0) s_chosen = static_cast<long long>(rng_u64(seed, sub, 1, ctr) % static_cast<uint64_t>(n));
                ctr++;
                __syncthreads();
                const long long ch = s_chosen;
                for (int t = tid; t < sd; t += kPPThreads) cent[c * sd + t] = base[ch * dim + t];
                __syncthreads();
                continue;
            }
            const float target = rng_f32(rng_u64(seed, sub, 1, ctr)) * sum;
            ctr++;
            if (tid == 0) {
                s_cum = 0.0f;
                s_chosen = -1;
            }
            __syncthreads();
            // sequential cumsum (pq.go:315-323), staged through LDS chunk by chunk
            for (int64_t c0 = 0; c0 < n; c0 += kPPChunk) {
                if (s_chosen >= 0) break;
                const int len = static_cast<int>(n - c0 < kPPChunk ? n - c0 : kPPChunk);
                for (int t = tid; t < len; t += kPPThreads) chunk[t] = mind[c0 + t];
                __syncthreads();
                if (tid == 0) {
                    float cum = s_cum;
                    long long ch = -1;
                    for (int t = 0; t < len; t++) {
                        cum += chunk[t];
                        if (cum >= target) {
                            ch = c0 + t;
                            break;
                        }
                    }
                    s_cum = cum;
                    s_chosen = ch;
                }
                __syncthreads();
            }
            const long long ch = s_chosen >= 0 ? s_chosen : 0;  // pq.go:317 `chosen := 0`
            for (int t = tid; t < sd; t += kPPThreads) {
                cur[t] = base[ch * dim + t];
                cent[c * sd + t] = cur[t];
            }
            __syncthreads();
        }
        if (c == k - 1 && c >= 1) {
            // the reference still updates minDistSq after the last centroid; the values are
            // never read again, skip the pass
            break;
        }
        // update minDistSq with the new centroid, then the sequential running sum
        if (tid == 0) s_sum = 0.0f;
        __syncthreads();
        for (int64_t c0 = 0; c0 < n; c0 += kPPChunk) {
            const int len = static_cast<int>(n - c0 < kPPChunk ? n - c0 : kPPChunk);
            for (int t = tid; t < len; t += kPPThreads) {
                const float d = l2_avx512_thread(base + (c0 + t) * dim, cur, sd);
                float mv = d;
                if (c >= 1) {
                    const float old = mind[c0 + t];
                    mv = d < old ? d : old;
                }
                mind[c0 + t] = mv;
                chunk[t] = mv;
            }
            __syncthreads();
            if (tid == 0) {
                float sum = s_sum;
                for (int t = 0; t < len; t++) sum += chunk[t];
                s_sum = sum;
            }
            __syncthreads();
        }
    }
}

// ---- Lloyd assignment (pq.go:353-386, :416-433): thread per (point, sub-quantizer) ----------
__global__ __launch_bounds__(256) void pq_assign_kernel(const float *__restrict__ vectors, int64_t n,
                                                        int dim, int sd, int k,
                                                        const float *__restrict__ cent_all,
                                                        int32_t *__restrict__ assign_all,
                                                        int *__restrict__ changed,
                                                        const int *__restrict__ done, int sub0)
{
    extern __shared__ float cent[];  // k*sd
    const int ls = blockIdx.y, sub = sub0 + ls;
    if (done[ls]) return;
    for (int t = threadIdx.x; t < k * sd; t += blockDim.x)
        cent[t] = cent_all[static_cast<int64_t>(ls) * k * sd + t];
    __syncthreads();
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *v = vectors + i * dim + static_cast<int64_t>(sub) * sd;
    float best = 3.40282346638528859811704183484516925440e+38f;
    int bi = 0;
    for (int c = 0; c < k; c++) {
        const float d = l2_avx512_thread(v, cent + c * sd, sd);
        if (d < best) {
            best = d;
            bi = c;
        }
    }
    int32_t *a = assign_all + static_cast<int64_t>(ls) * n + i;
    if (*a != bi) {
        *a = bi;
        changed[ls] = 1;
    }
}

// ---- Lloyd update (pq.go:388-414): thread per (sub-quantizer, cluster, coordinate); the sum
// runs over the points in index order exactly like the reference's single loop ----------------
__global__ __launch_bounds__(256) void pq_update_kernel(const float *__restrict__ vectors, int64_t n,
                                                        int dim, int sd, int k, int iter, uint64_t seed,
                                                        const int32_t *__restrict__ assign_all,
                                                        float *__restrict__ cent_all,
                                                        const int *__restrict__ changed,
                                                        const int *__restrict__ done, int sub0)
{
    const int ls = blockIdx.y, sub = sub0 + ls;
    if (done[ls] || !changed[ls]) return;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= k * sd) return;
    const int c = t / sd, j = t % sd;
    const int32_t *assign = assign_all + static_cast<int64_t>(ls) * n;
    const float *col = vectors + static_cast<int64_t>(sub) * sd + j;
    float sum = 0.0f;
    int64_t count = 0;
    for (int64_t i = 0; i < n; i++) {
        if (assign[i] == c) {
            sum += col[i * dim];
            count++;
        }
    }
    float *dst = cent_all + (static_cast<int64_t>(ls) * k + c) * sd + j;
    if (count > 0) {
        *dst = sum / static_cast<float>(count);
    } else {  // pq.go:408-411 re-seed an empty cluster with a random vector
        const int64_t idx = static_cast<int64_t>(rng_u64(seed, sub, 2 + static_cast<uint64_t>(iter), c) %
                                                 static_cast<uint64_t>(n));
        *dst = col[idx * dim];
    }
}

// after assign+update of one iteration: a sub-quantizer whose assignments did not change stops
// (pq.go:346-348 `break`); `changed` is cleared for the next iteration
__global__ void pq_iter_end_kernel(int m, int *__restrict__ changed, int *__restrict__ done)
{
    const int sub = blockIdx.x * blockDim.x + threadIdx.x;
    if (sub >= m) return;
    if (!done[sub] && !changed[sub]) done[sub] = 1;
    changed[sub] = 0;
}

// ---- int8 quantisation of the trained centroids (pq.go:97-136) --------------------------------
__global__ __launch_bounds__(256) void pq_quantize_kernel(const float *__restrict__ cent_all, int k,
                                                          int sd, int8_t *__restrict__ codebooks,
                                                          float *__restrict__ scales,
                                                          float *__restrict__ offsets, int sub0)
{
    __shared__ float smin[256], smax[256];
    const int ls = blockIdx.x, sub = sub0 + ls;
    const int tid = threadIdx.x;
    const int cnt = k * sd;
    const float *cent = cent_all + static_cast<int64_t>(ls) * cnt;
    float mn = 3.40282346638528859811704183484516925440e+38f, mx = -mn;
    for (int t = tid; t < cnt; t += 256) {
        const float v = cent[t];
        if (v < mn) mn = v;
        if (v > mx) mx = v;
    }
    smin[tid] = mn;
    smax[tid] = mx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            if (smin[tid + s] < smin[tid]) smin[tid] = smin[tid + s];
            if (smax[tid + s] > smax[tid]) smax[tid] = smax[tid + s];
        }
        __syncthreads();
    }
    mn = smin[0];
    mx = smax[0];
    if (mx == mn) mx = mn + 1e-6f;
    const float scale = (mx - mn) / 255.0f;
    const float offset = mn + 128.0f * scale;
    if (tid == 0) {
        scales[sub] = scale;
        offsets[sub] = offset;
    }
    for (int t = tid; t < cnt; t += 256) {
        const float q = (cent[t] - mn) / scale;
        // math.Round(float64(q)): half away from zero
        int val = static_cast<int>(round(static_cast<double>(q)));
        if (val < 0) val = 0;
        if (val > 255) val = 255;
        codebooks[static_cast<int64_t>(sub) * cnt + t] = static_cast<int8_t>(val - 128);
    }
}

// ---- Encode (pq.go:147-176): workgroup = (sub-quantizer, 256 rows); the sub-quantizer's
// dequantised codebook sits in LDS and is read as a broadcast -----------------------------------
__global__ __launch_bounds__(256) void pq_encode_kernel(const float *__restrict__ vectors, int64_t n,
                                                        int dim, int m, int sd, int k,
                                                        const int8_t *__restrict__ codebooks,
                                                        const float *__restrict__ scales,
                                                        const float *__restrict__ offsets,
                                                        uint8_t *__restrict__ codes)
{
    extern __shared__ float deq[];  // k*sd
    const int sub = blockIdx.y;
    const float scale = scales[sub], offset = offsets[sub];
    for (int t = threadIdx.x; t < k * sd; t += blockDim.x) {
        float v = static_cast<float>(codebooks[static_cast<int64_t>(sub) * k * sd + t]) * scale;
        deq[t] = v + offset;
    }
    __syncthreads();
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *v = vectors + i * dim + static_cast<int64_t>(sub) * sd;
    float q[16];
    const bool small = sd <= 16;
    if (small)
        for (int t = 0; t < sd; t++) q[t] = v[t];
    const float *qp = small ? q : v;
    int best = 0;
    float bd = l2_deq_thread(qp, deq, sd);
    for (int c = 1; c < k; c++) {
        const float d = l2_deq_thread(qp, deq + c * sd, sd);
        if (d < bd) {
            bd = d;
            best = c;
        }
    }
    codes[i * m + sub] = static_cast<uint8_t>(best);
}

// Encode for sub-vector dims that are a multiple of 4 (8 at the BASELINE shape): the dequantised
// centroid is read from LDS as float4 broadcasts (sd/4 reads instead of sd) and every thread scores
// kEncRows rows against it, so the 256 x sd table walk is amortised over kEncRows rows.  Same
// arithmetic as pq_encode_kernel: per dimension d = q - v, dd = d * d, sum = sum + dd, in order.
constexpr int kEncRows = 4;
template <int SD>
__global__ __launch_bounds__(256) void pq_encode_vec_kernel(const float *__restrict__ vectors, int64_t n,
                                                            int dim, int m, int k,
                                                            const int8_t *__restrict__ codebooks,
                                                            const float *__restrict__ scales,
                                                            const float *__restrict__ offsets,
                                                            uint8_t *__restrict__ codes)
{
    extern __shared__ float deq[];  // k*SD
    const int sub = blockIdx.y;
    const float scale = scales[sub], offset = offsets[sub];
    for (int t = threadIdx.x; t < k * SD; t += blockDim.x) {
        float v = static_cast<float>(codebooks[static_cast<int64_t>(sub) * k * SD + t]) * scale;
        deq[t] = v + offset;
    }
    __syncthreads();
    const int64_t i0 = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) * kEncRows;
    if (i0 >= n) return;
    float q[kEncRows][SD];
#pragma unroll
    for (int r = 0; r < kEncRows; r++) {
        const int64_t i = i0 + r < n ? i0 + r : n - 1;
        const float4 *v4 = reinterpret_cast<const float4 *>(vectors + i * dim + static_cast<int64_t>(sub) * SD);
#pragma unroll
        for (int t = 0; t < SD / 4; t++) {
            const float4 x = v4[t];
            q[r][4 * t] = x.x; q[r][4 * t + 1] = x.y; q[r][4 * t + 2] = x.z; q[r][4 * t + 3] = x.w;
        }
    }
    int best[kEncRows];
    float bd[kEncRows];
#pragma unroll
    for (int r = 0; r < kEncRows; r++) {
        best[r] = 0;
        bd[r] = 3.40282346638528859811704183484516925440e+38f;
    }
    for (int c = 0; c < k; c++) {
        float cv[SD];
        const float4 *c4 = reinterpret_cast<const float4 *>(deq + c * SD);
#pragma unroll
        for (int t = 0; t < SD / 4; t++) {
            const float4 x = c4[t];
            cv[4 * t] = x.x; cv[4 * t + 1] = x.y; cv[4 * t + 2] = x.z; cv[4 * t + 3] = x.w;
        }
#pragma unroll
        for (int r = 0; r < kEncRows; r++) {
            float sum = 0.0f;
#pragma unroll
            for (int t = 0; t < SD; t++) {
                const float d = q[r][t] - cv[t];
                const float dd = d * d;
                sum = sum + dd;
            }
            // FindNearestCentroidInt8 (kernels.go:376-396): centroid 0 seeds the minimum whatever its
            // distance is (NaN included), later ones need a strict '<'
            if (c == 0 || sum < bd[r]) {
                bd[r] = sum;
                best[r] = c;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < kEncRows; r++)
        if (i0 + r < n) codes[(i0 + r) * m + sub] = static_cast<uint8_t>(best[r]);
}

__global__ void pq_decode_kernel(const uint8_t *__restrict__ codes, int64_t n, int dim, int m, int sd,
                                 int k, const int8_t *__restrict__ codebooks,
                                 const float *__restrict__ scales, const float *__restrict__ offsets,
                                 float *__restrict__ out)
{
    const int64_t gid = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (gid >= n * dim) return;
    const int64_t i = gid / dim;
    const int d = static_cast<int>(gid % dim);
    const int sub = d / sd, t = d % sd;
    const int c = codes[i * m + sub];
    const float v = static_cast<float>(codebooks[(static_cast<int64_t>(sub) * k + c) * sd + t]) * scales[sub];
    out[gid] = v + offsets[sub];
}

// ComputeAsymmetricDistance (pq.go:234-260): thread per code row, terms added sequentially over m
__global__ void pq_asym_kernel(const float *__restrict__ query, const uint8_t *__restrict__ codes,
                               int64_t n, int dim, int m, int sd, int k,
                               const int8_t *__restrict__ codebooks, const float *__restrict__ scales,
                               const float *__restrict__ offsets, float *__restrict__ out)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float distance = 0.0f;
    for (int sub = 0; sub < m; sub++) {
        const int c = codes[i * m + sub];
        const int8_t *cb = codebooks + (static_cast<int64_t>(sub) * k + c) * sd;
        const float scale = scales[sub], offset = offsets[sub];
        float sum = 0.0f;
        for (int t = 0; t < sd; t++) {
            float v = static_cast<float>(cb[t]) * scale;
            v = v + offset;
            const float d = query[sub * sd + t] - v;
            const float dd = d * d;
            sum = sum + dd;
        }
        distance = distance + sum;
    }
    out[i] = distance;
}

}  // namespace vg

VG_API int32_t vg_pq_train_subset(vg_pq *pq, const float *vectors, int64_t n, int32_t iters, uint64_t seed,
                                  int32_t sub_begin, int32_t sub_count, void *stream)
{
    VG_CHECK(pq, VG_ERR_INVALID_ARG, "vg_pq_train: NULL quantizer");
    VG_CHECK(n > 0 && vectors, VG_ERR_INVALID_ARG, "no vectors provided for training");  // pq.go:69-71
    VG_CHECK(iters >= 0, VG_ERR_INVALID_ARG, "vg_pq_train: iters < 0");
    VG_CHECK(sub_begin >= 0 && sub_count >= 0 && sub_begin + sub_count <= pq->m, VG_ERR_INVALID_ARG,
             "vg_pq_train_subset: sub-quantizer range [%d, %d) outside [0, %d)", sub_begin, sub_begin + sub_count,
             pq->m);
    VG_CHECK(pq->subdim <= 256, VG_ERR_UNSUPPORTED, "vg_pq_train: sub-vector dim %d > 256", pq->subdim);
    if (sub_count == 0) return VG_OK;
    VG_HIP(hipSetDevice(pq->ctx->device));
    hipStream_t st = vg::pick_stream(pq->ctx, stream);
    const int m = sub_count, k = pq->k, sd = pq->subdim, dim = pq->dim;
    vg::DevIn<float> v;
    VG_TRY(v.init(vectors, static_cast<size_t>(n) * dim, st));
    vg::DevTmp<float> mind, cent;
    vg::DevTmp<int32_t> assign;
    vg::DevTmp<int> flags;
    VG_TRY(mind.init(static_cast<size_t>(m) * n, st));
    VG_TRY(cent.init(static_cast<size_t>(m) * k * sd, st));
    VG_TRY(assign.init(static_cast<size_t>(m) * n, st));
    VG_TRY(flags.init(static_cast<size_t>(2 * m), st));
    VG_HIP(hipMemsetAsync(assign.ptr, 0, sizeof(int32_t) * static_cast<size_t>(m) * n, st));
    VG_HIP(hipMemsetAsync(flags.ptr, 0, sizeof(int) * 2 * m, st));
    int *changed = flags.ptr, *done = flags.ptr + m;

    VG_LAUNCH(vg::pq_kmeanspp_kernel, dim3(m), dim3(vg::kPPThreads), 0, st, v.ptr, n, dim, sd,
                       k, seed, mind.ptr, cent.ptr, sub_begin);
    const size_t lds = static_cast<size_t>(k) * sd * sizeof(float);
    VG_CHECK(lds <= 64 * 1024, VG_ERR_UNSUPPORTED, "vg_pq_train: codebook of one sub-quantizer exceeds 64 KiB");
    const unsigned gx = static_cast<unsigned>((n + 255) / 256);
    const unsigned ux = static_cast<unsigned>((k * sd + 255) / 256);
    for (int it = 0; it < iters; it++) {
        VG_LAUNCH(vg::pq_assign_kernel, dim3(gx, m), dim3(256), lds, st, v.ptr, n, dim, sd, k,
                           cent.ptr, assign.ptr, changed, done, sub_begin);
        VG_LAUNCH(vg::pq_update_kernel, dim3(ux, m), dim3(256), 0, st, v.ptr, n, dim, sd, k, it,
                           seed, assign.ptr, cent.ptr, changed, done, sub_begin);
        VG_LAUNCH(vg::pq_iter_end_kernel, dim3((m + 63) / 64), dim3(64), 0, st, m, changed, done);
    }
    VG_LAUNCH(vg::pq_quantize_kernel, dim3(m), dim3(256), 0, st, cent.ptr, k, sd,
                       pq->d_codebooks, pq->d_scales, pq->d_offsets, sub_begin);
    VG_HIP(hipStreamSynchronize(st));
    if (sub_begin == 0 && sub_count == pq->m) pq->trained = true;
    return VG_OK;
}

VG_API int32_t vg_pq_train(vg_pq *pq, const float *vectors, int64_t n, int32_t iters, uint64_t seed,
                           void *stream)
{
    VG_CHECK(pq, VG_ERR_INVALID_ARG, "vg_pq_train: NULL quantizer");
    return vg_pq_train_subset(pq, vectors, n, iters, seed, 0, pq->m, stream);
}

VG_API int32_t vg_pq_encode(vg_pq *pq, const float *vectors, int64_t n, uint8_t *codes, void *stream)
{
    VG_CHECK(pq, VG_ERR_INVALID_ARG, "vg_pq_encode: NULL quantizer");
    VG_CHECK(pq->trained, VG_ERR_NOT_TRAINED, "ProductQuantizer not trained");
    VG_CHECK(n >= 0, VG_ERR_INVALID_ARG, "vg_pq_encode: n < 0");
    if (n == 0) return VG_OK;
    VG_CHECK(vectors && codes, VG_ERR_INVALID_ARG, "vg_pq_encode: NULL buffer");
    VG_HIP(hipSetDevice(pq->ctx->device));
    hipStream_t st = vg::pick_stream(pq->ctx, stream);
    vg::DevIn<float> v;
    vg::DevOut<uint8_t> c;
    VG_TRY(v.init(vectors, static_cast<size_t>(n) * pq->dim, st));
    VG_TRY(c.init(codes, static_cast<size_t>(n) * pq->m, st));
    const size_t lds = static_cast<size_t>(pq->k) * pq->subdim * sizeof(float);
    VG_CHECK(lds <= 64 * 1024, VG_ERR_UNSUPPORTED, "vg_pq_encode: codebook of one sub-quantizer exceeds 64 KiB");
    // grid.y = m <= 65535 is guaranteed by dim limits; grid.x up to 2^31
    const bool vec_ok = pq->dim % 4 == 0 && (reinterpret_cast<uintptr_t>(v.ptr) & 15) == 0;
    const unsigned gx_vec = static_cast<unsigned>((n + 256 * vg::kEncRows - 1) / (256 * vg::kEncRows));
    if (vec_ok && pq->subdim == 8)
        VG_LAUNCH(vg::pq_encode_vec_kernel<8>, dim3(gx_vec, pq->m), dim3(256), lds, st, v.ptr, n, pq->dim, pq->m, pq->k,
                  pq->d_codebooks, pq->d_scales, pq->d_offsets, c.ptr);
    else if (vec_ok && pq->subdim == 4)
        VG_LAUNCH(vg::pq_encode_vec_kernel<4>, dim3(gx_vec, pq->m), dim3(256), lds, st, v.ptr, n, pq->dim, pq->m, pq->k,
                  pq->d_codebooks, pq->d_scales, pq->d_offsets, c.ptr);
    else if (vec_ok && pq->subdim == 16)
        VG_LAUNCH(vg::pq_encode_vec_kernel<16>, dim3(gx_vec, pq->m), dim3(256), lds, st, v.ptr, n, pq->dim, pq->m,
                  pq->k, pq->d_codebooks, pq->d_scales, pq->d_offsets, c.ptr);
    else
        VG_LAUNCH(vg::pq_encode_kernel, dim3(static_cast<unsigned>((n + 255) / 256), pq->m), dim3(256),
                  lds, st, v.ptr, n, pq->dim, pq->m, pq->subdim, pq->k, pq->d_codebooks, pq->d_scales,
                  pq->d_offsets, c.ptr);
    VG_TRY(c.finish());
    if (c.on_host()) VG_HIP(hipStreamSynchronize(st));
    return VG_OK;
}

VG_API int32_t vg_pq_decode(vg_pq *pq, const uint8_t *codes, int64_t n, float *out, void *stream)
{
    VG_CHECK(pq, VG_ERR_INVALID_ARG, "vg_pq_decode: NULL quantizer");
    VG_CHECK(pq->trained, VG_ERR_NOT_TRAINED, "ProductQuantizer not trained");
    VG_CHECK(n >= 0, VG_ERR_INVALID_ARG, "vg_pq_decode: n < 0");
    if (n == 0) return VG_OK;
    VG_CHECK(codes && out, VG_ERR_INVALID_ARG, "vg_pq_decode: NULL buffer");
    VG_HIP(hipSetDevice(pq->ctx->device));
    hipStream_t st = vg::pick_stream(pq->ctx, stream);
    vg::DevIn<uint8_t> c;
    vg::DevOut<float> o;
    VG_TRY(c.init(codes, static_cast<size_t>(n) * pq->m, st));
    VG_TRY(o.init(out, static_cast<size_t>(n) * pq->dim, st));
    const int64_t total = n * pq->dim;
    VG_LAUNCH(vg::pq_decode_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0,
                       st, c.ptr, n, pq->dim, pq->m, pq->subdim, pq->k, pq->d_codebooks, pq->d_scales,
                       pq->d_offsets, o.ptr);
    VG_TRY(o.finish());
    if (o.on_host()) VG_HIP(hipStreamSynchronize(st));
    return VG_OK;
}

VG_API int32_t vg_pq_asymmetric_distance_batch(vg_pq *pq, const float *query, const uint8_t *codes,
                                               int64_t n, float *out, void *stream)
{
    VG_CHECK(pq, VG_ERR_INVALID_ARG, "vg_pq_asymmetric_distance_batch: NULL quantizer");
    VG_CHECK(pq->trained, VG_ERR_NOT_TRAINED, "ProductQuantizer not trained");
    VG_CHECK(n >= 0, VG_ERR_INVALID_ARG, "vg_pq_asymmetric_distance_batch: n < 0");
    if (n == 0) return VG_OK;
    VG_CHECK(query && codes && out, VG_ERR_INVALID_ARG, "vg_pq_asymmetric_distance_batch: NULL buffer");
    VG_HIP(hipSetDevice(pq->ctx->device));
    hipStream_t st = vg::pick_stream(pq->ctx, stream);
    vg::DevIn<float> q;
    vg::DevIn<uint8_t> c;
    vg::DevOut<float> o;
    VG_TRY(q.init(query, static_cast<size_t>(pq->dim), st));
    VG_TRY(c.init(codes, static_cast<size_t>(n) * pq->m, st));
    VG_TRY(o.init(out, static_cast<size_t>(n), st));
    VG_LAUNCH(vg::pq_asym_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, st,
                       q.ptr, c.ptr, n, pq->dim, pq->m, pq->subdim, pq->k, pq->d_codebooks, pq->d_scales,
                       pq->d_offsets, o.ptr);
    VG_TRY(o.finish());
    if (o.on_host()) VG_HIP(hipStreamSynchronize(st));
    return VG_OK;
}
