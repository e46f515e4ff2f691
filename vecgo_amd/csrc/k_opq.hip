// k_opq.hip — OptimizedProductQuantizer (internal/quantization/opq.go, svd.go): a block-diagonal rotation in
// front of the ProductQuantizer.
//   block size rule            opq.go:38-58   (multiple of the sub-vector size dividing dim, nearest 32)
//   rotateVector               opq.go:196-215 dst[b*bs + i] = simd.Dot(rotation_b[i], src_b)  (dotProductAvx512 order)
//   Encode / ComputeAsymmetricDistance   rotate, then the ProductQuantizer's (opq.go:218-231, :272-286)
//   Decode                     opq.go:234-269 PQ decode, then dst[i] = sum_j rotation[j][i] * src[j], left to right
//   Train                      opq.go:89-193  per outer iteration: rotate everything, ProductQuantizer.Train from
//                              scratch, M_b = sum_i x_i,b^T * yhat_i,b (yhat = Decode(Encode(rotated_i))), R_b =
//                              Procrustes(M_b) (svd.go: one-sided Jacobi SVD, R = U V^T, reflection fixed on the
//                              smallest singular value)
// Where the time goes: rotation of n vectors (a 32x32 block GEMV per block — 24 blocks at d = 768), the PQ training
// and encoding kernels of k_pq_train.hip / k_pq.hip, and the M accumulation.  The reference adds the n terms of
// every M entry in vector order in fp32; the kernel keeps that order — one thread owns one entry and walks the
// vectors, x and yhat blocks staged through LDS — so M, and with it the rotations, are bit-identical to a
// letter-by-letter CPU run (the test oracle).  The Procrustes solve itself is 24 matrices of 32 x 32: host code,
// the reference's arithmetic restated (fp32 ops separately rounded, float64 sqrt).
#include <algorithm>
#include <cmath>
#include <vector>

#include "vg_device.hpp"
#include "vg_internal.hpp"

namespace vg {

// dotProductAvx512 (floats_avx512.c:12-65) of two n-vectors by ONE thread: 4 x 16 lane accumulators over the
// 64-float blocks, (a0+a1)+(a2+a3), the reduce_add tree, then the FMA-contracted scalar tail.  n < 64 (the usual
// OPQ block of 32) is the tail alone.
__device__ inline float dot_avx512_order(const float *__restrict__ a, const float *__restrict__ b, int n)
{
    float total = 0.0f;
    int j = 0;
    if (n >= 64) {
        float acc[4][16];
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int l = 0; l < 16; l++) acc[k][l] = 0.0f;
        for (; j + 64 <= n; j += 64)
#pragma unroll
            for (int k = 0; k < 4; k++)
#pragma unroll
                for (int l = 0; l < 16; l++) acc[k][l] = __builtin_fmaf(a[j + k * 16 + l], b[j + k * 16 + l], acc[k][l]);
        float s[16];
#pragma unroll
        for (int l = 0; l < 16; l++) s[l] = (acc[0][l] + acc[1][l]) + (acc[2][l] + acc[3][l]);
        total = reduce16_regs(s);
    }
    for (; j < n; j++) total = __builtin_fmaf(a[j], b[j], total);
    return total;
}

// dst[row][b*bs + i] = Dot(rot[b][i], src[row][b*bs ..]); one thread per output element
__global__ void opq_rotate_kernel(const float *__restrict__ rot, int dim, int bs, const float *__restrict__ src, int64_t n,
                                  float *__restrict__ dst)
{
    const int64_t gid = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (gid >= n * dim) return;
    const int64_t row = gid / dim;
    const int e = static_cast<int>(gid % dim);
    const int b = e / bs, i = e % bs;
    dst[gid] = dot_avx512_order(rot + (static_cast<int64_t>(b) * bs + i) * bs, src + row * dim + static_cast<int64_t>(b) * bs, bs);
}

// dst[row][b*bs + i] = sum_j rot[b][j][i] * src[row][b*bs + j], j ascending, mul and add separately rounded
__global__ void opq_unrotate_kernel(const float *__restrict__ rot, int dim, int bs, const float *__restrict__ src, int64_t n,
                                    float *__restrict__ dst)
{
    const int64_t gid = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (gid >= n * dim) return;
    const int64_t row = gid / dim;
    const int e = static_cast<int>(gid % dim);
    const int b = e / bs, i = e % bs;
    const float *r = rot + static_cast<int64_t>(b) * bs * bs;
    const float *s = src + row * dim + static_cast<int64_t>(b) * bs;
    float sum = 0.0f;
    for (int j = 0; j < bs; j++) {
        const float p = r[static_cast<int64_t>(j) * bs + i] * s[j];
        sum = sum + p;
    }
    dst[gid] = sum;
}

// M_b[r][c] = sum_i x[i][b*bs + r] * y[i][b*bs + c], i ascending (opq.go:150-176).  One workgroup per block b; a thread
// owns entries e = tid, tid + 256, ... of the bs x bs matrix and walks the vectors, 32 at a time through LDS.
constexpr int kOpqThreads = 256;
constexpr int kOpqChunk = 32;
constexpr int kOpqMaxPerThread = 16;  // bs <= 64

__global__ __launch_bounds__(kOpqThreads) void opq_accumulate_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                                     int64_t n, int dim, int bs, float *__restrict__ m_out)
{
    extern __shared__ float sh[];  // [2][kOpqChunk][bs]
    float *sx = sh, *sy = sh + kOpqChunk * bs;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int ne = bs * bs;
    float acc[kOpqMaxPerThread];
#pragma unroll
    for (int u = 0; u < kOpqMaxPerThread; u++) acc[u] = 0.0f;
    for (int64_t i0 = 0; i0 < n; i0 += kOpqChunk) {
        const int cnt = static_cast<int>(n - i0 < kOpqChunk ? n - i0 : kOpqChunk);
        for (int t = tid; t < cnt * bs; t += kOpqThreads) {
            const int64_t src = (i0 + t / bs) * dim + static_cast<int64_t>(b) * bs + t % bs;
            sx[t] = x[src];
            sy[t] = y[src];
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < kOpqMaxPerThread; u++) {
            const int e = tid + u * kOpqThreads;
            if (e < ne) {
                const int r = e / bs, c = e % bs;
                float a = acc[u];
                for (int i = 0; i < cnt; i++) {
                    const float p = sx[i * bs + r] * sy[i * bs + c];
                    a = a + p;
                }
                acc[u] = a;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < kOpqMaxPerThread; u++) {
        const int e = tid + u * kOpqThreads;
        if (e < ne) m_out[static_cast<int64_t>(b) * ne + e] = acc[u];
    }
}

// ---- svd.go on the host ---------------------------------------------------------------------------------
static bool jacobi_sweep(float *u, float *v, int m, int n, double tol)
{
    bool changed = false;
    for (int i = 0; i < n - 1; i++)
        for (int j = i + 1; j < n; j++) {
            float alpha = 0.0f, beta = 0.0f, gamma = 0.0f;
            for (int k = 0; k < m; k++) {
                const float a = u[k * n + i], b = u[k * n + j];
                const float aa = a * a, bb = b * b, ab = a * b;
                alpha = alpha + aa;
                beta = beta + bb;
                gamma = gamma + ab;
            }
            if (alpha < 1e-12f || beta < 1e-12f) continue;
            const float ab = alpha * beta;
            if (std::fabs(static_cast<double>(gamma)) < tol * std::sqrt(static_cast<double>(ab))) continue;
            changed = true;
            const float num = beta - alpha, den = 2.0f * gamma;
            const float zeta = num / den;
            const float zz = zeta * zeta;
            const float one_zz = 1.0f + zz;
            const float root = static_cast<float>(std::sqrt(static_cast<double>(one_zz)));
            const float t = zeta > 0.0f ? 1.0f / (zeta + root) : -1.0f / (-zeta + root);
            const float tt = t * t;
            const float one_tt = 1.0f + tt;
            const float c = 1.0f / static_cast<float>(std::sqrt(static_cast<double>(one_tt)));
            const float s = c * t;
            for (int k = 0; k < m; k++) {
                const float t1 = u[k * n + i], t2 = u[k * n + j];
                const float c1 = c * t1, s2 = s * t2, s1 = s * t1, c2 = c * t2;
                u[k * n + i] = c1 - s2;
                u[k * n + j] = s1 + c2;
            }
            for (int k = 0; k < n; k++) {
                const float t1 = v[k * n + i], t2 = v[k * n + j];
                const float c1 = c * t1, s2 = s * t2, s1 = s * t1, c2 = c * t2;
                v[k * n + i] = c1 - s2;
                v[k * n + j] = s1 + c2;
            }
        }
    return changed;
}

static float determinant(const float *matrix, int n)
{
    if (n == 0) return 0.0f;
    std::vector<float> t(matrix, matrix + static_cast<size_t>(n) * n);
    std::vector<int> rowp(static_cast<size_t>(n));
    for (int i = 0; i < n; i++) rowp[i] = i;
    float det = 1.0f;
    for (int i = 0; i < n; i++) {
        int pivot = i;
        for (int j = i + 1; j < n; j++)
            if (std::fabs(static_cast<double>(t[rowp[j] * n + i])) > std::fabs(static_cast<double>(t[rowp[pivot] * n + i]))) pivot = j;
        if (pivot != i) {
            std::swap(rowp[i], rowp[pivot]);
            det = det * -1.0f;
        }
        float *ri = t.data() + static_cast<size_t>(rowp[i]) * n;
        if (ri[i] == 0.0f) return 0.0f;
        det = det * ri[i];
        for (int j = i + 1; j < n; j++) {
            float *rj = t.data() + static_cast<size_t>(rowp[j]) * n;
            const float factor = rj[i] / ri[i];
            for (int k = i + 1; k < n; k++) {
                const float p = factor * ri[k];
                rj[k] = rj[k] - p;
            }
        }
    }
    return det;
}

// computeProcrustesRotation (svd.go:126-178); mm (n x n) is destroyed
static void procrustes(float *mm, int n, float *r)
{
    std::vector<float> v(static_cast<size_t>(n) * n, 0.0f), sigma(static_cast<size_t>(n));
    for (int i = 0; i < n; i++) v[i * n + i] = 1.0f;
    float *u = mm;
    for (int it = 0; it < 100; it++)
        if (!jacobi_sweep(u, v.data(), n, n, 1e-5)) break;
    for (int j = 0; j < n; j++) {
        float sum = 0.0f;
        for (int i = 0; i < n; i++) {
            const float p = u[i * n + j] * u[i * n + j];
            sum = sum + p;
        }
        sigma[j] = static_cast<float>(std::sqrt(static_cast<double>(sum)));
        if (sigma[j] > 1e-10f) {
            const float inv = 1.0f / sigma[j];
            for (int i = 0; i < n; i++) u[i * n + j] = u[i * n + j] * inv;
        }
    }
    int min_idx = 0;
    float min_sigma = sigma[0];
    for (int i = 1; i < n; i++)
        if (sigma[i] < min_sigma) {
            min_sigma = sigma[i];
            min_idx = i;
        }
    for (int pass = 0; pass < 2; pass++) {
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) {
                float sum = 0.0f;
                for (int k = 0; k < n; k++) {
                    const float p = u[i * n + k] * v[j * n + k];
                    sum = sum + p;
                }
                r[i * n + j] = sum;
            }
        if (pass == 1 || !(determinant(r, n) < 0.0f)) break;
        for (int i = 0; i < n; i++) u[i * n + min_idx] = u[i * n + min_idx] * -1.0f;
    }
}

}  // namespace vg

struct vg_opq {
    vg_ctx *ctx = nullptr;
    vg_pq *pq = nullptr;
    int32_t dim = 0, m = 0, k = 0, block = 0, nblocks = 0, iters = 0;
    bool trained = false;
    float *d_rot = nullptr;         // nblocks * block * block
    std::vector<float> h_rot;
};

VG_API int32_t vg_opq_block_size(int32_t dim, int32_t m)
{
    if (dim <= 0 || m <= 0 || dim % m) return 0;
    const int sub = dim / m;
    int block = dim;
    if (dim > 64) {
        int best = 1000;
        for (int b = sub; b <= dim; b += sub)
            if (dim % b == 0) {
                const int diff = std::abs(b - 32);
                if (diff < best) {
                    best = diff;
                    block = b;
                }
            }
    }
    return block;
}

static int32_t opq_upload_rotations(vg_opq *o, hipStream_t st)
{
    VG_HIP(hipMemcpyAsync(o->d_rot, o->h_rot.data(), o->h_rot.size() * 4, hipMemcpyHostToDevice, st));
    VG_HIP(hipStreamSynchronize(st));
    return VG_OK;
}

VG_API int32_t vg_opq_create(vg_ctx *ctx, int32_t dim, int32_t m, int32_t k, int32_t num_iterations, vg_opq **out)
{
    VG_CHECK(ctx && out, VG_ERR_INVALID_ARG, "vg_opq_create: NULL argument");
    VG_CHECK(dim > 0 && m > 0 && dim % m == 0, VG_ERR_INVALID_ARG, "dimension must be divisible by numSubvectors");
    VG_CHECK(num_iterations >= 0, VG_ERR_INVALID_ARG, "vg_opq_create: negative iteration count");
    const int block = vg_opq_block_size(dim, m);
    VG_CHECK(block <= 64, VG_ERR_UNSUPPORTED, "vg_opq_create: rotation block of %d dimensions exceeds 64", block);
    vg_pq *pq = nullptr;
    VG_TRY(vg_pq_create(ctx, dim, m, k, &pq));
    vg_opq *o = new vg_opq;
    o->ctx = ctx;
    o->pq = pq;
    o->dim = dim;
    o->m = m;
    o->k = k;
    o->block = block;
    o->nblocks = dim / block;
    o->iters = num_iterations;
    o->h_rot.assign(static_cast<size_t>(o->nblocks) * block * block, 0.0f);
    for (int b = 0; b < o->nblocks; b++)
        for (int i = 0; i < block; i++) o->h_rot[(static_cast<size_t>(b) * block + i) * block + i] = 1.0f;  // identityMatrix
    hipError_t e = hipSetDevice(ctx->device);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&o->d_rot), o->h_rot.size() * 4);
    if (e != hipSuccess) {
        vg::set_error("vg_opq_create: %s", hipGetErrorString(e));
        vg_pq_destroy(pq);
        delete o;
        return e == hipErrorOutOfMemory ? VG_ERR_OUT_OF_MEMORY : VG_ERR_HIP;
    }
    const int32_t s = opq_upload_rotations(o, ctx->stream);
    if (s != VG_OK) {
        (void)hipFree(o->d_rot);
        vg_pq_destroy(pq);
        delete o;
        return s;
    }
    *out = o;
    return VG_OK;
}

VG_API int32_t vg_opq_destroy(vg_opq *o)
{
    if (!o) return VG_OK;
    if (o->d_rot) (void)hipFree(o->d_rot);
    vg_pq_destroy(o->pq);
    delete o;
    return VG_OK;
}

VG_API vg_pq *vg_opq_pq(vg_opq *o) { return o ? o->pq : nullptr; }
VG_API int32_t vg_opq_is_trained(vg_opq *o) { return o && o->trained ? 1 : 0; }

VG_API int32_t vg_opq_get_rotations(vg_opq *o, int32_t *block, int32_t *nblocks, float *rotations)
{
    VG_CHECK(o, VG_ERR_INVALID_ARG, "vg_opq_get_rotations: NULL handle");
    if (block) *block = o->block;
    if (nblocks) *nblocks = o->nblocks;
    if (rotations) std::memcpy(rotations, o->h_rot.data(), o->h_rot.size() * 4);
    return VG_OK;
}

VG_API int32_t vg_opq_set_rotations(vg_opq *o, const float *rotations)
{
    VG_CHECK(o && rotations, VG_ERR_INVALID_ARG, "vg_opq_set_rotations: NULL argument");
    VG_HIP(hipSetDevice(o->ctx->device));
    std::memcpy(o->h_rot.data(), rotations, o->h_rot.size() * 4);
    VG_TRY(opq_upload_rotations(o, o->ctx->stream));
    o->trained = true;  // as Train does before its first iteration (opq.go:102)
    return VG_OK;
}

static int32_t opq_rotate_dev(vg_opq *o, const float *d_src, int64_t n, float *d_dst, bool inverse, hipStream_t st)
{
    const int64_t total = n * o->dim;
    if (total == 0) return VG_OK;
    if (inverse)
        VG_LAUNCH(vg::opq_unrotate_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, st, o->d_rot, o->dim,
                  o->block, d_src, n, d_dst);
    else
        VG_LAUNCH(vg::opq_rotate_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, st, o->d_rot, o->dim,
                  o->block, d_src, n, d_dst);
    return VG_OK;
}

VG_API int32_t vg_opq_rotate(vg_opq *o, const float *vectors, int64_t n, float *out, void *stream)
{
    VG_CHECK(o, VG_ERR_INVALID_ARG, "vg_opq_rotate: NULL handle");
    VG_CHECK(n >= 0, VG_ERR_INVALID_ARG, "vg_opq_rotate: negative n");
    if (n == 0) return VG_OK;
    VG_CHECK(vectors && out, VG_ERR_INVALID_ARG, "vg_opq_rotate: NULL buffer");
    VG_HIP(hipSetDevice(o->ctx->device));
    hipStream_t st = vg::pick_stream(o->ctx, stream);
    vg::DevIn<float> v;
    vg::DevOut<float> d;
    VG_TRY(v.init(vectors, static_cast<size_t>(n) * o->dim, st));
    VG_TRY(d.init(out, static_cast<size_t>(n) * o->dim, st));
    VG_TRY(opq_rotate_dev(o, v.ptr, n, d.ptr, false, st));
    VG_TRY(d.finish());
    if (d.on_host()) VG_HIP(hipStreamSynchronize(st));
    return VG_OK;
}

VG_API int32_t vg_opq_encode(vg_opq *o, const float *vectors, int64_t n, uint8_t *codes, void *stream)
{
    VG_CHECK(o, VG_ERR_INVALID_ARG, "vg_opq_encode: NULL handle");
    VG_CHECK(o->trained, VG_ERR_NOT_READY, "OptimizedProductQuantizer not trained");
    VG_CHECK(n >= 0, VG_ERR_INVALID_ARG, "vg_opq_encode: negative n");
    if (n == 0) return VG_OK;
    VG_CHECK(vectors && codes, VG_ERR_INVALID_ARG, "vg_opq_encode: NULL buffer");
    VG_HIP(hipSetDevice(o->ctx->device));
    hipStream_t st = vg::pick_stream(o->ctx, stream);
    vg::DevIn<float> v;
    vg::DevTmp<float> rotated;
    VG_TRY(v.init(vectors, static_cast<size_t>(n) * o->dim, st));
    VG_TRY(rotated.init(static_cast<size_t>(n) * o->dim, st));
    VG_TRY(opq_rotate_dev(o, v.ptr, n, rotated.ptr, false, st));
    return vg_pq_encode(o->pq, rotated.ptr, n, codes, stream);
}

VG_API int32_t vg_opq_decode(vg_opq *o, const uint8_t *codes, int64_t n, float *out, void *stream)
{
    VG_CHECK(o, VG_ERR_INVALID_ARG, "vg_opq_decode: NULL handle");
    VG_CHECK(o->trained, VG_ERR_NOT_READY, "OptimizedProductQuantizer not trained");
    VG_CHECK(n >= 0, VG_ERR_INVALID_ARG, "vg_opq_decode: negative n");
    if (n == 0) return VG_OK;
    VG_CHECK(codes && out, VG_ERR_INVALID_ARG, "vg_opq_decode: NULL buffer");
    VG_HIP(hipSetDevice(o->ctx->device));
    hipStream_t st = vg::pick_stream(o->ctx, stream);
    vg::DevTmp<float> rotated;
    vg::DevOut<float> d;
    VG_TRY(rotated.init(static_cast<size_t>(n) * o->dim, st));
    VG_TRY(d.init(out, static_cast<size_t>(n) * o->dim, st));
    VG_TRY(vg_pq_decode(o->pq, codes, n, rotated.ptr, stream));
    VG_TRY(opq_rotate_dev(o, rotated.ptr, n, d.ptr, true, st));
    VG_TRY(d.finish());
    if (d.on_host()) VG_HIP(hipStreamSynchronize(st));
    return VG_OK;
}

VG_API int32_t vg_opq_asymmetric_distance_batch(vg_opq *o, const float *query, const uint8_t *codes, int64_t n,
                                                float *out, void *stream)
{
    VG_CHECK(o, VG_ERR_INVALID_ARG, "vg_opq_asymmetric_distance_batch: NULL handle");
    VG_CHECK(o->trained, VG_ERR_NOT_READY, "OptimizedProductQuantizer not trained");
    VG_CHECK(query, VG_ERR_INVALID_ARG, "vg_opq_asymmetric_distance_batch: NULL query");
    VG_HIP(hipSetDevice(o->ctx->device));
    hipStream_t st = vg::pick_stream(o->ctx, stream);
    vg::DevIn<float> q;
    vg::DevTmp<float> rq;
    VG_TRY(q.init(query, static_cast<size_t>(o->dim), st));
    VG_TRY(rq.init(static_cast<size_t>(o->dim), st));
    VG_TRY(opq_rotate_dev(o, q.ptr, 1, rq.ptr, false, st));
    return vg_pq_asymmetric_distance_batch(o->pq, rq.ptr, codes, n, out, stream);
}

VG_API int32_t vg_opq_train(vg_opq *o, const float *vectors, int64_t n, int32_t pq_iters, uint64_t seed, void *stream)
{
    VG_CHECK(o, VG_ERR_INVALID_ARG, "vg_opq_train: NULL handle");
    VG_CHECK(n > 0, VG_ERR_INVALID_ARG, "no vectors provided for training");
    VG_CHECK(vectors, VG_ERR_INVALID_ARG, "vg_opq_train: NULL vectors");
    VG_HIP(hipSetDevice(o->ctx->device));
    hipStream_t st = vg::pick_stream(o->ctx, stream);
    const int bs = o->block, nb = o->nblocks;
    const size_t count = static_cast<size_t>(n) * o->dim;
    vg::DevIn<float> v;
    VG_TRY(v.init(vectors, count, st));
    vg::DevTmp<float> rotated, recon, d_m;
    vg::DevTmp<uint8_t> codes;
    VG_TRY(rotated.init(count, st));
    VG_TRY(recon.init(count, st));
    VG_TRY(codes.init(static_cast<size_t>(n) * o->m, st));
    VG_TRY(d_m.init(static_cast<size_t>(nb) * bs * bs, st));
    // rotations back to identity, trained from here on (opq.go:98-102)
    std::fill(o->h_rot.begin(), o->h_rot.end(), 0.0f);
    for (int b = 0; b < nb; b++)
        for (int i = 0; i < bs; i++) o->h_rot[(static_cast<size_t>(b) * bs + i) * bs + i] = 1.0f;
    VG_TRY(opq_upload_rotations(o, st));
    o->trained = true;
    std::vector<float> h_m(static_cast<size_t>(nb) * bs * bs);
    const size_t lds = 2 * vg::kOpqChunk * static_cast<size_t>(bs) * sizeof(float);
    for (int it = 0; it < o->iters; it++) {
        VG_TRY(opq_rotate_dev(o, v.ptr, n, rotated.ptr, false, st));
        VG_TRY(vg_pq_train(o->pq, rotated.ptr, n, pq_iters, seed + static_cast<uint64_t>(it), stream));
        VG_TRY(vg_pq_encode(o->pq, rotated.ptr, n, codes.ptr, stream));
        VG_TRY(vg_pq_decode(o->pq, codes.ptr, n, recon.ptr, stream));
        VG_LAUNCH(vg::opq_accumulate_kernel, dim3(static_cast<unsigned>(nb)), dim3(vg::kOpqThreads), lds, st, v.ptr, recon.ptr, n,
                  o->dim, bs, d_m.ptr);
        VG_HIP(hipMemcpyAsync(h_m.data(), d_m.ptr, h_m.size() * 4, hipMemcpyDeviceToHost, st));
        VG_HIP(hipStreamSynchronize(st));
        for (int b = 0; b < nb; b++)
            vg::procrustes(h_m.data() + static_cast<size_t>(b) * bs * bs, bs, o->h_rot.data() + static_cast<size_t>(b) * bs * bs);
        VG_TRY(opq_upload_rotations(o, st));
    }
    return VG_OK;
}
