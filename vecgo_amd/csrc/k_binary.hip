// k_binary.hip — BinaryQuantizer (internal/quantization/binary.go:23-262) and NormalizeL2InPlace
// (distance/distance.go:40-53 -> dotProductAvx512, simd.Sqrt, scaleAvx512 floats_avx512.c:174-218).
//
//   Train        threshold = float32(sum of every component as float64 / count)        binary.go:59-79
//   Encode       bit i of the code = (v[i] >= threshold); ceil(dim/64) little-endian uint64 words
//                (Encode's bytes and EncodeUint64Into's words are the same memory)    binary.go:86-154
//   Decode       threshold +- 0.5                                                       binary.go:173-188
//   ComputeHammingDistance   encode the query, popcount(xor)                            binary.go:158-171, :221-239
//
// Train's reference is ONE float64 accumulator walked over n*dim values in order.  Here every workgroup sums a
// contiguous chunk in float64 and the chunk sums are added in chunk order: a different rounding sequence, i.e.
// the threshold can differ from the sequential one in the last bits of the float64 — after the conversion to
// float32 the two agree except when the float64 mean sits within ~1e-13 (relative) of a float32 rounding
// boundary.  Parity for Train is therefore "equal on every tested input, <= 1 float32 ulp by construction";
// everything else in this file is bit-exact.
#include "vg_device.hpp"
#include "vg_exact.hpp"
#include "vg_internal.hpp"

namespace vg {

constexpr int kBinThreads = 256;

__global__ __launch_bounds__(kBinThreads) void binary_sum_kernel(const float *__restrict__ v, int64_t count, int64_t per_block,
                                                                 double *__restrict__ partial)
{
    __shared__ double sh[kBinThreads];
    const int64_t lo = static_cast<int64_t>(blockIdx.x) * per_block;
    const int64_t hi = lo + per_block < count ? lo + per_block : count;
    double s = 0.0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += kBinThreads) s += static_cast<double>(v[i]);
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = kBinThreads / 2; w > 0; w >>= 1) {
        if (static_cast<int>(threadIdx.x) < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}

__global__ void binary_finish_kernel(const double *__restrict__ partial, int blocks, int64_t count, float *__restrict__ threshold)
{
    if (threadIdx.x || blockIdx.x) return;
    double s = 0.0;
    for (int b = 0; b < blocks; b++) s += partial[b];
    *threshold = static_cast<float>(s / static_cast<double>(count));
}

// 16 lanes per vector, lane L owns elements 4L..4L+3 of every 64-element word (one nibble), as rabitq_encode_kernel
__global__ __launch_bounds__(256) void binary_encode_kernel(const float *__restrict__ vectors, int64_t n, int dim,
                                                            const float *__restrict__ threshold_dev, float threshold_host,
                                                            uint8_t *__restrict__ codes)
{
    const int64_t row = static_cast<int64_t>(blockIdx.x) * 16 + (threadIdx.x >> 4);
    if (row >= n) return;
    const float th = threshold_dev ? *threshold_dev : threshold_host;
    const int L = threadIdx.x & 15;
    const float *v = vectors + row * dim;
    const int nw = (dim + 63) / 64;
    uint8_t *out = codes + row * static_cast<int64_t>(nw) * 8;
    for (int w = 0; w < nw; w++) {
        uint32_t nib = 0;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int e = w * 64 + 4 * L + t;
            if (e < dim && v[e] >= th) nib |= 1u << t;
        }
        const uint32_t other = static_cast<uint32_t>(
            __builtin_amdgcn_update_dpp(0, static_cast<int>(nib), kDppQuadXor1, 0xF, 0xF, false));
        if ((L & 1) == 0) out[w * 8 + (L >> 1)] = static_cast<uint8_t>(nib | (other << 4));
    }
}

__global__ void binary_decode_kernel(const uint8_t *__restrict__ codes, int64_t n, int dim, int code_bytes, float threshold,
                                     float *__restrict__ out)
{
    const int64_t gid = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (gid >= n * dim) return;
    const int64_t row = gid / dim;
    const int i = static_cast<int>(gid % dim);
    const int byte = i / 8;
    const bool set = byte < code_bytes && (codes[row * code_bytes + byte] & (1u << (i % 8))) != 0;
    out[gid] = set ? threshold + 0.5f : threshold - 0.5f;
}

// HammingDistance(query words, code words) for n codes: one lane per code, dwords
__global__ void binary_hamming_kernel(const uint8_t *__restrict__ qcode, const uint8_t *__restrict__ codes, int64_t n,
                                      int words, int32_t *__restrict__ out)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t *c = reinterpret_cast<const uint64_t *>(codes) + i * words;
    const uint64_t *q = reinterpret_cast<const uint64_t *>(qcode);
    int h = 0;
    for (int w = 0; w < words; w++) h += __popcll(c[w] ^ q[w]);
    out[i] = h;
}

// NormalizeL2InPlace of n rows: 16 lanes per row; ok[row] = 0 for a zero norm (row left untouched)
__global__ __launch_bounds__(256) void normalize_l2_kernel(float *__restrict__ vectors, int64_t n, int dim,
                                                           uint8_t *__restrict__ ok)
{
    const int64_t row = static_cast<int64_t>(blockIdx.x) * 16 + (threadIdx.x >> 4);
    if (row >= n) return;
    const Sub16 sub = Sub16::make(threadIdx.x);
    const int L = threadIdx.x & 15;
    float *v = vectors + row * dim;
    const float norm2 = exact_pair16<true, kPair>(v, v, dim, sub);  // simd.Dot(v, v)
    if (norm2 == 0.0f) {
        if (L == 0 && ok) ok[row] = 0;
        return;
    }
    const float inv = 1.0f / static_cast<float>(sqrt(static_cast<double>(norm2)));  // 1 / simd.Sqrt(norm2)
    for (int i = L; i < dim; i += 16) v[i] = v[i] * inv;                                // scaleAvx512: a[i] *= s
    if (L == 0 && ok) ok[row] = 1;
}

}  // namespace vg

VG_API int64_t vg_binary_code_bytes(int32_t dim) { return static_cast<int64_t>((dim + 63) / 64) * 8; }

VG_API int32_t vg_binary_train(vg_ctx *ctx, int32_t dim, const float *vectors, int64_t n, float *threshold, void *stream)
{
    VG_CHECK(ctx, VG_ERR_INVALID_ARG, "vg_binary_train: ctx is NULL");
    VG_CHECK(threshold, VG_ERR_INVALID_ARG, "vg_binary_train: NULL threshold");
    VG_CHECK(n > 0, VG_ERR_INVALID_ARG, "no vectors provided for training");
    VG_CHECK(dim > 0 && vectors, VG_ERR_INVALID_ARG, "vg_binary_train: bad dim or NULL vectors");
    VG_HIP(hipSetDevice(ctx->device));
    hipStream_t st = vg::pick_stream(ctx, stream);
    const int64_t count = n * dim;
    vg::DevIn<float> v;
    vg::DevOut<float> th;
    VG_TRY(v.init(vectors, static_cast<size_t>(count), st));
    VG_TRY(th.init(threshold, 1, st));
    const int blocks = static_cast<int>(std::min<int64_t>(1024, (count + 65535) / 65536));
    const int64_t per_block = (count + blocks - 1) / blocks;
    vg::DevTmp<double> partial;
    VG_TRY(partial.init(static_cast<size_t>(blocks), st));
    VG_LAUNCH(vg::binary_sum_kernel, dim3(static_cast<unsigned>(blocks)), dim3(vg::kBinThreads), 0, st, v.ptr, count, per_block,
              partial.ptr);
    VG_LAUNCH(vg::binary_finish_kernel, dim3(1), dim3(1), 0, st, partial.ptr, blocks, count, th.ptr);
    VG_TRY(th.finish());
    if (th.on_host()) VG_HIP(hipStreamSynchronize(st));
    return VG_OK;
}

VG_API int32_t vg_binary_encode(vg_ctx *ctx, int32_t dim, float threshold, const float *vectors, int64_t n, uint8_t *codes,
                                void *stream)
{
    VG_CHECK(ctx, VG_ERR_INVALID_ARG, "vg_binary_encode: ctx is NULL");
    VG_CHECK(dim > 0 && n >= 0, VG_ERR_INVALID_ARG, "vg_binary_encode: bad dim or n");
    if (n == 0) return VG_OK;
    VG_CHECK(vectors && codes, VG_ERR_INVALID_ARG, "vg_binary_encode: NULL buffer");
    VG_HIP(hipSetDevice(ctx->device));
    hipStream_t st = vg::pick_stream(ctx, stream);
    vg::DevIn<float> v;
    vg::DevOut<uint8_t> c;
    VG_TRY(v.init(vectors, static_cast<size_t>(n) * dim, st));
    VG_TRY(c.init(codes, static_cast<size_t>(n) * vg_binary_code_bytes(dim), st));
    VG_LAUNCH(vg::binary_encode_kernel, dim3(static_cast<unsigned>((n + 15) / 16)), dim3(256), 0, st, v.ptr, n, dim,
              static_cast<const float *>(nullptr), threshold, c.ptr);
    VG_TRY(c.finish());
    if (c.on_host()) VG_HIP(hipStreamSynchronize(st));
    return VG_OK;
}

VG_API int32_t vg_binary_decode(vg_ctx *ctx, int32_t dim, float threshold, const uint8_t *codes, int64_t n,
                                int32_t code_bytes, float *out, void *stream)
{
    VG_CHECK(ctx, VG_ERR_INVALID_ARG, "vg_binary_decode: ctx is NULL");
    VG_CHECK(dim > 0 && n >= 0 && code_bytes >= 0, VG_ERR_INVALID_ARG, "vg_binary_decode: bad dim, n or code_bytes");
    if (n == 0) return VG_OK;
    VG_CHECK(out && (codes || code_bytes == 0), VG_ERR_INVALID_ARG, "vg_binary_decode: NULL buffer");
    VG_HIP(hipSetDevice(ctx->device));
    hipStream_t st = vg::pick_stream(ctx, stream);
    vg::DevIn<uint8_t> c;
    vg::DevOut<float> o;
    VG_TRY(c.init(codes, static_cast<size_t>(n) * code_bytes, st));
    VG_TRY(o.init(out, static_cast<size_t>(n) * dim, st));
    const int64_t total = n * dim;
    VG_LAUNCH(vg::binary_decode_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, st, c.ptr, n, dim,
              code_bytes, threshold, o.ptr);
    VG_TRY(o.finish());
    if (o.on_host()) VG_HIP(hipStreamSynchronize(st));
    return VG_OK;
}

VG_API int32_t vg_binary_hamming_batch(vg_ctx *ctx, int32_t dim, float threshold, const float *query, const uint8_t *codes,
                                       int64_t n, int32_t *out, void *stream)
{
    VG_CHECK(ctx, VG_ERR_INVALID_ARG, "vg_binary_hamming_batch: ctx is NULL");
    VG_CHECK(dim > 0 && n >= 0, VG_ERR_INVALID_ARG, "vg_binary_hamming_batch: bad dim or n");
    if (n == 0) return VG_OK;
    VG_CHECK(query && codes && out, VG_ERR_INVALID_ARG, "vg_binary_hamming_batch: NULL buffer");
    VG_HIP(hipSetDevice(ctx->device));
    hipStream_t st = vg::pick_stream(ctx, stream);
    const int64_t cb = vg_binary_code_bytes(dim);
    vg::DevIn<float> q;
    vg::DevIn<uint8_t> c;
    vg::DevOut<int32_t> o;
    vg::DevTmp<uint8_t> qc;
    VG_TRY(q.init(query, static_cast<size_t>(dim), st));
    VG_TRY(c.init(codes, static_cast<size_t>(n) * cb, st));
    VG_TRY(o.init(out, static_cast<size_t>(n), st));
    VG_TRY(qc.init(static_cast<size_t>(cb), st));
    VG_LAUNCH(vg::binary_encode_kernel, dim3(1), dim3(256), 0, st, q.ptr, int64_t(1), dim, static_cast<const float *>(nullptr),
              threshold, qc.ptr);
    VG_LAUNCH(vg::binary_hamming_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, st, qc.ptr, c.ptr, n,
              static_cast<int>(cb / 8), o.ptr);
    VG_TRY(o.finish());
    if (o.on_host()) VG_HIP(hipStreamSynchronize(st));
    return VG_OK;
}

VG_API int32_t vg_normalize_l2(vg_ctx *ctx, float *vectors, int64_t n, int32_t dim, uint8_t *ok, void *stream)
{
    VG_CHECK(ctx, VG_ERR_INVALID_ARG, "vg_normalize_l2: ctx is NULL");
    VG_CHECK(dim >= 0 && n >= 0, VG_ERR_INVALID_ARG, "vg_normalize_l2: bad dim or n");
    if (n == 0) return VG_OK;
    VG_HIP(hipSetDevice(ctx->device));
    hipStream_t st = vg::pick_stream(ctx, stream);
    if (dim == 0) {  // len(v) == 0 -> false (distance.go:43-45)
        if (ok) {
            vg::DevOut<uint8_t> o0;
            VG_TRY(o0.init(ok, static_cast<size_t>(n), st));
            VG_HIP(hipMemsetAsync(o0.ptr, 0, static_cast<size_t>(n), st));
            VG_TRY(o0.finish());
        }
        return VG_OK;
    }
    VG_CHECK(vectors, VG_ERR_INVALID_ARG, "vg_normalize_l2: NULL vectors");
    const size_t count = static_cast<size_t>(n) * dim;
    float *dv = vectors;
    vg::DevTmp<float> staged;
    const bool host = !vg::is_device_ptr(vectors);
    if (host) {
        VG_TRY(staged.init(count, st));
        VG_HIP(hipMemcpyAsync(staged.ptr, vectors, count * 4, hipMemcpyHostToDevice, st));
        dv = staged.ptr;
    }
    vg::DevOut<uint8_t> o;
    VG_TRY(o.init(ok, ok ? static_cast<size_t>(n) : 0, st));
    VG_LAUNCH(vg::normalize_l2_kernel, dim3(static_cast<unsigned>((n + 15) / 16)), dim3(256), 0, st, dv, n, dim, o.ptr);
    if (host) VG_HIP(hipMemcpyAsync(vectors, dv, count * 4, hipMemcpyDeviceToHost, st));
    VG_TRY(o.finish());
    if (host || o.on_host()) VG_HIP(hipStreamSynchronize(st));
    return VG_OK;
}
