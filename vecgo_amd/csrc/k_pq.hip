// k_pq.hip — quantization.ProductQuantizer on the device (internal/quantization/pq.go).
#include "vg_device.hpp"
#include "vg_hnsw_layer.hpp"
#include "vg_internal.hpp"

namespace vg {

// One (query, subspace) per workgroup, one centroid per thread.
// Entry = squaredL2Int8DequantizedGeneric (internal/simd/kernels.go:354-362):
//   v = float32(code)*scale + offset ; d = q - v ; sum += d*d   — five separately
// rounded fp32 ops per term, sequential over the sub-dimension (Go on amd64 does not
// fuse; this TU is built with -ffp-contract=off).
// SCAN_LAYOUT: write the image the ADC scan keeps in LDS (k_adc.hip): the full 16-wide groups
// in pairs, word = ((g>>1)*256 + c)*32 + (g&1)*16 + l (an odd last group leaves half a block
// unused), then the m%16 tail rows as [j][c].  Same values.
template <bool SCAN_LAYOUT>
__global__ void pq_build_table_kernel(const float *__restrict__ queries,
                                      const int8_t *__restrict__ codebooks,
                                      const float *__restrict__ scales,
                                      const float *__restrict__ offsets, int dim, int m, int k,
                                      int subdim, float *__restrict__ tables)
{
    const int q = blockIdx.y;
    const int j = blockIdx.x;
    const float scale = scales[j];
    const float offset = offsets[j];
    const float *qs = queries + static_cast<int64_t>(q) * dim + j * subdim;
    for (int c = threadIdx.x; c < k; c += blockDim.x) {
        const int8_t *cb = codebooks + (static_cast<int64_t>(j) * k + c) * subdim;
        float sum = 0.0f;
        for (int i = 0; i < subdim; i++) {
            float v = static_cast<float>(cb[i]) * scale;
            v = v + offset;
            float d = qs[i] - v;
            float dd = d * d;
            sum = sum + dd;
        }
        if (SCAN_LAYOUT) {
            const int gfull = m >> 4;
            const int64_t words = static_cast<int64_t>((gfull + 1) >> 1) * 8192 + static_cast<int64_t>(m & 15) * 256;
            int64_t at;
            if (j < gfull * 16)
                at = (static_cast<int64_t>(j >> 5) * 256 + c) * 32 + ((j >> 4) & 1) * 16 + (j & 15);
            else
                at = static_cast<int64_t>((gfull + 1) >> 1) * 8192 + static_cast<int64_t>(j - gfull * 16) * 256 + c;
            tables[static_cast<int64_t>(q) * words + at] = sum;
        } else {
            tables[static_cast<int64_t>(q) * m * k + static_cast<int64_t>(j) * k + c] = sum;
        }
    }
}

// The graph walks' tables ([sub-quantizer][centroid], k = 256, sub-dimension 8): one workgroup per query, thread c
// owns centroid c of every sub-quantizer — one 8-byte codebook load and one coalesced 1 KiB row store per
// sub-quantizer, the query's sub-vector, scale and offset through scalar loads.  (The kernel above launches m
// workgroups of one table row each per query: 786 k workgroups for 8192 queries, 2.4 ms of which nearly all is
// workgroup scheduling.)  Same five separately rounded operations per dimension, same order.
__global__ __launch_bounds__(256) void pq_build_table_rows8_kernel(const float *__restrict__ queries,
                                                                    const int8_t *__restrict__ codebooks,
                                                                    const float *__restrict__ scales,
                                                                    const float *__restrict__ offsets, int m,
                                                                    float *__restrict__ tables)
{
    // Two sub-quantizers per packed-fp32 instruction (pq_term8_pair, vg_hnsw_layer.hpp: each half is the scalar
    // operation of its own sub-quantizer — the same five rounded operations per dimension, in order): 28 instead of 48
    // vector instructions per table entry.  The kernel is bound by exactly that count (4096 waves x 96 x 48 instructions
    // at one per 4 cycles and SIMD = the 40 us it took; 8 codebook entries in flight per thread changed nothing).
    extern __shared__ __attribute__((aligned(16))) float lut_qprep[];  // (m / 2) * kPqPairFloats: the query's constants
    const int64_t q = blockIdx.x;
    const int c = threadIdx.x;
    const float *qv = queries + q * m * 8;
    const uint2 *cb = reinterpret_cast<const uint2 *>(codebooks);
    float *out = tables + q * m * 256;
    pq_direct_prepare(lut_qprep, qv, scales, offsets, m & ~1, c & 63);  // (every wave writes the same image)
    __syncthreads();
    int j = 0;
    for (; j + 2 <= m; j += 2) {
        const vg_f2 t = pq_term8_pair(cb[j * 256 + c], cb[(j + 1) * 256 + c], lut_qprep + (j >> 1) * kPqPairFloats);
        out[j * 256 + c] = t.x;
        out[(j + 1) * 256 + c] = t.y;
    }
    if (j < m) out[j * 256 + c] = pq_term8(cb[j * 256 + c], qv + j * 8, scales[j], offsets[j]);
}

int32_t launch_pq_build_table(const vg_pq *pq, const float *d_queries, int64_t nq,
                              float *d_tables, bool scan_layout, hipStream_t st)
{
    if (nq == 0) return VG_OK;
    if (!scan_layout && pq->k == 256 && pq->subdim == 8 && (reinterpret_cast<uintptr_t>(pq->d_codebooks) & 7) == 0) {
        VG_LAUNCH(pq_build_table_rows8_kernel, dim3(static_cast<unsigned>(nq)), dim3(256),
                  static_cast<size_t>(pq->m >> 1) * kPqPairFloats * sizeof(float), st, d_queries, pq->d_codebooks, pq->d_scales,
                  pq->d_offsets, pq->m, d_tables);
        return VG_OK;
    }
    const int64_t maxy = 65535;
    for (int64_t q0 = 0; q0 < nq; q0 += maxy) {
        int64_t cnt = nq - q0 < maxy ? nq - q0 : maxy;
        dim3 grid(static_cast<unsigned>(pq->m), static_cast<unsigned>(cnt));
        auto kern = scan_layout ? pq_build_table_kernel<true> : pq_build_table_kernel<false>;
        VG_LAUNCH(kern, grid, dim3(256), 0, st, d_queries + q0 * pq->dim,
                           pq->d_codebooks, pq->d_scales, pq->d_offsets, pq->dim, pq->m, pq->k,
                           pq->subdim,
                           d_tables + q0 * (scan_layout ? (static_cast<int64_t>(((pq->m >> 4) + 1) >> 1) * 8192 +
                                                           static_cast<int64_t>(pq->m & 15) * 256)
                                                        : static_cast<int64_t>(pq->m) * pq->k));
    }
    return VG_OK;
}

}  // namespace vg

VG_API int32_t vg_pq_create(vg_ctx *ctx, int32_t dim, int32_t m, int32_t k, vg_pq **out)
{
    VG_CHECK(ctx && out, VG_ERR_INVALID_ARG, "vg_pq_create: NULL argument");
    *out = nullptr;
    // NewProductQuantizer pq.go:36-50
    VG_CHECK(dim > 0 && m > 0, VG_ERR_INVALID_ARG,
             "dimension and numSubvectors must be positive");
    VG_CHECK(dim % m == 0, VG_ERR_INVALID_ARG, "dimension must be divisible by numSubvectors");
    VG_CHECK(k > 0, VG_ERR_INVALID_ARG, "numCentroids must be positive");
    VG_CHECK(k <= 256, VG_ERR_INVALID_ARG, "numCentroids must be <= 256 for uint8 encoding");
    VG_HIP(hipSetDevice(ctx->device));
    vg_pq *pq = new vg_pq();
    pq->ctx = ctx;
    pq->dim = dim;
    pq->m = m;
    pq->k = k;
    pq->subdim = dim / m;
    size_t cb = static_cast<size_t>(m) * k * pq->subdim;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&pq->d_codebooks), cb);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&pq->d_scales), sizeof(float) * m);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&pq->d_offsets), sizeof(float) * m);
    if (e != hipSuccess) {
        vg::set_error("vg_pq_create: hipMalloc failed: %s", hipGetErrorString(e));
        vg_pq_destroy(pq);
        return VG_ERR_OUT_OF_MEMORY;
    }
    *out = pq;
    return VG_OK;
}

VG_API int32_t vg_pq_destroy(vg_pq *pq)
{
    if (!pq) return VG_OK;
    (void)hipSetDevice(pq->ctx->device);
    if (pq->d_codebooks) (void)hipFree(pq->d_codebooks);
    if (pq->d_scales) (void)hipFree(pq->d_scales);
    if (pq->d_offsets) (void)hipFree(pq->d_offsets);
    delete pq;
    return VG_OK;
}

VG_API int32_t vg_pq_set_codebooks(vg_pq *pq, const int8_t *codebooks, const float *scales,
                                   const float *offsets)
{
    VG_CHECK(pq && codebooks && scales && offsets, VG_ERR_INVALID_ARG,
             "vg_pq_set_codebooks: NULL argument");
    VG_HIP(hipSetDevice(pq->ctx->device));
    hipStream_t st = pq->ctx->stream;
    size_t cb = static_cast<size_t>(pq->m) * pq->k * pq->subdim;
    VG_HIP(hipMemcpyAsync(pq->d_codebooks, codebooks, cb, hipMemcpyDefault, st));
    VG_HIP(hipMemcpyAsync(pq->d_scales, scales, sizeof(float) * pq->m, hipMemcpyDefault, st));
    VG_HIP(hipMemcpyAsync(pq->d_offsets, offsets, sizeof(float) * pq->m, hipMemcpyDefault, st));
    VG_HIP(hipStreamSynchronize(st));
    pq->trained = true;
    return VG_OK;
}

VG_API int32_t vg_pq_get_codebooks(vg_pq *pq, int8_t *codebooks, float *scales, float *offsets)
{
    VG_CHECK(pq, VG_ERR_INVALID_ARG, "vg_pq_get_codebooks: NULL quantizer");
    VG_CHECK(pq->trained, VG_ERR_NOT_TRAINED, "ProductQuantizer not trained");
    VG_HIP(hipSetDevice(pq->ctx->device));
    hipStream_t st = pq->ctx->stream;
    size_t cb = static_cast<size_t>(pq->m) * pq->k * pq->subdim;
    if (codebooks) VG_HIP(hipMemcpyAsync(codebooks, pq->d_codebooks, cb, hipMemcpyDefault, st));
    if (scales) VG_HIP(hipMemcpyAsync(scales, pq->d_scales, sizeof(float) * pq->m, hipMemcpyDefault, st));
    if (offsets) VG_HIP(hipMemcpyAsync(offsets, pq->d_offsets, sizeof(float) * pq->m, hipMemcpyDefault, st));
    VG_HIP(hipStreamSynchronize(st));
    return VG_OK;
}

VG_API int32_t vg_pq_get_codebooks_range(vg_pq *pq, int32_t sub_begin, int32_t sub_count, int8_t *codebooks,
                                         float *scales, float *offsets)
{
    VG_CHECK(pq, VG_ERR_INVALID_ARG, "vg_pq_get_codebooks_range: NULL quantizer");
    VG_CHECK(sub_begin >= 0 && sub_count >= 0 && sub_begin + sub_count <= pq->m, VG_ERR_INVALID_ARG,
             "vg_pq_get_codebooks_range: range outside [0, %d)", pq->m);
    if (sub_count == 0) return VG_OK;
    VG_HIP(hipSetDevice(pq->ctx->device));
    hipStream_t st = pq->ctx->stream;
    const size_t per = static_cast<size_t>(pq->k) * pq->subdim;
    if (codebooks)
        VG_HIP(hipMemcpyAsync(codebooks, pq->d_codebooks + sub_begin * per, sub_count * per, hipMemcpyDefault, st));
    if (scales)
        VG_HIP(hipMemcpyAsync(scales, pq->d_scales + sub_begin, sizeof(float) * sub_count, hipMemcpyDefault, st));
    if (offsets)
        VG_HIP(hipMemcpyAsync(offsets, pq->d_offsets + sub_begin, sizeof(float) * sub_count, hipMemcpyDefault, st));
    VG_HIP(hipStreamSynchronize(st));
    return VG_OK;
}

VG_API int32_t vg_pq_is_trained(vg_pq *pq) { return pq && pq->trained ? 1 : 0; }

VG_API int32_t vg_pq_build_distance_table(vg_pq *pq, const float *queries, int64_t nq,
                                          float *tables, void *stream)
{
    VG_CHECK(pq, VG_ERR_INVALID_ARG, "vg_pq_build_distance_table: NULL quantizer");
    VG_CHECK(pq->trained, VG_ERR_NOT_TRAINED, "ProductQuantizer not trained");
    VG_CHECK(nq >= 0, VG_ERR_INVALID_ARG, "vg_pq_build_distance_table: nq < 0");
    if (nq == 0) return VG_OK;
    VG_CHECK(queries && tables, VG_ERR_INVALID_ARG, "vg_pq_build_distance_table: NULL buffer");
    VG_HIP(hipSetDevice(pq->ctx->device));
    hipStream_t st = vg::pick_stream(pq->ctx, stream);
    vg::DevIn<float> q;
    vg::DevOut<float> t;
    VG_TRY(q.init(queries, static_cast<size_t>(nq) * pq->dim, st));
    VG_TRY(t.init(tables, static_cast<size_t>(nq) * pq->m * pq->k, st));
    {
        vg::ProfScope prof(pq->ctx, "pq_build_table", st);
        VG_TRY(vg::launch_pq_build_table(pq, q.ptr, nq, t.ptr, false, st));
    }
    VG_TRY(t.finish());
    if (t.on_host()) VG_HIP(hipStreamSynchronize(st));
    return VG_OK;
}
