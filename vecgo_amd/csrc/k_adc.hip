// k_adc.hip — PQ-ADC flat scan with fused top-k (flat/segment.go:476-483, :678-689,
// :714-721) for gfx950.
//
// HBM layout of the codes ("tiles"): the reference keeps codes row-major,
// codes[i*m:(i+1)*m].  A lane-per-row scan of that layout issues 16-byte loads at a
// 96-byte stride (48 partially used lines per wave-instruction).  set_pq_codes re-tiles
// once: tile t = rows [64t, 64t+64); within a tile, group g (16 consecutive
// sub-quantizers) of all 64 rows is contiguous:
//     byte address = ((t*G + g)*64 + lane)*16,   G = ceil(m/16)
// so every wave-level load is one fully coalesced 1 KiB request and the scan streams
// exactly N*G*16 bytes (= N*m when 16 | m).
//
// Per-row arithmetic = pqAdcLookupAvx512 (internal/simd/src/floats_avx512.c:135-167):
// 16 lane accumulators acc[l] += table[(16g+l)*256 + code[16g+l]] for g ascending, the
// _mm512_reduce_add_ps tree, then the m%16 tail added sequentially.  fp32 adds only.
#include "vg_device.hpp"
#include "vg_internal.hpp"

namespace vg {

constexpr int kAdcWaves = 16;                 // 1024 threads: one workgroup per CU (LDS-bound)
constexpr int kAdcThreads = kAdcWaves * kWave;
constexpr int kAdcSyncEvery = 2;              // iterations between candidate-buffer checks
constexpr int kAdcBuf = 4096;                 // candidate keys in LDS (32 KiB)
constexpr int kAdcMaxK = 1024;

__global__ void pq_retile_kernel(const uint8_t *__restrict__ codes, int64_t n, int m, int groups,
                                 int64_t n_tiles, uint4 *__restrict__ tiles)
{
    int64_t gid = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    int64_t total = n_tiles * groups * 64;
    if (gid >= total) return;
    int lane = static_cast<int>(gid & 63);
    int64_t tg = gid >> 6;
    int g = static_cast<int>(tg % groups);
    int64_t t = tg / groups;
    int64_t row = t * 64 + lane;
    uint32_t w[4] = {0, 0, 0, 0};
    if (row < n) {
        const uint8_t *src = codes + row * m + g * 16;
        int cnt = m - g * 16;
        if (cnt > 16) cnt = 16;
        for (int b = 0; b < cnt; b++) w[b >> 2] |= static_cast<uint32_t>(src[b]) << (8 * (b & 3));
    }
    tiles[gid] = make_uint4(w[0], w[1], w[2], w[3]);
}

__device__ __forceinline__ uint32_t code_byte(const uint4 &c, int l)
{
    uint32_t w = (l < 4) ? c.x : (l < 8) ? c.y : (l < 12) ? c.z : c.w;
    return (w >> (8 * (l & 3))) & 0xFFu;
}

struct AdcShared {
    // dynamic LDS: float lut[m*256]; uint64 buf[kAdcBuf]; then these words
    int cnt;
    int pad;
    uint64_t tau;
};

// GF = number of full 16-wide groups when known at compile time (m = 16*GF exactly),
// or -1 for the generic shape (runtime full groups + tail).
template <int GF>
__global__ __launch_bounds__(kAdcThreads) void pq_adc_scan_kernel(
    const uint4 *__restrict__ tiles, int64_t n_rows, int64_t n_tiles, int m, int groups,
    const float *__restrict__ tables, int slices, int nq, int k, uint64_t *__restrict__ partial)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *lut = reinterpret_cast<float *>(smem);
    uint64_t *buf = reinterpret_cast<uint64_t *>(smem + static_cast<size_t>(m) * 256 * sizeof(float));
    AdcShared *sh = reinterpret_cast<AdcShared *>(buf + kAdcBuf);

    // XCD-aware mapping: blocks b and b+8 share an XCD (and its L2).  Consecutive blocks of
    // one XCD take different queries over the SAME row slice, so the slice is fetched from
    // HBM once per XCD and served to the other queries from that XCD's L2.
    const int b = blockIdx.x;
    const int xcd = b & 7;
    const int o = b >> 3;
    const int q = o % nq;
    const int s = (o / nq) * 8 + xcd;
    const int64_t t0 = n_tiles * s / slices;
    const int64_t t1 = n_tiles * (s + 1) / slices;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    {   // stage this query's lookup table (m KiB) into LDS with 16-byte loads
        const float4 *src = reinterpret_cast<const float4 *>(tables + static_cast<int64_t>(q) * m * 256);
        float4 *dst = reinterpret_cast<float4 *>(lut);
        for (int i = tid; i < m * 64; i += kAdcThreads) dst[i] = src[i];
    }
    if (tid == 0) {
        sh->cnt = 0;
        sh->tau = kKeyMax;
    }
    __syncthreads();
    uint64_t tau = kKeyMax;

    const int gfull = (GF >= 0) ? GF : (m >> 4);
    const int tail = (GF >= 0) ? 0 : (m & 15);
    const int64_t span = t1 - t0;
    const int iters = static_cast<int>((span + kAdcWaves - 1) / kAdcWaves);

    for (int it = 0; it < iters; it++) {
        const int64_t tile = t0 + static_cast<int64_t>(it) * kAdcWaves + wave;
        if (tile < t1) {
            const uint4 *tp = tiles + (tile * groups) * 64 + lane;
            float acc[16];
#pragma unroll
            for (int l = 0; l < 16; l++) acc[l] = 0.0f;
            if (GF >= 0) {
                uint4 c[GF > 0 ? GF : 1];
#pragma unroll
                for (int g = 0; g < GF; g++) c[g] = tp[g * 64];
#pragma unroll
                for (int g = 0; g < GF; g++) {
#pragma unroll
                    for (int l = 0; l < 16; l++)
                        acc[l] = acc[l] + lut[(g * 16 + l) * 256 + code_byte(c[g], l)];
                }
            } else {
                for (int g = 0; g < gfull; g++) {
                    uint4 c = tp[g * 64];
#pragma unroll
                    for (int l = 0; l < 16; l++)
                        acc[l] = acc[l] + lut[(g * 16 + l) * 256 + code_byte(c, l)];
                }
            }
            float total = reduce16_regs(acc);
            if (tail) {
                uint4 c = tp[gfull * 64];
                for (int l = 0; l < tail; l++)
                    total = total + lut[(gfull * 16 + l) * 256 + code_byte(c, l)];
            }
            const int64_t row = tile * 64 + lane;
            if (row < n_rows) {
                uint64_t key = make_key(total, static_cast<uint32_t>(row), false);
                if (key < tau) {
                    int pos = atomicAdd(&sh->cnt, 1);
                    buf[pos] = key;  // pos < kAdcBuf by the sync protocol below
                }
            }
        }
        // Every kAdcSyncEvery iterations at most kAdcSyncEvery*1024 keys were appended;
        // compact when the next round could overflow the buffer.
        if ((it % kAdcSyncEvery) == kAdcSyncEvery - 1 || it == iters - 1) {
            __syncthreads();
            int c = sh->cnt;
            const bool last = (it == iters - 1);
            if (last || c > kAdcBuf - kAdcSyncEvery * kAdcThreads) {
                for (int i = c + tid; i < kAdcBuf; i += kAdcThreads) buf[i] = kKeyMax;
                __syncthreads();
                bitonic_sort_lds(buf, kAdcBuf, tid, kAdcThreads);
                if (tid == 0) {
                    int keep = c < k ? c : k;
                    sh->cnt = keep;
                    sh->tau = (c >= k) ? buf[k - 1] : kKeyMax;
                }
                __syncthreads();
            }
            tau = sh->tau;
        }
    }
    uint64_t *out = partial + (static_cast<int64_t>(q) * slices + s) * k;
    const int have = sh->cnt;
    for (int i = tid; i < k; i += kAdcThreads) out[i] = (i < have) ? buf[i] : kKeyMax;
}

// One workgroup per query: merges `lists` sorted-or-not key lists of length k each into
// the k best keys, best first, and decodes them to (id, score).
constexpr int kMergeThreads = 256;
constexpr int kMergeBuf = 4096;
__global__ __launch_bounds__(kMergeThreads) void topk_merge_kernel(
    const uint64_t *__restrict__ partial, int lists, int k, bool descending,
    uint32_t *__restrict__ ids, float *__restrict__ scores)
{
    __shared__ uint64_t buf[kMergeBuf];
    const int q = blockIdx.x;
    const int tid = threadIdx.x;
    const uint64_t *src = partial + static_cast<int64_t>(q) * lists * k;
    const int64_t total = static_cast<int64_t>(lists) * k;
    int have = 0;  // running best keys occupy buf[0..have)
    int64_t pos = 0;
    do {
        int room = kMergeBuf - have;
        int take = static_cast<int>((total - pos) < room ? (total - pos) : room);
        for (int i = tid; i < take; i += kMergeThreads) buf[have + i] = src[pos + i];
        for (int i = have + take + tid; i < kMergeBuf; i += kMergeThreads) buf[i] = kKeyMax;
        __syncthreads();
        bitonic_sort_lds(buf, kMergeBuf, tid, kMergeThreads);
        pos += take;
        have = k;  // k <= kAdcMaxK < kMergeBuf; slots past the real keys hold kKeyMax
    } while (pos < total);
    for (int i = tid; i < k; i += kMergeThreads) {
        uint64_t key = buf[i];
        if (key == kKeyMax) {
            ids[static_cast<int64_t>(q) * k + i] = VG_INVALID_ID;
            scores[static_cast<int64_t>(q) * k + i] = descending ? -INFINITY : INFINITY;
        } else {
            ids[static_cast<int64_t>(q) * k + i] = key_row(key);
            scores[static_cast<int64_t>(q) * k + i] = key_score(key, descending);
        }
    }
}

int32_t launch_topk_merge(const uint64_t *partial, int64_t nq, int lists, int k, bool descending,
                          uint32_t *ids, float *scores, hipStream_t st)
{
    if (nq == 0 || k == 0) return VG_OK;
    hipLaunchKernelGGL(topk_merge_kernel, dim3(static_cast<unsigned>(nq)), dim3(kMergeThreads), 0,
                       st, partial, lists, k, descending, ids, scores);
    VG_HIP(hipGetLastError());
    return VG_OK;
}

int32_t launch_pq_build_table(const vg_pq *pq, const float *d_queries, int64_t nq,
                              float *d_tables, hipStream_t st);

static int adc_slices(int64_t nq, int64_t n_tiles, int cus)
{
    // smallest multiple of 8 (one group per XCD) with slices*nq >= #CUs, at least one
    // workgroup-iteration of tiles per slice
    int64_t s = (cus + nq - 1) / nq;
    s = ((s + 7) / 8) * 8;
    int64_t max_s = (n_tiles + kAdcWaves - 1) / kAdcWaves;
    max_s = (max_s / 8) * 8;
    if (max_s < 8) max_s = 8;
    if (s > max_s) s = max_s;
    if (s < 8) s = 8;
    return static_cast<int>(s);
}

template <int GF>
static int32_t launch_scan(const vg_index *idx, const float *tables, int64_t nq, int k,
                           int slices, uint64_t *partial, hipStream_t st)
{
    const vg_pq *pq = idx->pq;
    size_t lds = static_cast<size_t>(pq->m) * 256 * sizeof(float) + kAdcBuf * sizeof(uint64_t) +
                 sizeof(AdcShared);
    auto kern = pq_adc_scan_kernel<GF>;
    VG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                               hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    // grid.x limit is 2^31-1; chunk the queries if needed
    const int64_t max_q = (1ll << 30) / slices;
    for (int64_t q0 = 0; q0 < nq; q0 += max_q) {
        int64_t cnt = nq - q0 < max_q ? nq - q0 : max_q;
        hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(cnt * slices)), dim3(kAdcThreads), lds,
                           st, reinterpret_cast<const uint4 *>(idx->d_pq_tiles), idx->n,
                           idx->n_tiles, pq->m, idx->pq_groups,
                           tables + q0 * pq->m * 256, slices, static_cast<int>(cnt), k,
                           partial + q0 * slices * k);
    }
    VG_HIP(hipGetLastError());
    return VG_OK;
}

}  // namespace vg

VG_API int32_t vg_index_set_pq_codes(vg_index *idx, vg_pq *pq, const uint8_t *codes, void *stream)
{
    VG_CHECK(idx && pq, VG_ERR_INVALID_ARG, "vg_index_set_pq_codes: NULL handle");
    VG_CHECK(pq->trained, VG_ERR_NOT_TRAINED, "ProductQuantizer not trained");
    VG_CHECK(pq->dim == idx->dim, VG_ERR_DIM_MISMATCH, "vector dimension mismatch");
    VG_CHECK(idx->n == 0 || codes, VG_ERR_INVALID_ARG, "vg_index_set_pq_codes: codes is NULL");
    VG_HIP(hipSetDevice(idx->ctx->device));
    hipStream_t st = vg::pick_stream(idx->ctx, stream);
    if (idx->d_pq_tiles) {
        VG_HIP(hipStreamSynchronize(st));
        VG_HIP(hipFree(idx->d_pq_tiles));
        idx->d_pq_tiles = nullptr;
    }
    idx->pq = pq;
    idx->pq_groups = (pq->m + 15) / 16;
    idx->n_tiles = (idx->n + 63) / 64;
    if (idx->n == 0) return VG_OK;
    size_t bytes = static_cast<size_t>(idx->n_tiles) * idx->pq_groups * 64 * 16;
    VG_HIP(hipMalloc(reinterpret_cast<void **>(&idx->d_pq_tiles), bytes));
    vg::DevIn<uint8_t> in;
    VG_TRY(in.init(codes, static_cast<size_t>(idx->n) * pq->m, st));
    int64_t total = idx->n_tiles * idx->pq_groups * 64;
    hipLaunchKernelGGL(vg::pq_retile_kernel, dim3(static_cast<unsigned>((total + 255) / 256)),
                       dim3(256), 0, st, in.ptr, idx->n, pq->m, idx->pq_groups, idx->n_tiles,
                       reinterpret_cast<uint4 *>(idx->d_pq_tiles));
    VG_HIP(hipGetLastError());
    VG_HIP(hipStreamSynchronize(st));
    return VG_OK;
}

VG_API int32_t vg_search_pq_adc(vg_index *idx, const float *queries, int64_t nq, int32_t k,
                                uint32_t *ids, float *scores, void *stream)
{
    VG_CHECK(idx, VG_ERR_INVALID_ARG, "vg_search_pq_adc: NULL index");
    VG_CHECK(idx->pq != nullptr, VG_ERR_NOT_READY, "vg_search_pq_adc: index has no PQ codes");
    const vg_pq *pq = idx->pq;
    VG_CHECK(pq->trained, VG_ERR_NOT_TRAINED, "ProductQuantizer not trained");
    VG_CHECK(nq >= 0 && k >= 0, VG_ERR_INVALID_ARG, "vg_search_pq_adc: negative nq or k");
    if (nq == 0 || k == 0) return VG_OK;
    VG_CHECK(queries && ids && scores, VG_ERR_INVALID_ARG, "vg_search_pq_adc: NULL buffer");
    VG_CHECK(k <= vg::kAdcMaxK, VG_ERR_UNSUPPORTED, "vg_search_pq_adc: k=%d exceeds %d", k,
             vg::kAdcMaxK);
    // The reference builds the table with stride K and looks it up with stride 256
    // (pq.go:474 vs internal/simd/kernels.go:249); only K == 256 is self-consistent.
    VG_CHECK(pq->k == 256, VG_ERR_UNSUPPORTED,
             "vg_search_pq_adc: LUT scan needs numCentroids == 256 (got %d)", pq->k);
    size_t lds = static_cast<size_t>(pq->m) * 1024 + vg::kAdcBuf * 8 + sizeof(vg::AdcShared);
    VG_CHECK(lds <= 160 * 1024, VG_ERR_UNSUPPORTED,
             "vg_search_pq_adc: m=%d lookup table does not fit the 160 KiB LDS", pq->m);
    VG_HIP(hipSetDevice(idx->ctx->device));
    hipStream_t st = vg::pick_stream(idx->ctx, stream);

    vg::DevIn<float> q;
    vg::DevOut<uint32_t> oid;
    vg::DevOut<float> osc;
    VG_TRY(q.init(queries, static_cast<size_t>(nq) * idx->dim, st));
    VG_TRY(oid.init(ids, static_cast<size_t>(nq) * k, st));
    VG_TRY(osc.init(scores, static_cast<size_t>(nq) * k, st));

    if (idx->n == 0) {
        vg::DevTmp<uint64_t> none;
        VG_TRY(none.init(static_cast<size_t>(nq) * k, st));
        VG_HIP(hipMemsetAsync(none.ptr, 0xFF, static_cast<size_t>(nq) * k * 8, st));
        VG_TRY(vg::launch_topk_merge(none.ptr, nq, 1, k, false, oid.ptr, osc.ptr, st));
    } else {
        const int slices = vg::adc_slices(nq, idx->n_tiles, idx->ctx->compute_units);
        vg::DevTmp<float> tables;
        vg::DevTmp<uint64_t> partial;
        VG_TRY(tables.init(static_cast<size_t>(nq) * pq->m * 256, st));
        VG_TRY(partial.init(static_cast<size_t>(nq) * slices * k, st));
        VG_TRY(vg::launch_pq_build_table(pq, q.ptr, nq, tables.ptr, st));
        if (pq->m == 96)
            VG_TRY(vg::launch_scan<6>(idx, tables.ptr, nq, k, slices, partial.ptr, st));
        else
            VG_TRY(vg::launch_scan<-1>(idx, tables.ptr, nq, k, slices, partial.ptr, st));
        VG_TRY(vg::launch_topk_merge(partial.ptr, nq, slices, k, false, oid.ptr, osc.ptr, st));
    }
    VG_TRY(oid.finish());
    VG_TRY(osc.finish());
    if (oid.on_host() || osc.on_host()) VG_HIP(hipStreamSynchronize(st));
    return VG_OK;
}
