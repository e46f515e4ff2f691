// k_sq8.hip — 8-bit scalar quantization:
//   quantization.ScalarQuantizer   internal/quantization/quantizer.go:27-250
//   simd.Sq8uL2BatchPerDimension   internal/simd/src/sq8_avx512.c:59-103
//   flat.Segment.Search, SQ8 branch internal/segment/flat/segment.go:517-604
// Numerics contract (sq8_avx512.c): per row 16 lane accumulators over 16-element blocks,
//   rec = fma(float(code), invScale[j], min[j]); diff = q[j] - rec; sum[l] = fma(diff, diff, sum[l])
// then the _mm512_reduce_add_ps tree and an FMA-contracted scalar tail.  Here ONE GPU lane owns a
// row and keeps the 16 accumulators in registers, so a wave scores 64 rows at a time; the scan
// reads codes re-tiled to [tile of 64 rows][16-byte group][lane] (one coalesced 1 KiB request per
// wave-instruction, exactly dim bytes per row when 16 | dim).  q, min and invScale are the same
// for every lane: they are read through wave-uniform (scalar) loads, not per lane.
#include <algorithm>

#include <type_traits>

#include "vg_device.hpp"
#include "vg_internal.hpp"
#include "vg_cand_replay.hpp"

struct vg_int4 {
    vg_ctx *ctx = nullptr;
    int32_t dim = 0;
    bool trained = false;
    float *d_min = nullptr, *d_diff = nullptr;  // [dim] each
    float *d_table = nullptr;                   // [dim * 16] BuildInt4LookupTable
};

struct vg_sq8 {
    vg_ctx *ctx = nullptr;
    int32_t dim = 0;
    bool trained = false;
    float *d_mins = nullptr, *d_maxs = nullptr, *d_scales = nullptr, *d_inv = nullptr;  // [dim] each
};

namespace vg {

int32_t launch_page_patch(int64_t nq, int k, int off, int kk, bool descending, const int *always_one,
                          const uint32_t *fids, const float *fscores, uint32_t *ids, float *scores, uint64_t *min_keys,
                          hipStream_t st);
int32_t launch_topk_merge(const uint64_t *partial, int64_t nq, int lists, int k, bool descending,
                          uint32_t *ids, float *scores, hipStream_t st, const int *only_if = nullptr,
                          const int *always = nullptr);

constexpr float kF32Max = 3.40282346638528859811704183484516925440e+38f;

// ---- Train (quantizer.go:127-180) ----------------------------------------------------------------
// stage 1: thread = (row chunk, dimension): min / max over the chunk's rows (order-free, exact)
__global__ void sq8_minmax_kernel(const float *__restrict__ v, int64_t n, int dim, int chunks,
                                  float *__restrict__ pmin, float *__restrict__ pmax)
{
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    if (d >= dim) return;
    const int64_t r0 = n * c / chunks, r1 = n * (c + 1) / chunks;
    float mn = kF32Max, mx = -kF32Max;
    int64_t i = r0;
    for (; i + 8 <= r1; i += 8) {  // 8 rows in flight per thread; the data is read once
        float x[8];
#pragma unroll
        for (int u = 0; u < 8; u++) x[u] = __builtin_nontemporal_load(v + (i + u) * dim + d);
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (x[u] < mn) mn = x[u];
            if (x[u] > mx) mx = x[u];
        }
    }
    for (; i < r1; i++) {
        const float x = v[i * dim + d];
        if (x < mn) mn = x;
        if (x > mx) mx = x;
    }
    pmin[static_cast<int64_t>(c) * dim + d] = mn;
    pmax[static_cast<int64_t>(c) * dim + d] = mx;
}

// stage 2 + scales.  from_train: a constant dimension gets max = min + 1e-6 (quantizer.go:168-170);
// SetBounds instead zeroes both scales when max - min < 1e-9 (quantizer.go:64-72).
__global__ void sq8_finish_kernel(const float *__restrict__ pmin, const float *__restrict__ pmax, int chunks,
                                  int dim, bool from_train, float *__restrict__ mins, float *__restrict__ maxs,
                                  float *__restrict__ scales, float *__restrict__ inv)
{
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= dim) return;
    float mn, mx;
    if (from_train) {
        mn = kF32Max;
        mx = -kF32Max;
        for (int c = 0; c < chunks; c++) {
            const float a = pmin[static_cast<int64_t>(c) * dim + d], b = pmax[static_cast<int64_t>(c) * dim + d];
            if (a < mn) mn = a;
            if (b > mx) mx = b;
        }
        if (mn == mx) mx = mn + 1e-6f;
        const float range = mx - mn;
        scales[d] = 255.0f / range;
        inv[d] = range / 255.0f;
    } else {
        mn = pmin[d];
        mx = pmax[d];
        const float diff = mx - mn;
        if (diff < 1e-9f) {
            scales[d] = 0.0f;
            inv[d] = 0.0f;
        } else {
            scales[d] = 255.0f / diff;
            inv[d] = diff / 255.0f;
        }
    }
    mins[d] = mn;
    maxs[d] = mx;
}

// ---- EncodeInto / DecodeInto (quantizer.go:198-250): thread per element -----------------------------
__global__ void sq8_encode_kernel(const float *__restrict__ v, int64_t total, int dim,
                                  const float *__restrict__ mins, const float *__restrict__ maxs,
                                  const float *__restrict__ scales, uint8_t *__restrict__ out)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int d = static_cast<int>(i % dim);
    float val = v[i];
    const float mn = mins[d], mx = maxs[d];
    if (val < mn)
        val = mn;
    else if (val > mx)
        val = mx;
    const float normalized = (val - mn) * scales[d];
    const float r = normalized + 0.5f;
    out[i] = static_cast<uint8_t>(static_cast<int>(r));  // Go uint8(float32): truncation
}

__global__ void sq8_decode_kernel(const uint8_t *__restrict__ codes, int64_t total, int dim,
                                  const float *__restrict__ mins, const float *__restrict__ inv,
                                  float *__restrict__ out)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int d = static_cast<int>(i % dim);
    const float t = static_cast<float>(codes[i]) * inv[d];
    out[i] = t + mins[d];
}

// The element kernels above spend their time on `i % dim` (a 64-bit division per element): 1.6 - 1.75 TB/s.  dim % 4
// == 0 and 16-byte aligned buffers: a thread owns four consecutive dimensions (its parameters loaded once) and walks
// kRowsPerThread rows — no division, 16-byte loads / stores, the same operations per element.
constexpr int kRowsPerThread = 16;  // at least; more when n / 16 exceeds the grid's y range
static inline int rows_per_thread(int64_t n) { return static_cast<int>(std::max<int64_t>(kRowsPerThread, (n + 65534) / 65535)); }
__global__ __launch_bounds__(256) void sq8_encode4_kernel(const float *__restrict__ v, int64_t n, int dim,
                                                          const float *__restrict__ mins, const float *__restrict__ maxs,
                                                          const float *__restrict__ scales, uint8_t *__restrict__ out, int rpt)
{
    const int cg = blockIdx.x * blockDim.x + threadIdx.x;
    if (cg * 4 >= dim) return;
    const float4 mn = *reinterpret_cast<const float4 *>(mins + cg * 4), mx = *reinterpret_cast<const float4 *>(maxs + cg * 4),
                 sc = *reinterpret_cast<const float4 *>(scales + cg * 4);
    const int64_t r0 = static_cast<int64_t>(blockIdx.y) * rpt;
    auto enc = [](float val, float lo, float hi, float s) -> uint32_t {
        if (val < lo)
            val = lo;
        else if (val > hi)
            val = hi;
        const float normalized = (val - lo) * s;
        const float r = normalized + 0.5f;
        return static_cast<uint32_t>(static_cast<uint8_t>(static_cast<int>(r)));  // Go uint8(float32): truncation
    };
    for (int64_t row = r0; row < r0 + rpt && row < n; row++) {
        const float4 x = *reinterpret_cast<const float4 *>(v + row * dim + cg * 4);
        const uint32_t w = enc(x.x, mn.x, mx.x, sc.x) | (enc(x.y, mn.y, mx.y, sc.y) << 8) | (enc(x.z, mn.z, mx.z, sc.z) << 16) |
                           (enc(x.w, mn.w, mx.w, sc.w) << 24);
        *reinterpret_cast<uint32_t *>(out + row * dim + cg * 4) = w;
    }
}
__global__ __launch_bounds__(256) void sq8_decode4_kernel(const uint8_t *__restrict__ codes, int64_t n, int dim,
                                                          const float *__restrict__ mins, const float *__restrict__ inv,
                                                          float *__restrict__ out, int rpt)
{
    const int cg = blockIdx.x * blockDim.x + threadIdx.x;
    if (cg * 4 >= dim) return;
    const float4 mn = *reinterpret_cast<const float4 *>(mins + cg * 4), iv = *reinterpret_cast<const float4 *>(inv + cg * 4);
    const int64_t r0 = static_cast<int64_t>(blockIdx.y) * rpt;
    for (int64_t row = r0; row < r0 + rpt && row < n; row++) {
        const uint32_t w = *reinterpret_cast<const uint32_t *>(codes + row * dim + cg * 4);
        float4 o;
        float t;
        t = static_cast<float>(w & 0xFFu) * iv.x;
        o.x = t + mn.x;
        t = static_cast<float>((w >> 8) & 0xFFu) * iv.y;
        o.y = t + mn.y;
        t = static_cast<float>((w >> 16) & 0xFFu) * iv.z;
        o.z = t + mn.z;
        t = static_cast<float>(w >> 24) * iv.w;
        o.w = t + mn.w;
        *reinterpret_cast<float4 *>(out + row * dim + cg * 4) = o;
    }
}

// ---- the row kernel --------------------------------------------------------------------------------
// 16 bytes = one 16-element block of one row: lane accumulators l = 0..15 get one FMA each.
// qv / mn / iv point at the block's 16 floats and are wave-uniform.
__device__ __forceinline__ void sq8_block16(float (&acc)[16], const uint4 c, const float *__restrict__ qv,
                                            const float *__restrict__ mn, const float *__restrict__ iv)
{
    const uint32_t w[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
    for (int l = 0; l < 16; l++) {
        const float cf = static_cast<float>((w[l >> 2] >> (8 * (l & 3))) & 0xFFu);
        const float rec = __builtin_fmaf(cf, iv[l], mn[l]);
        const float diff = qv[l] - rec;
        acc[l] = __builtin_fmaf(diff, diff, acc[l]);
    }
}

// the tail of a row (dim % 16 elements, bytes in the low lanes of the last group)
__device__ __forceinline__ float sq8_tail(float total, const uint4 c, int cnt, const float *__restrict__ qv,
                                          const float *__restrict__ mn, const float *__restrict__ iv)
{
    const uint32_t w[4] = {c.x, c.y, c.z, c.w};
    for (int l = 0; l < cnt; l++) {
        const float cf = static_cast<float>((w[l >> 2] >> (8 * (l & 3))) & 0xFFu);
        const float rec = __builtin_fmaf(cf, iv[l], mn[l]);
        const float diff = qv[l] - rec;
        total = __builtin_fmaf(diff, diff, total);
    }
    return total;
}

// ScalarQuantizer.DotProduct (quantizer.go:109-119) over `cnt` (<= 16) elements of one row: a plain Go
// loop — val = mins[i] + float32(code[i])*invScales[i], dot += q[i]*val, four separately rounded
// operations (no FMA on amd64), one running sum in element order.
__device__ __forceinline__ float sq8_dot16(float total, const uint4 c, int cnt, const float *__restrict__ qv,
                                           const float *__restrict__ mn, const float *__restrict__ iv)
{
    const uint32_t w[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
    for (int l = 0; l < 16; l++) {
        if (l < cnt) {
            const float cf = static_cast<float>((w[l >> 2] >> (8 * (l & 3))) & 0xFFu);
            const float t = cf * iv[l];
            const float val = mn[l] + t;
            const float prod = qv[l] * val;
            total = total + prod;
        }
    }
    return total;
}

// one row's score from its tile pieces: L2 = the 16 lane accumulators of sq8u_l2_batch + tail,
// DOT = the sequential sum above
template <bool DOT>
__device__ __forceinline__ float sq8_row_score(const uint4 *__restrict__ tp, int groups, int full, int tail,
                                               const float *__restrict__ qv, const float *__restrict__ mins,
                                               const float *__restrict__ inv);

// L2DistanceBatch on the reference layout (codes n*dim): lane per row, 16 bytes at a time.  The
// interface path for small batches (the reference calls it with 256 rows, flat/segment.go:487,550).
__global__ __launch_bounds__(256) void sq8_l2_batch_kernel(const float *__restrict__ query,
                                                           const uint8_t *__restrict__ codes, int64_t n, int dim,
                                                           const float *__restrict__ mins,
                                                           const float *__restrict__ inv, float *__restrict__ out)
{
    const int64_t row = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (row >= n) return;
    const uint8_t *cp = codes + row * dim;
    float acc[16];
#pragma unroll
    for (int l = 0; l < 16; l++) acc[l] = 0.0f;
    int j = 0;
    for (; j + 16 <= dim; j += 16) {
        uint32_t w[4];
#pragma unroll
        for (int t = 0; t < 4; t++)  // rows are only byte-aligned
            w[t] = cp[j + 4 * t] | (cp[j + 4 * t + 1] << 8) | (cp[j + 4 * t + 2] << 16) |
                   (static_cast<uint32_t>(cp[j + 4 * t + 3]) << 24);
        sq8_block16(acc, make_uint4(w[0], w[1], w[2], w[3]), query + j, mins + j, inv + j);
    }
    float total = reduce16_regs(acc);
    for (; j < dim; j++) {
        const float rec = __builtin_fmaf(static_cast<float>(cp[j]), inv[j], mins[j]);
        const float diff = query[j] - rec;
        total = __builtin_fmaf(diff, diff, total);
    }
    out[row] = total;
}

// The same distances as a streaming scan (dim % 128 == 0, 16-byte aligned codes): the kernel above reads its row a byte
// at a time at a dim-byte stride (1.07 TB/s of codes at dim 768).  Here a wave takes 64 rows, 128-byte pieces of them
// arrive as whole lines (8 lanes per row; the next piece in flight while this one is scored) and are turned through
// the wave's LDS (row stride 144 bytes = 16 x 9: conflict-free ds_read_b128), each lane then walks ITS row with
// sq8_block16 — the same 16 accumulators in the same order.
constexpr int kSqTurnWaves = 4;
constexpr int kSqTurnStride = 144;
__global__ __launch_bounds__(kSqTurnWaves * 64) void sq8_l2_batch_turn_kernel(const float *__restrict__ query,
                                                                              const uint8_t *__restrict__ codes, int64_t n,
                                                                              int dim, const float *__restrict__ mins,
                                                                              const float *__restrict__ inv,
                                                                              float *__restrict__ out)
{
    __shared__ __attribute__((aligned(16))) unsigned char stage_all[kSqTurnWaves][64 * kSqTurnStride];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row0 = (static_cast<int64_t>(blockIdx.x) * kSqTurnWaves + wave) * 64;
    if (row0 >= n) return;
    unsigned char *stage = stage_all[wave];
    const int r = lane >> 3, part = lane & 7;
    auto row_ptr = [&](int k) {
        const int64_t row = row0 + r + 8 * k < n ? row0 + r + 8 * k : n - 1;  // past n: row n - 1 again, not stored
        return codes + row * dim + part * 16;
    };
    const uint8_t *s0 = row_ptr(0), *s1 = row_ptr(1), *s2 = row_ptr(2), *s3 = row_ptr(3), *s4 = row_ptr(4),
                  *s5 = row_ptr(5), *s6 = row_ptr(6), *s7 = row_ptr(7);
#define VG_SQ_LD(P, OFF) load_stream(reinterpret_cast<const uint4 *>((P) + (OFF)))
    uint4 u0 = VG_SQ_LD(s0, 0), u1 = VG_SQ_LD(s1, 0), u2 = VG_SQ_LD(s2, 0), u3 = VG_SQ_LD(s3, 0), u4 = VG_SQ_LD(s4, 0),
          u5 = VG_SQ_LD(s5, 0), u6 = VG_SQ_LD(s6, 0), u7 = VG_SQ_LD(s7, 0);
    float acc[16];
#pragma unroll
    for (int l = 0; l < 16; l++) acc[l] = 0.0f;
    unsigned char *wr = stage + r * kSqTurnStride + part * 16;
    for (int cb0 = 0; cb0 < dim; cb0 += 128) {
        *reinterpret_cast<uint4 *>(wr) = u0;
        *reinterpret_cast<uint4 *>(wr + 8 * kSqTurnStride) = u1;
        *reinterpret_cast<uint4 *>(wr + 16 * kSqTurnStride) = u2;
        *reinterpret_cast<uint4 *>(wr + 24 * kSqTurnStride) = u3;
        *reinterpret_cast<uint4 *>(wr + 32 * kSqTurnStride) = u4;
        *reinterpret_cast<uint4 *>(wr + 40 * kSqTurnStride) = u5;
        *reinterpret_cast<uint4 *>(wr + 48 * kSqTurnStride) = u6;
        *reinterpret_cast<uint4 *>(wr + 56 * kSqTurnStride) = u7;
        const int nxt = cb0 + 128 < dim ? cb0 + 128 : cb0;  // (the last piece again: unused)
        u0 = VG_SQ_LD(s0, nxt);
        u1 = VG_SQ_LD(s1, nxt);
        u2 = VG_SQ_LD(s2, nxt);
        u3 = VG_SQ_LD(s3, nxt);
        u4 = VG_SQ_LD(s4, nxt);
        u5 = VG_SQ_LD(s5, nxt);
        u6 = VG_SQ_LD(s6, nxt);
        u7 = VG_SQ_LD(s7, nxt);
#undef VG_SQ_LD
        for (int piece = 0; piece < 8; piece++) {
            const uint4 c = *reinterpret_cast<const uint4 *>(stage + lane * kSqTurnStride + piece * 16);
            const int j = cb0 + piece * 16;
            sq8_block16(acc, c, query + j, mins + j, inv + j);
        }
    }
    const float total = reduce16_regs(acc);
    if (row0 + lane < n) out[row0 + lane] = total;
}

// reference layout -> [tile][group][lane] 16-byte pieces (zero padded past dim and past n)
__global__ void sq8_retile_kernel(const uint8_t *__restrict__ codes, int64_t n, int dim, int groups,
                                  int64_t n_tiles, uint4 *__restrict__ tiles)
{
    const int64_t gid = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const int64_t total = n_tiles * groups * 64;
    if (gid >= total) return;
    const int lane = static_cast<int>(gid & 63);
    const int64_t tg = gid >> 6;
    const int g = static_cast<int>(tg % groups);
    const int64_t row = (tg / groups) * 64 + lane;
    uint32_t w[4] = {0, 0, 0, 0};
    if (row < n) {
        const uint8_t *src = codes + row * dim;
        for (int b = 0; b < 16; b++) {
            const int at = g * 16 + b;
            if (at < dim) w[b >> 2] |= static_cast<uint32_t>(src[at]) << (8 * (b & 3));
        }
    }
    tiles[gid] = make_uint4(w[0], w[1], w[2], w[3]);
}

// Exhaustive SQ8 scan with fused top-k.  HBM-bound by design: 16*groups bytes per row.
constexpr int kSqAhead = 4;  // 16-byte code groups in flight per lane
constexpr int kSqWaves = 4;
constexpr int kSqThreads = kSqWaves * 64;
template <bool DOT>
__device__ __forceinline__ float sq8_row_score(const uint4 *__restrict__ tp, int groups, int full, int tail,
                                               const float *__restrict__ qv, const float *__restrict__ mins,
                                               const float *__restrict__ inv)
{
    float acc[16];
#pragma unroll
    for (int l = 0; l < 16; l++) acc[l] = 0.0f;
    float run = 0.0f;
    // kSqAhead groups of codes in flight per lane (one ahead left the wave waiting on HBM every
    // 64 VALU instructions); addresses past the row's last group are clamped to it
    uint4 ring[kSqAhead];
    const int glast = groups - 1;
#pragma unroll
    for (int a = 0; a < kSqAhead; a++) ring[a] = load_stream(tp + (a < glast ? a : glast) * 64);
    for (int g0 = 0; g0 < full; g0 += kSqAhead) {
#pragma unroll
        for (int a = 0; a < kSqAhead; a++) {
            const int g = g0 + a;
            const uint4 c = ring[a];
            const int gn = g + kSqAhead;
            ring[a] = load_stream(tp + (gn < glast ? gn : glast) * 64);
            if (g < full) {
                if (DOT)
                    run = sq8_dot16(run, c, 16, qv + g * 16, mins + g * 16, inv + g * 16);
                else
                    sq8_block16(acc, c, qv + g * 16, mins + g * 16, inv + g * 16);
            }
        }
    }
    if (DOT) {
        if (tail) run = sq8_dot16(run, tp[full * 64], tail, qv + full * 16, mins + full * 16, inv + full * 16);
        return run;
    }
    float total = reduce16_regs(acc);
    if (tail) total = sq8_tail(total, tp[full * 64], tail, qv + full * 16, mins + full * 16, inv + full * 16);
    return total;
}

// The same with the ring carried ACROSS tiles (dim % 64 == 0: no tail group, whole ring rounds): the last round of a
// tile refills the ring with the first groups of the wave's NEXT tile, so a tile no longer starts with kSqAhead loads
// and an exposed HBM round trip (~2 us of the ~33 us a wave spends on a tile: 6.05 -> 6.4 TB/s at 4M x 768).
// `ring` arrives holding groups 0 .. kSqAhead-1 of this tile and leaves holding those of `tp_next`.
template <bool DOT>
__device__ __forceinline__ float sq8_row_score_stream(const uint4 *__restrict__ tp, const uint4 *__restrict__ tp_next, int full,
                                                      uint4 (&ring)[kSqAhead], const float *__restrict__ qv,
                                                      const float *__restrict__ mins, const float *__restrict__ inv)
{
    float acc[16];
#pragma unroll
    for (int l = 0; l < 16; l++) acc[l] = 0.0f;
    float run = 0.0f;
    for (int g0 = 0; g0 < full; g0 += kSqAhead) {
        const uint4 *src = g0 + kSqAhead < full ? tp + (g0 + kSqAhead) * 64 : tp_next;  // (uniform)
#pragma unroll
        for (int a = 0; a < kSqAhead; a++) {
            const int g = g0 + a;
            const uint4 c = ring[a];
            ring[a] = load_stream(src + a * 64);
            if (DOT)
                run = sq8_dot16(run, c, 16, qv + g * 16, mins + g * 16, inv + g * 16);
            else
                sq8_block16(acc, c, qv + g * 16, mins + g * 16, inv + g * 16);
        }
    }
    return DOT ? run : reduce16_regs(acc);
}

template <bool DOT, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void sq8_scan_kernel(
    const uint4 *__restrict__ tiles, int64_t n_rows, int64_t n_tiles, int groups, int dim,
    const float *__restrict__ queries, const float *__restrict__ mins, const float *__restrict__ inv, int slices,
    int nq, int k, uint64_t *__restrict__ partial, const uint64_t *__restrict__ min_keys)
{
    __shared__ uint64_t lists[WAVES * 64];
    __shared__ int valid[WAVES];
    const int b = blockIdx.x;
    const int xcd = b & 7;
    const int o = b >> 3;
    const int q = o % nq;
    const int s = (o / nq) * 8 + xcd;
    const int64_t t0 = n_tiles * s / slices, t1 = n_tiles * (s + 1) / slices;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *qv = queries + static_cast<int64_t>(q) * dim;
    const int full = dim >> 4, tail = dim & 15;
    WaveTopK tk;
    tk.init(k);
    // one-query passes deal the tiles round-robin over the workgroups (one moving window of the code array, see
    // rabitq_scan_kernel); several queries keep the slice mapping and share the slice in their XCD's L2
    const bool dealt = nq == 1;
    const int64_t step = dealt ? static_cast<int64_t>(slices) * WAVES : WAVES;
    const int64_t end = dealt ? n_tiles : t1;
    int64_t tile = dealt ? static_cast<int64_t>(s) * WAVES + wave : t0 + wave;
    if (tail == 0 && full % kSqAhead == 0 && full >= kSqAhead && tile < end) {  // the ring runs on from tile to tile
        uint4 ring[kSqAhead];
        const uint4 *tp = tiles + (tile * groups) * 64 + lane;
#pragma unroll
        for (int a = 0; a < kSqAhead; a++) ring[a] = load_stream(tp + a * 64);
        for (; tile < end; tile += step) {
            const int64_t tn = tile + step < end ? tile + step : tile;  // (the last tile's own first groups again: unused)
            const uint4 *tpn = tiles + (tn * groups) * 64 + lane;
            const float total = sq8_row_score_stream<DOT>(tp, tpn, full, ring, qv, mins, inv);
            tp = tpn;
            const int64_t row = tile * 64 + lane;
            uint64_t key = row < n_rows ? make_key(total, static_cast<uint32_t>(row), DOT) : kKeyMax;
            if (min_keys && key <= min_keys[q]) key = kKeyMax;  // paged results: only keys after the previous page
            tk.offer(key, lane);
        }
    }
    for (; tile < end; tile += step) {
        const float total = sq8_row_score<DOT>(tiles + (tile * groups) * 64 + lane, groups, full, tail, qv, mins, inv);
        const int64_t row = tile * 64 + lane;
        uint64_t key = row < n_rows ? make_key(total, static_cast<uint32_t>(row), DOT) : kKeyMax;
        if (min_keys && key <= min_keys[q]) key = kKeyMax;  // paged results: only keys after the previous page
        tk.offer(key, lane);
    }
    wg_rank_merge<WAVES>(tk, lists, valid, wave, lane, tid, k,
                          partial + (static_cast<int64_t>(q) * slices + s) * k);
}

// Partition-probed SQ8 scan (flat/segment.go:727-744 over the :517-604 branch): workgroup =
// (slice of one probed partition's tiles, probe, query); rows outside the partition's range are masked.
template <bool DOT, bool MASKED>
__global__ __launch_bounds__(kSqThreads) void sq8_probe_kernel(
    const uint4 *__restrict__ tiles, int64_t n_rows, int groups, int dim, const float *__restrict__ queries,
    const float *__restrict__ mins, const float *__restrict__ inv, const uint32_t *__restrict__ probes,
    const uint32_t *__restrict__ part_off, int np, int sub, int k, uint64_t *__restrict__ partial,
    const uint64_t *__restrict__ min_keys, const uint8_t *__restrict__ mask, int64_t mask_stride)
{
    __shared__ uint64_t lists[kSqWaves * 64];
    __shared__ int valid[kSqWaves];
    const int s = blockIdx.x, j = blockIdx.y;
    const int64_t q = blockIdx.z;
    const uint32_t p = probes[q * np + j];
    const int64_t R0 = part_off[p], R1 = part_off[p + 1];
    const int64_t tt0 = R0 >> 6, tt1 = (R1 + 63) >> 6;
    const int64_t t0 = tt0 + (tt1 - tt0) * s / sub, t1 = tt0 + (tt1 - tt0) * (s + 1) / sub;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *qv = queries + q * dim;
    const uint8_t *mq = MASKED ? mask + q * mask_stride : nullptr;
    const int full = dim >> 4, tail = dim & 15;
    WaveTopK tk;
    tk.init(k);
    for (int64_t tile = t0 + wave; tile < t1; tile += kSqWaves) {
        const int64_t row = tile * 64 + lane;
        // filter.Matches after the batch was scored (segment.go:559-561): the candidates are the rows that pass
        const bool live = row >= R0 && row < R1 && row < n_rows && (!MASKED || mask_bit(mq, row));
        if (MASKED && !__any(live)) continue;  // a tile the filter leaves nothing of: its codes are not read
        const float total = sq8_row_score<DOT>(tiles + (tile * groups) * 64 + lane, groups, full, tail, qv, mins, inv);
        uint64_t key = live ? make_key(total, static_cast<uint32_t>(row), DOT) : kKeyMax;
        if (min_keys && key <= min_keys[q]) key = kKeyMax;  // paged results (k > 64)
        tk.offer(key, lane);
    }
    wg_rank_merge<kSqWaves>(tk, lists, valid, wave, lane, tid, k, partial + ((q * np + j) * sub + s) * k);
}

// The same with the pairs grouped by partition (k_probe.hip): a lane decodes its row's 16 codes of a
// dimension group ONCE and applies them to up to kProbeQB queries held in LDS — the decode (cvt + fma)
// and the code traffic are shared, each query keeps its own 16 lane accumulators (L2) or running sum
// (DotProduct), i.e. exactly the arithmetic of sq8_row_score per (row, query).
constexpr int kSqProbeQ = 4;  // queries per decode pass: 4 x 16 lane accumulators keep two waves per SIMD
template <bool DOT, bool FULL>
__device__ __forceinline__ void sq8_row_scores_mq(const uint4 *__restrict__ tp, int groups, int full, int tail, int cnt,
                                                  const float *qlds, int dimp, const float *__restrict__ mins,
                                                  const float *__restrict__ inv, float (&total)[kSqProbeQ])
{
    float acc[DOT ? 1 : kSqProbeQ][16];
    float run[kSqProbeQ];
#pragma unroll
    for (int qi = 0; qi < kSqProbeQ; qi++) {
        run[qi] = 0.0f;
        if (!DOT) {
#pragma unroll
            for (int l = 0; l < 16; l++) acc[qi][l] = 0.0f;
        }
    }
    uint4 ring[kSqAhead];
    const int glast = groups - 1;
#pragma unroll
    for (int a = 0; a < kSqAhead; a++) ring[a] = tp[(a < glast ? a : glast) * 64];
    const int ngr = full + (tail ? 1 : 0);
    for (int g0 = 0; g0 < ngr; g0 += kSqAhead) {
#pragma unroll
        for (int a = 0; a < kSqAhead; a++) {
            const int g = g0 + a;
            const uint4 c = ring[a];
            const int gn = g + kSqAhead;
            ring[a] = tp[(gn < glast ? gn : glast) * 64];
            if (g >= ngr) continue;
            if (g == full && !DOT) continue;  // the L2 tail is added after the lane tree, below
            const int lim = g < full ? 16 : tail;
            const uint32_t w[4] = {c.x, c.y, c.z, c.w};
            const float *mn = mins + g * 16, *iv = inv + g * 16;
            float rec[16];
#pragma unroll
            for (int l = 0; l < 16; l++) {
                const float cf = static_cast<float>((w[l >> 2] >> (8 * (l & 3))) & 0xFFu);
                if (DOT) {
                    const float t = cf * iv[l < lim ? l : 0];
                    rec[l] = mn[l < lim ? l : 0] + t;
                } else {
                    rec[l] = __builtin_fmaf(cf, iv[l], mn[l]);
                }
            }
#pragma unroll
            for (int qi = 0; qi < kSqProbeQ; qi++) {
                if (FULL || qi < cnt) {
                    const float4 *q4 = reinterpret_cast<const float4 *>(qlds + qi * dimp + g * 16);
                    float qv[16];
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        const float4 x = q4[t];
                        qv[4 * t] = x.x; qv[4 * t + 1] = x.y; qv[4 * t + 2] = x.z; qv[4 * t + 3] = x.w;
                    }
                    if (DOT) {
#pragma unroll
                        for (int l = 0; l < 16; l++)
                            if (l < lim) {
                                const float prod = qv[l] * rec[l];
                                run[qi] = run[qi] + prod;
                            }
                    } else {
#pragma unroll
                        for (int l = 0; l < 16; l++) {
                            const float diff = qv[l] - rec[l];
                            acc[qi][l] = __builtin_fmaf(diff, diff, acc[qi][l]);
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int qi = 0; qi < kSqProbeQ; qi++) {
        if (DOT) {
            total[qi] = run[qi];
        } else if (FULL || qi < cnt) {
            float t = reduce16_regs(acc[qi]);
            if (tail) t = sq8_tail(t, tp[full * 64], tail, qlds + qi * dimp + full * 16, mins + full * 16, inv + full * 16);
            total[qi] = t;
        } else {
            total[qi] = 0.0f;
        }
    }
}

// Exhaustive scan of several queries: workgroup = (group of kSqProbeQ queries, slice), every code decoded
// once per group.  Same block order as sq8_scan_kernel: the groups of one slice share an XCD's L2.
template <bool DOT>
__global__ __launch_bounds__(kSqThreads) void sq8_scan_mq_kernel(
    const uint4 *__restrict__ tiles, int64_t n_rows, int64_t n_tiles, int groups, int dim,
    const float *__restrict__ queries, const float *__restrict__ mins, const float *__restrict__ inv, int slices,
    int nq, int k, uint64_t *__restrict__ partial, const uint64_t *__restrict__ min_keys)
{
    extern __shared__ __attribute__((aligned(16))) float qlds[];  // kSqProbeQ * dimp floats, then the merge scratch
    const int dimp = groups * 16;
    uint64_t *lists = reinterpret_cast<uint64_t *>(qlds + static_cast<size_t>(kSqProbeQ) * dimp);
    int *valid = reinterpret_cast<int *>(lists + kSqWaves * 64);
    const int ng = (nq + kSqProbeQ - 1) / kSqProbeQ;
    const int b = blockIdx.x;
    const int xcd = b & 7;
    const int o = b >> 3;
    const int qg = o % ng;
    const int s = (o / ng) * 8 + xcd;
    const int q0 = qg * kSqProbeQ;
    const int cnt = nq - q0 < kSqProbeQ ? nq - q0 : kSqProbeQ;
    const int64_t t0 = n_tiles * s / slices, t1 = n_tiles * (s + 1) / slices;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int qi = 0; qi < cnt; qi++) {
        const float *src = queries + static_cast<int64_t>(q0 + qi) * dim;
        for (int t = tid; t < dimp; t += kSqThreads) qlds[qi * dimp + t] = t < dim ? src[t] : 0.0f;
    }
    __syncthreads();
    const int full = dim >> 4, tail = dim & 15;
    WaveTopK tk[kSqProbeQ];
#pragma unroll
    for (int qi = 0; qi < kSqProbeQ; qi++) tk[qi].init(k);
    for (int64_t tile = t0 + wave; tile < t1; tile += kSqWaves) {
        const uint4 *tp = tiles + (tile * groups) * 64 + lane;
        float total[kSqProbeQ];
        if (cnt == kSqProbeQ)
            sq8_row_scores_mq<DOT, true>(tp, groups, full, tail, cnt, qlds, dimp, mins, inv, total);
        else
            sq8_row_scores_mq<DOT, false>(tp, groups, full, tail, cnt, qlds, dimp, mins, inv, total);
        const int64_t row = tile * 64 + lane;
#pragma unroll
        for (int qi = 0; qi < kSqProbeQ; qi++)
            if (qi < cnt) {
                uint64_t key = row < n_rows ? make_key(total[qi], static_cast<uint32_t>(row), DOT) : kKeyMax;
                if (min_keys && key <= min_keys[q0 + qi]) key = kKeyMax;
                tk[qi].offer(key, lane);
            }
    }
#pragma unroll
    for (int qi = 0; qi < kSqProbeQ; qi++) {
        if (qi < cnt) {
            wg_rank_merge<kSqWaves>(tk[qi], lists, valid, wave, lane, tid, k,
                                    partial + (static_cast<int64_t>(q0 + qi) * slices + s) * k);
            __syncthreads();
        }
    }
}

template <bool DOT>
__global__ __launch_bounds__(kSqThreads) void sq8_probe_mq_kernel(
    const uint4 *__restrict__ tiles, int64_t n_rows, int groups, int dim, const float *__restrict__ queries,
    const float *__restrict__ mins, const float *__restrict__ inv, const uint32_t *__restrict__ part_off,
    const uint32_t *__restrict__ pair_of, const ProbeGroup *__restrict__ pgroups, const uint32_t *__restrict__ ngroups,
    int np, int sub, int k, uint64_t *__restrict__ partial, const uint64_t *__restrict__ min_keys,
    const uint8_t *__restrict__ mask, int64_t mask_stride)
{
    extern __shared__ __attribute__((aligned(16))) float qlds[];  // kProbeQB * dimp floats, then the merge scratch
    const int dimp = groups * 16;
    uint64_t *lists = reinterpret_cast<uint64_t *>(qlds + static_cast<size_t>(kProbeQB) * dimp);
    int *valid = reinterpret_cast<int *>(lists + kSqWaves * 64);
    __shared__ uint32_t pair[kProbeQB];
    if (blockIdx.y >= ngroups[0]) return;
    const ProbeGroup pg = pgroups[blockIdx.y];
    const int s = blockIdx.x, cnt = static_cast<int>(pg.count);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < cnt) pair[tid] = pair_of[pg.first + tid];
    __syncthreads();
    for (int qi = 0; qi < cnt; qi++) {
        const float *src = queries + static_cast<int64_t>(pair[qi] / np) * dim;
        for (int t = tid; t < dimp; t += kSqThreads) qlds[qi * dimp + t] = t < dim ? src[t] : 0.0f;
    }
    __syncthreads();
    const int64_t R0 = part_off[pg.part], R1 = part_off[pg.part + 1];
    const int64_t tt0 = R0 >> 6, tt1 = (R1 + 63) >> 6;
    const int64_t t0 = tt0 + (tt1 - tt0) * s / sub, t1 = tt0 + (tt1 - tt0) * (s + 1) / sub;
    const int full = dim >> 4, tail = dim & 15;
    WaveTopK tk[kProbeQB];
#pragma unroll
    for (int qi = 0; qi < kProbeQB; qi++) tk[qi].init(k);
    for (int64_t tile = t0 + wave; tile < t1; tile += kSqWaves) {
        const uint4 *tp = tiles + (tile * groups) * 64 + lane;
        const int64_t row = tile * 64 + lane;
        const bool live = row >= R0 && row < R1 && row < n_rows;
        if (mask && mask_stride == 0 && !__any(live && mask_bit(mask, row))) continue;  // nothing of the tile passes the filter
        // the group's queries in passes of kSqProbeQ (the second pass finds the tile's codes in L1 / L2)
#pragma unroll
        for (int qb = 0; qb < kProbeQB; qb += kSqProbeQ) {
            if (qb < cnt) {
                float total[kSqProbeQ];
                const int left = cnt - qb;
                if (left >= kSqProbeQ)
                    sq8_row_scores_mq<DOT, true>(tp, groups, full, tail, left, qlds + qb * dimp, dimp, mins, inv, total);
                else
                    sq8_row_scores_mq<DOT, false>(tp, groups, full, tail, left, qlds + qb * dimp, dimp, mins, inv, total);
#pragma unroll
                for (int qi = 0; qi < kSqProbeQ; qi++)
                    if (qb + qi < cnt) {
                        uint64_t key = live ? make_key(total[qi], static_cast<uint32_t>(row), DOT) : kKeyMax;
                        if (min_keys && key <= min_keys[pair[qb + qi] / np]) key = kKeyMax;  // paged results (k > 64)
                        if (mask && live && !mask_bit(mask + static_cast<int64_t>(pair[qb + qi] / np) * mask_stride, row))
                            key = kKeyMax;  // filter.Matches (segment.go:559-561), each query its own mask
                        tk[qb + qi].offer(key, lane);
                    }
            }
        }
    }
#pragma unroll
    for (int qi = 0; qi < kProbeQB; qi++) {
        if (qi < cnt) {
            wg_rank_merge<kSqWaves>(tk[qi], lists, valid, wave, lane, tid, k,
                                    partial + (static_cast<int64_t>(pair[qi]) * sub + s) * k);
            __syncthreads();
        }
    }
}

int32_t launch_probe_scan_sq8_grouped(const vg_index *idx, const float *queries, const uint32_t *part_off,
                                      const uint32_t *pair_of, const ProbeGroup *groups, const uint32_t *ngroups, unsigned gmax,
                                      int np, int sub, int k, uint64_t *partial, const uint64_t *min_keys, const uint8_t *mask,
                                      int64_t mask_stride, hipStream_t st)
{
    const bool dot = idx->metric != VG_METRIC_L2;
    auto kern = dot ? sq8_probe_mq_kernel<true> : sq8_probe_mq_kernel<false>;
    const size_t lds = sizeof(float) * kProbeQB * static_cast<size_t>(idx->sq_groups) * 16 + kSqWaves * 64 * sizeof(uint64_t) + 64;
    VG_CHECK(lds <= 152 * 1024, VG_ERR_UNSUPPORTED, "sq8 grouped probe: %d dimensions do not fit LDS", idx->dim);
    VG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                               static_cast<int>(lds)));
    ProfScope prof(idx->ctx, "sq8_probe", st);
    VG_LAUNCH(kern, dim3(static_cast<unsigned>(sub), gmax), dim3(kSqThreads), lds, st,
              reinterpret_cast<const uint4 *>(idx->d_sq_tiles), idx->n, idx->sq_groups, idx->dim, queries, idx->sq->d_mins,
              idx->sq->d_inv, part_off, pair_of, groups, ngroups, np, sub, k, partial, min_keys, mask, mask_stride);
    return VG_OK;
}

int32_t launch_probe_scan_sq8(const vg_index *idx, const float *queries, const uint32_t *probes, const uint32_t *part_off,
                              int64_t nq, int np, int sub, int k, uint64_t *partial, const uint64_t *min_keys,
                              const uint8_t *mask, int64_t mask_stride, hipStream_t st)
{
    for (int64_t q0 = 0; q0 < nq; q0 += 65535) {
        const int64_t cnt = nq - q0 < 65535 ? nq - q0 : 65535;
        ProfScope prof(idx->ctx, "sq8_probe", st);
        auto kern = mask ? (idx->metric != VG_METRIC_L2 ? sq8_probe_kernel<true, true> : sq8_probe_kernel<false, true>)
                         : (idx->metric != VG_METRIC_L2 ? sq8_probe_kernel<true, false> : sq8_probe_kernel<false, false>);
        VG_LAUNCH(kern, dim3(static_cast<unsigned>(sub), static_cast<unsigned>(np), static_cast<unsigned>(cnt)),
                  dim3(kSqThreads), 0, st, reinterpret_cast<const uint4 *>(idx->d_sq_tiles), idx->n, idx->sq_groups, idx->dim,
                  queries + q0 * idx->dim, idx->sq->d_mins, idx->sq->d_inv, probes + q0 * np, part_off, np, sub, k,
                  partial + q0 * np * sub * k, min_keys ? min_keys + q0 : nullptr, mask ? mask + q0 * mask_stride : nullptr,
                  mask_stride);
    }
    return VG_OK;
}

// ==== INT4 (internal/quantization/int4.go, internal/simd/src/int4_avx512.c) ===========================
// stage 2 of Train (int4.go:52-61): diff = max - min, 0 -> 1; then BuildInt4LookupTable
// (kernels.go:94-103): table[d*16+q] = (float32(q)/15.0)*diff + min, three rounded operations
__global__ void int4_finish_kernel(const float *__restrict__ pmin, const float *__restrict__ pmax, int chunks,
                                   int dim, bool from_train, float *__restrict__ mins, float *__restrict__ diff)
{
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= dim) return;
    if (from_train) {
        float mn = kF32Max, mx = -kF32Max;
        for (int c = 0; c < chunks; c++) {
            const float a = pmin[static_cast<int64_t>(c) * dim + d], b = pmax[static_cast<int64_t>(c) * dim + d];
            if (a < mn) mn = a;
            if (b > mx) mx = b;
        }
        const float df = mx - mn;
        mins[d] = mn;
        diff[d] = df == 0.0f ? 1.0f : df;
    }
}

__global__ void int4_table_kernel(const float *__restrict__ mins, const float *__restrict__ diff, int dim,
                                  float *__restrict__ table)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= dim * 16) return;
    const int d = t >> 4, q = t & 15;
    const float a = static_cast<float>(q) / 15.0f;
    const float b = a * diff[d];
    table[t] = b + mins[d];
}

__device__ __forceinline__ uint32_t int4_quant(float v, float mn, float df)
{
    float norm = (v - mn) / df;  // int4.go:75-81
    if (norm < 0.0f)
        norm = 0.0f;
    else if (norm > 1.0f)
        norm = 1.0f;
    return static_cast<uint32_t>(round(static_cast<double>(norm) * 15.0));  // math.Round(float64(norm) * 15)
}

// Encode (int4.go:65-105): thread per output byte
__global__ void int4_encode_kernel(const float *__restrict__ v, int64_t n, int dim, const float *__restrict__ mins,
                                   const float *__restrict__ diff, uint8_t *__restrict__ out)
{
    const int cs = (dim + 1) / 2;
    const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t >= n * cs) return;
    const int64_t row = t / cs;
    const int i = static_cast<int>(t % cs) * 2;
    const float *x = v + row * dim;
    const uint32_t q1 = int4_quant(x[i], mins[i], diff[i]);
    const uint32_t q2 = i + 1 < dim ? int4_quant(x[i + 1], mins[i + 1], diff[i + 1]) : 0u;
    out[t] = static_cast<uint8_t>((q1 << 4) | (q2 & 0x0Fu));
}

// Decode (int4.go:108-130): float32(q)/15.0*diff + min, left to right
__global__ void int4_decode_kernel(const uint8_t *__restrict__ codes, int64_t n, int dim,
                                   const float *__restrict__ mins, const float *__restrict__ diff,
                                   float *__restrict__ out)
{
    const int cs = (dim + 1) / 2;
    const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t >= n * dim) return;
    const int64_t row = t / dim;
    const int i = static_cast<int>(t % dim);
    const uint8_t b = codes[row * cs + i / 2];
    const float a = static_cast<float>((i & 1) ? (b & 0x0F) : (b >> 4)) / 15.0f;
    const float c = a * diff[i];
    out[t] = c + mins[i];
}

// dim % 8 == 0, aligned buffers: a thread owns eight consecutive dimensions = four code bytes (see sq8_encode4_kernel)
__global__ __launch_bounds__(256) void int4_encode8_kernel(const float *__restrict__ v, int64_t n, int dim,
                                                           const float *__restrict__ mins, const float *__restrict__ diff,
                                                           uint8_t *__restrict__ out, int rpt)
{
    const int cg = blockIdx.x * blockDim.x + threadIdx.x;
    if (cg * 8 >= dim) return;
    const float4 mn0 = *reinterpret_cast<const float4 *>(mins + cg * 8), mn1 = *reinterpret_cast<const float4 *>(mins + cg * 8 + 4),
                 df0 = *reinterpret_cast<const float4 *>(diff + cg * 8), df1 = *reinterpret_cast<const float4 *>(diff + cg * 8 + 4);
    const int cs = dim >> 1;
    const int64_t r0 = static_cast<int64_t>(blockIdx.y) * rpt;
    for (int64_t row = r0; row < r0 + rpt && row < n; row++) {
        const float4 a = *reinterpret_cast<const float4 *>(v + row * dim + cg * 8), b = *reinterpret_cast<const float4 *>(v + row * dim + cg * 8 + 4);
        const uint32_t b0 = (int4_quant(a.x, mn0.x, df0.x) << 4) | (int4_quant(a.y, mn0.y, df0.y) & 0x0Fu);
        const uint32_t b1 = (int4_quant(a.z, mn0.z, df0.z) << 4) | (int4_quant(a.w, mn0.w, df0.w) & 0x0Fu);
        const uint32_t b2 = (int4_quant(b.x, mn1.x, df1.x) << 4) | (int4_quant(b.y, mn1.y, df1.y) & 0x0Fu);
        const uint32_t b3 = (int4_quant(b.z, mn1.z, df1.z) << 4) | (int4_quant(b.w, mn1.w, df1.w) & 0x0Fu);
        *reinterpret_cast<uint32_t *>(out + row * cs + cg * 4) = (b0 & 0xFFu) | ((b1 & 0xFFu) << 8) | ((b2 & 0xFFu) << 16) | (b3 << 24);
    }
}
__global__ __launch_bounds__(256) void int4_decode8_kernel(const uint8_t *__restrict__ codes, int64_t n, int dim,
                                                           const float *__restrict__ mins, const float *__restrict__ diff,
                                                           float *__restrict__ out, int rpt)
{
    const int cg = blockIdx.x * blockDim.x + threadIdx.x;
    if (cg * 8 >= dim) return;
    const float4 mn0 = *reinterpret_cast<const float4 *>(mins + cg * 8), mn1 = *reinterpret_cast<const float4 *>(mins + cg * 8 + 4),
                 df0 = *reinterpret_cast<const float4 *>(diff + cg * 8), df1 = *reinterpret_cast<const float4 *>(diff + cg * 8 + 4);
    const int cs = dim >> 1;
    const int64_t r0 = static_cast<int64_t>(blockIdx.y) * rpt;
    auto dec = [](uint32_t q, float df, float mn) -> float {
        const float a = static_cast<float>(q) / 15.0f;
        const float c = a * df;
        return c + mn;
    };
    for (int64_t row = r0; row < r0 + rpt && row < n; row++) {
        const uint32_t w = *reinterpret_cast<const uint32_t *>(codes + row * cs + cg * 4);
        float4 o0, o1;
        o0.x = dec((w >> 4) & 0xFu, df0.x, mn0.x);
        o0.y = dec(w & 0xFu, df0.y, mn0.y);
        o0.z = dec((w >> 12) & 0xFu, df0.z, mn0.z);
        o0.w = dec((w >> 8) & 0xFu, df0.w, mn0.w);
        o1.x = dec((w >> 20) & 0xFu, df1.x, mn1.x);
        o1.y = dec((w >> 16) & 0xFu, df1.y, mn1.y);
        o1.z = dec(w >> 28, df1.z, mn1.z);
        o1.w = dec((w >> 24) & 0xFu, df1.w, mn1.w);
        *reinterpret_cast<float4 *>(out + row * dim + cg * 8) = o0;
        *reinterpret_cast<float4 *>(out + row * dim + cg * 8 + 4) = o1;
    }
}

__device__ __forceinline__ float int4_nib(const uint8_t *code, int j)
{
    const uint8_t b = code[j >> 1];
    return static_cast<float>((j & 1) ? (b & 0x0F) : (b >> 4));
}

// int4L2DistanceBatchAvx512 (int4_avx512.c:191-299), lane per row: 64-element blocks feed sub-blocks
// 0,1 into sum1 and 2,3 into sum2, 32-element blocks both into sum1; dq = fma(f * (1/15), diff, min)
__global__ __launch_bounds__(256) void int4_l2_batch_kernel(const float *__restrict__ query,
                                                            const uint8_t *__restrict__ codes, int64_t n, int dim,
                                                            const float *__restrict__ mins,
                                                            const float *__restrict__ diff, float *__restrict__ out)
{
    const int64_t row = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (row >= n) return;
    const uint8_t *code = codes + row * ((dim + 1) / 2);
    const float sc = __uint_as_float(0x3d888889u);  // int4_avx512.c:35
    float s1[16], s2[16];
#pragma unroll
    for (int l = 0; l < 16; l++) s1[l] = s2[l] = 0.0f;
    auto block = [&](float (&acc)[16], int base) {
#pragma unroll
        for (int l = 0; l < 16; l++) {
            const int j = base + l;
            const float f = int4_nib(code, j) * sc;
            const float dq = __builtin_fmaf(f, diff[j], mins[j]);
            const float d = query[j] - dq;
            acc[l] = __builtin_fmaf(d, d, acc[l]);
        }
    };
    int i = 0;
    for (; i <= dim - 64; i += 64) {
        block(s1, i);
        block(s1, i + 16);
        block(s2, i + 32);
        block(s2, i + 48);
    }
    for (; i <= dim - 32; i += 32) {
        block(s1, i);
        block(s1, i + 16);
    }
#pragma unroll
    for (int l = 0; l < 16; l++) s1[l] = s1[l] + s2[l];
    float total = reduce16_regs(s1);
    for (; i < dim; i++) {
        const float f = int4_nib(code, i) * sc;
        const float v = __builtin_fmaf(f, diff[i], mins[i]);
        const float d = query[i] - v;
        total = __builtin_fmaf(d, d, total);
    }
    out[row] = total;
}

__global__ __launch_bounds__(256) void int4_l2_precomputed_kernel(const float *__restrict__ query,
                                                                  const uint8_t *__restrict__ codes, int64_t n,
                                                                  int dim, const float *__restrict__ table,
                                                                  float *__restrict__ out)
{
    const int64_t row = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (row >= n) return;
    out[row] = int4_l2_precomputed(query, codes + row * ((dim + 1) / 2), dim, table);
}

// Both distances as a streaming scan (dim % 64 == 0): the lane-per-row kernels above read their rows 16 bytes at a
// time at a dim/2-byte stride (64 lines per wave-instruction; 1.0 / 2.2 TB/s of codes at dim 768) and look every value
// up in a 48 KiB table.  Here a wave takes 64 rows: 128-byte pieces of them (256 dimensions) arrive as whole lines
// (8 lanes per row) and are turned through the wave's LDS (row stride 144 bytes = 16 x 9: the 16 lanes of a
// ds_read_b128 group never share a bank slot), each lane then walks ITS row; a code byte becomes its two values by ONE
// read of a 256-entry pair table in LDS (PRE: float(v) / 15, the table's own factor — int4.go:152-163; batch order:
// float(v) * 0x3d888889 — int4_avx512.c:35), the two values of a byte are neighbouring AVX-512 lanes, so every step is
// one packed-fp32 instruction on the pair with scalar-loaded diff / min / query: 2 - 2.5 vector instructions per
// dimension.  Accumulators and their order are the kernels' above: PRE — both 16-element halves of a 32-block into
// sum[]; batch order — sub-blocks 0, 1 of a 64-block into s1, 2, 3 into s2, s1 += s2 at the end.
constexpr int kI4Waves = 4;
constexpr int kI4Stride = 144;  // LDS bytes per staged row piece (128 + 16)
template <bool PRE>
__global__ __launch_bounds__(kI4Waves * 64) void int4_scan_kernel(const float *__restrict__ query,
                                                                  const uint8_t *__restrict__ codes, int64_t n, int dim,
                                                                  const float *__restrict__ mins,
                                                                  const float *__restrict__ diff, float *__restrict__ out)
{
    __shared__ vg_f2v pairs[256];
    __shared__ __attribute__((aligned(16))) unsigned char stage_all[kI4Waves][64 * kI4Stride];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {
        const float sc = __uint_as_float(0x3d888889u);
        const int b = tid;  // 256 threads, 256 entries
        vg_f2v a;
        a.x = PRE ? static_cast<float>(b >> 4) / 15.0f : static_cast<float>(b >> 4) * sc;
        a.y = PRE ? static_cast<float>(b & 15) / 15.0f : static_cast<float>(b & 15) * sc;
        pairs[b] = a;
    }
    __syncthreads();
    const int64_t tile = static_cast<int64_t>(blockIdx.x) * kI4Waves + wave;
    const int64_t row0 = tile * 64;
    if (row0 >= n) return;
    unsigned char *stage = stage_all[wave];
    const int row_bytes = dim >> 1;
    vg_f2v s1[8], s2[8];
#pragma unroll
    for (int p = 0; p < 8; p++) s1[p] = s2[p] = vg_f2v{0.0f, 0.0f};
    // one 32-element block (16 code bytes, `piece` of the staged row piece) into the accumulators
    auto block32 = [&](int cb0, int piece, auto second_c) {
        const uint4 c = *reinterpret_cast<const uint4 *>(stage + lane * kI4Stride + piece * 16);
        const uint32_t w[4] = {c.x, c.y, c.z, c.w};
        const int j0 = (cb0 + piece * 16) * 2;    // first dimension of the block
        constexpr bool second = !PRE && decltype(second_c)::value;  // batch order: sub-blocks 2, 3 of the 64-block
#pragma unroll
        for (int b = 0; b < 16; b++) {
            const uint32_t byte = (w[b >> 2] >> (8 * (b & 3))) & 0xFFu;
            const int j = j0 + 2 * b;
            const vg_f2v a = pairs[byte];
            const vg_f2v df = *reinterpret_cast<const vg_f2v *>(diff + j);
            const vg_f2v mn = *reinterpret_cast<const vg_f2v *>(mins + j);
            const vg_f2v qq = *reinterpret_cast<const vg_f2v *>(query + j);
            vg_f2v t;
            if (PRE) {
                t = a * df;
                t = t + mn;
            } else {
                t = __builtin_elementwise_fma(a, df, mn);
            }
            const vg_f2v d = qq - t;
            if (second)
                s2[b & 7] = __builtin_elementwise_fma(d, d, s2[b & 7]);
            else
                s1[b & 7] = __builtin_elementwise_fma(d, d, s1[b & 7]);
        }
    };
    // rows past n re-read row n - 1 (their result is not stored); loads are unguarded (a guarded load makes hipcc
    // wait for the previous one at the join)
    if ((row_bytes & 127) == 0) {
        // whole 128-byte pieces: 8 lanes per row, the next piece's lines in flight while this one is scored
        const int r = lane >> 3, part = lane & 7;
        const uint8_t *src[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int64_t row = row0 + r + 8 * k < n ? row0 + r + 8 * k : n - 1;
            src[k] = codes + row * row_bytes + part * 16;
        }
        uint4 u[8];
#pragma unroll
        for (int k = 0; k < 8; k++) u[k] = load_stream(reinterpret_cast<const uint4 *>(src[k]));
        for (int cb0 = 0; cb0 < row_bytes; cb0 += 128) {
#pragma unroll
            for (int k = 0; k < 8; k++) *reinterpret_cast<uint4 *>(stage + (r + 8 * k) * kI4Stride + part * 16) = u[k];
            const int nxt = cb0 + 128 < row_bytes ? cb0 + 128 : cb0;  // (the last piece again: unused)
#pragma unroll
            for (int k = 0; k < 8; k++) u[k] = load_stream(reinterpret_cast<const uint4 *>(src[k] + nxt));
            for (int piece = 0; piece < 8; piece += 2) {
                block32(cb0, piece, std::false_type{});
                block32(cb0, piece + 1, std::true_type{});
            }
        }
    } else {
        for (int cb0 = 0; cb0 < row_bytes; cb0 += 128) {
            const int cb = row_bytes - cb0 < 128 ? row_bytes - cb0 : 128;  // bytes of this piece (a multiple of 32)
            const int per_row = cb >> 4;                                    // 16-byte units per row
            const int units = 64 * per_row;
            uint4 u[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int e = lane + 64 * k < units ? lane + 64 * k : units - 1;
                const int r = e / per_row, part = e - r * per_row;
                const int64_t row = row0 + r < n ? row0 + r : n - 1;
                u[k] = load_stream(reinterpret_cast<const uint4 *>(codes + row * row_bytes + cb0 + part * 16));
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int e = lane + 64 * k;
                if (e < units) {
                    const int r = e / per_row, part = e - r * per_row;
                    *reinterpret_cast<uint4 *>(stage + r * kI4Stride + part * 16) = u[k];
                }
            }
            for (int piece = 0; piece < per_row; piece += 2) {  // per_row is even (dim % 64 == 0)
                block32(cb0, piece, std::false_type{});
                block32(cb0, piece + 1, std::true_type{});
            }
        }
    }
    float s16[16];
#pragma unroll
    for (int p = 0; p < 8; p++) {
        const vg_f2v v = PRE ? s1[p] : s1[p] + s2[p];
        s16[2 * p] = v.x;
        s16[2 * p + 1] = v.y;
    }
    const float total = reduce16_regs(s16);
    if (row0 + lane < n) out[row0 + lane] = total;
}

// The same scan with the lookups free of bank conflicts (dim <= 1024): the pair table above puts a wave's 64 random
// bytes on 32 bank slots — 62 % of its LDS cycles were conflicts and the LDS array was 89 % busy (3.65 TB/s of codes).
// Here the workgroup (12 waves, one per CU, persistent over the tiles) builds the quantizer's own dim x 16 value table
// in LDS (48 KiB at dim 768; PRE: float(v) / 15 * diff + min as BuildInt4LookupTable does; batch order:
// fma(float(v) * 0x3d888889, diff, min)): the 64 lanes of a lookup share the dimension, so they touch at most 16
// consecutive dwords — distinct banks or the same address.  A lookup's address is ONE v_perm_b32 (byte k of the
// pre-masked nibbles under the block's base; the dimension's offset is the instruction's immediate), a dimension
// costs 2 vector instructions (the table holds query[j] - value: a launch serves one query) and 2 LDS cycles per wave
// instead of ~3.5.  Rows are staged a whole 128-byte line at a time (stride 144 = 16 x 9; 12 waves beside the table:
// with 64-byte pieces and 16 waves the second half of a line was requested a step after the first and had often left
// L2 by then — the lines in flight on an XCD are about its 4 MiB — 1.34x the codes' bytes crossed the fabric).
// (the lookups are issued in inline asm, eight at a time — the four code bytes of one dword — and a group is retired by
// a COUNTED wait while the next group's eight are in flight: hipcc re-used one register pair per lookup and waited out
// every LDS round trip.  hipcc does not track asm loads: every value is an in/out operand of the wait statement, so no
// consumer can be scheduled above it.  LDS operations retire in order, scalar loads do not: nothing in the loop may
// issue one, which is one reason the table holds query[j] - value and not the value)
template <int OFF>
__device__ __forceinline__ float i4_lds_read(uint32_t addr)
{
    float v;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
struct I4Vals8 {
    float v[8];  // (hi, lo) of code bytes 4 WI .. 4 WI + 3
};
template <int N>
__device__ __forceinline__ void i4_lds_wait(I4Vals8 &x)
{
    asm volatile("s_waitcnt lgkmcnt(%8)"
                 : "+v"(x.v[0]), "+v"(x.v[1]), "+v"(x.v[2]), "+v"(x.v[3]), "+v"(x.v[4]), "+v"(x.v[5]), "+v"(x.v[6]), "+v"(x.v[7])
                 : "n"(N));
}
template <int WI>
__device__ __forceinline__ void i4_issue_word(I4Vals8 &x, uint32_t w, uint32_t base)
{
    const uint32_t hi4 = (w >> 2) & 0x3C3C3C3Cu;  // byte k: 4 * high nibble of code byte 4 WI + k
    const uint32_t lo4 = (w << 2) & 0x3C3C3C3Cu;  //         4 * low nibble
#define VG_I4_ONE(K)                                                                                              \
    x.v[2 * K] = i4_lds_read<(2 * (4 * WI + K)) * 64>(__builtin_amdgcn_perm(base, hi4, 0x07060500u | K));         \
    x.v[2 * K + 1] = i4_lds_read<(2 * (4 * WI + K) + 1) * 64>(__builtin_amdgcn_perm(base, lo4, 0x07060500u | K));
    VG_I4_ONE(0)
    VG_I4_ONE(1)
    VG_I4_ONE(2)
    VG_I4_ONE(3)
#undef VG_I4_ONE
}

#ifdef VG_I4_TIMING  // stage probe (tools/build_variant.sh): s_memtime per phase, totals written over out[] by lane 0
#define VG_I4_T(var) const int64_t var = static_cast<int64_t>(__builtin_readcyclecounter())
#define VG_I4_TACC(acc, a, b) (acc) += (b) - (a)
#else
#define VG_I4_T(var)
#define VG_I4_TACC(acc, a, b)
#endif
constexpr int kI4TabWaves = 12;
constexpr int kI4TabStride = 144;
constexpr int kI4TabMaxDim = 1024;

// the 16 code bytes of a 32-element block, read from the wave's staging buffer under the same in-order accounting as
// the lookups ("memory": the staging writes before it stay before it, the next piece's writes stay after the last one)
typedef uint32_t i4_u4 __attribute__((ext_vector_type(4)));  // one register tuple as an asm operand: no sub-register copies
__device__ __forceinline__ i4_u4 i4_lds_read_block(uint32_t addr)
{
    i4_u4 c;
    asm volatile("ds_read_b128 %0, %1" : "=v"(c) : "v"(addr) : "memory");
    return c;
}
template <int N>
__device__ __forceinline__ void i4_lds_wait_c(I4Vals8 &x, i4_u4 &c)
{
    asm volatile("s_waitcnt lgkmcnt(%9)"
                 : "+v"(x.v[0]), "+v"(x.v[1]), "+v"(x.v[2]), "+v"(x.v[3]), "+v"(x.v[4]), "+v"(x.v[5]), "+v"(x.v[6]), "+v"(x.v[7]),
                   "+v"(c)
                 : "n"(N));
}
__device__ __forceinline__ void i4_lds_drain(i4_u4 &c)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c) : : "memory");
}

// r04: the lookups of a whole 128-byte piece (8 blocks x 4 words x 8 lookups) run as ONE software pipeline — two words
// (16 lookups) in flight across block boundaries, the next block's 16 code bytes requested (asm, same queue) while the
// current block's second word is still out — where r03 drained the queue at the end of every 32-element block and then
// waited out the read of the next block's bytes: two LDS round trips per 32 dimensions with nothing of this wave in
// flight (stage probe -DVG_I4_TIMING: 83 % of a wave's time is the lookup phase, and its rate was that of a loop with
// those bubbles, not that of the vector ALU: tools/ubench/lds_valu_overlap.hip).  LDS operations of a wave return in
// order, so "at most N outstanding" names exactly which word is back: per block the counted waits are 9, 9, 8, 8 (the
// 9s have the next block's ds_read_b128 behind them).  The rows' addresses are a per-tile scalar base + a per-lane
// 32-bit offset (global_load saddr form): no 64-bit pointer arithmetic or selects per piece (r03: 32 of the 544 vector
// instructions of a piece).  Arithmetic and its order are unchanged: block b, words 0..3, accumulator (4 wi + k) & 7.
template <bool PRE>
__global__ __launch_bounds__(kI4TabWaves * 64) void int4_scan_tab_kernel(const float *__restrict__ query,
                                                                         const uint8_t *__restrict__ codes, int64_t n, int dim,
                                                                         const float *__restrict__ mins,
                                                                         const float *__restrict__ diff, float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char i4smem[];
    float *table = reinterpret_cast<float *>(i4smem);  // [dim][16]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // a scalar: the tile's base address lives in SGPRs
    unsigned char *stage = i4smem + static_cast<size_t>(dim) * 64 + wave * (64 * kI4TabStride);
    const int waves = blockDim.x >> 6;  // 12, fewer when the table leaves less room (dim 1024: 10)
    for (int e = tid; e < dim * 16; e += blockDim.x) {
        const int d = e >> 4, v = e & 15;
        float t;
        if (PRE) {
            const float a = static_cast<float>(v) / 15.0f;  // int4_table_kernel
            const float b = a * diff[d];
            t = b + mins[d];
        } else {
            const float f = static_cast<float>(v) * __uint_as_float(0x3d888889u);  // int4_l2_batch_kernel
            t = __builtin_fmaf(f, diff[d], mins[d]);
        }
        table[e] = query[d] - t;  // the difference the kernels square: one query per launch
    }
    __syncthreads();
    const int row_bytes = dim >> 1;
    const int64_t n_tiles = (n + 63) / 64;
    const int r = lane >> 3, part = lane & 7;  // staging: 8 lanes per row (one 128-byte line), 8 rows per load
    const int64_t tile_step = static_cast<int64_t>(gridDim.x) * waves;
    int64_t tile = static_cast<int64_t>(blockIdx.x) * waves + wave;
    if (tile >= n_tiles) return;
    // the table is the first thing in LDS: its real address (0 unless something static ever lands in this kernel's LDS)
    // goes into every lookup's base, and must leave the low 11 bits of a block's base free for the nibble byte
    const uint32_t table_lds = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(table));
    const uint32_t stage_rd = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(stage)) + lane * kI4TabStride;
    if ((table_lds & 2047u) != 0) __builtin_trap();
    // per-lane byte offsets of the 8 rows this lane helps to load, relative to the tile's first row; rows past n
    // (last tile only) read row n - 1 again and are not stored
    auto offsets = [&](int64_t t, uint32_t (&vo)[8]) {
        const int64_t left = n - t * 64;  // rows in the tile
        const uint32_t lim = left >= 64 ? 63u : static_cast<uint32_t>(left - 1);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t rr = static_cast<uint32_t>(r + 8 * k);
            vo[k] = (rr < lim ? rr : lim) * static_cast<uint32_t>(row_bytes) + static_cast<uint32_t>(part * 16);
        }
    };
    uint32_t vo[8];
    offsets(tile, vo);
    const uint8_t *tbase = codes + tile * 64 * row_bytes;  // uniform
#define VG_I4_ROWS(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define VG_I4_DECL(K) uint4 u##K = load_stream(reinterpret_cast<const uint4 *>(tbase + vo[K]));
    VG_I4_ROWS(VG_I4_DECL)
#undef VG_I4_DECL
    unsigned char *wr = stage + r * kI4TabStride + part * 16;
#ifdef VG_I4_TIMING
    int64_t t_vm = 0, t_ld = 0, t_look = 0, t_tiles = 0;
    const int64_t t_begin = static_cast<int64_t>(__builtin_readcyclecounter());
#endif
    for (; tile < n_tiles; tile += tile_step) {
#ifdef VG_I4_TIMING
        t_tiles++;
#endif
        const int64_t row0 = tile * 64;
        // the wave's next tile (its first piece is requested while this tile's last one is scored); none: this tile again
        const int64_t tnext = tile + tile_step < n_tiles ? tile + tile_step : tile;
        const uint8_t *nbase = codes + tnext * 64 * row_bytes;
        uint32_t von[8];
        offsets(tnext, von);
        vg_f2v s1[8], s2[8];
#pragma unroll
        for (int p = 0; p < 8; p++) s1[p] = s2[p] = vg_f2v{0.0f, 0.0f};
        for (int cb0 = 0; cb0 < row_bytes; cb0 += 128) {
            VG_I4_T(tA);
#define VG_I4_PUT(K) *reinterpret_cast<uint4 *>(wr + 8 * K * kI4TabStride) = u##K;
            VG_I4_ROWS(VG_I4_PUT)
#undef VG_I4_PUT
            VG_I4_T(tB);
            VG_I4_TACC(t_vm, tA, tB);
            i4_u4 c = i4_lds_read_block(stage_rd);
            {   // the next piece of this tile, or the first piece of the wave's next tile: a scalar base and a per-lane
                // 32-bit offset either way (uniform selects; a branch here became per-lane 64-bit pointers again)
                const bool more = cb0 + 128 < row_bytes;
                const uint8_t *pb = more ? tbase + cb0 + 128 : nbase;
#define VG_I4_GET(K) u##K = load_stream(reinterpret_cast<const uint4 *>(pb + (more ? vo[K] : von[K])));
                VG_I4_ROWS(VG_I4_GET)
#undef VG_I4_GET
            }
            i4_lds_drain(c);  // the staging writes and the first block's bytes
            VG_I4_T(tC);
            VG_I4_TACC(t_ld, tB, tC);
            const uint32_t pbase = table_lds + static_cast<uint32_t>(cb0) * 128u;  // table row of the piece's first dimension: j * 64 bytes
            I4Vals8 va, vb;
            i4_issue_word<0>(va, c.x, pbase);
            i4_issue_word<1>(vb, c.y, pbase);
#pragma unroll
            for (int blk = 0; blk < 8; blk++) {
                const uint32_t base = pbase + static_cast<uint32_t>(blk) * 2048u;  // 32 dimensions x 64 bytes
                vg_f2v *acc = (!PRE && (blk & 1)) ? s2 : s1;
                auto take = [&](const I4Vals8 &x, int wi) {  // code bytes 4 wi .. 4 wi + 3: accumulator pairs (4 wi + k) & 7
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const vg_f2v d = {x.v[2 * k], x.v[2 * k + 1]};
                        acc[(4 * wi + k) & 7] = __builtin_elementwise_fma(d, d, acc[(4 * wi + k) & 7]);
                    }
                };
                i4_u4 cn = c;
                if (blk < 7) {
                    cn = i4_lds_read_block(stage_rd + (blk + 1) * 16);
                    i4_lds_wait<9>(va);  // behind word 0: word 1 and the next block's bytes
                    take(va, 0);
                    i4_issue_word<2>(va, c.z, base);
                    i4_lds_wait<9>(vb);  // behind word 1: the bytes and word 2
                    take(vb, 1);
                    i4_issue_word<3>(vb, c.w, base);
                    i4_lds_wait_c<8>(va, cn);  // behind word 2: word 3 — the next block's bytes are older, so they are back
                    take(va, 2);
                    i4_issue_word<0>(va, cn.x, base + 2048u);
                    i4_lds_wait<8>(vb);
                    take(vb, 3);
                    i4_issue_word<1>(vb, cn.y, base + 2048u);
                    c = cn;
                } else {
                    i4_lds_wait<8>(va);
                    take(va, 0);
                    i4_issue_word<2>(va, c.z, base);
                    i4_lds_wait<8>(vb);
                    take(vb, 1);
                    i4_issue_word<3>(vb, c.w, base);
                    i4_lds_wait<8>(va);
                    take(va, 2);
                    i4_lds_wait<0>(vb);
                    take(vb, 3);
                }
            }
            VG_I4_T(tD);
            VG_I4_TACC(t_look, tC, tD);
        }
        float s16[16];
#pragma unroll
        for (int p = 0; p < 8; p++) {
            const vg_f2v v = PRE ? s1[p] : s1[p] + s2[p];
            s16[2 * p] = v.x;
            s16[2 * p + 1] = v.y;
        }
        const float total = reduce16_regs(s16);
        if (row0 + lane < n) out[row0 + lane] = total;
        tbase = nbase;
#pragma unroll
        for (int k = 0; k < 8; k++) vo[k] = von[k];
    }
#undef VG_I4_ROWS
#ifdef VG_I4_TIMING
    if (lane == 0) {
        const int64_t t_all = static_cast<int64_t>(__builtin_readcyclecounter()) - t_begin;
        float *o = out + (static_cast<int64_t>(blockIdx.x) * waves + wave) * 8;
        o[0] = static_cast<float>(t_tiles);
        o[1] = static_cast<float>(t_all);
        o[2] = static_cast<float>(t_vm);
        o[3] = static_cast<float>(t_ld);
        o[4] = static_cast<float>(t_look);
    }
#endif
}

static int sq_slices(int64_t nq, int64_t n_tiles, int cus)
{
    int64_t s = (4 * static_cast<int64_t>(cus) + nq - 1) / nq;  // ~4 workgroups per CU
    s = ((s + 7) / 8) * 8;
    int64_t max_s = (n_tiles / 8) * 8;
    if (max_s < 8) max_s = 8;
    if (s > max_s) s = max_s;
    if (s < 8) s = 8;
    return static_cast<int>(s);
}

}  // namespace vg

// ---- C ABI --------------------------------------------------------------------------------------------
VG_API int32_t vg_sq8_create(vg_ctx *ctx, int32_t dim, vg_sq8 **out)
{
    VG_CHECK(out, VG_ERR_INVALID_ARG, "vg_sq8_create: out is NULL");
    *out = nullptr;
    VG_CHECK(ctx, VG_ERR_INVALID_ARG, "vg_sq8_create: ctx is NULL");
    VG_CHECK(dim > 0, VG_ERR_INVALID_ARG, "vg_sq8_create: dim must be positive");
    VG_HIP(hipSetDevice(ctx->device));
    vg_sq8 *sq = new vg_sq8;
    sq->ctx = ctx;
    sq->dim = dim;
    float *block = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&block), sizeof(float) * 4 * static_cast<size_t>(dim));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        delete sq;
        vg::set_error("vg_sq8_create: hipMalloc failed: %s", hipGetErrorString(e));
        return VG_ERR_HIP;
    }
    sq->d_mins = block;
    sq->d_maxs = block + dim;
    sq->d_scales = block + 2 * dim;
    sq->d_inv = block + 3 * dim;
    *out = sq;
    return VG_OK;
}

VG_API int32_t vg_sq8_destroy(vg_sq8 *sq)
{
    if (!sq) return VG_OK;
    (void)hipSetDevice(sq->ctx->device);
    if (sq->d_mins) (void)hipFree(sq->d_mins);
    delete sq;
    return VG_OK;
}

VG_API int32_t vg_sq8_is_trained(vg_sq8 *sq) { return sq && sq->trained ? 1 : 0; }

VG_API int32_t vg_sq8_train(vg_sq8 *sq, const float *vectors, int64_t n, void *stream)
{
    VG_CHECK(sq, VG_ERR_INVALID_ARG, "vg_sq8_train: NULL quantizer");
    VG_CHECK(n > 0 && vectors, VG_ERR_INVALID_ARG, "no vectors provided for training");  // quantizer.go:128-130
    VG_HIP(hipSetDevice(sq->ctx->device));
    hipStream_t st = vg::pick_stream(sq->ctx, stream);
    const int dim = sq->dim;
    vg::DevIn<float> v;
    VG_TRY(v.init(vectors, static_cast<size_t>(n) * dim, st));
    int chunks = static_cast<int>(std::min<int64_t>(n, 1024));
    vg::DevTmp<float> pmin, pmax;
    VG_TRY(pmin.init(static_cast<size_t>(chunks) * dim, st));
    VG_TRY(pmax.init(static_cast<size_t>(chunks) * dim, st));
    VG_LAUNCH(vg::sq8_minmax_kernel, dim3((dim + 255) / 256, chunks), dim3(256), 0, st, v.ptr, n, dim, chunks,
              pmin.ptr, pmax.ptr);
    VG_LAUNCH(vg::sq8_finish_kernel, dim3((dim + 255) / 256), dim3(256), 0, st, pmin.ptr, pmax.ptr, chunks, dim,
              true, sq->d_mins, sq->d_maxs, sq->d_scales, sq->d_inv);
    VG_HIP(hipStreamSynchronize(st));
    sq->trained = true;
    return VG_OK;
}

VG_API int32_t vg_sq8_set_bounds(vg_sq8 *sq, const float *mins, const float *maxs)
{
    VG_CHECK(sq, VG_ERR_INVALID_ARG, "vg_sq8_set_bounds: NULL quantizer");
    VG_CHECK(mins && maxs, VG_ERR_INVALID_ARG, "vg_sq8_set_bounds: NULL bounds");
    VG_HIP(hipSetDevice(sq->ctx->device));
    hipStream_t st = sq->ctx->stream;
    vg::DevIn<float> a, b;
    VG_TRY(a.init(mins, static_cast<size_t>(sq->dim), st));
    VG_TRY(b.init(maxs, static_cast<size_t>(sq->dim), st));
    VG_LAUNCH(vg::sq8_finish_kernel, dim3((sq->dim + 255) / 256), dim3(256), 0, st, a.ptr, b.ptr, 1, sq->dim, false,
              sq->d_mins, sq->d_maxs, sq->d_scales, sq->d_inv);
    VG_HIP(hipStreamSynchronize(st));
    sq->trained = true;
    return VG_OK;
}

VG_API int32_t vg_sq8_get_params(vg_sq8 *sq, float *mins, float *maxs, float *scales, float *inv_scales)
{
    VG_CHECK(sq, VG_ERR_INVALID_ARG, "vg_sq8_get_params: NULL quantizer");
    VG_CHECK(sq->trained, VG_ERR_NOT_TRAINED, "ScalarQuantizer not trained");
    VG_HIP(hipSetDevice(sq->ctx->device));
    hipStream_t st = sq->ctx->stream;
    const size_t b = sizeof(float) * static_cast<size_t>(sq->dim);
    if (mins) VG_HIP(hipMemcpyAsync(mins, sq->d_mins, b, hipMemcpyDefault, st));
    if (maxs) VG_HIP(hipMemcpyAsync(maxs, sq->d_maxs, b, hipMemcpyDefault, st));
    if (scales) VG_HIP(hipMemcpyAsync(scales, sq->d_scales, b, hipMemcpyDefault, st));
    if (inv_scales) VG_HIP(hipMemcpyAsync(inv_scales, sq->d_inv, b, hipMemcpyDefault, st));
    VG_HIP(hipStreamSynchronize(st));
    return VG_OK;
}

VG_API int32_t vg_sq8_encode(vg_sq8 *sq, const float *vectors, int64_t n, uint8_t *codes, void *stream)
{
    VG_CHECK(sq, VG_ERR_INVALID_ARG, "vg_sq8_encode: NULL quantizer");
    VG_CHECK(sq->trained, VG_ERR_NOT_TRAINED, "ScalarQuantizer not trained");  // quantizer.go:184-186
    VG_CHECK(n >= 0, VG_ERR_INVALID_ARG, "vg_sq8_encode: n < 0");
    if (n == 0) return VG_OK;
    VG_CHECK(vectors && codes, VG_ERR_INVALID_ARG, "vg_sq8_encode: NULL buffer");
    VG_HIP(hipSetDevice(sq->ctx->device));
    hipStream_t st = vg::pick_stream(sq->ctx, stream);
    const int64_t total = n * sq->dim;
    vg::DevIn<float> v;
    vg::DevOut<uint8_t> c;
    VG_TRY(v.init(vectors, static_cast<size_t>(total), st));
    VG_TRY(c.init(codes, static_cast<size_t>(total), st));
    if (sq->dim % 4 == 0 && ((reinterpret_cast<uintptr_t>(v.ptr) | reinterpret_cast<uintptr_t>(c.ptr)) & 15) == 0)
        VG_LAUNCH(vg::sq8_encode4_kernel, dim3((sq->dim / 4 + 255) / 256, static_cast<unsigned>((n + vg::rows_per_thread(n) - 1) / vg::rows_per_thread(n))),
                  dim3(256), 0, st, v.ptr, n, sq->dim, sq->d_mins, sq->d_maxs, sq->d_scales, c.ptr, vg::rows_per_thread(n));
    else
        VG_LAUNCH(vg::sq8_encode_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, st, v.ptr, total,
                  sq->dim, sq->d_mins, sq->d_maxs, sq->d_scales, c.ptr);
    VG_TRY(c.finish());
    return VG_OK;
}

VG_API int32_t vg_sq8_decode(vg_sq8 *sq, const uint8_t *codes, int64_t n, float *out, void *stream)
{
    VG_CHECK(sq, VG_ERR_INVALID_ARG, "vg_sq8_decode: NULL quantizer");
    VG_CHECK(sq->trained, VG_ERR_NOT_TRAINED, "ScalarQuantizer not trained");
    VG_CHECK(n >= 0, VG_ERR_INVALID_ARG, "vg_sq8_decode: n < 0");
    if (n == 0) return VG_OK;
    VG_CHECK(codes && out, VG_ERR_INVALID_ARG, "vg_sq8_decode: NULL buffer");
    VG_HIP(hipSetDevice(sq->ctx->device));
    hipStream_t st = vg::pick_stream(sq->ctx, stream);
    const int64_t total = n * sq->dim;
    vg::DevIn<uint8_t> c;
    vg::DevOut<float> o;
    VG_TRY(c.init(codes, static_cast<size_t>(total), st));
    VG_TRY(o.init(out, static_cast<size_t>(total), st));
    if (sq->dim % 4 == 0 && ((reinterpret_cast<uintptr_t>(c.ptr) | reinterpret_cast<uintptr_t>(o.ptr)) & 15) == 0)
        VG_LAUNCH(vg::sq8_decode4_kernel, dim3((sq->dim / 4 + 255) / 256, static_cast<unsigned>((n + vg::rows_per_thread(n) - 1) / vg::rows_per_thread(n))),
                  dim3(256), 0, st, c.ptr, n, sq->dim, sq->d_mins, sq->d_inv, o.ptr, vg::rows_per_thread(n));
    else
        VG_LAUNCH(vg::sq8_decode_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, st, c.ptr, total,
                  sq->dim, sq->d_mins, sq->d_inv, o.ptr);
    VG_TRY(o.finish());
    return VG_OK;
}

VG_API int32_t vg_sq8_l2_distance_batch(vg_sq8 *sq, const float *query, const uint8_t *codes, int64_t n,
                                        float *out, void *stream)
{
    VG_CHECK(sq, VG_ERR_INVALID_ARG, "vg_sq8_l2_distance_batch: NULL quantizer");
    VG_CHECK(sq->trained, VG_ERR_NOT_TRAINED, "ScalarQuantizer not trained");
    VG_CHECK(n >= 0, VG_ERR_INVALID_ARG, "vg_sq8_l2_distance_batch: n < 0");
    if (n == 0) return VG_OK;
    VG_CHECK(query && codes && out, VG_ERR_INVALID_ARG, "vg_sq8_l2_distance_batch: NULL buffer");
    VG_HIP(hipSetDevice(sq->ctx->device));
    hipStream_t st = vg::pick_stream(sq->ctx, stream);
    vg::DevIn<float> q;
    vg::DevIn<uint8_t> c;
    vg::DevOut<float> o;
    VG_TRY(q.init(query, static_cast<size_t>(sq->dim), st));
    VG_TRY(c.init(codes, static_cast<size_t>(n) * sq->dim, st));
    VG_TRY(o.init(out, static_cast<size_t>(n), st));
    if (sq->dim % 128 == 0 && (reinterpret_cast<uintptr_t>(c.ptr) & 15) == 0)
        VG_LAUNCH(vg::sq8_l2_batch_turn_kernel,
                  dim3(static_cast<unsigned>(((n + 63) / 64 + vg::kSqTurnWaves - 1) / vg::kSqTurnWaves)),
                  dim3(vg::kSqTurnWaves * 64), 0, st, q.ptr, c.ptr, n, sq->dim, sq->d_mins, sq->d_inv, o.ptr);
    else
        VG_LAUNCH(vg::sq8_l2_batch_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, st, q.ptr, c.ptr, n,
                  sq->dim, sq->d_mins, sq->d_inv, o.ptr);
    VG_TRY(o.finish());
    return VG_OK;
}

// ---- batched L2 search through a bfloat16 nomination (vg_index_enable_sq8_nomination) -------------------------------------------
// The multi-query scan decodes every code once per 4 queries and is bound by the vector ALU (44 ms per 1024 queries x 1M x 768).
// With the opt-in image — the dequantised rows x^ = fma(code, invScale, min) rounded to bfloat16, 2 bytes per code — the fused flat
// search's nomination runs on them (k_flat.hip flat_nominate_bf16: threshold from a row sample, bf16 MFMA GEMM, the 64 best per
// query), and this file re-scores those 64 with the reference's own arithmetic on the CODES (sq8_row_score) and proves that no row
// outside them can enter the k best: outside rows have GEMM score >= tau, and |GEMM score + |q|^2 - L2Distance| <= eps (bfloat16
// rounding of both operands, fp32 accumulation, the reference's own rounding).  A query whose proof fails is scanned as before.
namespace vg {
size_t flat_nominate_bf16_scratch(int64_t cnt, int64_t n, int dim, int sel_k);
int32_t flat_nominate_bf16(vg_ctx *ctx, const uint16_t *rows_bf16, const float *norms, int64_t n, int dim, int dim_pad, const float *queries,
                           int64_t cnt, char *scratch, float *thr, int *counts, uint32_t *cand_id, float *cand_sc, hipStream_t st,
                           bool dot, const uint8_t *mask, int64_t mask_stride, int sel_k, bool pick, const uint64_t **cand_keys, int *cap,
                           const float *norm_max);
// k <= 48: the 64 best nominees are re-scored (sq8_verify_kernel); up to 256: everything below the threshold
// (sq8_verify_sort_kernel) — the 8th best of the 1/64 row sample passes ~512 rows, the 16th ~1024, the 32nd ~2048.  (The proof
// wants the threshold 2^-7 (|q|^2 + |x^|^2) above the k-th score: ~512 rows for k = 100 left enough queries to the scan — random-
// normal rows, 1M x 768 — that the batch took 8.3 ms, ~1024 rows 3 ms.)
constexpr int kSq8PickMaxK = 48, kSq8NomMaxK = 256;
#ifndef VG_SQ8_NOM_MIN_Q
#define VG_SQ8_NOM_MIN_Q 5  // smallest batch the nomination takes: 1M x 768, scan / nominated ms: 4 queries 0.34 / 0.37, 6: 0.51 / 0.38, 16: 0.94 / 0.37
#endif
static int sq8_nominate_sel_k(int k) { return k <= kSq8PickMaxK ? 8 : k <= 128 ? 16 : 32; }

__device__ __forceinline__ uint16_t sq8_bf16_rne(float x)
{
    const uint32_t b = __float_as_uint(x);
    return static_cast<uint16_t>((b + 0x7FFFu + ((b >> 16) & 1u)) >> 16);
}

// one lane per row of a 64-row tile: dequantise, round, write the row's bf16 image and its norm
// (rows of dim_pad elements: the dimensions from dim on are zeros, which add nothing to a dot product)
__global__ __launch_bounds__(64) void sq8_dequant_bf16_kernel(const uint4 *__restrict__ tiles, int64_t n, int dim, int groups,
                                                              const float *__restrict__ mins, const float *__restrict__ inv,
                                                              uint16_t *__restrict__ out, int dim_pad, float *__restrict__ norms,
                                                              int *__restrict__ norm_max_bits)
{
    const int64_t tile = blockIdx.x;
    const int lane = threadIdx.x;
    const int64_t row = tile * 64 + lane;
    float nrm = 0.0f;
    if (row < n) {
        for (int g = 0; g < dim_pad / 16; g++) {
            const uint4 c = g < groups ? tiles[(tile * groups + g) * 64 + lane] : make_uint4(0, 0, 0, 0);
            const uint32_t w[4] = {c.x, c.y, c.z, c.w};
            uint32_t packed[8];
#pragma unroll
            for (int t = 0; t < 16; t += 2) {
                uint32_t pair = 0;
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int j = g * 16 + t + h;
                    float x = 0.0f;
                    if (j < dim) {
                        const float code = static_cast<float>((w[(t + h) >> 2] >> (8 * ((t + h) & 3))) & 0xFFu);
                        x = __builtin_fmaf(code, inv[j], mins[j]);
                    }
                    nrm = __builtin_fmaf(x, x, nrm);
                    pair |= static_cast<uint32_t>(sq8_bf16_rne(x)) << (16 * h);
                }
                packed[t >> 1] = pair;
            }
            uint4 *dst = reinterpret_cast<uint4 *>(out + row * dim_pad + g * 16);
            dst[0] = make_uint4(packed[0], packed[1], packed[2], packed[3]);
            dst[1] = make_uint4(packed[4], packed[5], packed[6], packed[7]);
        }
        norms[row] = nrm;
    }
    // a NaN norm (a NaN in mins / inv) must reach norm_max — fmaxf would drop it, and a finite bound over a row whose GEMM score is
    // NaN would let the proof pass: NaN -> +Inf (the largest bit pattern below), the proof's comparisons then fail and the scan answers
    float mx = row < n ? (nrm == nrm ? nrm : INFINITY) : 0.0f;
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if (lane == 0) atomicMax(norm_max_bits, __float_as_int(mx));  // non-negative floats order like their bits
}

// per query: exact L2Distance / DotProduct of its 64 nominated rows from the codes, the k best by (score, row id), and the proof
template <bool DOT>
__global__ __launch_bounds__(64) void sq8_verify_kernel(const uint4 *__restrict__ tiles, int groups, int dim, const float *__restrict__ mins,
                                                        const float *__restrict__ inv, const float *__restrict__ queries,
                                                        const float *__restrict__ norm_max, const uint32_t *__restrict__ cand_ids,
                                                        const float *__restrict__ cand_scores, int k, uint32_t *__restrict__ ids,
                                                        float *__restrict__ scores, int *__restrict__ fail, const float *__restrict__ thr,
                                                        const int *__restrict__ counts, int cap, int thr_stride)
{
    constexpr int kc = 64;
    const int64_t q = blockIdx.x;
    const int lane = threadIdx.x;
    const float *qv = queries + q * dim;
    const uint32_t id = cand_ids[q * kc + lane];
    uint64_t key = kKeyMax;
    if (id != VG_INVALID_ID) {
        const float d = sq8_row_score<DOT>(tiles + (static_cast<int64_t>(id >> 6) * groups) * 64 + (id & 63), groups, dim >> 4, dim & 15,
                                           qv, mins, inv);
        key = make_key(d, id, DOT);
    }
    WaveTopK tk;
    tk.init(k);
    tk.offer(key, lane);
    float qn = 0.0f;
    for (int j = lane; j < dim; j += 64) qn = __builtin_fmaf(qv[j], qv[j], qn);
    for (int off = 32; off > 0; off >>= 1) qn += __shfl_xor(qn, off);
    const uint64_t kth = readlane_u64(tk.list, k - 1);
    const float tq = thr[q * thr_stride + (thr_stride - 1)];
    const int cnt = counts[q];
    bool ok = cnt <= cap;  // overflow: rows below the threshold were dropped
    const float tau = cnt > kc ? fminf(tq, cand_scores[q * kc + (kc - 1)]) : tq;
    const bool have_all = tq == INFINITY && cnt <= kc;
    if (ok && !have_all && tau != INFINITY) {
        // |s~ + |q|^2 - L2Distance|: bfloat16 rounding of q and x^ ((2^-7 + 2^-16)(|q|^2 + |x^|^2), as for the fp32 rows' bf16
        // filter), the GEMM's fp32 accumulation and the reference's own 16-lane sums ((2 dim + dim/8 + 32) u of the same)
        // (Dot: the score is -q.x^, half the L2 form's cross term: 2^-8 in place of 2^-7)
        const float eps = (4.0f * (static_cast<float>(dim) * 5.9604645e-8f) + (DOT ? 0.00390625f : 0.0078125f) * 1.02f) * (qn + norm_max[0]) + 1e-30f;
        if (kth == kKeyMax)
            ok = false;
        else if (DOT)
            ok = key_score(kth, true) > (-tau) + eps;  // outside rows: -q.x >= tau
        else
            ok = key_score(kth, false) < (tau + qn) - eps;
    }
    if (lane < k) {
        const uint64_t e = tk.list;
        ids[q * k + lane] = e == kKeyMax ? VG_INVALID_ID : key_row(e);
        scores[q * k + lane] = e == kKeyMax ? (DOT ? -INFINITY : INFINITY) : key_score(e, DOT);
    }
    if (lane == 0) fail[q] = ok ? 0 : 1;
}

// The same for k beyond the 64-candidate budget (flat_verify_sort_kernel's counterpart): EVERY appended row is re-scored from
// the codes — one lane per row — and sorted; what is left to argue about is what the threshold excluded, so the proof compares
// the k-th exact score with the threshold itself.  Dynamic LDS: cap keys.
template <bool DOT>
__global__ __launch_bounds__(256) void sq8_verify_sort_kernel(const uint4 *__restrict__ tiles, int groups, int dim, const float *__restrict__ mins,
                                                              const float *__restrict__ inv, const float *__restrict__ queries,
                                                              const float *__restrict__ norm_max, const uint64_t *__restrict__ cand,
                                                              const int *__restrict__ counts, int cap, int k, uint32_t *__restrict__ ids,
                                                              float *__restrict__ scores, int *__restrict__ fail, const float *__restrict__ thr,
                                                              int thr_stride)
{
    extern __shared__ uint64_t sortbuf[];
    const int64_t q = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const float *qv = queries + q * dim;
    const int total = counts[q];
    const int cnt = total < cap ? total : cap;
    int n2 = 64;
    while (n2 < cnt) n2 <<= 1;
    for (int c = tid; c < cnt; c += 256) {
        const uint32_t id = key_row(cand[q * cap + c]);
        const float d = sq8_row_score<DOT>(tiles + (static_cast<int64_t>(id >> 6) * groups) * 64 + (id & 63), groups, dim >> 4, dim & 15,
                                           qv, mins, inv);
        sortbuf[c] = make_key(d, id, DOT);
    }
    for (int i = cnt + tid; i < n2; i += 256) sortbuf[i] = kKeyMax;
    __syncthreads();
    bitonic_sort_lds(sortbuf, n2, tid, 256);
    for (int i = tid; i < k; i += 256) {
        const uint64_t e = i < n2 ? sortbuf[i] : kKeyMax;
        ids[q * k + i] = e == kKeyMax ? VG_INVALID_ID : key_row(e);
        scores[q * k + i] = e == kKeyMax ? (DOT ? -INFINITY : INFINITY) : key_score(e, DOT);
    }
    if (tid >= 64) return;
    float qn = 0.0f;
    for (int j = lane; j < dim; j += 64) qn = __builtin_fmaf(qv[j], qv[j], qn);
    for (int off = 32; off > 0; off >>= 1) qn += __shfl_xor(qn, off);
    const uint64_t kth = k - 1 < n2 ? sortbuf[k - 1] : kKeyMax;
    const float tau = thr[q * thr_stride + (thr_stride - 1)];
    bool ok = total <= cap;
    if (ok && tau != INFINITY) {  // (tau == +Inf: no threshold was set, every accepted row was appended)
        // (the margin: sq8_verify_kernel's)
        const float eps = (4.0f * (static_cast<float>(dim) * 5.9604645e-8f) + (DOT ? 0.00390625f : 0.0078125f) * 1.02f) * (qn + norm_max[0]) + 1e-30f;
        if (kth == kKeyMax)
            ok = false;
        else if (DOT)
            ok = key_score(kth, true) > (-tau) + eps;
        else
            ok = key_score(kth, false) < (tau + qn) - eps;
    }
    if (lane == 0) fail[q] = ok ? 0 : 1;
}
}  // namespace vg

VG_API int32_t vg_index_enable_sq8_nomination(vg_index *idx, int32_t on, void *stream)
{
    VG_CHECK(idx, VG_ERR_INVALID_ARG, "vg_index_enable_sq8_nomination: NULL index");
    VG_HIP(hipSetDevice(idx->ctx->device));
    hipStream_t st = vg::pick_stream(idx->ctx, stream);
    if (idx->d_sq_bf16) {
        VG_HIP(hipStreamSynchronize(st));
        VG_HIP(hipFree(idx->d_sq_bf16));
        VG_HIP(hipFree(idx->d_sq_norms));
        VG_HIP(hipFree(idx->d_sq_norm_max));
        idx->d_sq_bf16 = nullptr;
        idx->d_sq_norms = idx->d_sq_norm_max = nullptr;
    }
    if (!on) return VG_OK;
    VG_CHECK(idx->sq && idx->d_sq_tiles, VG_ERR_NOT_READY, "vg_index_enable_sq8_nomination: index has no SQ8 codes");
    const int bdim = (idx->dim + 63) & ~63;  // whole K steps of the bf16 GEMM; the padding is zeros
    // the three arrays are published together, after the image is built: a failure half way leaves the index as it was (the
    // batch search tests d_sq_bf16 alone — ADVICE r05: a failed second allocation left it set next to null norms)
    uint16_t *img = nullptr;
    float *norms = nullptr, *norm_max = nullptr;
    auto give_up = [&](hipError_t e) {
        (void)hipFree(img);
        (void)hipFree(norms);
        (void)hipFree(norm_max);
        return e;
    };
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&img), static_cast<size_t>(idx->n) * bdim * sizeof(uint16_t));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&norms), static_cast<size_t>(idx->n) * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&norm_max), sizeof(float));
    if (e == hipSuccess) e = hipMemsetAsync(norm_max, 0, sizeof(float), st);
    if (e != hipSuccess) VG_HIP(give_up(e));
    hipLaunchKernelGGL(vg::sq8_dequant_bf16_kernel, dim3(static_cast<unsigned>(idx->n_tiles)), dim3(64), 0, st,
                       reinterpret_cast<const uint4 *>(idx->d_sq_tiles), idx->n, idx->dim, idx->sq_groups, idx->sq->d_mins, idx->sq->d_inv,
                       img, bdim, norms, reinterpret_cast<int *>(norm_max));
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) VG_HIP(give_up(e));
    idx->sq_bf16_dim = bdim;
    idx->d_sq_bf16 = img;
    idx->d_sq_norms = norms;
    idx->d_sq_norm_max = norm_max;
    return VG_OK;
}

VG_API int32_t vg_index_set_sq8_codes(vg_index *idx, vg_sq8 *sq, const uint8_t *codes, void *stream)
{
    VG_CHECK(idx && sq, VG_ERR_INVALID_ARG, "vg_index_set_sq8_codes: NULL index or quantizer");
    VG_CHECK(sq->trained, VG_ERR_NOT_TRAINED, "ScalarQuantizer not trained");
    VG_CHECK(sq->dim == idx->dim, VG_ERR_DIM_MISMATCH, "vector dimension mismatch");
    VG_CHECK(idx->n == 0 || codes, VG_ERR_INVALID_ARG, "vg_index_set_sq8_codes: codes is NULL");
    VG_HIP(hipSetDevice(idx->ctx->device));
    hipStream_t st = vg::pick_stream(idx->ctx, stream);
    if (idx->d_sq_tiles) {
        VG_HIP(hipStreamSynchronize(st));
        VG_HIP(hipFree(idx->d_sq_tiles));
        idx->d_sq_tiles = nullptr;
    }
    if (idx->d_sq_bf16) {  // the old codes' nomination image (vg_index_enable_sq8_nomination again after new codes)
        VG_HIP(hipFree(idx->d_sq_bf16));
        VG_HIP(hipFree(idx->d_sq_norms));
        VG_HIP(hipFree(idx->d_sq_norm_max));
        idx->d_sq_bf16 = nullptr;
        idx->d_sq_norms = idx->d_sq_norm_max = nullptr;
    }
    idx->sq = sq;
    idx->sq_groups = (idx->dim + 15) / 16;
    idx->n_tiles = (idx->n + 63) / 64;
    if (idx->n == 0) return VG_OK;
    const int64_t total = idx->n_tiles * idx->sq_groups * 64;
    VG_HIP(hipMalloc(reinterpret_cast<void **>(&idx->d_sq_tiles), static_cast<size_t>(total) * 16));
    vg::DevIn<uint8_t> in;
    VG_TRY(in.init(codes, static_cast<size_t>(idx->n) * idx->dim, st));
    VG_LAUNCH(vg::sq8_retile_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, st, in.ptr, idx->n,
              idx->dim, idx->sq_groups, idx->n_tiles, reinterpret_cast<uint4 *>(idx->d_sq_tiles));
    VG_HIP(hipStreamSynchronize(st));
    return VG_OK;
}

namespace vg {
// sq8_verify_kernel over nq query rows whose nomination (8 thresholds, count, 64 candidates each) another file produced: the
// partition-probed scan's (query, probe) pairs (k_probe.hip)
int32_t launch_sq8_verify(vg_index *idx, const float *queries, int64_t nq, const ProbeNominated &nom, int k, uint32_t *ids, float *scores,
                          int *fail, hipStream_t st)
{
    const bool dot = idx->metric != VG_METRIC_L2;
    if (k > kSq8PickMaxK) {
        auto kern = dot ? sq8_verify_sort_kernel<true> : sq8_verify_sort_kernel<false>;
        VG_LAUNCH(kern, dim3(static_cast<unsigned>(nq)), dim3(256), sizeof(uint64_t) * static_cast<size_t>(nom.cap), st,
                  reinterpret_cast<const uint4 *>(idx->d_sq_tiles), idx->sq_groups, idx->dim, idx->sq->d_mins, idx->sq->d_inv, queries,
                  idx->d_sq_norm_max, nom.cand, nom.counts, nom.cap, k, ids, scores, fail, nom.thr, nom.sel_k);
        return VG_OK;
    }
    auto kern = dot ? sq8_verify_kernel<true> : sq8_verify_kernel<false>;
    VG_LAUNCH(kern, dim3(static_cast<unsigned>(nq)), dim3(64), 0, st, reinterpret_cast<const uint4 *>(idx->d_sq_tiles), idx->sq_groups, idx->dim,
              idx->sq->d_mins, idx->sq->d_inv, queries, idx->d_sq_norm_max, nom.cand_id, nom.cand_sc, k, ids, scores, fail, nom.thr, nom.counts,
              nom.cap, nom.sel_k);
    return VG_OK;
}
// whether a batch takes the nomination (vg_index_enable_sq8_nomination; device queries)
bool sq8_nomination_applies(const vg_index *idx, const float *d_queries, int64_t nq, int k)
{
    return idx->d_sq_bf16 && nq >= VG_SQ8_NOM_MIN_Q && k <= kSq8NomMaxK && idx->n > k && (reinterpret_cast<uintptr_t>(d_queries) & 15) == 0;
}
// The nomination + exact re-score + proof for a batch (device buffers; mask: a device row filter per query / for the batch, or
// null): writes every query's k results and lists the queries whose proof failed — the caller scans those.  4096 queries a pass.
int32_t sq8_nominated_pass(vg_index *idx, const float *q, int64_t nq, int k, const uint8_t *mask, int64_t mask_stride, uint32_t *oid,
                           float *osc, hipStream_t st, std::vector<int> &failed)
{
    const bool dot = idx->metric != VG_METRIC_L2;
    for (int64_t q0 = 0; q0 < nq; q0 += 4096) {
        const int64_t cnt = std::min<int64_t>(4096, nq - q0);
        std::vector<int> h(static_cast<size_t>(cnt));
        {
            ArenaCall ar(idx->ctx, st);
            const int sel_k = sq8_nominate_sel_k(k);
            const int i_scr = ar.add(flat_nominate_bf16_scratch(cnt, idx->n, idx->sq_bf16_dim, sel_k));
            const int i_thr = ar.add(sizeof(float) * static_cast<size_t>(cnt) * sel_k);
            const int i_cnt = ar.add(sizeof(int) * static_cast<size_t>(cnt));
            const int i_cid = ar.add(sizeof(uint32_t) * static_cast<size_t>(cnt) * 64);
            const int i_csc = ar.add(sizeof(float) * static_cast<size_t>(cnt) * 64);
            const int i_fail = ar.add(sizeof(int) * static_cast<size_t>(cnt));
            VG_TRY(ar.commit());
            float *thr = ar.get<float>(i_thr), *csc = ar.get<float>(i_csc);
            int *counts = ar.get<int>(i_cnt), *fail = ar.get<int>(i_fail);
            uint32_t *cid = ar.get<uint32_t>(i_cid);
            ProbeNominated nom{thr, counts, cid, csc, 0, sel_k, nullptr};
            VG_TRY(flat_nominate_bf16(idx->ctx, idx->d_sq_bf16, idx->d_sq_norms, idx->n, idx->dim, idx->sq_bf16_dim, q + q0 * idx->dim, cnt, ar.get<char>(i_scr),
                                      thr, counts, cid, csc, st, dot, mask ? mask + q0 * mask_stride : nullptr, mask_stride, sel_k,
                                      k <= kSq8PickMaxK, &nom.cand, &nom.cap, idx->d_sq_norm_max));
            VG_TRY(launch_sq8_verify(idx, q + q0 * idx->dim, cnt, nom, k, oid + q0 * k, osc + q0 * k, fail, st));
            VG_HIP(hipMemcpyAsync(h.data(), fail, sizeof(int) * static_cast<size_t>(cnt), hipMemcpyDeviceToHost, st));
            VG_HIP(hipStreamSynchronize(st));
        }
        for (int64_t i = 0; i < cnt; i++)
            if (h[static_cast<size_t>(i)]) failed.push_back(static_cast<int>(q0 + i));
    }
    return VG_OK;
}
}  // namespace vg

namespace vg {
// vg_cand_replay.hpp's scorer for the SQ8 scan: sq.L2Distance / sq.DotProduct of a row's code (flat/segment.go:517-604, :659-667),
// one lane per row of the re-tiled codes.  At risk: a non-finite query value, minimum or inverse scale; magnitudes whose partial
// sums could overflow (|x^_j| <= 255 |inv_j| + |min_j|).
template <bool DOT>
struct Sq8Scorer {
    const uint4 *tiles;
    const float *mins, *inv;
    int groups, dim;
    __device__ bool risk(int64_t, const float *q, int tid) const
    {
        __shared__ int flag;
        __shared__ float bmax;
        if (tid == 0) bmax = 0.0f;
        __syncthreads();
        bool bad = false;
        float b = 0.0f;
        for (int j = tid; j < dim; j += kReplayThreads) {
            const float mn = mins[j], iv = inv[j];
            bad = bad || !is_finite_f32(q[j]) || !is_finite_f32(mn) || !is_finite_f32(iv);
            b = fmaxf(b, 255.0f * fabsf(iv) + fabsf(mn));
        }
        for (int off = 32; off > 0; off >>= 1) b = fmaxf(b, __shfl_xor(b, off));
        if ((tid & 63) == 0) atomicMax(reinterpret_cast<int *>(&bmax), __float_as_int(b));  // non-negative floats order like their bits
        __syncthreads();
        const float bm = bmax;
        for (int j = tid; j < dim; j += kReplayThreads) bad = bad || !(score_bound(fabsf(q[j]), bm, DOT) * static_cast<float>(dim) < 1e38f);
        return block_any(bad, &flag, tid);
    }
    __device__ void prepare(int64_t, const float *, int) const {}
    __device__ void score_chunk(int64_t, const float *q, int64_t row0, int64_t n, int tid, float *out) const
    {
        const int64_t row = row0 + tid;  // (row0 is a multiple of 256: whole tiles of 64)
        if (row >= n) return;
        out[tid] = sq8_row_score<DOT>(tiles + ((row >> 6) * groups) * 64 + (row & 63), groups, dim >> 4, dim & 15, q, mins, inv);
    }
};
}  // namespace vg

namespace vg {
// the replay for device buffers: the whole segment, or (probes: nq * np partition ids, part_off) the probed partitions; mask: a
// device row filter per query / for the batch, or null (k_probe.hip calls it for the filtered and the partition-probed scans)
int32_t sq8_nan_replay(vg_index *idx, const float *d_queries, int64_t nq, int k, const uint8_t *d_mask, int64_t mask_stride,
                       const uint32_t *d_probes, int np, const uint32_t *d_part_off, uint32_t *d_ids, float *d_scores, hipStream_t st)
{
    if (idx->n == 0) return VG_OK;
    const uint4 *tiles = reinterpret_cast<const uint4 *>(idx->d_sq_tiles);
    if (idx->metric != VG_METRIC_L2)
        return launch_cand_replay(Sq8Scorer<true>{tiles, idx->sq->d_mins, idx->sq->d_inv, idx->sq_groups, idx->dim}, d_queries, idx->dim, idx->n, nq, k,
                                  true, d_mask, mask_stride, d_ids, d_scores, st, nullptr, d_probes, np, d_part_off);
    return launch_cand_replay(Sq8Scorer<false>{tiles, idx->sq->d_mins, idx->sq->d_inv, idx->sq_groups, idx->dim}, d_queries, idx->dim, idx->n, nq, k,
                              false, d_mask, mask_stride, d_ids, d_scores, st, nullptr, d_probes, np, d_part_off);
}
}  // namespace vg

static int32_t sq8_search_impl(vg_index *idx, const float *queries, int64_t nq, int32_t k, uint32_t *ids, float *scores, void *stream,
                               bool allow_nomination);

VG_API int32_t vg_search_sq8(vg_index *idx, const float *queries, int64_t nq, int32_t k, uint32_t *ids,
                             float *scores, void *stream)
{
    return sq8_search_impl(idx, queries, nq, k, ids, scores, stream, true);
}

static int32_t sq8_search_impl(vg_index *idx, const float *queries, int64_t nq, int32_t k, uint32_t *ids, float *scores, void *stream,
                               bool allow_nomination)
{
    VG_CHECK(idx, VG_ERR_INVALID_ARG, "vg_search_sq8: NULL index");
    VG_CHECK(nq >= 0 && k >= 0, VG_ERR_INVALID_ARG, "vg_search_sq8: negative nq or k");
    if (nq == 0 || k == 0) return VG_OK;
    VG_CHECK(idx->metric != VG_METRIC_HAMMING, VG_ERR_UNSUPPORTED, "unsupported metric for float32: Hamming");
    const bool dot = idx->metric != VG_METRIC_L2;  // segment.go:659-667: sq.L2Distance or sq.DotProduct
    VG_CHECK(idx->n == 0 || idx->d_sq_tiles, VG_ERR_NOT_READY, "vg_search_sq8: index has no SQ8 codes");
    VG_CHECK(queries && ids && scores, VG_ERR_INVALID_ARG, "vg_search_sq8: NULL buffer");
    VG_CHECK(k <= 512, VG_ERR_UNSUPPORTED, "vg_search_sq8: k=%d exceeds 512", k);
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
    } else if (allow_nomination && vg::sq8_nomination_applies(idx, q.ptr, nq, k)) {
        std::vector<int> failed;
        VG_TRY(vg::sq8_nominated_pass(idx, q.ptr, nq, k, nullptr, 0, oid.ptr, osc.ptr, st, failed));
        if (!failed.empty()) {  // the scan kernels for the queries whose proof failed (ties at the k-th score, thresholds too tight)
            const int64_t nf = static_cast<int64_t>(failed.size());
            vg::DevTmp<float> fq;
            vg::DevTmp<uint32_t> fid;
            vg::DevTmp<float> fsc;
            VG_TRY(fq.init(static_cast<size_t>(nf) * idx->dim, st));
            VG_TRY(fid.init(static_cast<size_t>(nf) * k, st));
            VG_TRY(fsc.init(static_cast<size_t>(nf) * k, st));
            for (int64_t i = 0; i < nf; i++)
                VG_HIP(hipMemcpyAsync(fq.ptr + i * idx->dim, q.ptr + static_cast<int64_t>(failed[static_cast<size_t>(i)]) * idx->dim,
                                      sizeof(float) * idx->dim, hipMemcpyDeviceToDevice, st));
            VG_TRY(sq8_search_impl(idx, fq.ptr, nf, k, fid.ptr, fsc.ptr, st, false));
            for (int64_t i = 0; i < nf; i++) {
                const int64_t at = static_cast<int64_t>(failed[static_cast<size_t>(i)]) * k;
                VG_HIP(hipMemcpyAsync(oid.ptr + at, fid.ptr + i * k, sizeof(uint32_t) * k, hipMemcpyDeviceToDevice, st));
                VG_HIP(hipMemcpyAsync(osc.ptr + at, fsc.ptr + i * k, sizeof(float) * k, hipMemcpyDeviceToDevice, st));
            }
        }
    } else {
        // two or more queries: groups of kSqProbeQ share every decode (sq8_scan_mq_kernel)
        const size_t mq_lds = sizeof(float) * vg::kSqProbeQ * static_cast<size_t>(idx->sq_groups) * 16 +
                              vg::kSqWaves * 64 * sizeof(uint64_t) + 64;
        const bool mq = nq >= 2 && mq_lds <= 128 * 1024;
        const int64_t units = mq ? (nq + vg::kSqProbeQ - 1) / vg::kSqProbeQ : nq;  // workgroups per slice
        const int slices = vg::sq_slices(units, idx->n_tiles, idx->ctx->compute_units);
        // a wave keeps 64 keys: k > 64 comes in pages of 64, every page a scan for the keys after the previous
        // page's last one (ceil(k / 64) scans)
        const bool paged = k > 64;
        const int pk = paged ? 64 : k;
        vg::ArenaCall ar(idx->ctx, st);
        const int i_partial = ar.add(sizeof(uint64_t) * static_cast<size_t>(nq) * slices * pk);
        const int i_pid = ar.add(paged ? sizeof(uint32_t) * static_cast<size_t>(nq) * pk : 0);
        const int i_psc = ar.add(paged ? sizeof(float) * static_cast<size_t>(nq) * pk : 0);
        const int i_floor = ar.add(paged ? sizeof(uint64_t) * static_cast<size_t>(nq) : 0);
        const int i_one = ar.add(paged ? 256 : 0);
        VG_TRY(ar.commit());
        uint64_t *partial = ar.get<uint64_t>(i_partial), *floor_keys = ar.get<uint64_t>(i_floor);
        uint32_t *pid = ar.get<uint32_t>(i_pid);
        float *psc = ar.get<float>(i_psc);
        int *one = ar.get<int>(i_one);
        if (paged) VG_HIP(hipMemsetAsync(one, 1, sizeof(int), st));
        if (mq) {
            auto kern = dot ? vg::sq8_scan_mq_kernel<true> : vg::sq8_scan_mq_kernel<false>;
            VG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(mq_lds)));
        }
        for (int off = 0; off < k; off += 64) {
            const int kk = paged ? std::min(64, k - off) : k;
            const uint64_t *floor = off ? floor_keys : nullptr;
            if (mq) {
                auto kern = dot ? vg::sq8_scan_mq_kernel<true> : vg::sq8_scan_mq_kernel<false>;
                const int64_t max_q = ((1ll << 30) / slices) * vg::kSqProbeQ;  // whole groups per launch
                for (int64_t q0 = 0; q0 < nq; q0 += max_q) {
                    const int64_t cnt = nq - q0 < max_q ? nq - q0 : max_q;
                    const int64_t ng = (cnt + vg::kSqProbeQ - 1) / vg::kSqProbeQ;
                    vg::ProfScope prof(idx->ctx, "sq8_scan", st);
                    VG_LAUNCH(kern, dim3(static_cast<unsigned>(ng * slices)), dim3(vg::kSqThreads), mq_lds, st,
                              reinterpret_cast<const uint4 *>(idx->d_sq_tiles), idx->n, idx->n_tiles, idx->sq_groups, idx->dim,
                              q.ptr + q0 * idx->dim, idx->sq->d_mins, idx->sq->d_inv, slices, static_cast<int>(cnt), kk,
                              partial + q0 * slices * kk, floor ? floor + q0 : nullptr);
                }
            } else {
                const int64_t max_q = (1ll << 30) / slices;
                for (int64_t q0 = 0; q0 < nq; q0 += max_q) {
                    const int64_t cnt = nq - q0 < max_q ? nq - q0 : max_q;
                    vg::ProfScope prof(idx->ctx, "sq8_scan", st);
                    // one query: workgroups of 8 waves (32 waves per CU).  The row loop keeps fewer bytes in flight than
                    // its load ring suggests (the compiler drains it at every group), so the single pass wants the
                    // occupancy: 4M x 768 scan kernel 540 -> 518 us, call 590 -> 559 us (8 workgroups of 4 waves: 500 us,
                    // but the merge of twice the lists gives it back)
                    const bool wide = nq == 1 && idx->n_tiles >= static_cast<int64_t>(slices) * 8;
                    auto kern = wide ? (dot ? vg::sq8_scan_kernel<true, 8> : vg::sq8_scan_kernel<false, 8>)
                                     : (dot ? vg::sq8_scan_kernel<true, vg::kSqWaves> : vg::sq8_scan_kernel<false, vg::kSqWaves>);
                    VG_LAUNCH(kern, dim3(static_cast<unsigned>(cnt * slices)), dim3(wide ? 512 : vg::kSqThreads), 0, st,
                              reinterpret_cast<const uint4 *>(idx->d_sq_tiles), idx->n, idx->n_tiles, idx->sq_groups, idx->dim,
                              q.ptr + q0 * idx->dim, idx->sq->d_mins, idx->sq->d_inv, slices, static_cast<int>(cnt), kk,
                              partial + q0 * slices * kk, floor ? floor + q0 : nullptr);
                }
            }
            if (!paged) {
                VG_TRY(vg::launch_topk_merge(partial, nq, slices, k, dot, oid.ptr, osc.ptr, st));
            } else {
                VG_TRY(vg::launch_topk_merge(partial, nq, slices, kk, dot, pid, psc, st));
                VG_TRY(vg::launch_page_patch(nq, k, off, kk, dot, one, pid, psc, oid.ptr, osc.ptr, floor_keys, st));
            }
        }
    }
    // queries whose scores may hold a NaN: the reference's heap, operation by operation (vg_cand_replay.hpp; not for the
    // queries this function sends to itself after a failed proof: the caller's pass covers them)
    if (idx->n > 0 && allow_nomination) VG_TRY(vg::sq8_nan_replay(idx, q.ptr, nq, k, nullptr, 0, nullptr, 0, nullptr, oid.ptr, osc.ptr, st));
    VG_TRY(oid.finish());
    VG_TRY(osc.finish());
    return VG_OK;
}

// ---- INT4 C ABI ---------------------------------------------------------------------------------------
static int32_t int4_rebuild_table(vg_int4 *iq, hipStream_t st)
{
    VG_LAUNCH(vg::int4_table_kernel, dim3((iq->dim * 16 + 255) / 256), dim3(256), 0, st, iq->d_min, iq->d_diff,
              iq->dim, iq->d_table);
    VG_HIP(hipStreamSynchronize(st));
    iq->trained = true;
    return VG_OK;
}

VG_API int32_t vg_int4_create(vg_ctx *ctx, int32_t dim, vg_int4 **out)
{
    VG_CHECK(out, VG_ERR_INVALID_ARG, "vg_int4_create: out is NULL");
    *out = nullptr;
    VG_CHECK(ctx, VG_ERR_INVALID_ARG, "vg_int4_create: ctx is NULL");
    VG_CHECK(dim > 0, VG_ERR_INVALID_ARG, "vg_int4_create: dim must be positive");
    VG_HIP(hipSetDevice(ctx->device));
    vg_int4 *iq = new vg_int4;
    iq->ctx = ctx;
    iq->dim = dim;
    float *block = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&block), sizeof(float) * 18 * static_cast<size_t>(dim));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        delete iq;
        vg::set_error("vg_int4_create: hipMalloc failed: %s", hipGetErrorString(e));
        return VG_ERR_HIP;
    }
    iq->d_min = block;
    iq->d_diff = block + dim;
    iq->d_table = block + 2 * static_cast<size_t>(dim);
    *out = iq;
    return VG_OK;
}

VG_API int32_t vg_int4_destroy(vg_int4 *iq)
{
    if (!iq) return VG_OK;
    (void)hipSetDevice(iq->ctx->device);
    if (iq->d_min) (void)hipFree(iq->d_min);
    delete iq;
    return VG_OK;
}

VG_API int32_t vg_int4_is_trained(vg_int4 *iq) { return iq && iq->trained ? 1 : 0; }

VG_API int32_t vg_int4_train(vg_int4 *iq, const float *vectors, int64_t n, void *stream)
{
    VG_CHECK(iq, VG_ERR_INVALID_ARG, "vg_int4_train: NULL quantizer");
    VG_CHECK(n > 0 && vectors, VG_ERR_INVALID_ARG, "no vectors provided for training");  // int4.go:30-32
    VG_HIP(hipSetDevice(iq->ctx->device));
    hipStream_t st = vg::pick_stream(iq->ctx, stream);
    const int dim = iq->dim;
    vg::DevIn<float> v;
    VG_TRY(v.init(vectors, static_cast<size_t>(n) * dim, st));
    int chunks = static_cast<int>(std::min<int64_t>(n, 1024));
    vg::DevTmp<float> pmin, pmax;
    VG_TRY(pmin.init(static_cast<size_t>(chunks) * dim, st));
    VG_TRY(pmax.init(static_cast<size_t>(chunks) * dim, st));
    VG_LAUNCH(vg::sq8_minmax_kernel, dim3((dim + 255) / 256, chunks), dim3(256), 0, st, v.ptr, n, dim, chunks,
              pmin.ptr, pmax.ptr);
    VG_LAUNCH(vg::int4_finish_kernel, dim3((dim + 255) / 256), dim3(256), 0, st, pmin.ptr, pmax.ptr, chunks, dim, true,
              iq->d_min, iq->d_diff);
    return int4_rebuild_table(iq, st);
}

/* UnmarshalBinary (int4.go:190-219): min[dim], diff[dim] as stored, table rebuilt */
VG_API int32_t vg_int4_set_params(vg_int4 *iq, const float *min_val, const float *diff)
{
    VG_CHECK(iq, VG_ERR_INVALID_ARG, "vg_int4_set_params: NULL quantizer");
    VG_CHECK(min_val && diff, VG_ERR_INVALID_ARG, "vg_int4_set_params: NULL parameters");
    VG_HIP(hipSetDevice(iq->ctx->device));
    hipStream_t st = iq->ctx->stream;
    const size_t b = sizeof(float) * static_cast<size_t>(iq->dim);
    VG_HIP(hipMemcpyAsync(iq->d_min, min_val, b, hipMemcpyDefault, st));
    VG_HIP(hipMemcpyAsync(iq->d_diff, diff, b, hipMemcpyDefault, st));
    return int4_rebuild_table(iq, st);
}

VG_API int32_t vg_int4_get_params(vg_int4 *iq, float *min_val, float *diff, float *table)
{
    VG_CHECK(iq, VG_ERR_INVALID_ARG, "vg_int4_get_params: NULL quantizer");
    VG_CHECK(iq->trained, VG_ERR_NOT_TRAINED, "Int4Quantizer not trained");
    VG_HIP(hipSetDevice(iq->ctx->device));
    hipStream_t st = iq->ctx->stream;
    const size_t b = sizeof(float) * static_cast<size_t>(iq->dim);
    if (min_val) VG_HIP(hipMemcpyAsync(min_val, iq->d_min, b, hipMemcpyDefault, st));
    if (diff) VG_HIP(hipMemcpyAsync(diff, iq->d_diff, b, hipMemcpyDefault, st));
    if (table) VG_HIP(hipMemcpyAsync(table, iq->d_table, 16 * b, hipMemcpyDefault, st));
    VG_HIP(hipStreamSynchronize(st));
    return VG_OK;
}

VG_API int64_t vg_int4_code_bytes(int32_t dim) { return (static_cast<int64_t>(dim) + 1) / 2; }

VG_API int32_t vg_int4_encode(vg_int4 *iq, const float *vectors, int64_t n, uint8_t *codes, void *stream)
{
    VG_CHECK(iq, VG_ERR_INVALID_ARG, "vg_int4_encode: NULL quantizer");
    VG_CHECK(iq->trained, VG_ERR_NOT_TRAINED, "Int4Quantizer not trained");
    VG_CHECK(n >= 0, VG_ERR_INVALID_ARG, "vg_int4_encode: n < 0");
    if (n == 0) return VG_OK;
    VG_CHECK(vectors && codes, VG_ERR_INVALID_ARG, "vg_int4_encode: NULL buffer");
    VG_HIP(hipSetDevice(iq->ctx->device));
    hipStream_t st = vg::pick_stream(iq->ctx, stream);
    const int64_t cs = vg_int4_code_bytes(iq->dim);
    vg::DevIn<float> v;
    vg::DevOut<uint8_t> c;
    VG_TRY(v.init(vectors, static_cast<size_t>(n) * iq->dim, st));
    VG_TRY(c.init(codes, static_cast<size_t>(n * cs), st));
    if (iq->dim % 8 == 0 && ((reinterpret_cast<uintptr_t>(v.ptr) | reinterpret_cast<uintptr_t>(c.ptr)) & 15) == 0)
        VG_LAUNCH(vg::int4_encode8_kernel, dim3((iq->dim / 8 + 255) / 256, static_cast<unsigned>((n + vg::rows_per_thread(n) - 1) / vg::rows_per_thread(n))),
                  dim3(256), 0, st, v.ptr, n, iq->dim, iq->d_min, iq->d_diff, c.ptr, vg::rows_per_thread(n));
    else
        VG_LAUNCH(vg::int4_encode_kernel, dim3(static_cast<unsigned>((n * cs + 255) / 256)), dim3(256), 0, st, v.ptr, n,
                  iq->dim, iq->d_min, iq->d_diff, c.ptr);
    VG_TRY(c.finish());
    return VG_OK;
}

VG_API int32_t vg_int4_decode(vg_int4 *iq, const uint8_t *codes, int64_t n, float *out, void *stream)
{
    VG_CHECK(iq, VG_ERR_INVALID_ARG, "vg_int4_decode: NULL quantizer");
    VG_CHECK(iq->trained, VG_ERR_NOT_TRAINED, "Int4Quantizer not trained");
    VG_CHECK(n >= 0, VG_ERR_INVALID_ARG, "vg_int4_decode: n < 0");
    if (n == 0) return VG_OK;
    VG_CHECK(codes && out, VG_ERR_INVALID_ARG, "vg_int4_decode: NULL buffer");
    VG_HIP(hipSetDevice(iq->ctx->device));
    hipStream_t st = vg::pick_stream(iq->ctx, stream);
    const int64_t cs = vg_int4_code_bytes(iq->dim);
    vg::DevIn<uint8_t> c;
    vg::DevOut<float> o;
    VG_TRY(c.init(codes, static_cast<size_t>(n * cs), st));
    VG_TRY(o.init(out, static_cast<size_t>(n) * iq->dim, st));
    if (iq->dim % 8 == 0 && ((reinterpret_cast<uintptr_t>(c.ptr) | reinterpret_cast<uintptr_t>(o.ptr)) & 15) == 0)
        VG_LAUNCH(vg::int4_decode8_kernel, dim3((iq->dim / 8 + 255) / 256, static_cast<unsigned>((n + vg::rows_per_thread(n) - 1) / vg::rows_per_thread(n))),
                  dim3(256), 0, st, c.ptr, n, iq->dim, iq->d_min, iq->d_diff, o.ptr, vg::rows_per_thread(n));
    else
        VG_LAUNCH(vg::int4_decode_kernel, dim3(static_cast<unsigned>((n * iq->dim + 255) / 256)), dim3(256), 0, st, c.ptr, n,
                  iq->dim, iq->d_min, iq->d_diff, o.ptr);
    VG_TRY(o.finish());
    return VG_OK;
}

// precomputed = 0: L2DistanceBatch (int4.go:150-164, batch kernel order);
// precomputed = 1: L2Distance per code (int4.go:133-147, lookup-table kernel order)
VG_API int32_t vg_int4_l2_distance_batch(vg_int4 *iq, const float *query, const uint8_t *codes, int64_t n,
                                         int32_t precomputed, float *out, void *stream)
{
    VG_CHECK(iq, VG_ERR_INVALID_ARG, "vg_int4_l2_distance_batch: NULL quantizer");
    VG_CHECK(iq->trained, VG_ERR_NOT_TRAINED, "Int4Quantizer not trained");
    VG_CHECK(n >= 0, VG_ERR_INVALID_ARG, "vg_int4_l2_distance_batch: n < 0");
    if (n == 0) return VG_OK;
    VG_CHECK(query && codes && out, VG_ERR_INVALID_ARG, "vg_int4_l2_distance_batch: NULL buffer");
    VG_HIP(hipSetDevice(iq->ctx->device));
    hipStream_t st = vg::pick_stream(iq->ctx, stream);
    const int64_t cs = vg_int4_code_bytes(iq->dim);
    vg::DevIn<float> q;
    vg::DevIn<uint8_t> c;
    vg::DevOut<float> o;
    VG_TRY(q.init(query, static_cast<size_t>(iq->dim), st));
    VG_TRY(c.init(codes, static_cast<size_t>(n * cs), st));
    VG_TRY(o.init(out, static_cast<size_t>(n), st));
    const bool scan = iq->dim % 64 == 0 && (reinterpret_cast<uintptr_t>(c.ptr) & 15) == 0;
    const unsigned scan_blocks = static_cast<unsigned>(((n + 63) / 64 + vg::kI4Waves - 1) / vg::kI4Waves);
    vg::ProfScope prof(iq->ctx, "int4_scan", st);
    if (scan && iq->dim % 256 == 0 && iq->dim <= vg::kI4TabMaxDim) {
        const int waves = static_cast<int>(std::min<int64_t>(vg::kI4TabWaves, (160 * 1024 - static_cast<int64_t>(iq->dim) * 64) / (64 * vg::kI4TabStride)));
        const size_t lds = static_cast<size_t>(iq->dim) * 64 + static_cast<size_t>(waves) * 64 * vg::kI4TabStride;
        const int64_t tiles = (n + 63) / 64;
        const unsigned blocks = static_cast<unsigned>(std::min<int64_t>((tiles + waves - 1) / waves, std::max(iq->ctx->compute_units, 1)));
        auto kern = precomputed ? vg::int4_scan_tab_kernel<true> : vg::int4_scan_tab_kernel<false>;
        VG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   static_cast<int>(lds)));
        VG_LAUNCH(kern, dim3(blocks), dim3(waves * 64), lds, st, q.ptr, c.ptr, n, iq->dim, iq->d_min, iq->d_diff,
                  o.ptr);
    } else if (scan && precomputed)
        VG_LAUNCH(vg::int4_scan_kernel<true>, dim3(scan_blocks), dim3(vg::kI4Waves * 64), 0, st, q.ptr, c.ptr, n, iq->dim,
                  iq->d_min, iq->d_diff, o.ptr);
    else if (scan)
        VG_LAUNCH(vg::int4_scan_kernel<false>, dim3(scan_blocks), dim3(vg::kI4Waves * 64), 0, st, q.ptr, c.ptr, n, iq->dim,
                  iq->d_min, iq->d_diff, o.ptr);
    else if (precomputed)
        VG_LAUNCH(vg::int4_l2_precomputed_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, st, q.ptr,
                  c.ptr, n, iq->dim, iq->d_table, o.ptr);
    else
        VG_LAUNCH(vg::int4_l2_batch_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, st, q.ptr, c.ptr,
                  n, iq->dim, iq->d_min, iq->d_diff, o.ptr);
    VG_TRY(o.finish());
    return VG_OK;
}

// INT4 codes of a DiskANN segment, n * ceil(dim/2) bytes row-major (diskann/segment.go:378-416):
// kept in that layout, the graph search reads them by node id
VG_API int32_t vg_index_set_int4_codes(vg_index *idx, vg_int4 *iq, const uint8_t *codes, void *stream)
{
    VG_CHECK(idx && iq, VG_ERR_INVALID_ARG, "vg_index_set_int4_codes: NULL index or quantizer");
    VG_CHECK(iq->trained, VG_ERR_NOT_TRAINED, "Int4Quantizer not trained");
    VG_CHECK(iq->dim == idx->dim, VG_ERR_DIM_MISMATCH, "dimension mismatch");
    VG_CHECK(idx->n == 0 || codes, VG_ERR_INVALID_ARG, "vg_index_set_int4_codes: codes is NULL");
    VG_HIP(hipSetDevice(idx->ctx->device));
    hipStream_t st = vg::pick_stream(idx->ctx, stream);
    if (idx->d_int4_rows) {
        VG_HIP(hipStreamSynchronize(st));
        VG_HIP(hipFree(idx->d_int4_rows));
        idx->d_int4_rows = nullptr;
    }
    idx->int4_table = iq->d_table;
    idx->int4_min = iq->d_min;
    idx->int4_diff = iq->d_diff;
    if (idx->n == 0) return VG_OK;
    const size_t bytes = static_cast<size_t>(idx->n) * static_cast<size_t>(vg_int4_code_bytes(idx->dim));
    VG_HIP(hipMalloc(reinterpret_cast<void **>(&idx->d_int4_rows), bytes));
    VG_HIP(hipMemcpyAsync(idx->d_int4_rows, codes, bytes, hipMemcpyDefault, st));
    VG_HIP(hipStreamSynchronize(st));
    return VG_OK;
}
