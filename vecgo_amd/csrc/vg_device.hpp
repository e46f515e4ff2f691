// vg_device.hpp — device helpers shared by the gfx950 kernels.
//
// Numerics contract: this library is compiled with -ffp-contract=off.  Every FMA
// the reference's AVX-512 kernels issue is an explicit __builtin_fmaf here and every
// other fp32 op is a separately rounded IEEE op, so per-candidate distances are
// bit-identical to the reference's (see DESIGN.md "Summation order").
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace vg {

constexpr int kWave = 64;
constexpr uint64_t kKeyMax = 0xFFFFFFFFFFFFFFFFull;

// fp32 → uint32 whose unsigned order equals the float order (ascending).
__device__ __forceinline__ uint32_t f32_ordered(float f)
{
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ordered_f32(uint32_t u)
{
    uint32_t v = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
    return __uint_as_float(v);
}

// 64-bit selection key: smaller key == better candidate under
// searcher.InternalCandidateBetter (candidate_queue.go:12-23) for one segment:
// score first (ascending for L2-like, descending for Dot), then RowID ascending.
__device__ __forceinline__ uint64_t make_key(float score, uint32_t row, bool descending)
{
    uint32_t s = f32_ordered(score);
    if (descending) s = ~s;
    return (static_cast<uint64_t>(s) << 32) | row;
}
__device__ __forceinline__ float key_score(uint64_t key, bool descending)
{
    uint32_t s = static_cast<uint32_t>(key >> 32);
    if (descending) s = ~s;
    return ordered_f32(s);
}
__device__ __forceinline__ uint32_t key_row(uint64_t key) { return static_cast<uint32_t>(key); }

// Bitonic sort (ascending) of n = power-of-two 64-bit keys in LDS by the whole
// workgroup.  Caller guarantees a __syncthreads() before; ends with one.
__device__ __forceinline__ void bitonic_sort_lds(uint64_t *buf, int n, int tid, int nthreads)
{
    for (int size = 2; size <= n; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < (n >> 1); t += nthreads) {
                int lo = ((t / stride) * (stride << 1)) + (t % stride);
                int hi = lo + stride;
                bool up = ((lo & size) == 0);
                uint64_t a = buf[lo], b = buf[hi];
                if ((a > b) == up) {
                    buf[lo] = b;
                    buf[hi] = a;
                }
            }
            __syncthreads();
        }
    }
}

// _mm512_reduce_add_ps order over 16 per-lane partial sums held in registers:
// (i,i+8) → (i,i+4) → (i,i+2) → (0,1)   [internal/simd/floats_avx512.s]
__device__ __forceinline__ float reduce16_regs(const float (&s)[16])
{
    float a[8], b[4], c[2];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = s[i] + s[i + 8];
#pragma unroll
    for (int i = 0; i < 4; i++) b[i] = a[i] + a[i + 4];
#pragma unroll
    for (int i = 0; i < 2; i++) c[i] = b[i] + b[i + 2];
    return c[0] + c[1];
}

// ---------------------------------------------------------------------------------
// Per-wave top-k (k <= 64) held in registers: lane i owns the i-th best key seen by this
// wave.  Insertion is branch-uniform and touches no LDS memory, so a streaming scan needs no
// workgroup barrier.  After the first tile the pass rate is ~k/rows_seen, so the insert
// loop is cold.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int lane)
{
    uint32_t lo = __builtin_amdgcn_readlane(static_cast<uint32_t>(v), lane);
    uint32_t hi = __builtin_amdgcn_readlane(static_cast<uint32_t>(v >> 32), lane);
    return (static_cast<uint64_t>(hi) << 32) | lo;
}

// A 16-byte load of data that is read once (a scan's codes / rows): the nontemporal hint keeps the
// stream from displacing what the kernel does reuse and measures 5 % faster on gfx950 (3 GiB read:
// 6.62 vs 6.31 TB/s, tools/ubench/stream_read.hip).  HIP's uint4 / float4 are union structs the builtin
// does not take, hence the native vector types.
typedef unsigned int vg_u4v __attribute__((ext_vector_type(4)));
typedef float vg_f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 load_stream(const uint4 *p)
{
    const vg_u4v v = __builtin_nontemporal_load(reinterpret_cast<const vg_u4v *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float4 load_stream(const float4 *p)
{
    const vg_f4v v = __builtin_nontemporal_load(reinterpret_cast<const vg_f4v *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}

// row filters: bit i of byte i/8 set = row i takes part; no mask = every row
__device__ __forceinline__ bool mask_bit(const uint8_t *__restrict__ mask, int64_t i)
{
    return mask == nullptr || ((mask[i >> 3] >> (i & 7)) & 1);
}

struct WaveTopK {
    uint64_t list;  // lane i: i-th smallest key of this wave so far
    uint64_t tau;   // wave-uniform: key at lane k-1 (kKeyMax until k keys were seen)
    int k;
    __device__ __forceinline__ void init(int k_)
    {
        list = kKeyMax;
        tau = kKeyMax;
        k = k_;
    }
    // every lane offers one key (kKeyMax = nothing)
    __device__ __forceinline__ void offer(uint64_t key, int lane)
    {
        uint64_t mask = __ballot(key < tau);
        while (mask) {
            const int j = __builtin_ctzll(mask);
            mask &= mask - 1;
            const uint64_t c = readlane_u64(key, j);
            if (c < tau) {  // wave-uniform: tau may have dropped since the ballot
                uint32_t llo = __shfl_up(static_cast<uint32_t>(list), 1);
                uint32_t lhi = __shfl_up(static_cast<uint32_t>(list >> 32), 1);
                const uint64_t left = (static_cast<uint64_t>(lhi) << 32) | llo;
                const bool keep = list < c;
                const bool left_lt = (lane == 0) || (left < c);
                list = keep ? list : (left_lt ? c : left);
                tau = readlane_u64(list, k - 1);
            }
        }
    }
};

// int4L2DistancePrecomputedAvx512 (int4_avx512.c:127-189) of one code: what Int4Quantizer.L2Distance
// runs (int4.go:140-147) and therefore the DiskANN node scorer
__device__ __forceinline__ float int4_l2_precomputed(const float *__restrict__ query, const uint8_t *__restrict__ code, int dim,
                                                     const float *__restrict__ table)
{
    float sum[16];
#pragma unroll
    for (int l = 0; l < 16; l++) sum[l] = 0.0f;
    int i = 0;
    // 32 elements at a time when the code is 16-byte aligned: one 16-byte load of nibbles, then 32
    // table reads in flight; the two 16-element blocks still feed sum[] in order
    if ((reinterpret_cast<uintptr_t>(code) & 15) == 0) {
        for (; i <= dim - 32; i += 32) {
            const uint4 c = *reinterpret_cast<const uint4 *>(code + (i >> 1));
            const uint32_t w[4] = {c.x, c.y, c.z, c.w};
            float t[32];
#pragma unroll
            for (int l = 0; l < 32; l++) {
                const uint32_t b = (w[l >> 3] >> (8 * ((l >> 1) & 3))) & 0xFFu;
                const int qv = (l & 1) ? (b & 0x0F) : (b >> 4);
                t[l] = table[(i + l) * 16 + qv];
            }
#pragma unroll
            for (int l = 0; l < 16; l++) {
                const float d = query[i + l] - t[l];
                sum[l] = __builtin_fmaf(d, d, sum[l]);
            }
#pragma unroll
            for (int l = 0; l < 16; l++) {
                const float d = query[i + 16 + l] - t[16 + l];
                sum[l] = __builtin_fmaf(d, d, sum[l]);
            }
        }
    }
    for (; i <= dim - 16; i += 16) {
#pragma unroll
        for (int l = 0; l < 16; l++) {
            const int j = i + l;
            const uint8_t b = code[j >> 1];
            const int qv = (j & 1) ? (b & 0x0F) : (b >> 4);
            const float d = query[j] - table[j * 16 + qv];
            sum[l] = __builtin_fmaf(d, d, sum[l]);
        }
    }
    float total = reduce16_regs(sum);
    for (; i < dim; i++) {
        const uint8_t b = code[i >> 1];
        const int qv = (i & 1) ? (b & 0x0F) : (b >> 4);
        const float d = query[i] - table[i * 16 + qv];
        total = __builtin_fmaf(d, d, total);
    }
    return total;
}

// The same distance without the dim x 16 table (48 KiB at dim 768: more than a CU's L1, so a walk's lookups
// are L2 round trips): table[j][v] = float(v) / 15 * diff[j] + min[j] (int4.go:152-163) evaluated in place.
// `pairs` (LDS, 256 entries, int4_fill_pairs) maps a code byte to (float(hi) / 15, float(lo) / 15): the two
// values a byte holds are neighbouring AVX-512 lanes, so every step is one packed-fp32 instruction on the
// pair; diff / min / query are wave-uniform (scalar loads).  dim % 32 == 0 and 16-byte aligned codes.
typedef float vg_f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void int4_fill_pairs(vg_f2v *pairs, int tid, int nthreads)
{
    for (int b = tid; b < 256; b += nthreads) {
        vg_f2v a;
        a.x = static_cast<float>(b >> 4) / 15.0f;
        a.y = static_cast<float>(b & 15) / 15.0f;
        pairs[b] = a;
    }
}
__device__ __forceinline__ float int4_l2_direct(const float *__restrict__ query, const uint8_t *__restrict__ code, int dim,
                                                const float *__restrict__ mins, const float *__restrict__ diff,
                                                const vg_f2v *pairs)
{
    vg_f2v sum[8];
#pragma unroll
    for (int p = 0; p < 8; p++) sum[p] = vg_f2v{0.0f, 0.0f};
    for (int i = 0; i < dim; i += 32) {
        const uint4 c = *reinterpret_cast<const uint4 *>(code + (i >> 1));
        const uint32_t w[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
        for (int h = 0; h < 2; h++) {  // the two 16-element blocks feed sum[] in order
#pragma unroll
            for (int p = 0; p < 8; p++) {
                const int byte = 8 * h + p;
                const uint32_t b = (w[byte >> 2] >> (8 * (byte & 3))) & 0xFFu;
                const int j = i + 2 * byte;
                const vg_f2v a = pairs[b];
                const vg_f2v df = *reinterpret_cast<const vg_f2v *>(diff + j);
                const vg_f2v mn = *reinterpret_cast<const vg_f2v *>(mins + j);
                const vg_f2v qq = *reinterpret_cast<const vg_f2v *>(query + j);
                vg_f2v t = a * df;
                t = t + mn;
                const vg_f2v d = qq - t;
                sum[p] = __builtin_elementwise_fma(d, d, sum[p]);
            }
        }
    }
    float s16[16];
#pragma unroll
    for (int p = 0; p < 8; p++) {
        s16[2 * p] = sum[p].x;
        s16[2 * p + 1] = sum[p].y;
    }
    return reduce16_regs(s16);
}

// Rank-merge of the per-wave sorted lists of one workgroup into `out[0..k)` (ascending,
// kKeyMax padded): a key's final position is its own index plus the number of smaller keys in
// every other wave's list (keys are unique).  `lists` = WAVES*64 keys of LDS, `valid` = WAVES ints.
template <int WAVES>
__device__ __forceinline__ void wg_rank_merge(const WaveTopK &tk, uint64_t *lists, int *valid,
                                              int wave, int lane, int tid, int k, uint64_t *out)
{
    lists[wave * 64 + lane] = tk.list;
    const int nvalid = __popcll(__ballot(tk.list != kKeyMax));
    if (lane == 0) valid[wave] = nvalid;
    __syncthreads();
    const uint64_t e = tk.list;
    if (e != kKeyMax) {
        // the WAVES binary searches are independent: run them step by step side by side (7 rounds of WAVES
        // LDS reads in flight instead of 7*WAVES dependent ones); in its own list a key's count is its lane
        int pos[WAVES];
#pragma unroll
        for (int w = 0; w < WAVES; w++) pos[w] = 0;
#pragma unroll
        for (int step = 32; step > 0; step >>= 1) {
#pragma unroll
            for (int w = 0; w < WAVES; w++)
                if (lists[w * 64 + pos[w] + step - 1] < e) pos[w] += step;
        }
        int rank = 0;
#pragma unroll
        for (int w = 0; w < WAVES; w++) rank += pos[w] + (lists[w * 64 + pos[w]] < e ? 1 : 0);
        if (rank < k) out[rank] = e;
    }
    int total = 0;
    for (int w = 0; w < WAVES; w++) total += valid[w];
    for (int i = total + tid; i < k; i += WAVES * 64) out[i] = kKeyMax;
}

}  // namespace vg
