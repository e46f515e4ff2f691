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

}  // namespace vg
