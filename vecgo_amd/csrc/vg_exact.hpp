// vg_exact.hpp — fp32 L2 / dot product of one (query, row) pair by a 16-lane group, bit-identical
// to the reference's AVX-512 kernels:
//   PAIR   squaredL2Avx512 / dotProductAvx512   (internal/simd/src/floats_avx512.c:12-129)
//   BATCH  squaredL2BatchAvx512 / dotBatchAvx512 (internal/simd/src/batch_avx512.c:19-143)
//   BOUND  squaredL2BoundedAvx512                (internal/simd/src/bounded_l2_avx512.c:19-108)
//
// The reference keeps 4 zmm accumulators x 16 lanes: element e*64 + k*16 + l of block e goes to
// acc[k][l] with one FMA.  Here 16 GPU lanes share a row; each owns one 16-byte piece of every
// 64-float block, i.e. accumulator k and lanes l = 4*lq .. 4*lq+3 of it (4 fp32 accumulators
// per GPU lane).  The lane -> (k, lq) map is chosen so that every step of the reference's
// reduction is ONE DPP add inside the 16-lane row:
//   (acc0+acc1), (acc2+acc3)        partner lane i^7   row_half_mirror
//   their sum                       partner lane 15-i  row_mirror
//   reduce_add (l,l+8)              lq^2               quad_perm [2,3,0,1]
//   reduce_add (l,l+4)              lq^1               quad_perm [1,0,3,2]
//   reduce_add (l,l+2), (0,1)       in-lane
// fp32 addition is commutative, so both partners of a step hold the same bits afterwards.
#pragma once

#include <hip/hip_runtime.h>

#include "vg_device.hpp"

namespace vg {

constexpr int kDppQuadXor1 = 0xB1;       // quad_perm [1,0,3,2]
constexpr int kDppQuadXor2 = 0x4E;       // quad_perm [2,3,0,1]
constexpr int kDppRowMirror = 0x140;     // lane i <- lane 15-i
constexpr int kDppRowHalfMirror = 0x141; // lane i <- lane i^7 (within 8)

template <int CTRL>
__device__ __forceinline__ float dpp_partner_add(float x)
{
    int y = __builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, false);
    return x + __int_as_float(y);
}

struct Sub16 {
    int f4;  // which float4 of a 64-float block this lane owns (= k*4 + lq)
    int lq;  // which float4 of a 16-float block this lane owns in the 16-wide tail
    __device__ __forceinline__ static Sub16 make(int lane)
    {
        const int i = lane & 15;
        const int j = (i & 8) ? 15 - i : i;
        const int lq = (j & 4) ? 3 - (j & 3) : (j & 3);
        const int k = ((i & 8) ? 2 : 0) + ((j & 4) ? 1 : 0);
        Sub16 s;
        s.f4 = k * 4 + lq;
        s.lq = lq;
        return s;
    }
};

enum ExactMode { kPair = 0, kBatch = 1, kBounded = 2 };
constexpr int kExactChunk = 12;      // 64-float blocks whose loads are issued together (768 floats = one chunk)
constexpr int kExactChunkSmall = 4;  // ... and the size tried next for what is left
#ifndef VG_QLDS_CHUNK
#define VG_QLDS_CHUNK 6
#endif
// the same for the query-in-LDS form (split-heap walks at 4 waves per SIMD): 6 blocks = 91 registers, nothing spilled
// (12: 128 with 20 spilled; ef 1024 / 2048 per 8192 queries: 42.7 / 84.6 ms with 6, 43.3 / 86.2 with 8, 44.3 / 89.3 with 12)
constexpr int kExactChunkQ = VG_QLDS_CHUNK;

// One pair.  `row` and `q` point at dim floats (16-byte aligned when dim % 4 == 0, which the
// fast path requires; other dims take the scalar route below).  All 16 lanes of the group
// return the same value.
template <bool DOT, int MODE, bool STREAM = false, bool QLDS = false>
__device__ __forceinline__ float exact_pair16(const float *__restrict__ row,
                                              const float *__restrict__ q, int dim, Sub16 sub)
{
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const int nblk = dim >> 6;
    const float4 *r4 = reinterpret_cast<const float4 *>(row) + sub.f4;
    const float4 *q4 = reinterpret_cast<const float4 *>(q) + sub.f4;
    // kExactChunk blocks at a time: their 2 * kExactChunk loads are issued together (a one-block loop
    // serialises one memory round trip per 64 floats: 12 per 768-d row, which is what bounded the
    // graph searches).  The FMAs still run in block order, so every accumulator chain is unchanged.
    auto step = [&](const float4 a, const float4 b) {
        if (DOT) {
            acc[0] = __builtin_fmaf(a.x, b.x, acc[0]);
            acc[1] = __builtin_fmaf(a.y, b.y, acc[1]);
            acc[2] = __builtin_fmaf(a.z, b.z, acc[2]);
            acc[3] = __builtin_fmaf(a.w, b.w, acc[3]);
        } else {
            const float d0 = a.x - b.x, d1 = a.y - b.y, d2 = a.z - b.z, d3 = a.w - b.w;
            acc[0] = __builtin_fmaf(d0, d0, acc[0]);
            acc[1] = __builtin_fmaf(d1, d1, acc[1]);
            acc[2] = __builtin_fmaf(d2, d2, acc[2]);
            acc[3] = __builtin_fmaf(d3, d3, acc[3]);
        }
    };
    int e = 0;
    if (QLDS) {  // the query in LDS: read where used, not held beside the row (see exact_l2_both16)
        for (; e + kExactChunkQ <= nblk; e += kExactChunkQ) {
            float4 b[kExactChunkQ];
#pragma unroll
            for (int u = 0; u < kExactChunkQ; u++) b[u] = STREAM ? load_stream(r4 + (e + u) * 16) : r4[(e + u) * 16];
#pragma unroll
            for (int u = 0; u < kExactChunkQ; u++) {
                step(q4[(e + u) * 16], b[u]);
                if ((u & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    for (; !QLDS && e + kExactChunk <= nblk; e += kExactChunk) {
        float4 a[kExactChunk], b[kExactChunk];
#pragma unroll
        for (int u = 0; u < kExactChunk; u++) {
            a[u] = q4[(e + u) * 16];
            b[u] = STREAM ? load_stream(r4 + (e + u) * 16) : r4[(e + u) * 16];
        }
#pragma unroll
        for (int u = 0; u < kExactChunk; u++) step(a[u], b[u]);
    }
    for (; e + kExactChunkSmall <= nblk; e += kExactChunkSmall) {
        float4 a[kExactChunkSmall], b[kExactChunkSmall];
#pragma unroll
        for (int u = 0; u < kExactChunkSmall; u++) {
            a[u] = q4[(e + u) * 16];
            b[u] = STREAM ? load_stream(r4 + (e + u) * 16) : r4[(e + u) * 16];
        }
#pragma unroll
        for (int u = 0; u < kExactChunkSmall; u++) step(a[u], b[u]);
    }
    for (; e < nblk; e++) step(q4[e * 16], STREAM ? load_stream(r4 + e * 16) : r4[e * 16]);
    float s[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const float p = dpp_partner_add<kDppRowHalfMirror>(acc[t]);  // (acc0+acc1) | (acc2+acc3)
        s[t] = dpp_partner_add<kDppRowMirror>(p);                    // sum of the two
    }
    int j = nblk << 6;
    if (MODE == kBatch) {  // batch_avx512.c:60-66: 16-wide blocks go into the combined register
        for (; j + 16 <= dim; j += 16) {
            const float4 a = *(reinterpret_cast<const float4 *>(q + j) + sub.lq);
            const float4 b = *(reinterpret_cast<const float4 *>(row + j) + sub.lq);
            if (DOT) {
                s[0] = __builtin_fmaf(a.x, b.x, s[0]);
                s[1] = __builtin_fmaf(a.y, b.y, s[1]);
                s[2] = __builtin_fmaf(a.z, b.z, s[2]);
                s[3] = __builtin_fmaf(a.w, b.w, s[3]);
            } else {
                const float d0 = a.x - b.x, d1 = a.y - b.y, d2 = a.z - b.z, d3 = a.w - b.w;
                s[0] = __builtin_fmaf(d0, d0, s[0]);
                s[1] = __builtin_fmaf(d1, d1, s[1]);
                s[2] = __builtin_fmaf(d2, d2, s[2]);
                s[3] = __builtin_fmaf(d3, d3, s[3]);
            }
        }
    }
    float b[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const float a = dpp_partner_add<kDppQuadXor2>(s[t]);  // (l, l+8)
        b[t] = dpp_partner_add<kDppQuadXor1>(a);              // (l, l+4)
    }
    float total;
    if (MODE == kBounded) {
        total = (b[0] + b[1]) + (b[2] + b[3]);  // hsum512: two _mm_hadd_ps
        for (; j + 8 <= dim; j += 8) {          // AVX2 8-wide remainder: mul, hadd tree
            float qv[8];
#pragma unroll
            for (int l = 0; l < 8; l++) {
                const float d = q[j + l] - row[j + l];
                qv[l] = d * d;
            }
            const float p0 = qv[0] + qv[4], p1 = qv[1] + qv[5], p2 = qv[2] + qv[6], p3 = qv[3] + qv[7];
            total = total + ((p0 + p1) + (p2 + p3));
        }
    } else {
        total = (b[0] + b[2]) + (b[1] + b[3]);  // reduce_add: (l,l+2) then (0,1)
    }
    for (; j < dim; j++) {  // scalar tail: clang contracts `total += d*d` to one FMA
        if (DOT) {
            total = __builtin_fmaf(q[j], row[j], total);
        } else {
            const float d = q[j] - row[j];
            total = __builtin_fmaf(d, d, total);
        }
    }
    return total;
}

// squaredL2BoundedAvx512's EARLY EXIT (bounded_l2_avx512.c:19-108): the partial total it returns when a 64-float
// block's running sum first exceeds `bound` — the reduction (combine4 + hsum512) evaluated after every block, as the
// reference does.  Only called for pairs already known to exceed (exact_pair16<false, kBounded> > bound), so the
// block loop always terminates by its own test when the excess is reached inside the 64-blocks; an excess that only
// the 8-wide / scalar remainder produces returns the full sum, like the reference.  All 16 lanes return the value.
__device__ __forceinline__ float exact_l2_bounded_partial16(const float *__restrict__ row, const float *__restrict__ q,
                                                            int dim, Sub16 sub, float bound, float full)
{
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const int nblk = dim >> 6;
    const float4 *r4 = reinterpret_cast<const float4 *>(row) + sub.f4;
    const float4 *q4 = reinterpret_cast<const float4 *>(q) + sub.f4;
    for (int e = 0; e < nblk; e++) {
        const float4 a = q4[e * 16], b = r4[e * 16];
        const float d0 = a.x - b.x, d1 = a.y - b.y, d2 = a.z - b.z, d3 = a.w - b.w;
        acc[0] = __builtin_fmaf(d0, d0, acc[0]);
        acc[1] = __builtin_fmaf(d1, d1, acc[1]);
        acc[2] = __builtin_fmaf(d2, d2, acc[2]);
        acc[3] = __builtin_fmaf(d3, d3, acc[3]);
        float t[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const float p = dpp_partner_add<kDppRowHalfMirror>(acc[u]);
            const float s = dpp_partner_add<kDppRowMirror>(p);
            const float x = dpp_partner_add<kDppQuadXor2>(s);
            t[u] = dpp_partner_add<kDppQuadXor1>(x);
        }
        const float total = (t[0] + t[1]) + (t[2] + t[3]);
        if (total > bound) return total;  // the same bits in all 16 lanes: a uniform exit for the group
    }
    return full;
}

// L2 of one pair in BOTH reduction orders at once: `pair` = squaredL2Avx512 (what
// distFunc/SquaredL2 returns), `bnd` = squaredL2BoundedAvx512 run to completion.  The HNSW
// layer search needs both because which one the reference calls depends on heap state that
// evolves inside the neighbour loop (hnsw.go:1353-1376).  Since every partial sum of the
// bounded kernel is non-decreasing in the block index (squares are >= +0 and fp32 FMA / add
// are monotone), "some 64-block partial > bound" <=> "bnd > bound": the early exit does not
// have to be replayed.
// QLDS: `q` points into LDS (the wave's copy of the query): the query's pieces are then read where they are used
// (an LDS read is ~100 cycles, the compiler keeps a few ahead) instead of being held in registers next to the row's
// for a whole chunk — 48 registers less, which is what lets the split-heap walk run a fifth wave per SIMD.
template <bool STREAM = false, bool QLDS = false>
__device__ __forceinline__ void exact_l2_both16(const float *__restrict__ row,
                                                const float *__restrict__ q, int dim, Sub16 sub,
                                                float &pair, float &bnd)
{
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const int nblk = dim >> 6;
    const float4 *r4 = reinterpret_cast<const float4 *>(row) + sub.f4;
    const float4 *q4 = reinterpret_cast<const float4 *>(q) + sub.f4;
    auto step = [&](const float4 a, const float4 b) {
        const float d0 = a.x - b.x, d1 = a.y - b.y, d2 = a.z - b.z, d3 = a.w - b.w;
        acc[0] = __builtin_fmaf(d0, d0, acc[0]);
        acc[1] = __builtin_fmaf(d1, d1, acc[1]);
        acc[2] = __builtin_fmaf(d2, d2, acc[2]);
        acc[3] = __builtin_fmaf(d3, d3, acc[3]);
    };
    int e = 0;
    if (QLDS) {
        for (; e + kExactChunkQ <= nblk; e += kExactChunkQ) {
            float4 rb[kExactChunkQ];
#pragma unroll
            for (int u = 0; u < kExactChunkQ; u++) rb[u] = STREAM ? load_stream(r4 + (e + u) * 16) : r4[(e + u) * 16];
#pragma unroll
            for (int u = 0; u < kExactChunkQ; u++) {
                step(q4[(e + u) * 16], rb[u]);
                if ((u & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // keep the LDS reads from being hoisted into registers
            }
        }
    }
    for (; !QLDS && e + kExactChunk <= nblk; e += kExactChunk) {  // see exact_pair16
        float4 qa[kExactChunk], rb[kExactChunk];
#pragma unroll
        for (int u = 0; u < kExactChunk; u++) {
            qa[u] = q4[(e + u) * 16];
            rb[u] = STREAM ? load_stream(r4 + (e + u) * 16) : r4[(e + u) * 16];
        }
#pragma unroll
        for (int u = 0; u < kExactChunk; u++) step(qa[u], rb[u]);
    }
    for (; e + kExactChunkSmall <= nblk; e += kExactChunkSmall) {
        float4 qa[kExactChunkSmall], rb[kExactChunkSmall];
#pragma unroll
        for (int u = 0; u < kExactChunkSmall; u++) {
            qa[u] = q4[(e + u) * 16];
            rb[u] = STREAM ? load_stream(r4 + (e + u) * 16) : r4[(e + u) * 16];
        }
#pragma unroll
        for (int u = 0; u < kExactChunkSmall; u++) step(qa[u], rb[u]);
    }
    for (; e < nblk; e++) step(q4[e * 16], STREAM ? load_stream(r4 + e * 16) : r4[e * 16]);
    float b[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const float p = dpp_partner_add<kDppRowHalfMirror>(acc[t]);
        const float s = dpp_partner_add<kDppRowMirror>(p);
        const float a = dpp_partner_add<kDppQuadXor2>(s);
        b[t] = dpp_partner_add<kDppQuadXor1>(a);
    }
    float tp = (b[0] + b[2]) + (b[1] + b[3]);
    float tb = (b[0] + b[1]) + (b[2] + b[3]);
    int j = nblk << 6;
    int jb = j;
    for (; jb + 8 <= dim; jb += 8) {
        float qv[8];
#pragma unroll
        for (int l = 0; l < 8; l++) {
            const float d = q[jb + l] - row[jb + l];
            qv[l] = d * d;
        }
        const float p0 = qv[0] + qv[4], p1 = qv[1] + qv[5], p2 = qv[2] + qv[6], p3 = qv[3] + qv[7];
        tb = tb + ((p0 + p1) + (p2 + p3));
    }
    for (; jb < dim; jb++) {
        const float d = q[jb] - row[jb];
        tb = __builtin_fmaf(d, d, tb);
    }
    for (; j < dim; j++) {
        const float d = q[j] - row[j];
        tp = __builtin_fmaf(d, d, tp);
    }
    pair = tp;
    bnd = tb;
}

// One pair with the row already in registers (dim <= 1024, dim % 4 == 0: 16 float4 per lane of the
// 16-lane group, loaded by the caller) and the query in LDS; kPair order.
typedef float exact_f2 __attribute__((ext_vector_type(2)));
template <bool DOT>
__device__ __forceinline__ float exact_rowregs16(const float4 (&rr)[16], int nblk, const float *__restrict__ row,
                                                 const float *__restrict__ q, int dim, Sub16 sub)
{
    // the four accumulator chains of a lane go through the packed fp32 ops two at a time
    // (v_pk_add_f32 / v_pk_fma_f32): every half is the IEEE operation of its own chain
    exact_f2 lo = {0.0f, 0.0f}, hi = {0.0f, 0.0f};
    const float4 *q4 = reinterpret_cast<const float4 *>(q) + sub.f4;
#pragma unroll
    for (int e = 0; e < 16; e++) {
        if (e < nblk) {
            const float4 a = q4[e * 16];
            const float4 b = rr[e];
            const exact_f2 alo = {a.x, a.y}, ahi = {a.z, a.w}, blo = {b.x, b.y}, bhi = {b.z, b.w};
            if (DOT) {
                lo = __builtin_elementwise_fma(alo, blo, lo);
                hi = __builtin_elementwise_fma(ahi, bhi, hi);
            } else {
                const exact_f2 dlo = alo - blo, dhi = ahi - bhi;
                lo = __builtin_elementwise_fma(dlo, dlo, lo);
                hi = __builtin_elementwise_fma(dhi, dhi, hi);
            }
        }
    }
    const float acc[4] = {lo.x, lo.y, hi.x, hi.y};
    float b[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const float p = dpp_partner_add<kDppRowHalfMirror>(acc[t]);
        const float s2 = dpp_partner_add<kDppRowMirror>(p);
        const float a = dpp_partner_add<kDppQuadXor2>(s2);
        b[t] = dpp_partner_add<kDppQuadXor1>(a);
    }
    float total = (b[0] + b[2]) + (b[1] + b[3]);
    for (int j = nblk << 6; j < dim; j++) {  // scalar tail (FMA-contracted in the reference)
        if (DOT) {
            total = __builtin_fmaf(q[j], row[j], total);
        } else {
            const float d = q[j] - row[j];
            total = __builtin_fmaf(d, d, total);
        }
    }
    return total;
}

// The same for TWO rows held in registers: every LDS read of the query serves both rows (LDS
// bandwidth, not VALU issue, bounds the one-row form once several queries share a row pass).
template <bool DOT>
__device__ __forceinline__ void exact_rowregs16x2(const float4 (&ra)[16], const float4 (&rb)[16], int nblk,
                                                  const float *__restrict__ rowa, const float *__restrict__ rowb,
                                                  const float *__restrict__ q, int dim, Sub16 sub, float &va, float &vb)
{
    exact_f2 alo_acc = {0.0f, 0.0f}, ahi_acc = {0.0f, 0.0f}, blo_acc = {0.0f, 0.0f}, bhi_acc = {0.0f, 0.0f};
    const float4 *q4 = reinterpret_cast<const float4 *>(q) + sub.f4;
#pragma unroll
    for (int e = 0; e < 16; e++) {
        if (e < nblk) {
            const float4 x = q4[e * 16];
            const exact_f2 xlo = {x.x, x.y}, xhi = {x.z, x.w};
            const exact_f2 alo = {ra[e].x, ra[e].y}, ahi = {ra[e].z, ra[e].w};
            const exact_f2 blo = {rb[e].x, rb[e].y}, bhi = {rb[e].z, rb[e].w};
            if (DOT) {
                alo_acc = __builtin_elementwise_fma(xlo, alo, alo_acc);
                ahi_acc = __builtin_elementwise_fma(xhi, ahi, ahi_acc);
                blo_acc = __builtin_elementwise_fma(xlo, blo, blo_acc);
                bhi_acc = __builtin_elementwise_fma(xhi, bhi, bhi_acc);
            } else {
                const exact_f2 d0 = xlo - alo, d1 = xhi - ahi, d2 = xlo - blo, d3 = xhi - bhi;
                alo_acc = __builtin_elementwise_fma(d0, d0, alo_acc);
                ahi_acc = __builtin_elementwise_fma(d1, d1, ahi_acc);
                blo_acc = __builtin_elementwise_fma(d2, d2, blo_acc);
                bhi_acc = __builtin_elementwise_fma(d3, d3, bhi_acc);
            }
        }
    }
    const float acc[2][4] = {{alo_acc.x, alo_acc.y, ahi_acc.x, ahi_acc.y}, {blo_acc.x, blo_acc.y, bhi_acc.x, bhi_acc.y}};
    float tot[2];
#pragma unroll
    for (int r = 0; r < 2; r++) {
        float b[4];
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const float p = dpp_partner_add<kDppRowHalfMirror>(acc[r][t]);
            const float s2 = dpp_partner_add<kDppRowMirror>(p);
            const float a = dpp_partner_add<kDppQuadXor2>(s2);
            b[t] = dpp_partner_add<kDppQuadXor1>(a);
        }
        tot[r] = (b[0] + b[2]) + (b[1] + b[3]);
    }
    for (int j = nblk << 6; j < dim; j++) {  // scalar tail (FMA-contracted in the reference)
        if (DOT) {
            tot[0] = __builtin_fmaf(q[j], rowa[j], tot[0]);
            tot[1] = __builtin_fmaf(q[j], rowb[j], tot[1]);
        } else {
            const float da = q[j] - rowa[j], db = q[j] - rowb[j];
            tot[0] = __builtin_fmaf(da, da, tot[0]);
            tot[1] = __builtin_fmaf(db, db, tot[1]);
        }
    }
    va = tot[0];
    vb = tot[1];
}

}  // namespace vg
