// vg_flat_gemm.hpp — the fp32 MFMA GEMM that nominates candidates for the flat search (k_flat.hip).
// Kept in a header so that tools/ubench/gemm_probe.hip can time variants of the same code.
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "vg_device.hpp"

namespace vg {

// ---- 1. GEMM ---------------------------------------------------------------------------------
// C tile 128 (queries) x 128 (rows) per workgroup, K step 32, 4 waves as 2x2, each wave 64x64 =
// 2x2 v_mfma_f32_32x32x2_f32 tiles.  Operands sit row-major in LDS with a leading dimension of
// 33 floats: the MFMA operand read (lane l: row l&31, k = l>>5) then touches 32 consecutive
// banks.  Registers prefetch the next K step while the current one is multiplied.
constexpr int kGemmBM = 128, kGemmBN = 128, kGemmBK = 32, kGemmLd = kGemmBK + 1;
constexpr int kGemmPasses = kGemmBM * kGemmBK / 4 / 256;  // float4 loads per thread per operand tile
constexpr int kGemmThreads = 256;
constexpr int kGemmTile = kGemmBM * kGemmLd;  // floats of one operand tile in LDS
constexpr size_t kGemmLdsBytes = 4 * kGemmTile * sizeof(float);  // A,B double-buffered: 66 KiB
using f32x16 = __attribute__((ext_vector_type(16))) float;

// Epilogue shared by both GEMM kernels.  C/D map of the 32x32 MFMA: col = lane&31 (row index n),
// row = (r&3) + 8*(r>>2) + 4*(lane>>5).  MODE 2 compares every element with its query's threshold.
// Read in place the thresholds were 64 dependent global loads per lane (6 % of the kernel); read
// one by one from LDS between the branches of the append they were 64 serialised LDS round trips.
// Now: thr_reg (thread t < 128: threshold of query q0+t) and xn (||x||^2 of this lane's two
// columns) are loaded by the caller BEFORE the K loop; the epilogue stages the 128 thresholds in
// LDS (`lds_thr`, free once the K loop is over) and every lane pulls its 32 with 8 ds_read_b128.
template <bool DOT, int MODE>
__device__ __forceinline__ void gemm_epilogue(const f32x16 (&acc)[2][2], float *lds_thr, float thr_reg,
                                              const float (&xn)[2], int tid, int lane, int wr, int wc, int64_t q0,
                                              int64_t nq, int64_t n0, int64_t n, int64_t tn,
                                              float *__restrict__ scores, int64_t out_cols,
                                              int *__restrict__ counts, uint64_t *__restrict__ cand, int cap,
                                              const uint8_t *__restrict__ mask, int64_t mask_stride, uint32_t row_base = 0,
                                              const int64_t *__restrict__ mask_off = nullptr /* grouped: byte offset of query
                                              row qq's filter in `mask` (a pair's query), instead of qq * mask_stride */)
{
    auto filter_of = [&](int64_t qq) { return mask_off ? mask + mask_off[qq] : mask + qq * mask_stride; };
    float4 t4[2][4];  // thresholds of rows i*32 + 8*g + 4*(lane>>5) + 0..3
    if (MODE == 2) {
        if (tid < kGemmBM) lds_thr[tid] = thr_reg;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int g = 0; g < 4; g++)
                t4[i][g] = *reinterpret_cast<const float4 *>(lds_thr + wr * 64 + i * 32 + 8 * g + 4 * (lane >> 5));
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int col = wc * 64 + j * 32 + (lane & 31);
        const int64_t nn = n0 + col;
#pragma unroll
        for (int i = 0; i < 2; i++) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int ql = wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int64_t qq = q0 + ql;
                const float dotv = acc[i][j][r];
                const float sc = DOT ? -dotv : __builtin_fmaf(-2.0f, dotv, xn[j]);
                if (MODE == 0) {
                    if (qq < nq && nn < n) scores[qq * n + nn] = sc;
                } else if (MODE == 1) {
                    // (a row the query's filter rejects is not in the sample: the threshold is a quantile of the rows it wants)
                    if (qq < nq)
                        scores[qq * out_cols + tn * kGemmBN + col] =
                            nn < n && (mask == nullptr || mask_bit(filter_of(qq), nn + row_base)) ? sc : INFINITY;
                } else {
                    const float4 tv = t4[i][r >> 2];
                    const float t = (r & 3) == 0 ? tv.x : (r & 3) == 1 ? tv.y : (r & 3) == 2 ? tv.z : tv.w;
                    // the filter bit is looked at only for the few elements below the threshold
                    if (nn < n && sc < t && (mask == nullptr || mask_bit(filter_of(qq), nn + row_base))) {
                        const int pos = atomicAdd(&counts[qq], 1);
                        if (pos < cap) cand[qq * cap + pos] = make_key(sc, static_cast<uint32_t>(nn) + row_base, false);
                    }
                }
            }
        }
    }
}

// the caller-side loads that go with it
template <bool DOT, int MODE>
__device__ __forceinline__ void gemm_epilogue_inputs(float &thr_reg, float (&xn)[2], int tid, int lane, int wc,
                                                     int64_t q0, int64_t nq, int64_t n0, int64_t n,
                                                     const float *__restrict__ norms,
                                                     const float *__restrict__ thr, int thr_stride, int thr_off)
{
    thr_reg = -INFINITY;  // -Inf: nothing passes (queries past nq)
    if (MODE == 2 && tid < kGemmBM && q0 + tid < nq) thr_reg = thr[(q0 + tid) * thr_stride + thr_off];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int64_t nn = n0 + wc * 64 + j * 32 + (lane & 31);
        xn[j] = (!DOT && nn < n) ? norms[nn] : 0.0f;
    }
}

// MODE 0: scores[q][n] for every row.  MODE 1: only every tile_stride-th row tile, written
// compactly (row length out_cols; columns past n hold +Inf) — the sample that sets the per-query
// threshold.  MODE 2: no score matrix at all: an element below its query's threshold is appended
// (64-bit key) to that query's candidate buffer.  Whatever the threshold, every row NOT appended
// has score >= threshold, which is all the proof in flat_verify_kernel needs.
// PROBE (tools/ubench/gemm_probe.hip only; the library instantiates PROBE = 0): bit 0 drops the
// epilogue, bit 1 the global loads of the K loop, bit 2 its LDS stores, bit 3 its barrier, bit 4
// its LDS operand reads — each variant computes garbage and exists to price that stage.
template <bool DOT, int MODE, int PROBE = 0>
__global__ __launch_bounds__(kGemmThreads) void flat_gemm_kernel(
    const float *__restrict__ queries, int64_t nq, const float *__restrict__ base, int64_t n,
    int dim, const float *__restrict__ norms, float *__restrict__ scores, int tile_stride,
    int64_t out_cols, const float *__restrict__ thr, int thr_stride, int thr_off,
    int *__restrict__ counts, uint64_t *__restrict__ cand, int cap, const uint8_t *__restrict__ mask = nullptr,
    int64_t mask_stride = 0)
{
    extern __shared__ float gemm_lds[];  // A0 | B0 | A1 | B1, kGemmLdsBytes
    float *const As0 = gemm_lds, *const Bs0 = gemm_lds + kGemmTile;
    float *const As1 = gemm_lds + 2 * kGemmTile, *const Bs1 = gemm_lds + 3 * kGemmTile;
    // XCD-aware block order: blocks b, b+8, ... share an XCD (and its L2).  All query tiles of
    // one row tile go to the same XCD, back to back, so the row tile crosses the fabric once
    // instead of once per query tile (measured: FETCH_SIZE 12.4 GB -> see DESIGN.md §4).
    const int mtiles = static_cast<int>((nq + kGemmBM - 1) / kGemmBM);
    const int64_t ntiles = MODE == 1 ? (((n + kGemmBN - 1) / kGemmBN) + tile_stride - 1) / tile_stride
                                     : (n + kGemmBN - 1) / kGemmBN;
    const int64_t bt = blockIdx.x;
    const int64_t xcd = bt & 7, jx = bt >> 3;
    const int64_t tn = (jx / mtiles) * 8 + xcd;
    const int tm = static_cast<int>(jx % mtiles);
    if (tn >= ntiles) return;
    const int64_t q0 = static_cast<int64_t>(tm) * kGemmBM;
    const int64_t n0 = (MODE == 1 ? tn * tile_stride : tn) * kGemmBN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;

    // staging map: BK/4 lanes cover one row's BK floats, 256/(BK/4) rows per pass
    constexpr int kLanesPerRow = kGemmBK / 4;
    constexpr int kRowsPerPass = kGemmThreads / kLanesPerRow;
    const int srow = tid / kLanesPerRow;
    const int sk = (tid % kLanesPerRow) * 4;
    const float *aptr[kGemmPasses];
    const float *bptr[kGemmPasses];
#pragma unroll
    for (int p = 0; p < kGemmPasses; p++) {
        int64_t qa = q0 + p * kRowsPerPass + srow;
        if (qa >= nq) qa = nq - 1;
        int64_t nb = n0 + p * kRowsPerPass + srow;
        if (nb >= n) nb = n - 1;
        aptr[p] = queries + qa * dim + sk;
        bptr[p] = base + nb * dim + sk;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    float4 ra[kGemmPasses], rb[kGemmPasses];
    const int ksteps = (dim + kGemmBK - 1) / kGemmBK;
    const int full_steps = dim / kGemmBK;  // K steps with no ragged edge (uniform per kernel)
    auto load_tile = [&](int kt) {
        const int k0 = kt * kGemmBK;
        if (kt < full_steps) {  // unguarded: 8 independent 16-byte loads in flight
#pragma unroll
            for (int p = 0; p < kGemmPasses; p++) {
                ra[p] = *reinterpret_cast<const float4 *>(aptr[p] + k0);
                rb[p] = *reinterpret_cast<const float4 *>(bptr[p] + k0);
            }
        } else {  // ragged K edge: element-wise with zero fill
#pragma unroll
            for (int p = 0; p < kGemmPasses; p++) {
                float ta[4] = {0, 0, 0, 0}, tb[4] = {0, 0, 0, 0};
                for (int e = 0; e < 4; e++)
                    if (k0 + sk + e < dim) {
                        ta[e] = aptr[p][k0 + e];
                        tb[e] = bptr[p][k0 + e];
                    }
                ra[p] = make_float4(ta[0], ta[1], ta[2], ta[3]);
                rb[p] = make_float4(tb[0], tb[1], tb[2], tb[3]);
            }
        }
    };
    // One operand pass (a float4 of A and of B per thread) from registers into an LDS buffer.
    auto store_pass = [&](float *As, float *Bs, int p) {
        float *da = As + (p * kRowsPerPass + srow) * kGemmLd + sk;
        float *db = Bs + (p * kRowsPerPass + srow) * kGemmLd + sk;
        da[0] = ra[p].x; da[1] = ra[p].y; da[2] = ra[p].z; da[3] = ra[p].w;
        db[0] = rb[p].x; db[1] = rb[p].y; db[2] = rb[p].z; db[3] = rb[p].w;
    };
    static_assert(kGemmPasses == 4 && kGemmBK == 32, "pipeline below is written for 4 passes and 8 groups");
    float thr_reg, xn[2];
    gemm_epilogue_inputs<DOT, MODE>(thr_reg, xn, tid, lane, wc, q0, nq, n0, n, norms, thr, thr_stride, thr_off);
    load_tile(0);
#pragma unroll
    for (int p = 0; p < kGemmPasses; p++) store_pass(As0, Bs0, p);
    __syncthreads();
    const int a_off = (wr * 64 + (lane & 31)) * kGemmLd + (lane >> 5);
    const int b_off = (wc * 64 + (lane & 31)) * kGemmLd + (lane >> 5);
    // K loop, LDS double-buffered: while buffer kt&1 is multiplied, the global loads of step kt+1
    // are in flight and land in the other buffer during the second half of the step, so a step
    // costs ONE barrier and the MFMA pipe never waits for a tile.  Inside a step the operand
    // fragments of group g+1 (two MFMA k-steps) are read from LDS before group g's 8 MFMAs issue.
    for (int kt = 0; kt < ksteps; kt++) {
        const bool more = kt + 1 < ksteps;
        const float *a_base = ((kt & 1) ? As1 : As0) + a_off;
        const float *b_base = ((kt & 1) ? Bs1 : Bs0) + b_off;
        float *An = (kt & 1) ? As0 : As1, *Bn = (kt & 1) ? Bs0 : Bs1;
        if (more && !(PROBE & 2)) load_tile(kt + 1);
        float fa[2][4], fb[2][4];  // [parity][a0 k, a0 k+2, a1 k, a1 k+2]
        fa[0][0] = a_base[0]; fa[0][1] = a_base[2];
        fa[0][2] = a_base[32 * kGemmLd]; fa[0][3] = a_base[32 * kGemmLd + 2];
        fb[0][0] = b_base[0]; fb[0][1] = b_base[2];
        fb[0][2] = b_base[32 * kGemmLd]; fb[0][3] = b_base[32 * kGemmLd + 2];
#pragma unroll
        for (int g = 0; g < 8; g++) {
            const int c = g & 1, nx = c ^ 1;
            if (g < 7 && !(PROBE & 16)) {
                const int kk = 4 * (g + 1);
                fa[nx][0] = a_base[kk]; fa[nx][1] = a_base[kk + 2];
                fa[nx][2] = a_base[32 * kGemmLd + kk]; fa[nx][3] = a_base[32 * kGemmLd + kk + 2];
                fb[nx][0] = b_base[kk]; fb[nx][1] = b_base[kk + 2];
                fb[nx][2] = b_base[32 * kGemmLd + kk]; fb[nx][3] = b_base[32 * kGemmLd + kk + 2];
            }
            if (more && g >= 4 && !(PROBE & 4)) store_pass(An, Bn, g - 4);
            __builtin_amdgcn_sched_barrier(0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][0], fb[c][0], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][0], fb[c][2], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][2], fb[c][0], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][2], fb[c][2], acc[1][1], 0, 0, 0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][1], fb[c][1], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][1], fb[c][3], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][3], fb[c][1], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][3], fb[c][3], acc[1][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!(PROBE & 8)) __syncthreads();
    }
    if (PROBE & 1) {  // keep the accumulators alive without an epilogue
        float t = 0.0f;
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 2; j++)
                for (int r = 0; r < 16; r++) t += acc[i][j][r];
        if (t == 123.456f) scores[0] = t;
        return;
    }
    gemm_epilogue<DOT, MODE>(acc, gemm_lds, thr_reg, xn, tid, lane, wr, wc, q0, nq, n0, n, tn, scores, out_cols,
                             counts, cand, cap, mask, mask_stride);
}

// ---- the same GEMM with LDS-DMA staging (dim % 4 == 0) -----------------------------------------
// Tiles go global -> LDS by `global_load_lds_dwordx4`: no staging registers, no ds_write pass, and
// the transfer of step kt+1 has the whole of step kt to land.  One wave-instruction fills 1 KiB =
// 8 rows x 32 floats, lane l -> row l>>3, 16-byte slot l&7.  The image is unpadded (row = 128 B)
// and XOR-swizzled: slot s of row R holds k-granule s ^ ((R>>1)&7).  The DMA destination is
// lane-linear, so the swizzle is applied to each lane's SOURCE address; a row's 8 lanes still
// cover one 128-byte line.  Operands are read back with ds_read_b128: lane (row, h = lane>>5)
// takes granule 2j+h, i.e. k = 4(2j+h)+c for component c, and MFMA c of group j consumes
// component c from both operands — the two k-slots of an MFMA need not be adjacent, only equal
// on the A and B side.  With f(R) = (R>>1)&7 each 16-lane ds_read_b128 group ({0-3,12-15,20-27},
// ...) touches 16 distinct 16-byte slots of the 256-byte bank row: conflict-free.
constexpr int kDmaTile = kGemmBM * kGemmBK;                           // floats, no padding
constexpr size_t kDmaLdsBytes = 4 * kDmaTile * sizeof(float);         // 64 KiB: A0 | B0 | A1 | B1

// source of the zero fill past a ragged K edge
static __device__ float4 g_gemm_zero16;

// One LDS-DMA wave-instruction: lane l's 16 bytes at `src` land at LDS byte address lds_base + 16*l.
// Inline asm on purpose: hipcc counts a builtin LDS-DMA as a pending LDS write and drains it
// (vmcnt(0)) before the very next ds_read, which serialises transfer and compute; the asm form is
// invisible to that bookkeeping and is retired by the explicit vmcnt(0) in front of the barrier
// that publishes the tile.  (Uncounted extra loads only make compiler-emitted vmcnt(N) waits
// stronger, never weaker.)  M0 is saved and restored inside the statement.
__device__ __forceinline__ void glds16(const float *src, uint32_t lds_base)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(lds_base)
                 : "memory");
}

// Same, addressed as wave-uniform base (SGPR pair) + per-lane 32-bit byte offset.  Measured next to
// back-to-back fp32 MFMAs (tools/ubench/mfma_vmem.hip, 2 waves per SIMD): this form takes 16 cycles
// of MFMA issue per instruction, the 64-bit per-lane address form 53 — the K loop uses this one.
__device__ __forceinline__ void glds16(const float *base, uint32_t byte_off, uint32_t lds_base)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(byte_off), "s"(lds_base), "s"(base)
                 : "memory");
}

// The same with the nontemporal hint, for base rows a kernel reads exactly once (the small-batch tile:
// one workgroup per row tile, nobody else touches those rows).
__device__ __forceinline__ void glds16_stream(const float *base, uint32_t byte_off, uint32_t lds_base)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(byte_off), "s"(lds_base), "s"(base)
                 : "memory");
}

// BF16 = true: `queries` and `base` hold bfloat16 rows and `dim` counts 4-byte words per row (elements / 2): the tiles
// are moved exactly as above (a K step is 128 bytes of a row either way, a 16-byte granule now 8 elements), only the
// matrix instruction changes — v_mfma_f32_32x32x16_bf16 takes one granule per lane (lane (row, h): the 8 elements
// 16j + 8h ..) where the fp32 form takes one float, so a K step is 16 instead of 64 matrix instructions per wave.
// The scores are a FILTER for the exact fp32 re-score (vg_index_enable_bf16_filter, k_flat.hip); elements / 2 must be
// a multiple of kGemmBK.
typedef __bf16 vg_bf16x8 __attribute__((ext_vector_type(8)));
// (the kernel's body as a function: flat_gemm_dma_kernel runs it on one problem, flat_gemm_dma_grouped_kernel on the problem its
// workgroup belongs to.  bt = the workgroup's index within its problem; GROUPED: row ids in the appended keys are row_base + the
// row's index in `base`)
template <bool DOT, int MODE, int PROBE, bool BF16, bool GROUPED>
__device__ __forceinline__ void flat_gemm_dma_body(
    const float *__restrict__ queries, int64_t nq, const float *__restrict__ base, int64_t n,
    int dim, const float *__restrict__ norms, float *__restrict__ scores, int tile_stride,
    int64_t out_cols, const float *__restrict__ thr, int thr_stride, int thr_off,
    int *__restrict__ counts, uint64_t *__restrict__ cand, int cap, const uint8_t *__restrict__ mask,
    int64_t mask_stride, int64_t bt, uint32_t row_base, const int64_t *__restrict__ mask_off = nullptr)
{
    extern __shared__ float gemm_lds[];
    const int mtiles = static_cast<int>((nq + kGemmBM - 1) / kGemmBM);
    const int64_t ntiles = MODE == 1 ? (((n + kGemmBN - 1) / kGemmBN) + tile_stride - 1) / tile_stride
                                     : (n + kGemmBN - 1) / kGemmBN;
    const int64_t xcd = bt & 7, jx = bt >> 3;
    const int64_t tn = (jx / mtiles) * 8 + xcd;
    const int tm = static_cast<int>(jx % mtiles);
    if (tn >= ntiles) return;
    const int64_t q0 = static_cast<int64_t>(tm) * kGemmBM;
    const int64_t n0 = (MODE == 1 ? tn * tile_stride : tn) * kGemmBN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    // DMA map: pass p, wave w fill rows p*32 + w*8 .. +8 of a tile
    const int drow = wave * 8 + (lane >> 3);                       // + p*32
    const int dgl = (lane & 7) ^ ((wave * 4 + (lane >> 4)) & 7);   // k-granule this lane fetches
    // source = tile base (wave-uniform, advanced by the K step) + per-lane byte offset (loop-invariant;
    // < 128 * dim * 4, so 32 bits do); rows past the end of the matrix re-read its last row
    const float *const abase = queries + q0 * dim;
    const float *const bbase = base + n0 * dim;
    uint32_t aoff[kGemmPasses], boff[kGemmPasses];
#pragma unroll
    for (int p = 0; p < kGemmPasses; p++) {
        int64_t qa = q0 + p * 32 + drow;
        if (qa >= nq) qa = nq - 1;
        int64_t nb = n0 + p * 32 + drow;
        if (nb >= n) nb = n - 1;
        aoff[p] = static_cast<uint32_t>(((qa - q0) * dim + dgl * 4) * 4);
        boff[p] = static_cast<uint32_t>(((nb - n0) * dim + dgl * 4) * 4);
    }
    const int ksteps = (dim + kGemmBK - 1) / kGemmBK;
    const int full_steps = dim / kGemmBK;  // K steps with no ragged edge (uniform per kernel)
    const int kstart = BF16 ? (tm * 5) % ksteps : 0;  // (5: coprime with the 12 steps of d = 768; 1-7 measured alike, 8 % over none)
    const uint32_t lds0 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(
        (__attribute__((address_space(3))) void *)gemm_lds));
    // (LDS byte address of operand tile t of buffer b) + this wave's 1 KiB piece of pass p
    auto piece = [&](int b, int t, int p) {
        return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(
            static_cast<int>(lds0 + ((b * 2 + t) * kDmaTile + (p * 32 + wave * 8) * kGemmBK) * 4)));
    };
    auto dma_tile = [&](int kt) {
        // BF16: the query tiles of one row tile (same XCD, resident together) walk K from staggered starting steps, so
        // that a K slice of the rows is a miss for one of them and an L2 hit for the others instead of all of them
        // waiting on the same miss at every step (the accumulation order differs per query tile: the scores are a
        // filter with an order-free error bound).  The LDS buffer alternates with kt as before.
        const int kk = BF16 ? (kt + kstart) % ksteps : kt;
        const int k0 = (PROBE & 32) ? 0 : kk * kGemmBK, b = kt & 1;  // PROBE bit 5: re-fetch tile 0 (cache-hot)
        if (kt < full_steps) {
#pragma unroll
            for (int p = 0; p < kGemmPasses; p++) {
                glds16(abase + k0, aoff[p], piece(b, 0, p));
#ifdef VG_GEMM_B_NT
                if (!(PROBE & 64)) glds16_stream(bbase + k0, boff[p], piece(b, 1, p));
#else
                if (!(PROBE & 64)) glds16(bbase + k0, boff[p], piece(b, 1, p));  // PROBE bit 6: A tiles only
#endif
            }
        } else {  // ragged K edge; dim % 4 == 0, so a granule is inside or outside as a whole
            const float *zeros = reinterpret_cast<const float *>(&g_gemm_zero16);
            const bool in = k0 + dgl * 4 < dim;
#pragma unroll
            for (int p = 0; p < kGemmPasses; p++) {
                glds16(in ? abase + k0 + aoff[p] / 4 : zeros, piece(b, 0, p));
                glds16(in ? bbase + k0 + boff[p] / 4 : zeros, piece(b, 1, p));
            }
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    // operand read offsets (floats): row block base + swizzled granule of group j
    const int h = lane >> 5, f = (lane >> 1) & 7;
    int goff[4];
#pragma unroll
    for (int j = 0; j < 4; j++) goff[j] = ((2 * j + h) ^ f) * 4;
    const int a_row = (wr * 64 + (lane & 31)) * kGemmBK;
    const int b_row = (wc * 64 + (lane & 31)) * kGemmBK;

    float thr_reg, xn[2];
    gemm_epilogue_inputs<DOT, MODE>(thr_reg, xn, tid, lane, wc, q0, nq, n0, n, norms, thr, thr_stride, thr_off);
    dma_tile(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < ksteps; kt++) {
        const float *As = gemm_lds + (kt & 1) * 2 * kDmaTile, *Bs = As + kDmaTile;
        if (kt + 1 < ksteps && !(PROBE & 2)) dma_tile(kt + 1);
        float4 fa[2][2], fb[2][2];  // [parity][row block]
        if (!(PROBE & 16)) {
            fa[0][0] = *reinterpret_cast<const float4 *>(As + a_row + goff[0]);
            fa[0][1] = *reinterpret_cast<const float4 *>(As + a_row + 32 * kGemmBK + goff[0]);
            fb[0][0] = *reinterpret_cast<const float4 *>(Bs + b_row + goff[0]);
            fb[0][1] = *reinterpret_cast<const float4 *>(Bs + b_row + 32 * kGemmBK + goff[0]);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int c = j & 1, nx = c ^ 1;
            if (j < 3 && !(PROBE & 16)) {
                fa[nx][0] = *reinterpret_cast<const float4 *>(As + a_row + goff[j + 1]);
                fa[nx][1] = *reinterpret_cast<const float4 *>(As + a_row + 32 * kGemmBK + goff[j + 1]);
                fb[nx][0] = *reinterpret_cast<const float4 *>(Bs + b_row + goff[j + 1]);
                fb[nx][1] = *reinterpret_cast<const float4 *>(Bs + b_row + 32 * kGemmBK + goff[j + 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
#define VG_MFMA4(comp)                                                                                     \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][0].comp, fb[c][0].comp, acc[0][0], 0, 0, 0);    \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][0].comp, fb[c][1].comp, acc[0][1], 0, 0, 0);    \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][1].comp, fb[c][0].comp, acc[1][0], 0, 0, 0);    \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][1].comp, fb[c][1].comp, acc[1][1], 0, 0, 0);
            if constexpr (BF16) {
#pragma unroll
                for (int ai = 0; ai < 2; ai++)
#pragma unroll
                    for (int bi = 0; bi < 2; bi++)
                        acc[ai][bi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(vg_bf16x8, fa[c][ai]),
                                                                             __builtin_bit_cast(vg_bf16x8, fb[c][bi]),
                                                                             acc[ai][bi], 0, 0, 0);
            } else {
                VG_MFMA4(x)
                VG_MFMA4(y)
                VG_MFMA4(z)
                VG_MFMA4(w)
            }
#undef VG_MFMA4
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!(PROBE & 128)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of tile kt+1 have landed
        if (!(PROBE & 8)) __syncthreads();                // ... everyone's have; tile kt is free
    }
    if (PROBE & 1) {
        float t = 0.0f;
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 2; j++)
                for (int r = 0; r < 16; r++) t += acc[i][j][r];
        if (t == 123.456f) scores[0] = t;
        return;
    }
    gemm_epilogue<DOT, MODE>(acc, gemm_lds, thr_reg, xn, tid, lane, wr, wc, q0, nq, n0, n, tn, scores, out_cols,
                             counts, cand, cap, mask, mask_stride, GROUPED ? row_base : 0u, GROUPED ? mask_off : nullptr);
}

template <bool DOT, int MODE, int PROBE = 0, bool BF16 = false>
__global__ __launch_bounds__(kGemmThreads) void flat_gemm_dma_kernel(
    const float *__restrict__ queries, int64_t nq, const float *__restrict__ base, int64_t n,
    int dim, const float *__restrict__ norms, float *__restrict__ scores, int tile_stride,
    int64_t out_cols, const float *__restrict__ thr, int thr_stride, int thr_off,
    int *__restrict__ counts, uint64_t *__restrict__ cand, int cap, const uint8_t *__restrict__ mask = nullptr,
    int64_t mask_stride = 0)
{
    flat_gemm_dma_body<DOT, MODE, PROBE, BF16, false>(queries, nq, base, n, dim, norms, scores, tile_stride, out_cols, thr, thr_stride,
                                                       thr_off, counts, cand, cap, mask, mask_stride, blockIdx.x, 0u);
}

// One launch over MANY problems that share the row matrix (the partition-probed flat search, k_probe.hip: problem p = the queries
// probing partition p x that partition's rows): group g multiplies the a_cnt query rows from a_off on (of `queries`, the pairs'
// query vectors bucketed by partition) with the b_cnt rows from b_off on; scores / thresholds / counts / candidate lists are
// indexed by the PAIR (a_off + local row), the appended keys carry global row ids.  first_block[g] (a multiple of 8, so that the
// XCD-aware tile order holds inside a group) = the group's first workgroup; first_block[ngroups] = their total.
struct GemmGroup {
    int64_t a_off, b_off;
    int32_t a_cnt, b_cnt;
};
template <bool DOT, int MODE, bool BF16 = false>
__global__ __launch_bounds__(kGemmThreads) void flat_gemm_dma_grouped_kernel(
    const GemmGroup *__restrict__ groups, const int64_t *__restrict__ first_block, int ngroups,
    const float *__restrict__ queries, const float *__restrict__ base, int dim, const float *__restrict__ norms,
    float *__restrict__ scores, int tile_stride, int64_t out_cols, const float *__restrict__ thr, int thr_stride, int thr_off,
    int *__restrict__ counts, uint64_t *__restrict__ cand, int cap, const uint8_t *__restrict__ mask /* row filters (bit per
    GLOBAL row), or nullptr */, const int64_t *__restrict__ mask_off /* [pairs]: where pair i's filter starts in `mask` */)
{
    const int64_t b = blockIdx.x;
    if (b >= first_block[ngroups]) return;
    int lo = 0, hi = ngroups - 1;  // the last group whose first workgroup is <= b
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (first_block[mid] <= b)
            lo = mid;
        else
            hi = mid - 1;
    }
    const GemmGroup g = groups[lo];
    if (g.a_cnt == 0 || g.b_cnt == 0) return;
    flat_gemm_dma_body<DOT, MODE, 0, BF16, true>(
        queries + g.a_off * dim, g.a_cnt, base + g.b_off * dim, g.b_cnt, dim, norms + g.b_off,
        scores ? scores + g.a_off * out_cols : nullptr, tile_stride, out_cols, thr ? thr + g.a_off * thr_stride : nullptr, thr_stride,
        thr_off, counts ? counts + g.a_off : nullptr, cand ? cand + g.a_off * cap : nullptr, cap, mask, 0, b - first_block[lo],
        static_cast<uint32_t>(g.b_off), mask ? mask_off + g.a_off : nullptr);
}

// ---- bfloat16, more than 128 queries: a 256 x 256 tile, 8 waves, one persistent workgroup per CU ---------------------------
// The 128 x 128 tile above moves 32 KiB into LDS per K step for 16 matrix instructions per wave: at the bf16 rate (16x the fp32
// one) two resident workgroups ask the L2 -> LDS path for 64 B per clock and CU, and the kernel ran at a third of the matrix
// peak waiting for tiles (r05: 834 TFLOP/s; profiles/r06_pmc_gemm_bf16_*.csv).  Here a workgroup is 8 waves as 2 (queries) x 4
// (rows), each wave 128 x 64 = 4 x 2 accumulators of v_mfma_f32_32x32x16_bf16: 64 KiB per K step for 32 instructions per wave
// and two waves per SIMD — half the bytes per flop through the fill path, 6 instead of 8 ds_read_b128 per 8 matrix instructions.
// Tiles, swizzle and operand reads are flat_gemm_dma_body's (a K step = 128 bytes of a row; piece = 8 rows x 128 B per
// wave-instruction; slot s of row R holds granule s ^ ((R >> 1) & 7)).  What differs:
//  * one workgroup per CU (160 KiB of LDS), so nothing else hides a barrier: the K loop is rotated by one operand group — the
//    wait for the next step's tiles and the barrier sit BEFORE the last 8 matrix instructions of a step, whose operands are
//    already in registers; they and the first operand reads of the next step overlap the barrier skew;
//  * the row tiles (first touch = an HBM miss) are fetched TWO K steps ahead into a ring of three buffers (NB = 3), the query
//    tiles (L2-resident) one step ahead into two;
//  * PERSISTENT: the grid is one workgroup per CU and a workgroup walks its tiles (bt, bt + grid, ...: the same XCD, the same
//    XCD-aware order as above) as ONE stream of K steps — the fills of the next tile's first steps are in flight while the
//    current tile finishes, and its epilogue runs under them.  With a workgroup per tile every tile paid a launch, a cold first
//    fill and a drained pipe: 3.9 of 14.4 us per tile with nothing but matrix instructions in the loop (tools/ubench/gemm_bf16_probe);
//  * the epilogue is a sign test: the accumulators START at (t_q - |x_r|^2) / 2 (L2; t_q for Dot: t = the query's threshold) —
//    eight extra matrix instructions per tile on a 3-way bfloat16 split of t/2 against ones and of -|x|^2/2 — so that
//    score < t  <=>  accumulator > 0, found with 8 v_max3 per 16 scores; only a (wave, 32 x 32 block) with a hit looks at its
//    elements.  (fma + compare + branch per score cost 0.35 of 1.78 ms.)  The appended key's score is t - 2 acc (t - acc);
//  * MODE 2 only (the append): the row sample of MODE 1 is 1/64 of the work and stays on the 128 x 128 tile.
// The scores differ from the 128-tile kernel's by roundings only (the accumulation starts from another value): they are a filter
// whose error bound does not depend on the order (flat_verify_kernel's eps_extra).
// WAR / RAW on the tiles as MI355X_MICROARCH.md prescribes for LDS-DMA: a wave waits for its own pieces (counted vmcnt), then the
// barrier publishes them; a buffer is refilled only after a barrier that every wave passed with its reads retired (lgkmcnt(0)).
constexpr int kBigBM = 256, kBigBN = 256, kBigThreads = 512;
constexpr int kBigTile = kBigBM * kGemmBK;  // floats per operand tile (256 rows x 128 B = 32 KiB)
template <int NB>
constexpr size_t big_lds_bytes() { return (2 + NB) * kBigTile * sizeof(float); }

// x = p0 + p1 + p2 exactly (three bfloat16 = 24 significant bits; x finite), the three as the low halves of p0, p1, p2
__device__ __forceinline__ void big_split3(float x, uint32_t &p0, uint32_t &p1, uint32_t &p2)
{
    auto rne = [](float v) {
        const uint32_t u = __float_as_uint(v);
        return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
    };
    p0 = rne(x);
    const float r1 = x - __uint_as_float(p0 << 16);
    p1 = rne(r1);
    const float r2 = r1 - __uint_as_float(p1 << 16);
    p2 = rne(r2);
}
__device__ __forceinline__ int big_max3_i32(int a, int b, int c)
{
    int r;
    asm("v_max3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

template <bool DOT, int NB, int PROBE = 0>
__global__ __launch_bounds__(kBigThreads) void flat_gemm_bf16_big_kernel(
    const float *__restrict__ queries, int64_t nq, const float *__restrict__ base, int64_t n, int dim /* 4-byte words per row */,
    const float *__restrict__ norms, const float *__restrict__ thr, int thr_stride, int thr_off, int *__restrict__ counts,
    uint64_t *__restrict__ cand, int cap, const uint8_t *__restrict__ mask, int64_t mask_stride,
    int count_stride /* ints between two queries' counters: 32 = a 128-byte line each (see launch_gemm_t) */)
{
    static_assert(NB == 2 || NB == 3, "two or three row-tile buffers");
    extern __shared__ float gemm_lds[];
    const int mtiles = static_cast<int>((nq + kBigBM - 1) / kBigBM);
    const int ntiles = static_cast<int>((n + kBigBN - 1) / kBigBN);
    // Tile slots: slot bt = 8 jx + xcd holds (query tile tm = jx % mtiles, row tile tn = 8 (jx / mtiles) + xcd) — all query tiles
    // of a row tile on one XCD, back to back (see flat_gemm_kernel); the last 8 row tiles may have holes (tn >= ntiles).  A
    // workgroup's slots are blockIdx.x + i * gridDim.x (gridDim.x a multiple of 8: always the same XCD); a position keeps jx's
    // quotient and remainder by mtiles and steps them by the constants dq / dr — no division in the K-step loop (a 64-bit one is
    // ~120 scalar instructions, and the loop needed eight of them at every tile boundary, in front of the matrix instructions).
    const int total = mtiles * ((ntiles + 7) / 8) * 8, G = static_cast<int>(gridDim.x), xcd = static_cast<int>(blockIdx.x & 7);
    const int dq = (G >> 3) / mtiles, dr = (G >> 3) % mtiles;
    struct Pos {
        int bt, quo, rem;  // slot; (bt >> 3) / mtiles and % mtiles
    };
    auto tm_of = [&](const Pos &p) { return p.rem; };
    auto tn_of = [&](const Pos &p) { return p.quo * 8 + xcd; };
    auto step = [&](Pos &p) {  // this workgroup's next slot that holds a tile (bt >= total: none)
        do {
            p.bt += G;
            p.quo += dq;
            p.rem += dr;
            if (p.rem >= mtiles) {
                p.rem -= mtiles;
                p.quo++;
            }
        } while (p.bt < total && tn_of(p) >= ntiles);
    };
    Pos pos0;
    pos0.bt = static_cast<int>(blockIdx.x);
    pos0.quo = (pos0.bt >> 3) / mtiles;
    pos0.rem = (pos0.bt >> 3) % mtiles;
    if (pos0.bt < total && tn_of(pos0) >= ntiles) step(pos0);
    if (pos0.bt >= total) return;
    int my_tiles = 0;
    for (Pos p = pos0; p.bt < total; step(p)) my_tiles++;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int ksteps = dim / kGemmBK;  // whole K steps (the bf16 images are padded to them)
    const int S = my_tiles * ksteps;   // this workgroup's stream of K steps

    // DMA map: pass p (0..3), wave w (0..7) fill rows p*64 + w*8 .. +8 of a tile
    const int drow = wave * 8 + (lane >> 3);
    const int dgl = (lane & 7) ^ ((wave * 4 + (lane >> 4)) & 7);
    const uint32_t lds0 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(
        (__attribute__((address_space(3))) void *)gemm_lds));
    // tile buffers: A0 A1 | B0 .. B(NB-1)
    auto piece = [&](int buf, int p) {
        return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(
            static_cast<int>(lds0 + (buf * kBigTile + (p * 64 + wave * 8) * kGemmBK) * 4)));
    };
    // fetch cursors: the (tile, K step) the next query / row tile fill is for, its source and the buffer it goes to.  A piece's
    // source = a wave-uniform address (tile base + K step + the piece's first row: scalar arithmetic) + ONE per-lane offset
    // (row within the piece, swizzled granule) for all pieces; only a tile that sticks out of the matrix (a_last / b_last < 255:
    // its rows past the end re-read the last one — their results are never appended) computes a per-piece offset.
    Pos a_pos = pos0, b_pos = pos0;
    int a_k = 0, b_k = 0, ia = 0, ib = 0;
    const float *abase, *bbase;
    int a_last, b_last;  // last row of the tile that exists (255 for a whole tile)
    const uint32_t off0 = static_cast<uint32_t>((drow * dim + dgl * 4) * 4);
    auto set_a = [&](const Pos &ps) {
        const int64_t q0 = static_cast<int64_t>(tm_of(ps)) * kBigBM;
        abase = queries + q0 * dim;
        a_last = static_cast<int>(nq - q0 < kBigBM ? nq - q0 : kBigBM) - 1;
    };
    auto set_b = [&](const Pos &ps) {
        const int64_t n0 = static_cast<int64_t>(tn_of(ps)) * kBigBN;
        bbase = base + n0 * dim;
        b_last = static_cast<int>(n - n0 < kBigBN ? n - n0 : kBigBN) - 1;
    };
    auto fill = [&](const float *src /* tile base + K step */, int last, int buf) {
        if (last == kBigBM - 1) {
#pragma unroll
            for (int p = 0; p < 4; p++) glds16(src + static_cast<int64_t>(p) * 64 * dim, off0, piece(buf, p));
        } else {
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const int row = p * 64 + drow < last ? p * 64 + drow : last;
                glds16(src, static_cast<uint32_t>((row * dim + dgl * 4) * 4), piece(buf, p));
            }
        }
    };
    auto stage_a = [&]() {
        fill(abase + a_k * kGemmBK, a_last, ia);
        ia ^= 1;
        if (++a_k == ksteps) {
            a_k = 0;
            step(a_pos);
            if (a_pos.bt < total) set_a(a_pos);
        }
    };
    auto stage_b = [&]() {
        if (!(PROBE & 64)) fill(bbase + b_k * kGemmBK, b_last, 2 + ib);
        ib = ib + 1 == NB ? 0 : ib + 1;
        if (++b_k == ksteps) {
            b_k = 0;
            step(b_pos);
            if (b_pos.bt < total) set_b(b_pos);
        }
    };
    set_a(pos0);
    set_b(pos0);

    f32x16 acc[4][2];
    const int h = lane >> 5, f = (lane >> 1) & 7;
    int lpart[4];  // floats: this lane's row within a 32-row block + its swizzled granule of group j
#pragma unroll
    for (int j = 0; j < 4; j++) lpart[j] = (lane & 31) * kGemmBK + ((2 * j + h) ^ f) * 4;
    const int a_wave = wr * 128 * kGemmBK, b_wave = wc * 64 * kGemmBK;  // wave-uniform

    // Per tile: lane l keeps the thresholds of the wave's query rows l and 64 + l and the norm of its row column l, loaded a
    // matrix group before the tile starts; tile_init hands every lane those of ITS rows (32 i + (l & 31)) and columns
    // (32 j + (l & 31)) by ds_bpermute and turns them into the accumulators' starting values (see the header).
    constexpr float kThrClamp = 1e30f;  // +-Inf thresholds (no sample / queries past nq) stay finite in the matrix unit
    Pos c_pos = pos0;                   // the tile being computed
    // (every load is unconditional — indices clamped, the values of rows / queries that do not exist replaced afterwards — and a
    // tile's loads are always followed by its tile_init: a load under a condition leaves the compiler with a "maybe pending"
    // register, and it drains the whole vector-memory queue, the fills in flight included, before the next tile overwrites it)
    float tq2[2], xq1;
    auto tile_loads = [&](const Pos &ps) {
        const int64_t q0 = static_cast<int64_t>(tm_of(ps)) * kBigBM, n0 = static_cast<int64_t>(tn_of(ps)) * kBigBN;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int64_t qq = q0 + wr * 128 + i * 64 + lane;
            tq2[i] = thr[(qq < nq ? qq : nq - 1) * thr_stride + thr_off];
        }
        const int64_t nn = n0 + wc * 64 + lane;
        xq1 = norms[nn < n ? nn : n - 1];
    };
    auto tile_init = [&]() {
        const uint32_t one = 0x3F80u;  // bfloat16 1.0
        uint4 ta[4], xb[2];
        // (the tile being initialised: c_pos; queries past nq get -Inf = nothing passes, rows past n are never appended)
        const int64_t iq0 = static_cast<int64_t>(tm_of(c_pos)) * kBigBM;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            float t = __shfl(tq2[i >> 1], (i & 1) * 32 + (lane & 31));
            t = iq0 + wr * 128 + i * 32 + (lane & 31) < nq ? t : -INFINITY;
            t = t < kThrClamp ? t : kThrClamp;    // (NaN -> the clamp: everything passes, the search falls back to its exact kernel)
            t = t > -kThrClamp ? t : -kThrClamp;
            uint32_t p0, p1, p2;
            big_split3(DOT ? t : 0.5f * t, p0, p1, p2);
            ta[i] = h == 0 ? make_uint4(p0 | (p1 << 16), p2 | (one << 16), one | (one << 16), 0u) : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int j = 0; j < 2; j++) {
            uint32_t p0 = 0, p1 = 0, p2 = 0;
            if (!DOT) big_split3(-0.5f * __shfl(xq1, j * 32 + (lane & 31)), p0, p1, p2);
            xb[j] = h == 0 ? make_uint4(one | (one << 16), one | (p0 << 16), p1 | (p2 << 16), 0u) : make_uint4(0u, 0u, 0u, 0u);
        }
        const f32x16 zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(vg_bf16x8, ta[i]),
                                                                   __builtin_bit_cast(vg_bf16x8, xb[j]), zero, 0, 0, 0);
    };
    // The appends.  The elements of a finished tile whose accumulator is positive go to their queries' candidate lists — about four
    // per wave and tile (0.05 % of 1024 x 8192).  An append is a RETURNING atomic on the query's counter, and vector-memory
    // operations retire in order: the fills of the following K steps cannot be seen to have landed before the atomic is back.
    // Waited for where it is issued, every passing element was a 1.5 us round trip with the matrix pipe idle (4.9 of a tile's
    // 20 us); parked for one tile and finished behind the next scan it still sat in front of the next step's fills and took
    // longer than a step (3.3 us per tile).  So a lane parks its passing elements (query, row, accumulator) ACROSS tiles, up to
    // kPark of them, and the whole workgroup appends every kFlushEvery-th tile: the atomics and threshold loads of all parked
    // elements of a wave go out together and are waited for once.  Measured (tools/ubench/gemm_bf16_probe): the
    // kernel without any append 1.13 ms, with the scan's maxima 1.17, elements collected and dropped 1.24, all of it 1.3x.
    // Also measured and not kept: a log per wave with a scatter kernel behind (a plain store to a line that has left the L2
    // retires as slowly as an atomic: 1.59 + 0.06 ms; nontemporal 1.48 + 0.06) and a pool of registers per wave written out
    // once (v_readlane / v_writelane per element, scalar registers spilled: 1.69 + 0.06).
#ifndef VG_BIG_PARK
#define VG_BIG_PARK 4
#endif
#ifndef VG_BIG_FLUSH_EVERY
#define VG_BIG_FLUSH_EVERY 6
#endif
    constexpr int kPark = VG_BIG_PARK;
    int pend = 0;  // parked elements of this lane (0..kPark)
    uint32_t p_q[kPark], p_row[kPark];
    float p_a[kPark];
#pragma unroll
    for (int e = 0; e < kPark; e++) {
        p_q[e] = 0u;
        p_row[e] = 0u;
        p_a[e] = 0.0f;
    }
    constexpr int kFlushEvery = VG_BIG_FLUSH_EVERY;
    int flushes = 0, tiles_done = static_cast<int>(blockIdx.x >> 3) % kFlushEvery;  // (the workgroups take turns: all of them
                                                                                     // at once is 45 k atomics on 1024 counters)
    auto finish = [&](uint32_t qq, uint32_t row, float a, float t, int pos) {
        t = t < kThrClamp ? t : kThrClamp;
        const float sc = DOT ? t - a : __builtin_fmaf(-2.0f, a, t);
        if (pos < cap) cand[qq * static_cast<uint32_t>(cap) + static_cast<uint32_t>(pos)] = make_key(sc, row, false);
    };
    // every parked element of the wave: the atomics and the threshold loads of all of them go out together, ONE round trip
    auto flush_parked = [&]() {
        int pos[kPark];
        float tt[kPark];
#pragma unroll
        for (int e = 0; e < kPark; e++) {
            pos[e] = 0;
            tt[e] = 0.0f;
            if (pend > e) {
                pos[e] = (PROBE & 4096) ? e : atomicAdd(&counts[p_q[e] * static_cast<uint32_t>(count_stride)], 1);  // (stage probe 4096: no atomics, slot e of every list)
                tt[e] = thr[p_q[e] * static_cast<uint32_t>(thr_stride) + static_cast<uint32_t>(thr_off)];
            }
        }
        // (ONE wait, and one the compiler's bookkeeping sees: left to place its own it waits in front of every store for results
        // that "may be pending" under the conditions above — and so for the store before it, four write round trips in a row)
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
#pragma unroll
        for (int e = 0; e < kPark; e++)
            if (pend > e) finish(p_q[e], p_row[e], p_a[e], tt[e], pos[e]);
        pend = 0;
        flushes++;
    };
    auto tile_append = [&](const Pos &ps) {
        const int64_t q0 = static_cast<int64_t>(tm_of(ps)) * kBigBM, n0 = static_cast<int64_t>(tn_of(ps)) * kBigBN;
        const int n_in = static_cast<int>(n - n0 < kBigBN ? n - n0 : kBigBN);
        // (everything below that depends only on the lane is derived from these two INSIDE the K-step loop: left to itself the
        // compiler computes the 64 query indices of a lane once, ahead of the loop, and keeps them in 64 registers)
        int hv = 4 * h, lv = lane & 31;
        asm volatile("" : "+v"(hv), "+v"(lv));
#pragma unroll
        for (int i = 0; i < 4; i++) {
#pragma unroll
            for (int j = 0; j < 2; j++) {
                // the largest of a lane's 16 elements, through the largest of each group of four (registers 4g .. 4g+3 = four
                // consecutive queries): 10 instructions; a block in which some lane has a positive one (two in five) then looks at
                // the four groups, and at the elements of a group only where a lane's group maximum is positive
                // (a positive float is a positive int; a NaN with a clear sign bit gets here too and fails the float test below)
                int gm[4];
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int m3 = big_max3_i32(__float_as_int(acc[i][j][4 * g]), __float_as_int(acc[i][j][4 * g + 1]),
                                                __float_as_int(acc[i][j][4 * g + 2]));
                    gm[g] = m3 > __float_as_int(acc[i][j][4 * g + 3]) ? m3 : __float_as_int(acc[i][j][4 * g + 3]);
                }
                int mx = big_max3_i32(gm[0], gm[1], gm[2]);
                mx = mx > gm[3] ? mx : gm[3];
                if (__builtin_amdgcn_ballot_w64(mx > 0) == 0) continue;  // wave-uniform: nothing of this 32 x 32 block passes
                if (PROBE & 256) {  // (stage probe: the maxima only, kept alive)
                    if (mx == 0x12345678) counts[0] = 1;
                    continue;
                }
                // the few passing elements: 32-bit offsets from the (wave-uniform) array bases (nq * cap < 2^31: the launcher checks)
                const int nl = wc * 64 + j * 32 + lv;  // row index inside the tile
                if (nl >= n_in) continue;              // (per lane: a row past the end of the matrix)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    if (gm[g] <= 0) continue;
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int r = 4 * g + e;
                        const float a = acc[i][j][r];
                        if (a > 0.0f) {
                            const uint32_t qq = static_cast<uint32_t>(q0) + static_cast<uint32_t>(wr * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + hv);
                            const int64_t nn = n0 + nl;
                            // Queries past nq started at -1e30 and get here only when a row's dot product is +Inf (an Inf or a 1e38
                            // in the row times the re-read last query): they have no list — r06's first version appended to
                            // counts[qq] / cand[qq * cap] beyond the arrays (a memory fault on Dot segments with such rows, found
                            // by tools/fuzz_nonfinite.py).  (The filter bit is looked at only for the few elements that pass.)
                            if (static_cast<int64_t>(qq) < nq && (mask == nullptr || mask_bit(mask + static_cast<int64_t>(qq) * mask_stride, nn))) {
                                if (pend < kPark) {
#pragma unroll
                                    for (int s2 = 0; s2 < kPark; s2++) {
                                        const bool here = pend == s2;
                                        p_q[s2] = here ? qq : p_q[s2];
                                        p_row[s2] = here ? static_cast<uint32_t>(nn) : p_row[s2];
                                        p_a[s2] = here ? a : p_a[s2];
                                    }
                                    pend++;
                                } else {
                                    finish(qq, static_cast<uint32_t>(nn), a, thr[qq * static_cast<uint32_t>(thr_stride) + static_cast<uint32_t>(thr_off)],
                                           atomicAdd(&counts[qq * static_cast<uint32_t>(count_stride)], 1));
                                }
                            }
                        }
                    }
                }
            }
        }
        if (PROBE & 2048) pend = 0;  // (stage probe: the passing elements are collected and dropped)
        // Every kFlushEvery-th tile, ALL waves of the workgroup append what they have parked: one wave waiting for its atomics
        // holds the other seven at the next barrier, so flushes decided wave by wave (some lane nearly full: 9 per wave, 72 per
        // workgroup and 61 tiles) stalled the workgroup at nearly every tile.  (A lane that fills its kPark slots in between
        // takes the immediate path.)
        if (++tiles_done % kFlushEvery == 0) flush_parked();
    };

    float4 fa[2][4], fb[2][2];  // [set][block]: operand granules of one group (16 k) — the group being multiplied / the next one
    auto read_group = [&](const float *As, const float *Bs, int j, int set) {
        const float *pa = As + a_wave + lpart[j], *pb = Bs + b_wave + lpart[j];
#pragma unroll
        for (int i = 0; i < 4; i++) fa[set][i] = *reinterpret_cast<const float4 *>(pa + i * 32 * kGemmBK);
#pragma unroll
        for (int i = 0; i < 2; i++) fb[set][i] = *reinterpret_cast<const float4 *>(pb + i * 32 * kGemmBK);
    };
    auto mfma_group = [&](int set) {
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(vg_bf16x8, fa[set][i]),
                                                                   __builtin_bit_cast(vg_bf16x8, fb[set][j]), acc[i][j], 0, 0, 0);
    };

    // prologue: the first step's tiles (NB = 3: and the second step's row tile) in flight, the first tile's starting values
    // computed under them, then published
    tile_loads(pos0);
    stage_a();
    stage_b();
    if (NB == 3 && S > 1) stage_b();
    tile_init();
    if (NB == 3 && S > 1)
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int ra = 0, rb = 0;  // buffers the current step is read from
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): no scalar load pending at the loop's entry (see the wait inside it)
    read_group(gemm_lds, gemm_lds + 2 * kBigTile, 0, 0);
    // Step g: the four operand groups of the step's tiles, group j+1 read from LDS while group j is multiplied; the fills of the
    // NEXT step's tiles — query tile g+1, row tile g+NB-1, into the buffers the barrier at the end of step g-1 freed — go out
    // behind the first two groups' matrix instructions (an LDS-DMA piece costs the issuing wave ~60 cycles: eight of them right
    // behind the barrier would leave the matrix pipe empty there), queries first: the wait below counts on that order.
    // The body is the same straight line for every step — the last one also passes the barrier and reads "the next step's" first
    // group (stale LDS, never multiplied): with a branch around that block the compiler merges two LDS-counter states in front
    // of the last matrix group and waits lgkmcnt(0) there, i.e. for the reads just issued — the overlap the rotation exists for.
    int t = 0;  // the K step of the tile being computed (c_pos)
    for (int g = 0; g < S; g++) {
        const float *As = gemm_lds + ra * kBigTile, *Bs = gemm_lds + (2 + rb) * kBigTile;
        const bool last = t == ksteps - 1;
        Pos n_pos = c_pos;
        if (last) {
            // the next tile's thresholds and norms, a whole K step before tile_init wants them and OLDER in the vector-memory
            // queue than this step's fills: the wait at the end of the step covers them (the compiler's own wait in front of
            // tile_init counts only what it can see and so asks for the fills issued behind them as well: by then a step old)
            step(n_pos);
            if (n_pos.bt >= total) n_pos = c_pos;  // (after the last tile: its own values once more, 8 idle matrix instructions)
            tile_loads(n_pos);
        }
        if (!(PROBE & 16)) read_group(As, Bs, 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(0);
        __builtin_amdgcn_sched_barrier(0);
        if (g + 1 < S && !(PROBE & 2)) stage_a();
        if (g + NB - 1 < S && !(PROBE & 2) && (PROBE & 512)) stage_b();  // (stage probe: the row tile's fill one group earlier)
        if (!(PROBE & 16)) read_group(As, Bs, 2, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(1);
        __builtin_amdgcn_sched_barrier(0);
        if (g + NB - 1 < S && !(PROBE & 2) && !(PROBE & 512)) stage_b();
        if (!(PROBE & 16)) read_group(As, Bs, 3, 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(0);
        __builtin_amdgcn_sched_barrier(0);
        ra ^= 1;
        rb = rb + 1 == NB ? 0 : rb + 1;
        // lgkmcnt(0): this wave's reads of step g are retired (its buffers are refilled in the next step).  It is the builtin, which
        // the compiler's own counter bookkeeping sees: after it, it knows that no LDS read — and no scalar load, which would force
        // every later LDS wait to lgkmcnt(0) — is pending.  The DMA wait stays asm (the compiler knows nothing of the fills):
        // step g+1's tiles = everything this wave has in flight but (NB = 3) the row tile g+2 issued after query tile g+1.
        __builtin_amdgcn_s_waitcnt(0xC07F);
        if (NB == 3 && g + 2 < S)
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!(PROBE & 8)) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (!(PROBE & 16)) read_group(gemm_lds + ra * kBigTile, gemm_lds + (2 + rb) * kBigTile, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(1);  // group 3 of step g, from registers
        __builtin_amdgcn_sched_barrier(0);
        if (last) {
            if (!(PROBE & 1)) {
                tile_append(c_pos);
            } else {  // stage probe: keep the accumulators alive without an epilogue
                float sum = 0.0f;
                for (int i = 0; i < 4; i++)
                    for (int j = 0; j < 2; j++)
                        for (int r = 0; r < 16; r++) sum += acc[i][j][r];
                if (sum == 123.456f) counts[0] = 1;
            }
            c_pos = n_pos;
            t = 0;
            if (PROBE & 1024) {  // (stage probe: the accumulators start at zero — no thresholds, no norms, no extra matrix group)
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 2; j++)
#pragma unroll
                        for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;
            } else {
                tile_init();
            }
        } else {
            t++;
        }
    }
    flush_parked();
    if ((PROBE & 32768) && lane == 0) atomicAdd(&counts[nq * count_stride], flushes);  // (stage probe: how often a wave flushed, counts[nq])
}

// ---- 5..64 queries: the same pipeline with a 32 x 128 or 64 x 128 tile ------------------------------
// A 128-query tile does 128 queries' worth of MFMA work whatever it holds (1.65 ms per pass over
// 1M x 768 rows).  With RB = 1 or 2 blocks of 32 query rows the MFMA time drops 4x / 2x and the pass
// becomes HBM-bound like the small-batch scan: 4 waves as 1 x 4, each RB accumulators of 32 x 32; per
// K step a workgroup moves 16 KiB of rows and RB * 4 KiB of queries by LDS-DMA for RB * 16 MFMAs per
// wave.  LDS per workgroup: 2 x (RB * 4 + 16) KiB, so three to four workgroups share a CU.
// nq <= RB * 32 only (one query tile; the grid covers the row tiles).  MODE 1 / 2 as above.
constexpr int kG32BM = 32;
#ifndef VG_PROBE_RB
#define VG_PROBE_RB 4
#endif
constexpr int kProbeRB = VG_PROBE_RB;  // blocks of 32 query rows in a tile of the grouped (partition-probed) form: see probe_bucket_scan_kernel
template <int RB>
constexpr size_t g32_lds_bytes() { return 2 * (RB * kG32BM * kGemmBK + kDmaTile) * sizeof(float); }

// BF16: as in flat_gemm_dma_kernel (rows of bfloat16, `dim` in 4-byte words) — the pass is HBM-bound, so half the bytes
// per row is half the time.
// (body / kernel / grouped kernel as for flat_gemm_dma_kernel: tn = the workgroup's row tile within its problem, row_base as there)
template <bool DOT, int MODE, int RB, bool BF16, bool GROUPED>
__device__ __forceinline__ void flat_gemm_dma32_body(
    const float *__restrict__ queries, int64_t nq, const float *__restrict__ base, int64_t n,
    int dim, const float *__restrict__ norms, float *__restrict__ scores, int tile_stride,
    int64_t out_cols, const float *__restrict__ thr, int thr_stride, int thr_off,
    int *__restrict__ counts, uint64_t *__restrict__ cand, int cap, const uint8_t *__restrict__ mask,
    int64_t mask_stride, int64_t tn, uint32_t row_base, const int64_t *__restrict__ mask_off = nullptr)
{
    auto filter_of = [&](int ql) { return GROUPED && mask_off ? mask + mask_off[ql] : mask + ql * mask_stride; };
    extern __shared__ float gemm_lds[];
    constexpr int kATile = RB * kG32BM * kGemmBK;  // floats
    const int64_t ntiles = MODE == 1 ? (((n + kGemmBN - 1) / kGemmBN) + tile_stride - 1) / tile_stride
                                     : (n + kGemmBN - 1) / kGemmBN;
    if (tn >= ntiles) return;
    const int64_t n0 = (MODE == 1 ? tn * tile_stride : tn) * kGemmBN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // DMA map of the B tile as in flat_gemm_dma_kernel (pass p, wave w: rows p*32 + w*8 .. +8); the
    // A tile is RB passes of the same shape over the query rows
    const int drow = wave * 8 + (lane >> 3);
    const int dgl = (lane & 7) ^ ((wave * 4 + (lane >> 4)) & 7);
    const float *const bbase = base + n0 * dim;
    uint32_t boff[kGemmPasses], aoff[RB];
#pragma unroll
    for (int p = 0; p < kGemmPasses; p++) {
        int64_t nb = n0 + p * 32 + drow;
        if (nb >= n) nb = n - 1;
        boff[p] = static_cast<uint32_t>(((nb - n0) * dim + dgl * 4) * 4);
    }
#pragma unroll
    for (int p = 0; p < RB; p++) {
        int64_t qa = p * 32 + drow;
        if (qa >= nq) qa = nq - 1;
        aoff[p] = static_cast<uint32_t>((qa * dim + dgl * 4) * 4);
    }
    const int ksteps = (dim + kGemmBK - 1) / kGemmBK;
    const int full_steps = dim / kGemmBK;
    const uint32_t lds0 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(
        (__attribute__((address_space(3))) void *)gemm_lds));
    constexpr int kBuf = kATile + kDmaTile;  // floats per buffer: A (RB * 32 rows) then B (128 rows)
    auto a_piece = [&](int b, int p) {
        return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(
            static_cast<int>(lds0 + (b * kBuf + (p * 32 + wave * 8) * kGemmBK) * 4)));
    };
    auto b_piece = [&](int b, int p) {
        return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(
            static_cast<int>(lds0 + (b * kBuf + kATile + (p * 32 + wave * 8) * kGemmBK) * 4)));
    };
    auto dma_tile = [&](int kt) {
        const int k0 = kt * kGemmBK, b = kt & 1;
        if (kt < full_steps) {
#pragma unroll
            for (int p = 0; p < RB; p++) glds16(queries + k0, aoff[p], a_piece(b, p));
#pragma unroll
            for (int p = 0; p < kGemmPasses; p++) glds16_stream(bbase + k0, boff[p], b_piece(b, p));
        } else {  // ragged K edge; dim % 4 == 0, so a granule is inside or outside as a whole
            const float *zeros = reinterpret_cast<const float *>(&g_gemm_zero16);
            const bool in = k0 + dgl * 4 < dim;
#pragma unroll
            for (int p = 0; p < RB; p++) glds16(in ? queries + k0 + aoff[p] / 4 : zeros, a_piece(b, p));
#pragma unroll
            for (int p = 0; p < kGemmPasses; p++) glds16(in ? bbase + k0 + boff[p] / 4 : zeros, b_piece(b, p));
        }
    };
    f32x16 acc[RB];
#pragma unroll
    for (int i = 0; i < RB; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][r] = 0.0f;
    const int h = lane >> 5, f = (lane >> 1) & 7;
    int goff[4];
#pragma unroll
    for (int j = 0; j < 4; j++) goff[j] = ((2 * j + h) ^ f) * 4;
    const int a_row = (lane & 31) * kGemmBK;
    const int b_row = (wave * 32 + (lane & 31)) * kGemmBK;

    // epilogue inputs before the K loop: thread t < RB*32 holds the threshold of query t, every lane
    // the norm of its column
    float thr_reg = -INFINITY;
    if (MODE == 2 && tid < RB * kG32BM && tid < nq) thr_reg = thr[static_cast<int64_t>(tid) * thr_stride + thr_off];
    const int col = wave * 32 + (lane & 31);
    const int64_t nn = n0 + col;
    const float xn = (!DOT && nn < n) ? norms[nn] : 0.0f;

    const int used = GROUPED ? static_cast<int>((nq + kG32BM - 1) / kG32BM) : RB;  // blocks of 32 query rows that hold a query
    dma_tile(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < ksteps; kt++) {
        const float *As = gemm_lds + (kt & 1) * kBuf, *Bs = As + kATile;
        if (kt + 1 < ksteps) dma_tile(kt + 1);
        float4 fa[2][RB], fb[2];  // [parity]
#pragma unroll
        for (int i = 0; i < RB; i++) fa[0][i] = *reinterpret_cast<const float4 *>(As + i * 32 * kGemmBK + a_row + goff[0]);
        fb[0] = *reinterpret_cast<const float4 *>(Bs + b_row + goff[0]);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int c = j & 1, nx = c ^ 1;
            if (j < 3) {
#pragma unroll
                for (int i = 0; i < RB; i++)
                    fa[nx][i] = *reinterpret_cast<const float4 *>(As + i * 32 * kGemmBK + a_row + goff[j + 1]);
                fb[nx] = *reinterpret_cast<const float4 *>(Bs + b_row + goff[j + 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            // (a block of 32 query rows that holds no query — the grouped form's last tile of a group is half empty on average — is
            // not multiplied: `used` is uniform over the workgroup)
            if constexpr (BF16) {
#pragma unroll
                for (int i = 0; i < RB; i++)
                    if (i < used)
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(vg_bf16x8, fa[c][i]),
                                                                        __builtin_bit_cast(vg_bf16x8, fb[c]), acc[i], 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < RB; i++)
                    if (i < used) {
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][i].x, fb[c].x, acc[i], 0, 0, 0);
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][i].y, fb[c].y, acc[i], 0, 0, 0);
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][i].z, fb[c].z, acc[i], 0, 0, 0);
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][i].w, fb[c].w, acc[i], 0, 0, 0);
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    float4 t4[RB][4];  // thresholds of rows i*32 + 8*g + 4*(lane>>5) + 0..3
    if (MODE == 2) {
        if (tid < RB * kG32BM) gemm_lds[tid] = thr_reg;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < RB; i++)
#pragma unroll
            for (int g = 0; g < 4; g++)
                t4[i][g] = *reinterpret_cast<const float4 *>(gemm_lds + i * 32 + 8 * g + 4 * (lane >> 5));
    }
#pragma unroll
    for (int i = 0; i < RB; i++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int ql = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const float dotv = acc[i][r];
            const float sc = DOT ? -dotv : __builtin_fmaf(-2.0f, dotv, xn);
            if (MODE == 1) {
                if (ql < nq)
                    scores[static_cast<int64_t>(ql) * out_cols + tn * kGemmBN + col] =
                        nn < n && (mask == nullptr || mask_bit(filter_of(ql), nn + (GROUPED ? row_base : 0u))) ? sc : INFINITY;
            } else {
                const float4 tv = t4[i][r >> 2];
                const float t = (r & 3) == 0 ? tv.x : (r & 3) == 1 ? tv.y : (r & 3) == 2 ? tv.z : tv.w;
                if (nn < n && sc < t && (mask == nullptr || mask_bit(filter_of(ql), nn + (GROUPED ? row_base : 0u)))) {
                    const int pos = atomicAdd(&counts[ql], 1);
                    if (pos < cap)
                        cand[static_cast<int64_t>(ql) * cap + pos] = make_key(sc, static_cast<uint32_t>(nn) + (GROUPED ? row_base : 0u), false);
                }
            }
        }
    }
}

template <bool DOT, int MODE, int RB, bool BF16 = false>
__global__ __launch_bounds__(kGemmThreads) void flat_gemm_dma32_kernel(
    const float *__restrict__ queries, int64_t nq, const float *__restrict__ base, int64_t n,
    int dim, const float *__restrict__ norms, float *__restrict__ scores, int tile_stride,
    int64_t out_cols, const float *__restrict__ thr, int thr_stride, int thr_off,
    int *__restrict__ counts, uint64_t *__restrict__ cand, int cap, const uint8_t *__restrict__ mask = nullptr,
    int64_t mask_stride = 0)
{
    flat_gemm_dma32_body<DOT, MODE, RB, BF16, false>(queries, nq, base, n, dim, norms, scores, tile_stride, out_cols, thr, thr_stride,
                                                      thr_off, counts, cand, cap, mask, mask_stride, blockIdx.x, 0u);
}

// the grouped form: a group's query rows in tiles of RB * 32 (one workgroup per query tile and row tile; first_block as above)
template <bool DOT, int MODE, int RB, bool BF16 = false>
__global__ __launch_bounds__(kGemmThreads) void flat_gemm_dma32_grouped_kernel(
    const GemmGroup *__restrict__ groups, const int64_t *__restrict__ first_block, int ngroups,
    const float *__restrict__ queries, const float *__restrict__ base, int dim, const float *__restrict__ norms,
    float *__restrict__ scores, int tile_stride, int64_t out_cols, const float *__restrict__ thr, int thr_stride, int thr_off,
    int *__restrict__ counts, uint64_t *__restrict__ cand, int cap, const uint8_t *__restrict__ mask,
    const int64_t *__restrict__ mask_off)
{
    const int64_t b = blockIdx.x;
    if (b >= first_block[ngroups]) return;
    int lo = 0, hi = ngroups - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (first_block[mid] <= b)
            lo = mid;
        else
            hi = mid - 1;
    }
    const GemmGroup g = groups[lo];
    if (g.a_cnt == 0 || g.b_cnt == 0) return;
    // a group's workgroups: query tile by query tile (RB * 32 query rows each), a tile's row tiles (MODE 1: its sampled ones) in order
    const int64_t rel = b - first_block[lo];
    const int64_t nt = (static_cast<int64_t>(g.b_cnt) + kGemmBN - 1) / kGemmBN;
    const int64_t per = MODE == 1 ? (nt + tile_stride - 1) / tile_stride : nt;
    const int64_t qt = rel / per, tn = rel - qt * per;
    const int64_t a_off = g.a_off + qt * (RB * kG32BM);
    const int64_t a_cnt = std::min<int64_t>(RB * kG32BM, g.a_cnt - qt * (RB * kG32BM));
    if (a_cnt <= 0) return;
    flat_gemm_dma32_body<DOT, MODE, RB, BF16, true>(
        queries + a_off * dim, a_cnt, base + g.b_off * dim, g.b_cnt, dim, norms + g.b_off,
        scores ? scores + a_off * out_cols : nullptr, tile_stride, out_cols, thr ? thr + a_off * thr_stride : nullptr, thr_stride,
        thr_off, counts ? counts + a_off : nullptr, cand ? cand + a_off * cap : nullptr, cap, mask, 0, tn,
        static_cast<uint32_t>(g.b_off), mask ? mask_off + a_off : nullptr);
}

}  // namespace vg
