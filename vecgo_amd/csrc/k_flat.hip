// k_flat.hip — exact brute-force search (flat/segment.go:691-721, hnsw.go:2021-2101):
//   1. scores[q][n] = ||x_n||^2 - 2 q.x_n  (L2)  or  -q.x_n  (Dot)   fp32 MFMA GEMM
//   2. per query: the kc smallest scores                              streaming select
//   3. exact re-scoring of those kc rows in the reference's summation order, top-k by
//      (Score, RowID), and a proof that no other row can belong to the top-k (k > 48: every row whose
//      GEMM score is under the query's threshold is re-scored, the proof is against the threshold)
//   4. queries whose proof fails are recomputed by the exhaustive exact kernel
// Steps 1-2 only nominate candidates; every reported id/score comes from step 3/4.
#include <algorithm>

#include "vg_device.hpp"
#include "vg_exact.hpp"
#include "vg_cand_replay.hpp"
#include "vg_flat_gemm.hpp"
#include "vg_internal.hpp"

namespace vg {

int32_t launch_topk_merge(const uint64_t *partial, int64_t nq, int lists, int k, bool descending,
                          uint32_t *ids, float *scores, hipStream_t st, const int *only_if = nullptr,
                          const int *always = nullptr);

// One GEMM launch.  `dma` picks the LDS-DMA kernel (16-byte aligned operands, dim % 4 == 0);
// both kernels need more dynamic LDS than the 64 KiB a kernel gets by default.
struct GemmArgs {
    const float *queries;
    int64_t nq;
    const float *base;
    int64_t n;
    int dim;
    const float *norms;
    float *scores;
    int tile_stride;
    int64_t out_cols;
    const float *thr;
    int thr_stride, thr_off;
    int *counts;
    uint64_t *cand;
    int cap;
    const uint8_t *mask = nullptr;  // row filter of a filtered search (k_probe.hip): bit per (query, row), or one per row
    int64_t mask_stride = 0;
    int cus = 256;  // compute units of the device (the persistent bf16 tile launches one workgroup per CU)
    // nq * kCountLine ints of scratch: the persistent bf16 tile's counters, one 128-byte line per query.  An append is a returning
    // atomic on its query's counter; 1024 counters in 32 lines are 16 k atomics per line and launch, and a line serves them one
    // after the other (1024 queries x 1M x 768: 0.16 of the kernel's 1.35 ms went there — "keys stored without atomics" in
    // profiles/r06_gemm_bf16_probe.txt).  Null: the tile counts in `counts` itself.
    int *counts_wide = nullptr;
};
constexpr int kCountLine = 32;
__global__ void counts_narrow_kernel(const int *__restrict__ wide, int64_t nq, int *__restrict__ counts)
{
    const int64_t q = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (q < nq) counts[q] = wide[q * kCountLine];
}

template <bool DOT, int MODE>
static int32_t launch_gemm_t(bool dma, unsigned blocks, hipStream_t st, const GemmArgs &a, bool bf16)
{
    if (bf16) {  // rows of bfloat16 (a.dim = 4-byte words per row): the LDS-DMA tiles only
        constexpr int M = MODE == 0 ? 1 : MODE;
        // up to 128 queries: the tile of 1 .. 4 blocks of 32 query rows (HBM-bound at the bf16 rate: 65 .. 128 queries 0.44 .. 0.49 ms on
        // the 128 x 128 tile, 0.40 .. 0.45 on this one — tools/mid_batch_time.py)
        if (a.nq <= 4 * kG32BM && !hook(kHookFlatNoSmallTile)) {
            const bool one = a.nq <= kG32BM;
            const int rb = static_cast<int>((a.nq + kG32BM - 1) / kG32BM);
            auto kern = one ? flat_gemm_dma32_kernel<DOT, M, 1, true> : rb == 2 ? flat_gemm_dma32_kernel<DOT, M, 2, true> : rb == 3 ? flat_gemm_dma32_kernel<DOT, M, 3, true> : flat_gemm_dma32_kernel<DOT, M, 4, true>;
            const size_t lds = one ? g32_lds_bytes<1>() : rb == 2 ? g32_lds_bytes<2>() : rb == 3 ? g32_lds_bytes<3>() : g32_lds_bytes<4>();
            VG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lds)));
            const int64_t tiles = (a.n + kGemmBN - 1) / kGemmBN;
            const int64_t grid = MODE == 1 ? (tiles + a.tile_stride - 1) / a.tile_stride : tiles;
            VG_LAUNCH(kern, dim3(static_cast<unsigned>(grid)), dim3(kGemmThreads), lds, st, a.queries, a.nq, a.base,
                      a.n, a.dim, a.norms, a.scores, a.tile_stride, a.out_cols, a.thr, a.thr_stride, a.thr_off, a.counts,
                      a.cand, a.cap, a.mask, a.mask_stride);
            return VG_OK;
        }
        // more than one 128-query tile: the persistent 256 x 256 tile (32-bit candidate offsets: nq * cap < 2^31)
        if (MODE == 2 && a.nq > kGemmBM && a.nq * static_cast<int64_t>(a.cap) < (int64_t(1) << 31) && a.thr_stride < 65536 &&
            !hook(kHookFlatNoBigTile)) {
            const bool two = hook(kHookFlatBigTile2);
            auto kern = two ? flat_gemm_bf16_big_kernel<DOT, 2> : hook(kHookFlatBigEarlyB) ? flat_gemm_bf16_big_kernel<DOT, 3, 512> : flat_gemm_bf16_big_kernel<DOT, 3>;
            const size_t lds = two ? big_lds_bytes<2>() : big_lds_bytes<3>();
            VG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lds)));
            const int64_t mt = (a.nq + kBigBM - 1) / kBigBM, nt = (a.n + kBigBN - 1) / kBigBN;
            const int64_t slots = mt * ((nt + 7) / 8) * 8, per_cu = std::max(a.cus / 8, 1) * 8;  // (one workgroup per CU: LDS)
            if (a.counts_wide) VG_HIP(hipMemsetAsync(a.counts_wide, 0, sizeof(int) * static_cast<size_t>(a.nq) * kCountLine, st));
            VG_LAUNCH(kern, dim3(static_cast<unsigned>(std::min(slots, per_cu))), dim3(kBigThreads), lds, st, a.queries, a.nq,
                      a.base, a.n, a.dim, a.norms, a.thr, a.thr_stride, a.thr_off, a.counts_wide ? a.counts_wide : a.counts, a.cand, a.cap, a.mask,
                      a.mask_stride, a.counts_wide ? kCountLine : 1);
            if (a.counts_wide)
                VG_LAUNCH(counts_narrow_kernel, dim3(static_cast<unsigned>((a.nq + 255) / 256)), dim3(256), 0, st, a.counts_wide, a.nq, a.counts);
            return VG_OK;
        }
        auto kern = flat_gemm_dma_kernel<DOT, M, 0, true>;
        VG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   static_cast<int>(kDmaLdsBytes)));
        VG_LAUNCH(kern, dim3(blocks), dim3(kGemmThreads), kDmaLdsBytes, st, a.queries, a.nq, a.base, a.n, a.dim,
                  a.norms, a.scores, a.tile_stride, a.out_cols, a.thr, a.thr_stride, a.thr_off, a.counts, a.cand,
                  a.cap, a.mask, a.mask_stride);
        return VG_OK;
    }
    // (test hook kHookFlatNoSmallTile: always the 128-query tile)
    // 1 - 3 blocks of 32 queries: the HBM-bound shapes, and 65 .. 96 queries at three quarters of the 128-query tile's matrix work
    // (1.65 -> 1.28 ms per call at 1M x 768; 97 .. 128 queries: four blocks run as long as the 128 x 128 tile, which keeps them)
    if (dma && MODE != 0 && a.nq <= 3 * kG32BM && !hook(kHookFlatNoSmallTile)) {
        constexpr int M = MODE == 0 ? 1 : MODE;
        const bool one = a.nq <= kG32BM;
        const int rb = static_cast<int>((a.nq + kG32BM - 1) / kG32BM);
        auto kern = one ? flat_gemm_dma32_kernel<DOT, M, 1> : rb == 2 ? flat_gemm_dma32_kernel<DOT, M, 2> : flat_gemm_dma32_kernel<DOT, M, 3>;
        const size_t lds = one ? g32_lds_bytes<1>() : rb == 2 ? g32_lds_bytes<2>() : g32_lds_bytes<3>();
        VG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   static_cast<int>(lds)));
        const int64_t tiles = (a.n + kGemmBN - 1) / kGemmBN;
        const int64_t grid = MODE == 1 ? (tiles + a.tile_stride - 1) / a.tile_stride : tiles;
        VG_LAUNCH(kern, dim3(static_cast<unsigned>(grid)), dim3(kGemmThreads), lds, st, a.queries, a.nq, a.base,
                  a.n, a.dim, a.norms, a.scores, a.tile_stride, a.out_cols, a.thr, a.thr_stride, a.thr_off, a.counts,
                  a.cand, a.cap, a.mask, a.mask_stride);
    } else if (dma) {
        auto kern = flat_gemm_dma_kernel<DOT, MODE>;
        VG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   static_cast<int>(kDmaLdsBytes)));
        VG_LAUNCH(kern, dim3(blocks), dim3(kGemmThreads), kDmaLdsBytes, st, a.queries, a.nq, a.base, a.n, a.dim,
                  a.norms, a.scores, a.tile_stride, a.out_cols, a.thr, a.thr_stride, a.thr_off, a.counts, a.cand,
                  a.cap, a.mask, a.mask_stride);
    } else {
        auto kern = flat_gemm_kernel<DOT, MODE>;
        VG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   static_cast<int>(kGemmLdsBytes)));
        VG_LAUNCH(kern, dim3(blocks), dim3(kGemmThreads), kGemmLdsBytes, st, a.queries, a.nq, a.base, a.n, a.dim,
                  a.norms, a.scores, a.tile_stride, a.out_cols, a.thr, a.thr_stride, a.thr_off, a.counts, a.cand,
                  a.cap, a.mask, a.mask_stride);
    }
    return VG_OK;
}

template <int MODE>
static int32_t launch_gemm(bool dot, bool dma, unsigned blocks, hipStream_t st, const GemmArgs &a, bool bf16 = false)
{
    return dot ? launch_gemm_t<true, MODE>(dma, blocks, st, a, bf16) : launch_gemm_t<false, MODE>(dma, blocks, st, a, bf16);
}

// fp32 -> bfloat16, round to nearest even (NaN stays NaN): rows of `dim` floats written as rows of `dim_pad` bfloat16, the tail
// zeros (operands of the bf16 GEMM are padded to whole K steps)
__global__ void f32_to_bf16_pad_kernel(const float *__restrict__ src, int64_t rows, int dim, int dim_pad, uint16_t *__restrict__ dst)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= rows * dim_pad) return;
    const int64_t r = i / dim_pad;
    const int j = static_cast<int>(i - r * dim_pad);
    const uint32_t u = j < dim ? __float_as_uint(src[r * dim + j]) : 0u;
    uint16_t v;
    if ((u & 0x7F800000u) == 0x7F800000u && (u & 0x007FFFFFu) != 0)
        v = static_cast<uint16_t>((u >> 16) | 0x0040u);
    else
        v = static_cast<uint16_t>((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
    dst[i] = v;
}

// After the proofs: the work list of step 4 (todo[0] = how many queries need the exhaustive scan,
// todo[1 + j] = the j-th of them, ascending) and the bookkeeping for vg_index_flat_stats
// (stats[0] += queries of the chunk, stats[1] += those sent to the exhaustive kernel).
__global__ __launch_bounds__(256) void flat_todo_kernel(const int *__restrict__ flags, const int *__restrict__ always,
                                                        int cnt, int *__restrict__ todo,
                                                        unsigned long long *__restrict__ stats)
{
    __shared__ int part[256];
    const int tid = threadIdx.x;
    const int per = (cnt + 255) / 256;
    const int lo = tid * per < cnt ? tid * per : cnt, hi = lo + per < cnt ? lo + per : cnt;
    if (!flags) {  // a path with no proof step: only count the queries
        if (tid == 0) atomicAdd(&stats[0], static_cast<unsigned long long>(cnt));
        return;
    }
    const bool all = *always != 0;
    int mine = 0;
    for (int i = lo; i < hi; i++) mine += (all || flags[i] != 0);
    part[tid] = mine;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int t = 0; t < 256; t++) {
            const int c = part[t];
            part[t] = run;
            run += c;
        }
        todo[0] = run;
        atomicAdd(&stats[0], static_cast<unsigned long long>(cnt));
        atomicAdd(&stats[1], static_cast<unsigned long long>(run));
    }
    __syncthreads();
    int at = part[tid];
    for (int i = lo; i < hi; i++)
        if (all || flags[i] != 0) todo[1 + at++] = i;
}

// per query: the kc best keys among the candidates the fused GEMM appended (count <= cap)
__global__ __launch_bounds__(256) void flat_pick_kernel(const uint64_t *__restrict__ cand,
                                                        const int *__restrict__ counts, int cap, int kc,
                                                        uint32_t *__restrict__ cand_ids,
                                                        float *__restrict__ cand_scores)
{
    __shared__ uint64_t lists[4 * 64];
    __shared__ int valid[4];
    __shared__ uint64_t best[64];
    const int64_t q = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int c = counts[q];
    if (c > cap) c = cap;
    const uint64_t *src = cand + q * cap;
    WaveTopK tk;
    tk.init(kc);
    for (int i0 = wave * 64; i0 < c; i0 += 256) {
        const int i = i0 + lane;
        tk.offer(i < c ? src[i] : kKeyMax, lane);
    }
    wg_rank_merge<4>(tk, lists, valid, wave, lane, tid, kc, best);
    __syncthreads();
    if (tid < kc) {
        const uint64_t e = best[tid];
        cand_ids[q * kc + tid] = e == kKeyMax ? VG_INVALID_ID : key_row(e);
        cand_scores[q * kc + tid] = e == kKeyMax ? INFINITY : key_score(e, false);
    }
}

__global__ void fill_f32_kernel(float *p, int64_t n, float v)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// A query's thresholds never above the largest score a row can have: 2 (|q|^2 + max |x|^2) >= |x|^2 + 2 |q| |x| (L2) >= |q.x|
// (Dot), with room for the bfloat16 rounding of both operands.  "No threshold" (+Inf: no sample of a small segment, a filter that
// leaves the sample short) means the same after it — every row passes — but the persistent bf16 tile (flat_gemm_bf16_big_kernel)
// starts its accumulators at (t - |x|^2) / 2 and reads a row's score back as t - 2 acc: with t = 1e30 standing in for +Inf the dot
// product is absorbed, every appended score comes back as ~0, and the 64 "best" of them were an arbitrary 64 (r06: Dot segments of
// up to 4096 rows, batches above 128 queries, answered wrongly — tests/test_gpu_nonfinite.py found it).  One wave per query.
__global__ __launch_bounds__(64) void flat_thr_cap_kernel(float *__restrict__ thr, int sel_k, const float *__restrict__ queries, int dim,
                                                          const float *__restrict__ norm_max)
{
    const int64_t q = blockIdx.x;
    const int lane = threadIdx.x;
    float qn = 0.0f;
    for (int j = lane; j < dim; j += 64) qn = __builtin_fmaf(queries[q * dim + j], queries[q * dim + j], qn);
    for (int off = 32; off > 0; off >>= 1) qn += __shfl_xor(qn, off);
    const float bound = 2.002f * (qn + norm_max[0]);
    if (!(bound < 1e30f)) return;  // not finite: the tile's own clamp stands in, and every proof fails on its (infinite) margin
    for (int i = lane; i < sel_k; i += 64) {
        const float t = thr[q * sel_k + i];
        if (!(t < bound)) thr[q * sel_k + i] = bound;  // (NaN too)
    }
}

// ---- 2. per-query streaming select of the kc smallest scores -------------------------------
constexpr int kSelWaves = 4;
constexpr int kSelThreads = kSelWaves * 64;
__global__ __launch_bounds__(kSelThreads) void flat_select_kernel(const float *__restrict__ scores,
                                                                  int64_t n, int slices, int kc,
                                                                  uint64_t *__restrict__ partial)
{
    __shared__ uint64_t lists[kSelWaves * 64];
    __shared__ int valid[kSelWaves];
    const int s = blockIdx.x;
    const int64_t q = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t r0 = n * s / slices, r1 = n * (s + 1) / slices;
    const float *row = scores + q * n;
    WaveTopK tk;
    tk.init(kc);
    // each wave sweeps chunks of 64 lanes x 4 loads x 4 floats; the 4 loads are unguarded and in
    // flight together (a guarded load would serialise on vmcnt(0)); a chunk whose minimum is not
    // below the wave's k-th key is skipped with one ballot
    const int64_t a0 = (r0 + 3) & ~int64_t(3);          // 16-byte aligned interior [a0, a1)
    const int64_t a1 = a0 + ((r1 - a0) / 1024) * 1024;  // whole 1024-float chunks
    if (a0 < r1 && a1 > a0) {
        const float4 *row4 = reinterpret_cast<const float4 *>(row + a0);
        const int64_t chunks = (a1 - a0) / 1024;
        for (int64_t c = wave; c < chunks; c += kSelWaves) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] = row4[c * 256 + u * 64 + lane];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int64_t i = a0 + c * 1024 + (u * 64 + lane) * 4;
                const float vals[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
                float mn = fminf(fminf(vals[0], vals[1]), fminf(vals[2], vals[3]));
                if (!__ballot(make_key(mn, 0u, false) < tk.tau)) continue;
#pragma unroll
                for (int e = 0; e < 4; e++)
                    tk.offer(make_key(vals[e], static_cast<uint32_t>(i + e), false), lane);
            }
        }
    }
    // ragged head and tail
    {
        const int64_t head_end = (a0 < r1 && a1 > a0) ? a0 : r1;
        for (int64_t i0 = r0 + wave * 64; i0 < head_end; i0 += kSelThreads) {
            const int64_t i = i0 + lane;
            tk.offer(i < head_end ? make_key(row[i], static_cast<uint32_t>(i), false) : kKeyMax, lane);
        }
        const int64_t tail_begin = (a0 < r1 && a1 > a0) ? a1 : r1;
        for (int64_t i0 = tail_begin + wave * 64; i0 < r1; i0 += kSelThreads) {
            const int64_t i = i0 + lane;
            tk.offer(i < r1 ? make_key(row[i], static_cast<uint32_t>(i), false) : kKeyMax, lane);
        }
    }
    wg_rank_merge<kSelWaves>(tk, lists, valid, wave, lane, tid, kc,
                             partial + (q * slices + s) * kc);
}

// ---- 3. exact re-score + proof -----------------------------------------------------------------
// cand_ids/cand_scores: the kc best GEMM scores per query, ascending.  One workgroup per query.
template <bool DOT>
__global__ __launch_bounds__(256) void flat_verify_kernel(
    const float *__restrict__ base, int64_t n, int dim, const float *__restrict__ queries,
    const float *__restrict__ norms_max /* [1] */, const uint32_t *__restrict__ cand_ids,
    const float *__restrict__ cand_scores, int kc, int k, uint32_t *__restrict__ ids,
    float *__restrict__ scores, int *__restrict__ fallback, const float *__restrict__ thr, int thr_stride,
    int thr_off, const int *__restrict__ counts, int cap, float eps_extra)
{
    __shared__ uint64_t keys[64];
    const Sub16 sub = Sub16::make(threadIdx.x);
    const int64_t q = blockIdx.x;
    const float *qv = queries + q * dim;
    for (int c = threadIdx.x >> 4; c < kc; c += 16) {
        const uint32_t id = cand_ids[q * kc + c];
        uint64_t key = kKeyMax;
        if (id != VG_INVALID_ID) {
            const float v = exact_pair16<DOT, kPair>(base + static_cast<int64_t>(id) * dim, qv, dim, sub);
            key = make_key(v, id, DOT);
        }
        if ((threadIdx.x & 15) == 0) keys[c] = key;
    }
    __syncthreads();
    if (threadIdx.x >= 64) return;
    const int lane = threadIdx.x;
    WaveTopK tk;
    tk.init(k);
    tk.offer(lane < kc ? keys[lane] : kKeyMax, lane);
    // ||q||^2 (any order: only feeds the error bound)
    float qn = 0.0f;
    for (int j = lane; j < dim; j += 64) qn = __builtin_fmaf(qv[j], qv[j], qn);
    for (int off = 32; off > 0; off >>= 1) qn += __shfl_xor(qn, off);
    const uint64_t kth = readlane_u64(tk.list, k - 1);
    bool ok = true;
    // tau: a lower bound on the GEMM score of every row that was NOT nominated
    float tau;
    bool have_all;
    if (thr) {  // fused path: rows not appended have score >= threshold; appended-but-not-picked
                // rows have score >= the kc-th picked score
        const float tq = thr[q * thr_stride + thr_off];
        const int cnt = counts[q];
        if (cnt > cap) ok = false;  // buffer overflow: some rows below the threshold were dropped
        tau = cnt > kc ? fminf(tq, cand_scores[q * kc + (kc - 1)]) : tq;
        have_all = (tq == INFINITY) && cnt <= kc;
    } else {
        tau = cand_scores[q * kc + (kc - 1)];  // worst nominated GEMM score
        have_all = (n <= kc);                  // every row is a candidate: nothing to prove
    }
    if (ok && !have_all) {
        const float xmax = norms_max[0];
        // fp32 GEMM-form vs exact: |err| <= ~ 2*dim*2^-24*(|q||x|) per dot; bound generously
        // (eps_extra: the bf16 filter's share, 0 for the fp32 GEMM — see vg_index_enable_bf16_filter)
        const float eps = (4.0f * (static_cast<float>(dim) * 5.9604645e-8f) + eps_extra) * (qn + xmax) + 1e-30f;
        if (tau == INFINITY) {
            ok = true;  // nothing was excluded
        } else if (kth == kKeyMax) {
            ok = false;
        } else if (DOT) {
            // outside rows: -q.x >= tau  =>  q.x <= -tau (+eps); need kth dot > that
            const float dk = key_score(kth, true);
            ok = dk > (-tau) + eps;
        } else {
            const float dk = key_score(kth, false);
            ok = dk < (tau + qn) - eps;
        }
    }
    if (lane < k) {
        const uint64_t e = tk.list;
        ids[q * k + lane] = e == kKeyMax ? VG_INVALID_ID : key_row(e);
        scores[q * k + lane] = e == kKeyMax ? (DOT ? -INFINITY : INFINITY) : key_score(e, DOT);
    }
    if (lane == 0) fallback[q] = ok ? 0 : 1;
}

// The same for k beyond the 64-candidate budget: EVERY appended row (GEMM score below the query's
// threshold, ~500 of them) is re-scored exactly, so the only rows left to argue about are the ones the
// threshold excluded: the proof compares the k-th exact score with the threshold itself.
template <bool DOT>
__global__ __launch_bounds__(256) void flat_verify_all_kernel(
    const float *__restrict__ base, int dim, const float *__restrict__ queries,
    const float *__restrict__ norms_max /* [1] */, const uint64_t *__restrict__ cand, const int *__restrict__ counts,
    int cap, int k, uint32_t *__restrict__ ids, float *__restrict__ scores, int *__restrict__ fallback,
    const float *__restrict__ thr, int thr_stride, int thr_off, float eps_extra)
{
    __shared__ uint64_t lists[4 * 64];
    __shared__ int valid[4];
    __shared__ uint64_t best[64];
    const Sub16 sub = Sub16::make(threadIdx.x);
    const int64_t q = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *qv = queries + q * dim;
    const int total = counts[q];
    const int cnt = total < cap ? total : cap;
    WaveTopK tk;
    tk.init(k);
    for (int c0 = wave * 4; c0 < cnt; c0 += 16) {  // 4 candidates per wave step, one per 16-lane group
        const int c = c0 + (lane >> 4);
        uint64_t key = kKeyMax;
        if (c < cnt) {
            const uint32_t id = key_row(cand[q * cap + c]);
            const float v = exact_pair16<DOT, kPair>(base + static_cast<int64_t>(id) * dim, qv, dim, sub);
            if ((lane & 15) == 0) key = make_key(v, id, DOT);
        }
        tk.offer(key, lane);
    }
    wg_rank_merge<4>(tk, lists, valid, wave, lane, tid, k, best);
    __syncthreads();
    if (tid >= 64) return;
    float qn = 0.0f;  // ||q||^2 (any order: only feeds the error bound)
    for (int j = lane; j < dim; j += 64) qn = __builtin_fmaf(qv[j], qv[j], qn);
    for (int off = 32; off > 0; off >>= 1) qn += __shfl_xor(qn, off);
    const uint64_t kth = best[k - 1];
    const float tau = thr[q * thr_stride + thr_off];  // every row that was not appended scores >= tau
    bool ok = total <= cap;                           // overflow: rows below the threshold were dropped
    if (ok && tau != INFINITY) {
        const float xmax = norms_max[0];
        // (eps_extra: the bf16 filter's share, 0 for the fp32 GEMM — see vg_index_enable_bf16_filter)
        const float eps = (4.0f * (static_cast<float>(dim) * 5.9604645e-8f) + eps_extra) * (qn + xmax) + 1e-30f;
        if (kth == kKeyMax)
            ok = false;
        else if (DOT)
            ok = key_score(kth, true) > (-tau) + eps;
        else
            ok = key_score(kth, false) < (tau + qn) - eps;
    }
    if (lane < k) {
        const uint64_t e = best[lane];
        ids[q * k + lane] = e == kKeyMax ? VG_INVALID_ID : key_row(e);
        scores[q * k + lane] = e == kKeyMax ? (DOT ? -INFINITY : INFINITY) : key_score(e, DOT);
    }
    if (lane == 0) fallback[q] = ok ? 0 : 1;
}

// k > 64 (one wave's register list holds 64 keys): the exact keys of every appended row are sorted in
// LDS instead; the proof is the same as above.  n2 = the power of two the workgroup sorts (>= appended rows).
template <bool DOT>
__global__ __launch_bounds__(256) void flat_verify_sort_kernel(
    const float *__restrict__ base, int dim, const float *__restrict__ queries,
    const float *__restrict__ norms_max /* [1] */, const uint64_t *__restrict__ cand, const int *__restrict__ counts,
    int cap, int k, uint32_t *__restrict__ ids, float *__restrict__ scores, int *__restrict__ fallback,
    const float *__restrict__ thr, int thr_stride, int thr_off, float eps_extra)
{
    extern __shared__ uint64_t sortbuf[];  // cap keys
    const Sub16 sub = Sub16::make(threadIdx.x);
    const int64_t q = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const float *qv = queries + q * dim;
    const int total = counts[q];
    const int cnt = total < cap ? total : cap;
    int n2 = 64;
    while (n2 < cnt) n2 <<= 1;
    for (int c0 = 0; c0 < cnt; c0 += 16) {  // 16 candidates per step, one per 16-lane group
        const int c = c0 + (tid >> 4);
        if (c < cnt) {
            const uint32_t id = key_row(cand[q * cap + c]);
            const float v = exact_pair16<DOT, kPair>(base + static_cast<int64_t>(id) * dim, qv, dim, sub);
            if ((tid & 15) == 0) sortbuf[c] = make_key(v, id, DOT);
        }
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
    const float tau = thr[q * thr_stride + thr_off];
    bool ok = total <= cap;
    if (ok && tau != INFINITY) {
        const float xmax = norms_max[0];
        // (eps_extra: the bf16 filter's share, 0 for the fp32 GEMM — see vg_index_enable_bf16_filter)
        const float eps = (4.0f * (static_cast<float>(dim) * 5.9604645e-8f) + eps_extra) * (qn + xmax) + 1e-30f;
        if (kth == kKeyMax)
            ok = false;
        else if (DOT)
            ok = key_score(kth, true) > (-tau) + eps;
        else
            ok = key_score(kth, false) < (tau + qn) - eps;
    }
    if (lane == 0) fallback[q] = ok ? 0 : 1;
}

// ---- 4. exhaustive exact scan for the queries whose proof failed ---------------------------------
// grid = (slices, slots): slot y takes the work-list entries y, y + slots, ...; with an empty list
// (the normal case) the whole launch is a few thousand workgroups that read one word and leave.
constexpr int kExactSlots = 32;
template <bool DOT>
__global__ __launch_bounds__(256) void flat_exact_kernel(const float *__restrict__ base, int64_t n,
                                                         int dim, const float *__restrict__ queries,
                                                         const int *__restrict__ todo, int slices, int k,
                                                         uint64_t *__restrict__ partial,
                                                         const uint64_t *__restrict__ min_keys = nullptr,
                                                         const uint8_t *__restrict__ mask = nullptr, int64_t mask_stride = 0)
{
    __shared__ uint64_t lists[4 * 64];
    __shared__ int valid[4];
    const int s = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntodo = todo[0];
    const Sub16 sub = Sub16::make(tid);
    const int64_t r0 = n * s / slices, r1 = n * (s + 1) / slices;
    for (int j = blockIdx.y; j < ntodo; j += gridDim.y) {
        const int64_t q = todo[1 + j];
        uint64_t *out = partial + (q * slices + s) * k;
        const float *qv = queries + q * dim;
        // paged results (k > 64): only keys after the last one of the previous page count
        const uint64_t floor_key = min_keys ? min_keys[q] : 0;
        const uint8_t *mq = mask ? mask + q * mask_stride : nullptr;  // filtered search: rows the filter rejects are skipped
        WaveTopK tk;
        tk.init(k);
        // 4 rows per wave step (one per 16-lane group)
        for (int64_t i0 = r0 + wave * 4; i0 < r1; i0 += 16) {
            const int64_t i = i0 + (lane >> 4);
            uint64_t key = kKeyMax;
            if (i < r1 && mask_bit(mq, i)) {
                const float v = exact_pair16<DOT, kPair>(base + i * dim, qv, dim, sub);
                if ((lane & 15) == 0) {
                    key = make_key(v, static_cast<uint32_t>(i), DOT);
                    if (min_keys && key <= floor_key) key = kKeyMax;
                }
            }
            tk.offer(key, lane);
        }
        wg_rank_merge<4>(tk, lists, valid, wave, lane, tid, k, out);
        __syncthreads();  // lists / valid are reused by the next work item
    }
}

// ---- small query batches: one HBM pass scores every row against up to 8 queries --------------------
// Below ~5 queries even the 32-query GEMM tile is slower than a plain scan (a 128-query tile does 128 queries'
// worth of MFMA work and 1/8 of the tiles' loads for whatever is in it: 1.65 ms at 1M x 768).
// This kernel is the HBM-bound alternative: a 16-lane group loads a row ONCE into registers
// (dim <= 1024: 16 float4 per lane) and scores it against QB queries held in LDS, each in the
// reference's summation order (vg_exact.hpp, kPair) — exact by construction, no proof step.
constexpr int kScanQB = 8;        // queries one pass can carry
constexpr int kGemmMaxK = 48;     // largest k the 64-candidate nomination + proof serves (above: flat_verify_all_kernel)
constexpr int kFlatMaxK = 512;    // candidates appended per query stay well under the 4096-key buffer (3k expected)
constexpr int kScanMaxBatch = 4;  // ... and the batch size up to which the scan beats the 32-query GEMM tile
template <bool DOT>
__global__ __launch_bounds__(256) void flat_scan_mq_kernel(const float *__restrict__ base, int64_t n, int dim,
                                                           const float *__restrict__ queries, int nq, int slices,
                                                           int k, uint64_t *__restrict__ partial)
{
    extern __shared__ float qlds[];  // nq * dim floats, then the merge scratch
    uint64_t *lists = reinterpret_cast<uint64_t *>(qlds + static_cast<size_t>(kScanQB) * dim);
    int *valid = reinterpret_cast<int *>(lists + 4 * 64);
    const int s = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int t = tid; t < nq * dim; t += 256) qlds[t] = queries[t];
    __syncthreads();
    const Sub16 sub = Sub16::make(tid);
    const int nblk = dim >> 6;
    const int64_t r0 = n * s / slices, r1 = n * (s + 1) / slices;
    WaveTopK tk[kScanQB];
#pragma unroll
    for (int qi = 0; qi < kScanQB; qi++) tk[qi].init(k);
    for (int64_t i0 = r0 + wave * 4; i0 < r1; i0 += 16) {
        const int64_t i = i0 + (lane >> 4);
        const bool live = i < r1;
        const float *row = base + (live ? i : r1 - 1) * dim;
        float4 rr[16];
        const float4 *r4 = reinterpret_cast<const float4 *>(row) + sub.f4;
#pragma unroll
        for (int e = 0; e < 16; e++)
            if (e < nblk) rr[e] = load_stream(r4 + e * 16);
#pragma unroll
        for (int qi = 0; qi < kScanQB; qi++) {
            if (qi < nq) {
                const float v = exact_rowregs16<DOT>(rr, nblk, row, qlds + static_cast<size_t>(qi) * dim, dim, sub);
                uint64_t key = kKeyMax;
                if (live && (lane & 15) == 0) key = make_key(v, static_cast<uint32_t>(i), DOT);
                tk[qi].offer(key, lane);
            }
        }
    }
#pragma unroll
    for (int qi = 0; qi < kScanQB; qi++) {
        if (qi < nq) {
            wg_rank_merge<4>(tk[qi], lists, valid, wave, lane, tid, k,
                             partial + (static_cast<int64_t>(qi) * slices + s) * k);
            __syncthreads();
        }
    }
}

// overwrite the results of the fallback queries from the exact scan's merged lists
__global__ void flat_patch_kernel(const int *__restrict__ fallback, const int *__restrict__ always, int k,
                                  const uint32_t *__restrict__ fids, const float *__restrict__ fscores,
                                  uint32_t *__restrict__ ids, float *__restrict__ scores)
{
    const int64_t q = blockIdx.x;
    if (!(always && always[0]) && !fallback[q]) return;
    for (int i = threadIdx.x; i < k; i += blockDim.x) {
        ids[q * k + i] = fids[q * k + i];
        scores[q * k + i] = fscores[q * k + i];
    }
}

// page `off / 64` of a paged exhaustive result (k > 64): kk results per flagged query land at ids[q*k + off ..),
// and the last key of the page becomes the floor of the next one (kKeyMax when the rows ran out)
__global__ void flat_page_patch_kernel(const int *__restrict__ fallback, const int *__restrict__ always, int k, int off,
                                       int kk, bool descending, const uint32_t *__restrict__ fids,
                                       const float *__restrict__ fscores, uint32_t *__restrict__ ids,
                                       float *__restrict__ scores, uint64_t *__restrict__ min_keys)
{
    const int64_t q = blockIdx.x;
    if (!(always && always[0]) && !(fallback && fallback[q])) return;
    for (int i = threadIdx.x; i < kk; i += blockDim.x) {
        ids[q * k + off + i] = fids[q * kk + i];
        scores[q * k + off + i] = fscores[q * kk + i];
    }
    if (threadIdx.x == 0) {
        const uint32_t last = fids[q * kk + kk - 1];
        min_keys[q] = last == VG_INVALID_ID ? kKeyMax : make_key(fscores[q * kk + kk - 1], last, descending);
    }
}

// all queries of a paged scan (k > 64 through a kernel that keeps 64 keys per wave): copy page `off / 64` and
// make its last key the floor of the next page
int32_t launch_page_patch(int64_t nq, int k, int off, int kk, bool descending, const int *always_one,
                          const uint32_t *fids, const float *fscores, uint32_t *ids, float *scores, uint64_t *min_keys,
                          hipStream_t st)
{
    VG_LAUNCH(flat_page_patch_kernel, dim3(static_cast<unsigned>(nq)), dim3(64), 0, st, nullptr, always_one, k, off, kk,
              descending, fids, fscores, ids, scores, min_keys);
    return VG_OK;
}

}  // namespace vg

namespace vg {
int32_t flat_search_masked(vg_index *idx, const float *queries, int64_t nq, int32_t k, const uint8_t *mask, int64_t mask_stride,
                           uint32_t *ids, float *scores, void *stream, bool l2_scores = false, bool cand_replay = true);
}

VG_API int32_t vg_search_flat(vg_index *idx, const float *queries, int64_t nq, int32_t k, uint32_t *ids,
                              float *scores, void *stream)
{
    return vg::flat_search_masked(idx, queries, nq, k, nullptr, 0, ids, scores, stream);
}

// vg_search_flat, and — with `mask` (a DEVICE pointer: bit i of byte i/8 of query q's mask at mask + q * mask_stride,
// stride 0 = one mask) — the batched fp32 leg of vg_search_flat_filtered (k_probe.hip): the rows a query's filter rejects
// are left out of its threshold sample and of its candidate list, so the proof argues about the rows it wants only
// ("every wanted row not appended scores at or above the threshold"); the exhaustive fallback skips them too.
// l2_scores: squared-L2 scores whatever the index's metric (vg_search_hnsw_brute on a Cosine index: hnsw's distance there is
// 0.5 * squared L2, flat.Segment's is the dot product)
int32_t vg::flat_search_masked(vg_index *idx, const float *queries, int64_t nq, int32_t k, const uint8_t *mask,
                               int64_t mask_stride, uint32_t *ids, float *scores, void *stream, bool l2_scores, bool cand_replay)
{
    VG_CHECK(idx, VG_ERR_INVALID_ARG, "vg_search_flat: NULL index");
    VG_CHECK(nq >= 0 && k >= 0, VG_ERR_INVALID_ARG, "vg_search_flat: negative nq or k");
    if (nq == 0 || k == 0) return VG_OK;
    VG_CHECK(idx->metric != VG_METRIC_HAMMING, VG_ERR_UNSUPPORTED, "unsupported metric for float32: Hamming");
    VG_CHECK(idx->n == 0 || idx->d_vectors, VG_ERR_NOT_READY, "vg_search_flat: index has no fp32 vectors");
    VG_CHECK(queries && ids && scores, VG_ERR_INVALID_ARG, "vg_search_flat: NULL buffer");
    VG_CHECK(k <= vg::kFlatMaxK, VG_ERR_UNSUPPORTED, "vg_search_flat: k=%d exceeds %d", k, vg::kFlatMaxK);
    VG_HIP(hipSetDevice(idx->ctx->device));
    hipStream_t st = vg::pick_stream(idx->ctx, stream);
    const bool dot = idx->metric != VG_METRIC_L2 && !l2_scores;
    const int64_t n = idx->n;
    const int dim = idx->dim;
    const int kc = 64;  // nominated candidates per query
    // k beyond what 64 nominated candidates can prove: the fused GEMM path re-scores every appended row
    // (flat_verify_all_kernel); the unfused score-matrix variant (test hook) stops at 64
    const bool unfused = vg::hook(vg::kHookFlatUnfused) && mask == nullptr;
    const bool big_k_scan = k > vg::kGemmMaxK && unfused;  // unfused: 64 nominated candidates cannot prove k near 64
    VG_CHECK(k <= 64 || !unfused, VG_ERR_UNSUPPORTED, "vg_search_flat: k=%d needs the fused GEMM path", k);

    vg::DevIn<float> q;
    vg::DevOut<uint32_t> oid;
    vg::DevOut<float> osc;
    VG_TRY(q.init(queries, static_cast<size_t>(nq) * dim, st));
    VG_TRY(oid.init(ids, static_cast<size_t>(nq) * k, st));
    VG_TRY(osc.init(scores, static_cast<size_t>(nq) * k, st));

    if (n == 0) {
        vg::DevTmp<uint64_t> none;
        VG_TRY(none.init(static_cast<size_t>(nq) * k, st));
        VG_HIP(hipMemsetAsync(none.ptr, 0xFF, static_cast<size_t>(nq) * k * 8, st));
        VG_TRY(vg::launch_topk_merge(none.ptr, nq, 1, k, dot, oid.ptr, osc.ptr, st));
    } else if (mask == nullptr && nq <= vg::kScanMaxBatch && k <= 64 && dim % 4 == 0 && dim <= 1024 && dim >= 64 && !vg::hook(vg::kHookFlatNoScan) &&
               !vg::hook(vg::kHookFlatForceExact) && !unfused && (reinterpret_cast<uintptr_t>(idx->d_vectors) & 15) == 0) {
        // small batch: HBM-bound exact scan, kScanQB queries per pass over the rows
        const int slices = static_cast<int>(std::min<int64_t>(4 * idx->ctx->compute_units, std::max<int64_t>(1, n / 64)));
        vg::ArenaCall ar(idx->ctx, st);
        const int i_partial = ar.add(sizeof(uint64_t) * static_cast<size_t>(nq) * slices * k);
        VG_TRY(ar.commit());
        uint64_t *partial = ar.get<uint64_t>(i_partial);
        const size_t lds = sizeof(float) * static_cast<size_t>(vg::kScanQB) * dim + 4 * 64 * sizeof(uint64_t) + 64;
        for (int64_t q0 = 0; q0 < nq; q0 += vg::kScanQB) {
            const int cnt = static_cast<int>(std::min<int64_t>(vg::kScanQB, nq - q0));
            vg::ProfScope prof(idx->ctx, "flat_scan", st);
            if (dot)
                VG_LAUNCH(vg::flat_scan_mq_kernel<true>, dim3(slices), dim3(256), lds, st, idx->d_vectors, n, dim,
                          q.ptr + q0 * dim, cnt, slices, k, partial + q0 * slices * k);
            else
                VG_LAUNCH(vg::flat_scan_mq_kernel<false>, dim3(slices), dim3(256), lds, st, idx->d_vectors, n, dim,
                          q.ptr + q0 * dim, cnt, slices, k, partial + q0 * slices * k);
        }
        VG_TRY(vg::launch_topk_merge(partial, nq, slices, k, dot, oid.ptr, osc.ptr, st));
        VG_LAUNCH(vg::flat_todo_kernel, dim3(1), dim3(256), 0, st, nullptr, nullptr, static_cast<int>(nq), nullptr,
                  idx->d_flat_stats);
    } else if (big_k_scan) {
        // kGemmMaxK < k <= 64 on rows the register scan does not take (dim % 4, dim < 64 or > 1024): the
        // exhaustive exact kernel for every query, one pass over the rows per query
        const int ex_slices = static_cast<int>(std::min<int64_t>(256, std::max<int64_t>(1, n / 64)));
        const int64_t qc = std::min<int64_t>(nq, 4096);
        vg::ArenaCall ar(idx->ctx, st);
        const int i_flags = ar.add(sizeof(int) * (static_cast<size_t>(qc) + 1));
        const int i_todo = ar.add(sizeof(int) * (static_cast<size_t>(qc) + 1));
        const int i_fpartial = ar.add(sizeof(uint64_t) * static_cast<size_t>(qc) * ex_slices * k);
        VG_TRY(ar.commit());
        int *flags = ar.get<int>(i_flags), *todo = ar.get<int>(i_todo);
        uint64_t *fpartial = ar.get<uint64_t>(i_fpartial);
        VG_HIP(hipMemsetAsync(flags, 0, sizeof(int) * static_cast<size_t>(qc), st));
        VG_HIP(hipMemsetAsync(flags + qc, 1, sizeof(int), st));  // "every query" switch of flat_todo_kernel
        for (int64_t q0 = 0; q0 < nq; q0 += qc) {
            const int64_t cnt = std::min(qc, nq - q0);
            VG_LAUNCH(vg::flat_todo_kernel, dim3(1), dim3(256), 0, st, flags, flags + qc, static_cast<int>(cnt), todo,
                      idx->d_flat_stats);
            const unsigned slots = static_cast<unsigned>(std::min<int64_t>(cnt, vg::kExactSlots));
            if (dot)
                VG_LAUNCH(vg::flat_exact_kernel<true>, dim3(ex_slices, slots), dim3(256), 0, st, idx->d_vectors, n, dim,
                          q.ptr + q0 * dim, todo, ex_slices, k, fpartial);
            else
                VG_LAUNCH(vg::flat_exact_kernel<false>, dim3(ex_slices, slots), dim3(256), 0, st, idx->d_vectors, n, dim,
                          q.ptr + q0 * dim, todo, ex_slices, k, fpartial);
            VG_TRY(vg::launch_topk_merge(fpartial, cnt, ex_slices, k, dot, oid.ptr + q0 * k, osc.ptr + q0 * k, st));
        }
    } else {
        const bool fused = !unfused;  // test hook: materialise the score matrix
        const int cap = 4096;          // candidate keys per query (fused path)
        // threshold = sample_j-th best score of the row sample: ~64 * sample_j rows pass it (k > 64: ~3k of them)
        const int sample_j = k <= 64 ? 8 : std::min(64, std::max(8, (3 * k + 63) / 64));
        const int sample_stride = 64;  // every 64th 128-row tile is sampled
        const int64_t nt = (n + vg::kGemmBN - 1) / vg::kGemmBN;
        const int64_t nst = (nt + sample_stride - 1) / sample_stride;  // sampled tiles
        const int64_t ns = nst * vg::kGemmBN;                          // sampled columns
        const bool use_sample = fused && n > cap;  // n <= cap: threshold +Inf, every row is appended

        // query chunk: whole 128-query tiles; unfused keeps the score matrix <= 2 GiB
        int64_t qc = fused ? 4096 : (int64_t(2) << 30) / (n * 4);
        if (qc > 4096) qc = 4096;
        if (qc >= vg::kGemmBM) qc = (qc / vg::kGemmBM) * vg::kGemmBM;
        if (qc < 1) qc = 1;
        if (qc > nq) qc = nq;
        const int64_t sel_n = fused ? ns : n;
        const int sel_slices = static_cast<int>(std::min<int64_t>(64, std::max<int64_t>(1, sel_n / 4096)));
        const int sel_k = fused ? sample_j : kc;
        const int ex_slices = static_cast<int>(std::min<int64_t>(256, std::max<int64_t>(1, n / 64)));

        vg::ArenaCall ar(idx->ctx, st);
        const int i_sc = ar.add(sizeof(float) * static_cast<size_t>(qc) * (fused ? (use_sample ? ns : 1) : n));
        const int i_partial = ar.add(sizeof(uint64_t) * static_cast<size_t>(qc) * sel_slices * sel_k);
        const int i_sid = ar.add(sizeof(uint32_t) * static_cast<size_t>(qc) * sel_k);
        const int i_thr = ar.add(sizeof(float) * static_cast<size_t>(qc) * sel_k);
        const int i_counts = ar.add(sizeof(int) * static_cast<size_t>(qc));
        const int i_cand = ar.add(fused ? sizeof(uint64_t) * static_cast<size_t>(qc) * cap : 0);
        const int i_cand_id = ar.add(sizeof(uint32_t) * static_cast<size_t>(qc) * kc);
        const int i_cand_sc = ar.add(sizeof(float) * static_cast<size_t>(qc) * kc);
        const int i_flags = ar.add(sizeof(int) * (static_cast<size_t>(qc) + 1));
        const int i_todo = ar.add(sizeof(int) * (static_cast<size_t>(qc) + 1));
        const int pk = std::min(k, 64);  // results per page of the exhaustive fallback
        const int i_fpartial = ar.add(sizeof(uint64_t) * static_cast<size_t>(qc) * ex_slices * pk);
        const int i_fid = ar.add(sizeof(uint32_t) * static_cast<size_t>(qc) * pk);
        const int i_fsc = ar.add(sizeof(float) * static_cast<size_t>(qc) * pk);
        const int i_minkeys = ar.add(sizeof(uint64_t) * static_cast<size_t>(qc));
        // the bfloat16 filter (vg_index_enable_bf16_filter): the two nomination GEMMs read bf16 copies of the rows and
        // of the queries; what they nominate is re-scored exactly as before and the proof widens its margin by the
        // rounding the copies can introduce.  bfloat16 keeps 8 significant bits: rounding to nearest moves an operand
        // by at most 2^-8 of its magnitude, a product q_i*x_i by at most (2^-7 + 2^-16)|q_i x_i|, the dot product by
        // (2^-7 + 2^-16)|q||x| <= (2^-8 + 2^-17)(|q|^2 + |x|^2).  A Dot score is the dot product; an L2 score is
        // |x|^2 - 2 q.x (norms from the fp32 rows), TWICE that: (2^-7 + 2^-16)(|q|^2 + |x|^2).  (r02 used the Dot
        // bound for both — half of what L2 needs: ADVICE r02, tests/test_gpu_flat_bf16.py
        // test_filter_worst_case_rounding.)  The verify kernels multiply eps_extra by |q|^2 + max|x|^2.
        // (any dim: the copies' rows are padded with zeros to whole K steps, vectors_bf16_dim)
        const bool bf16 = idx->d_vectors_bf16 != nullptr && fused && !vg::hook(vg::kHookFlatNoDma);
        const int bdim = idx->vectors_bf16_dim;
        const float eps_extra = !bf16 ? 0.0f : (dot ? 0.00390625f : 0.0078125f) * 1.02f;
        const int i_qbf = ar.add(bf16 ? sizeof(uint16_t) * static_cast<size_t>(qc) * bdim : 0);
        const int i_cwide = ar.add(bf16 ? sizeof(int) * static_cast<size_t>(qc) * vg::kCountLine : 0);
        VG_TRY(ar.commit());
        uint16_t *qbf = ar.get<uint16_t>(i_qbf);
        float *sc = ar.get<float>(i_sc), *thr = ar.get<float>(i_thr);
        float *cand_sc = ar.get<float>(i_cand_sc), *fsc = ar.get<float>(i_fsc);
        uint64_t *partial = ar.get<uint64_t>(i_partial), *fpartial = ar.get<uint64_t>(i_fpartial);
        uint64_t *cand = ar.get<uint64_t>(i_cand);
        uint64_t *min_keys = ar.get<uint64_t>(i_minkeys);
        uint32_t *sid = ar.get<uint32_t>(i_sid), *cand_id = ar.get<uint32_t>(i_cand_id), *fid = ar.get<uint32_t>(i_fid);
        int *counts = ar.get<int>(i_counts), *flags = ar.get<int>(i_flags), *todo = ar.get<int>(i_todo);
        int *counts_wide = bf16 ? ar.get<int>(i_cwide) : nullptr;

        int *always = flags + qc;  // test hook kHookFlatForceExact: run step 4 for every query
        VG_HIP(hipMemsetAsync(always, vg::hook(vg::kHookFlatForceExact) ? 1 : 0, sizeof(int), st));
        for (int64_t q0 = 0; q0 < nq; q0 += qc) {
            const int64_t cnt = std::min(qc, nq - q0);
            const float *qp = q.ptr + q0 * dim;
            const uint8_t *m0 = mask ? mask + q0 * mask_stride : nullptr;
            const int64_t mt = (cnt + vg::kGemmBM - 1) / vg::kGemmBM;
            const unsigned ucnt = static_cast<unsigned>(cnt);
            // (test hook kHookFlatNoDma: force the register-staged GEMM)
            const bool dma = dim % 4 == 0 && (reinterpret_cast<uintptr_t>(qp) & 15) == 0 &&
                             (reinterpret_cast<uintptr_t>(idx->d_vectors) & 15) == 0 && !vg::hook(vg::kHookFlatNoDma);
            // operands of the nomination GEMMs
            const float *ga = qp, *gb = idx->d_vectors;
            int gdim = dim;
            if (bf16) {
                VG_LAUNCH(vg::f32_to_bf16_pad_kernel, dim3(static_cast<unsigned>((cnt * bdim + 255) / 256)), dim3(256), 0, st, qp,
                          cnt, dim, bdim, qbf);
                ga = reinterpret_cast<const float *>(qbf);
                gb = reinterpret_cast<const float *>(idx->d_vectors_bf16);
                gdim = bdim / 2;
            }
            if (fused) {
                // (a) threshold per query from a row sample
                if (use_sample) {
                    VG_TRY(vg::launch_gemm<1>(dot, dma, static_cast<unsigned>(mt * ((nst + 7) / 8) * 8), st,
                                              {ga, cnt, gb, n, gdim, idx->d_norms, sc, sample_stride, ns,
                                               nullptr, 0, 0, nullptr, nullptr, 0, m0, mask_stride}, bf16));
                    VG_LAUNCH(vg::flat_select_kernel, dim3(sel_slices, ucnt), dim3(vg::kSelThreads), 0, st, sc, ns,
                              sel_slices, sel_k, partial);
                    VG_TRY(vg::launch_topk_merge(partial, cnt, sel_slices, sel_k, false, sid, thr, st));
                } else {
                    VG_LAUNCH(vg::fill_f32_kernel, dim3(static_cast<unsigned>((cnt * sel_k + 255) / 256)), dim3(256),
                              0, st, thr, cnt * sel_k, INFINITY);
                }
                if (bf16) VG_LAUNCH(vg::flat_thr_cap_kernel, dim3(ucnt), dim3(64), 0, st, thr, sel_k, qp, dim, idx->d_norm_max);
                // (b) the GEMM, appending every element below its query's threshold
                VG_HIP(hipMemsetAsync(counts, 0, sizeof(int) * static_cast<size_t>(cnt), st));
                {
                    vg::ProfScope prof(idx->ctx, "flat_gemm", st);
                    VG_TRY(vg::launch_gemm<2>(dot, dma, static_cast<unsigned>(mt * ((nt + 7) / 8) * 8), st,
                                              {ga, cnt, gb, n, gdim, idx->d_norms, nullptr, 1, 0, thr,
                                               sel_k, sel_k - 1, counts, cand, cap, m0, mask_stride, idx->ctx->compute_units, counts_wide}, bf16));
                }
                // (c) the kc best appended keys (k > kGemmMaxK: all of them go to the exact re-score below)
                if (k <= vg::kGemmMaxK)
                    VG_LAUNCH(vg::flat_pick_kernel, dim3(ucnt), dim3(256), 0, st, cand, counts, cap, kc, cand_id, cand_sc);
            } else {
                {
                    vg::ProfScope prof(idx->ctx, "flat_gemm", st);
                    VG_TRY(vg::launch_gemm<0>(dot, dma, static_cast<unsigned>(mt * ((nt + 7) / 8) * 8), st,
                                              {qp, cnt, idx->d_vectors, n, dim, idx->d_norms, sc, 1, n, nullptr, 0, 0,
                                               nullptr, nullptr, 0}));
                }
                {
                    vg::ProfScope prof(idx->ctx, "flat_select", st);
                    VG_LAUNCH(vg::flat_select_kernel, dim3(sel_slices, ucnt), dim3(vg::kSelThreads), 0, st, sc, n,
                              sel_slices, kc, partial);
                }
                VG_TRY(vg::launch_topk_merge(partial, cnt, sel_slices, kc, false, cand_id, cand_sc, st));
            }
            const float *vthr = fused ? thr : nullptr;
            if (fused && k > 64) {
                const size_t sort_lds = sizeof(uint64_t) * static_cast<size_t>(cap);
                auto vk = dot ? vg::flat_verify_sort_kernel<true> : vg::flat_verify_sort_kernel<false>;
                VG_LAUNCH(vk, dim3(ucnt), dim3(256), sort_lds, st, idx->d_vectors, dim, qp, idx->d_norm_max, cand, counts, cap,
                          k, oid.ptr + q0 * k, osc.ptr + q0 * k, flags, thr, sel_k, sel_k - 1, eps_extra);
            } else if (fused && k > vg::kGemmMaxK) {
                if (dot)
                    VG_LAUNCH(vg::flat_verify_all_kernel<true>, dim3(ucnt), dim3(256), 0, st, idx->d_vectors, dim, qp,
                              idx->d_norm_max, cand, counts, cap, k, oid.ptr + q0 * k, osc.ptr + q0 * k, flags, thr, sel_k,
                              sel_k - 1, eps_extra);
                else
                    VG_LAUNCH(vg::flat_verify_all_kernel<false>, dim3(ucnt), dim3(256), 0, st, idx->d_vectors, dim, qp,
                              idx->d_norm_max, cand, counts, cap, k, oid.ptr + q0 * k, osc.ptr + q0 * k, flags, thr, sel_k,
                              sel_k - 1, eps_extra);
            } else if (dot)
                VG_LAUNCH(vg::flat_verify_kernel<true>, dim3(ucnt), dim3(256), 0, st, idx->d_vectors, n, dim, qp,
                          idx->d_norm_max, cand_id, cand_sc, kc, k, oid.ptr + q0 * k, osc.ptr + q0 * k, flags, vthr,
                          sel_k, sel_k - 1, counts, cap, eps_extra);
            else
                VG_LAUNCH(vg::flat_verify_kernel<false>, dim3(ucnt), dim3(256), 0, st, idx->d_vectors, n, dim, qp,
                          idx->d_norm_max, cand_id, cand_sc, kc, k, oid.ptr + q0 * k, osc.ptr + q0 * k, flags, vthr,
                          sel_k, sel_k - 1, counts, cap, eps_extra);
            // step 4 always launches, on the work list the proofs left behind (normally empty)
            VG_LAUNCH(vg::flat_todo_kernel, dim3(1), dim3(256), 0, st, flags, always, static_cast<int>(cnt), todo,
                      idx->d_flat_stats);
            if (vg::hook(vg::kHookFlatDebug)) {
                std::vector<int> hf(cnt), hc(cnt);
                (void)hipMemcpyAsync(hf.data(), flags, sizeof(int) * cnt, hipMemcpyDeviceToHost, st);
                (void)hipMemcpyAsync(hc.data(), counts, sizeof(int) * cnt, hipMemcpyDeviceToHost, st);
                (void)hipStreamSynchronize(st);
                int64_t nf = 0, csum = 0, cmax = 0;
                for (int f : hf) nf += f != 0;
                for (int c : hc) { csum += c; cmax = std::max<int64_t>(cmax, c); }
                fprintf(stderr, "[vg_search_flat] chunk q0=%lld cnt=%lld: %lld queries failed the proof; "
                        "appended candidates avg %.1f max %lld\n", (long long)q0, (long long)cnt, (long long)nf,
                        fused ? double(csum) / cnt : 0.0, (long long)(fused ? cmax : 0));
            }
            // the exhaustive kernel keeps 64 keys per wave: k > 64 comes in pages of 64, each page scanning
            // for the keys after the previous page's last one
            const unsigned slots = static_cast<unsigned>(std::min<int64_t>(cnt, vg::kExactSlots));
            for (int off = 0; off < k; off += 64) {
                const int kk = std::min(64, k - off);
                const uint64_t *floor_keys = off ? min_keys : nullptr;
                if (dot)
                    VG_LAUNCH(vg::flat_exact_kernel<true>, dim3(ex_slices, slots), dim3(256), 0, st, idx->d_vectors, n,
                              dim, qp, todo, ex_slices, kk, fpartial, floor_keys, m0, mask_stride);
                else
                    VG_LAUNCH(vg::flat_exact_kernel<false>, dim3(ex_slices, slots), dim3(256), 0, st, idx->d_vectors, n,
                              dim, qp, todo, ex_slices, kk, fpartial, floor_keys, m0, mask_stride);
                VG_TRY(vg::launch_topk_merge(fpartial, cnt, ex_slices, kk, dot, fid, fsc, st, flags, always));
                if (k <= 64)
                    VG_LAUNCH(vg::flat_patch_kernel, dim3(ucnt), dim3(64), 0, st, flags, always, k, fid, fsc,
                              oid.ptr + q0 * k, osc.ptr + q0 * k);
                else
                    VG_LAUNCH(vg::flat_page_patch_kernel, dim3(ucnt), dim3(64), 0, st, flags, always, k, off, kk, dot, fid,
                              fsc, oid.ptr + q0 * k, osc.ptr + q0 * k, min_keys);
            }
        }
    }
    // queries whose scores may hold a NaN (a non-finite query value / row, an overflowing dot product): the reference's heap,
    // operation by operation (vg_cand_replay.hpp); every other query returns from this launch at once
    if (n > 0 && cand_replay)
        VG_TRY(vg::launch_cand_replay(vg::FlatF32Scorer{idx->d_vectors, idx->d_norm_max + 1, dim, dot, 0}, q.ptr, dim, n, nq, k, dot, mask, mask_stride,
                                      oid.ptr, osc.ptr, st));
    VG_TRY(oid.finish());
    VG_TRY(osc.finish());
    if (oid.on_host() || osc.on_host()) VG_HIP(hipStreamSynchronize(st));
    return VG_OK;
}

// The nomination + proof of vg_search_flat over MANY (queries, rows) problems in one set of launches: the partition-probed search
// of k_probe.hip.  pair_queries: the (query, probe) pairs' query vectors bucketed by partition ([pairs, dim]); groups[g]: partition
// g's pairs and rows; first_block_*: each group's first workgroup in the sample / main GEMM launch (device; the totals are not
// known to the host: the grids are the host's upper bounds, surplus workgroups leave at once).  Per pair: ids / scores [pairs, k]
// = its partition's k best rows by (score, row id) with exact scores, and fail[pair] = 1 where the proof does not hold (the caller
// answers those queries with the exact probe kernels).  k <= kGemmMaxK; fp32 rows, dim % 4 == 0, 16-byte aligned.
namespace vg {
// thresholds kept per pair: the 8th best of the 1/8 row sample passes ~64 rows — enough for the 64-candidate re-score (k <= 48);
// a larger k re-scores everything that passed and wants ~3k of them — ~5k under a bfloat16 nomination, whose proof needs the
// threshold 2^-7 (|q|^2 + |x|^2) above the k-th score: at 3k, random-normal rows of dim 768 in partitions of 8192 failed the
// proof for about every other query of 8 probes (the rank of a sampled threshold varies by 1 / sqrt(sel_k))
int probe_gemm_sel_k(int k, bool bf16) { return k <= kGemmMaxK ? 8 : std::min(64, std::max(8, ((bf16 ? 5 : 3) * k + 7) / 8)); }
// candidate keys per pair: ~64 rows fall below a pair's threshold (8th best of 1/8 of its partition); 2048 leaves 30x, and a
// batch of 65535 pairs holds 1 GiB of them (4096, the flat search's, would be 2)
constexpr int kProbeGemmCap = 2048;
// (scratch: one piece of the caller's arena — the caller holds the arena for the call — carved here; the layout in one place)
struct ProbeGemmLayout {
    size_t sc, partial, sid, thr, counts, cand, cid, csc, qbf, total;
    int sel_slices;
};
static ProbeGemmLayout probe_gemm_layout(int64_t pairs, int64_t ns_max, int k, int dim = 0 /* > 0: room for the pairs' queries in bfloat16 */)
{
    constexpr int kc = 64, cap = kProbeGemmCap;
    const int sel_k = probe_gemm_sel_k(k, dim > 0);
    ProbeGemmLayout l;
    l.sel_slices = static_cast<int>(std::min<int64_t>(8, std::max<int64_t>(1, ns_max / 1024)));
    size_t at = 0;
    auto piece = [&](size_t bytes) {
        const size_t o = at;
        at += (bytes + 255) & ~size_t(255);
        return o;
    };
    l.sc = piece(sizeof(float) * static_cast<size_t>(pairs) * ns_max);
    l.partial = piece(sizeof(uint64_t) * static_cast<size_t>(pairs) * l.sel_slices * sel_k);
    l.sid = piece(sizeof(uint32_t) * static_cast<size_t>(pairs) * sel_k);
    l.thr = piece(sizeof(float) * static_cast<size_t>(pairs) * sel_k);
    l.counts = piece(sizeof(int) * static_cast<size_t>(pairs));
    l.cand = piece(sizeof(uint64_t) * static_cast<size_t>(pairs) * cap);
    l.cid = piece(sizeof(uint32_t) * static_cast<size_t>(pairs) * kc);
    l.csc = piece(sizeof(float) * static_cast<size_t>(pairs) * kc);
    l.qbf = piece(sizeof(uint16_t) * static_cast<size_t>(pairs) * dim);
    l.total = at;
    return l;
}
size_t flat_probe_gemm_scratch_bytes(int64_t pairs, int64_t ns_max, int k, int bf16_dim) { return probe_gemm_layout(pairs, ns_max, k, bf16_dim).total; }

int32_t flat_probe_gemm(vg_index *idx, const float *pair_queries, int64_t pairs, const GemmGroup *groups, const int64_t *const first_block[4],
                        int ngroups, const int64_t grid[4], int sample_stride, int64_t ns_max, int k, uint32_t *pair_ids,
                        float *pair_scores, int *fail, char *scratch, const uint8_t *mask /* the batch's row filters, or null */,
                        const int64_t *mask_off /* [pairs]: byte offset of each bucketed pair's filter in `mask` */, hipStream_t st,
                        const uint16_t *rows_bf16, const float *rows_norms, ProbeNominated *nominated)
{
    // rows_bf16 != null: nominate on that bfloat16 row image (with its norms) instead of the fp32 rows, and leave the exact
    // re-score + proof to the caller: *nominated = where the per-pair thresholds / counts / 64 candidates are (the partition-probed
    // SQ8 scan, k_sq8.hip)
    // first_block / grid: [0] sample, [1] main launch of the 128-query tiles (groups of more than 64 pairs); [2], [3] the same of
    // the 64-query tiles (flat_gemm_dma32_grouped_kernel<.., 2>: a group of at most 64 pairs is one query tile, one workgroup per
    // row tile, HBM-bound — a 128-query tile would spend the matrix cores on padding)
    const bool dot = idx->metric != VG_METRIC_L2;
    const int dim = idx->dim;
    // (rows_bf16 == idx->d_vectors_bf16: the fp32 rows' own bf16 filter, vg_index_enable_bf16_filter — the nomination runs on it,
    // the re-score and the proof below stay, the proof's margin widened as in vg_search_flat)
    const bool bf16 = rows_bf16 != nullptr;
    const int kc = 64, cap = kProbeGemmCap, sel_k = probe_gemm_sel_k(k, bf16);
    const bool own_filter = bf16 && rows_bf16 == idx->d_vectors_bf16;
    const float *const queries_f32 = pair_queries;
    const int bdim = !bf16 ? 0 : own_filter ? idx->vectors_bf16_dim : idx->sq_bf16_dim;  // row length of the bfloat16 image
    const ProbeGemmLayout l = probe_gemm_layout(pairs, ns_max, k, bdim);
    const int sel_slices = l.sel_slices;
    const float *grows = idx->d_vectors, *gnorms = idx->d_norms;
    int gdim = idx->dim;
    if (bf16) {
        uint16_t *qbf = reinterpret_cast<uint16_t *>(scratch + l.qbf);
        VG_LAUNCH(f32_to_bf16_pad_kernel, dim3(static_cast<unsigned>((pairs * bdim + 255) / 256)), dim3(256), 0, st, pair_queries,
                  pairs, idx->dim, bdim, qbf);
        pair_queries = reinterpret_cast<const float *>(qbf);
        grows = reinterpret_cast<const float *>(rows_bf16);
        gnorms = rows_norms;
        gdim = bdim / 2;
    }
    float *sc = reinterpret_cast<float *>(scratch + l.sc), *thr = reinterpret_cast<float *>(scratch + l.thr);
    float *cand_sc = reinterpret_cast<float *>(scratch + l.csc);
    uint64_t *partial = reinterpret_cast<uint64_t *>(scratch + l.partial), *cand = reinterpret_cast<uint64_t *>(scratch + l.cand);
    uint32_t *sid = reinterpret_cast<uint32_t *>(scratch + l.sid), *cand_id = reinterpret_cast<uint32_t *>(scratch + l.cid);
    int *counts = reinterpret_cast<int *>(scratch + l.counts);
    const unsigned upairs = static_cast<unsigned>(pairs);
    // (a) a threshold per pair: the sel_k-th best score of every sample_stride-th row tile of its partition (columns no tile
    // writes — smaller partitions — hold +Inf)
    VG_LAUNCH(fill_f32_kernel, dim3(static_cast<unsigned>((pairs * ns_max + 255) / 256)), dim3(256), 0, st, sc, pairs * ns_max, INFINITY);
    {
        auto kern = bf16 ? (dot ? flat_gemm_dma_grouped_kernel<true, 1, true> : flat_gemm_dma_grouped_kernel<false, 1, true>)
                         : (dot ? flat_gemm_dma_grouped_kernel<true, 1> : flat_gemm_dma_grouped_kernel<false, 1>);
        VG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   static_cast<int>(kDmaLdsBytes)));
        if (grid[0])
            VG_LAUNCH(kern, dim3(static_cast<unsigned>(grid[0])), dim3(kGemmThreads), kDmaLdsBytes, st, groups, first_block[0], ngroups,
                      pair_queries, grows, gdim, gnorms, sc, sample_stride, ns_max, nullptr, 0, 0, nullptr, nullptr, 0, mask, mask_off);
        auto kern32 = bf16 ? (dot ? flat_gemm_dma32_grouped_kernel<true, 1, kProbeRB, true> : flat_gemm_dma32_grouped_kernel<false, 1, kProbeRB, true>)
                           : (dot ? flat_gemm_dma32_grouped_kernel<true, 1, kProbeRB> : flat_gemm_dma32_grouped_kernel<false, 1, kProbeRB>);
        VG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern32), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   static_cast<int>(g32_lds_bytes<kProbeRB>())));
        if (grid[2])
            VG_LAUNCH(kern32, dim3(static_cast<unsigned>(grid[2])), dim3(kGemmThreads), g32_lds_bytes<kProbeRB>(), st, groups, first_block[2], ngroups,
                      pair_queries, grows, gdim, gnorms, sc, sample_stride, ns_max, nullptr, 0, 0, nullptr, nullptr, 0, mask, mask_off);
    }
    VG_LAUNCH(flat_select_kernel, dim3(sel_slices, upairs), dim3(kSelThreads), 0, st, sc, ns_max, sel_slices, sel_k, partial);
    VG_TRY(launch_topk_merge(partial, pairs, sel_slices, sel_k, false, sid, thr, st));
    // (b) every element below its pair's threshold is appended to the pair's candidates
    VG_HIP(hipMemsetAsync(counts, 0, sizeof(int) * static_cast<size_t>(pairs), st));
    {
        ProfScope prof(idx->ctx, "flat_probe_gemm", st);
        auto kern = bf16 ? (dot ? flat_gemm_dma_grouped_kernel<true, 2, true> : flat_gemm_dma_grouped_kernel<false, 2, true>)
                         : (dot ? flat_gemm_dma_grouped_kernel<true, 2> : flat_gemm_dma_grouped_kernel<false, 2>);
        VG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   static_cast<int>(kDmaLdsBytes)));
        if (grid[1])
            VG_LAUNCH(kern, dim3(static_cast<unsigned>(grid[1])), dim3(kGemmThreads), kDmaLdsBytes, st, groups, first_block[1], ngroups,
                      pair_queries, grows, gdim, gnorms, nullptr, 1, 0, thr, sel_k, sel_k - 1, counts, cand, cap, mask, mask_off);
        auto kern32 = bf16 ? (dot ? flat_gemm_dma32_grouped_kernel<true, 2, kProbeRB, true> : flat_gemm_dma32_grouped_kernel<false, 2, kProbeRB, true>)
                           : (dot ? flat_gemm_dma32_grouped_kernel<true, 2, kProbeRB> : flat_gemm_dma32_grouped_kernel<false, 2, kProbeRB>);
        VG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern32), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   static_cast<int>(g32_lds_bytes<kProbeRB>())));
        if (grid[3])
            VG_LAUNCH(kern32, dim3(static_cast<unsigned>(grid[3])), dim3(kGemmThreads), g32_lds_bytes<kProbeRB>(), st, groups, first_block[3], ngroups,
                      pair_queries, grows, gdim, gnorms, nullptr, 1, 0, thr, sel_k, sel_k - 1, counts, cand, cap, mask, mask_off);
    }
    // (c) the kc best of them, (d) re-scored exactly, the k best, and the proof against everything not nominated
    // (k > kGemmMaxK: every appended row is re-scored — flat_verify_all / _sort, as in vg_search_flat)
    if (k <= kGemmMaxK) VG_LAUNCH(flat_pick_kernel, dim3(upairs), dim3(256), 0, st, cand, counts, cap, kc, cand_id, cand_sc);
    if (bf16 && !own_filter) {
        nominated->thr = thr;
        nominated->counts = counts;
        nominated->cand_id = cand_id;
        nominated->cand_sc = cand_sc;
        nominated->cap = cap;
        nominated->sel_k = sel_k;
        nominated->cand = cand;
        return VG_OK;
    }
    const float eps_extra = own_filter ? (dot ? 0.00390625f : 0.0078125f) * 1.02f : 0.0f;
    if (k > 64) {
        auto vk = dot ? flat_verify_sort_kernel<true> : flat_verify_sort_kernel<false>;
        VG_LAUNCH(vk, dim3(upairs), dim3(256), sizeof(uint64_t) * static_cast<size_t>(cap), st, idx->d_vectors, dim, queries_f32,
                  idx->d_norm_max, cand, counts, cap, k, pair_ids, pair_scores, fail, thr, sel_k, sel_k - 1, eps_extra);
    } else if (k > kGemmMaxK) {
        auto vk = dot ? flat_verify_all_kernel<true> : flat_verify_all_kernel<false>;
        VG_LAUNCH(vk, dim3(upairs), dim3(256), 0, st, idx->d_vectors, dim, queries_f32, idx->d_norm_max, cand, counts, cap, k, pair_ids,
                  pair_scores, fail, thr, sel_k, sel_k - 1, eps_extra);
    } else if (dot)
        VG_LAUNCH(flat_verify_kernel<true>, dim3(upairs), dim3(256), 0, st, idx->d_vectors, idx->n, dim, queries_f32, idx->d_norm_max,
                  cand_id, cand_sc, kc, k, pair_ids, pair_scores, fail, thr, sel_k, sel_k - 1, counts, cap, eps_extra);
    else
        VG_LAUNCH(flat_verify_kernel<false>, dim3(upairs), dim3(256), 0, st, idx->d_vectors, idx->n, dim, queries_f32, idx->d_norm_max,
                  cand_id, cand_sc, kc, k, pair_ids, pair_scores, fail, thr, sel_k, sel_k - 1, counts, cap, eps_extra);
    return VG_OK;
}
}  // namespace vg

// The nomination stage of the fused flat search over ANY bfloat16 row image with its norms: thresholds from a row sample, the
// bf16 MFMA GEMM appending what falls below them, the kc best per query — for a caller that re-scores and proves with its own
// exact distance (the SQ8 batch search, k_sq8.hip: rows = the dequantised codes rounded to bfloat16, norms = |x^|^2).  L2 scores.
// queries: cnt x dim fp32 (device), cnt <= 4096; rows_bf16: n x dim_pad (dim rounded up to a multiple of 64, the tail zeros).  Outputs (device, caller's):
// thr[cnt * sel_k] (the threshold is a query's last entry), counts[cnt], cand_id / cand_sc[cnt * 64] ascending (pick only).
namespace vg {
constexpr int kNomKc = 64, kNomCap = 4096, kNomStride = 64;
struct NominateLayout {
    size_t qbf, sc, partial, sid, cand, cwide, total;
    int sel_slices;
    int64_t ns;
};
static NominateLayout nominate_layout(int64_t cnt, int64_t n, int dim, int sel_k)
{
    NominateLayout l;
    const int64_t nt = (n + kGemmBN - 1) / kGemmBN, nst = (nt + kNomStride - 1) / kNomStride;
    l.ns = nst * kGemmBN;
    l.sel_slices = static_cast<int>(std::min<int64_t>(64, std::max<int64_t>(1, l.ns / 4096)));
    size_t at = 0;
    auto piece = [&](size_t bytes) {
        const size_t o = at;
        at += (bytes + 255) & ~size_t(255);
        return o;
    };
    l.qbf = piece(sizeof(uint16_t) * static_cast<size_t>(cnt) * dim);
    l.sc = piece(sizeof(float) * static_cast<size_t>(cnt) * l.ns);
    l.partial = piece(sizeof(uint64_t) * static_cast<size_t>(cnt) * l.sel_slices * sel_k);
    l.sid = piece(sizeof(uint32_t) * static_cast<size_t>(cnt) * sel_k);
    l.cand = piece(sizeof(uint64_t) * static_cast<size_t>(cnt) * kNomCap);
    l.cwide = piece(sizeof(int) * static_cast<size_t>(cnt) * kCountLine);
    l.total = at;
    return l;
}
size_t flat_nominate_bf16_scratch(int64_t cnt, int64_t n, int dim, int sel_k) { return nominate_layout(cnt, n, dim, sel_k).total; }

int32_t flat_nominate_bf16(vg_ctx *ctx, const uint16_t *rows_bf16, const float *norms, int64_t n, int dim, int dim_pad, const float *queries,
                           int64_t cnt, char *scratch, float *thr, int *counts, uint32_t *cand_id, float *cand_sc, hipStream_t st,
                           bool dot, const uint8_t *mask, int64_t mask_stride, int sel_k, bool pick, const uint64_t **cand_keys, int *cap,
                           const float *norm_max)
{
    // dot: scores are -q.x (the largest dot products first); mask: a row filter per query (mask + q * mask_stride) or for the
    // batch (stride 0) — rejected rows are left out of the sample and of the candidates, as in flat_search_masked
    // sel_k: thresholds kept per query (thr[cnt * sel_k], the last is the query's: ~64 * sel_k rows pass it); pick: the 64 best
    // appended rows into cand_id / cand_sc; *cand_keys / *cap: every appended key, cap per query (in `scratch`)
    // dim_pad: the image's row length (dim rounded up to whole 64-element K steps, the tail zeros); the queries are padded alike
    const NominateLayout l = nominate_layout(cnt, n, dim_pad, sel_k);
    uint16_t *qbf = reinterpret_cast<uint16_t *>(scratch + l.qbf);
    float *sc = reinterpret_cast<float *>(scratch + l.sc);
    uint64_t *partial = reinterpret_cast<uint64_t *>(scratch + l.partial), *cand = reinterpret_cast<uint64_t *>(scratch + l.cand);
    uint32_t *sid = reinterpret_cast<uint32_t *>(scratch + l.sid);
    const int64_t mt = (cnt + kGemmBM - 1) / kGemmBM, nt = (n + kGemmBN - 1) / kGemmBN, nst = l.ns / kGemmBN;
    VG_LAUNCH(f32_to_bf16_pad_kernel, dim3(static_cast<unsigned>((cnt * dim_pad + 255) / 256)), dim3(256), 0, st, queries, cnt, dim, dim_pad, qbf);
    const float *ga = reinterpret_cast<const float *>(qbf), *gb = reinterpret_cast<const float *>(rows_bf16);
    const int gdim = dim_pad / 2;
    if (n > kNomCap) {
        VG_TRY(launch_gemm<1>(dot, true, static_cast<unsigned>(mt * ((nst + 7) / 8) * 8), st,
                              {ga, cnt, gb, n, gdim, norms, sc, kNomStride, l.ns, nullptr, 0, 0, nullptr, nullptr, 0, mask, mask_stride}, true));
        VG_LAUNCH(flat_select_kernel, dim3(l.sel_slices, static_cast<unsigned>(cnt)), dim3(kSelThreads), 0, st, sc, l.ns, l.sel_slices,
                  sel_k, partial);
        VG_TRY(launch_topk_merge(partial, cnt, l.sel_slices, sel_k, false, sid, thr, st));
    } else {
        VG_LAUNCH(fill_f32_kernel, dim3(static_cast<unsigned>((cnt * sel_k + 255) / 256)), dim3(256), 0, st, thr, cnt * sel_k, INFINITY);
    }
    // (norm_max: the largest of `norms`, device)
    VG_LAUNCH(flat_thr_cap_kernel, dim3(static_cast<unsigned>(cnt)), dim3(64), 0, st, thr, sel_k, queries, dim, norm_max);
    VG_HIP(hipMemsetAsync(counts, 0, sizeof(int) * static_cast<size_t>(cnt), st));
    {
        ProfScope prof(ctx, "sq8_nominate_gemm", st);
        VG_TRY(launch_gemm<2>(dot, true, static_cast<unsigned>(mt * ((nt + 7) / 8) * 8), st,
                              {ga, cnt, gb, n, gdim, norms, nullptr, 1, 0, thr, sel_k, sel_k - 1, counts, cand, kNomCap, mask, mask_stride,
                               ctx->compute_units, reinterpret_cast<int *>(scratch + l.cwide)},
                              true));
    }
    if (pick) VG_LAUNCH(flat_pick_kernel, dim3(static_cast<unsigned>(cnt)), dim3(256), 0, st, cand, counts, kNomCap, kNomKc, cand_id, cand_sc);
    *cand_keys = cand;
    *cap = kNomCap;
    return VG_OK;
}
}  // namespace vg

VG_API int32_t vg_index_enable_bf16_filter(vg_index *idx, int32_t on, void *stream)
{
    VG_CHECK(idx, VG_ERR_INVALID_ARG, "vg_index_enable_bf16_filter: NULL index");
    VG_HIP(hipSetDevice(idx->ctx->device));
    hipStream_t st = vg::pick_stream(idx->ctx, stream);
    if (idx->d_vectors_bf16) {
        VG_HIP(hipStreamSynchronize(st));
        VG_HIP(hipFree(idx->d_vectors_bf16));
        idx->d_vectors_bf16 = nullptr;
    }
    if (!on) return VG_OK;
    VG_CHECK(idx->d_vectors, VG_ERR_NOT_READY, "vg_index_enable_bf16_filter: index has no fp32 vectors");
    // rows of whole K steps (2 * kGemmBK = 64 bfloat16): the dimensions from dim on are zeros, which add nothing to a dot product
    idx->vectors_bf16_dim = (idx->dim + 2 * vg::kGemmBK - 1) / (2 * vg::kGemmBK) * (2 * vg::kGemmBK);
    const int64_t count = idx->n * idx->vectors_bf16_dim;
    VG_HIP(hipMalloc(reinterpret_cast<void **>(&idx->d_vectors_bf16), static_cast<size_t>(count) * sizeof(uint16_t)));
    VG_LAUNCH(vg::f32_to_bf16_pad_kernel, dim3(static_cast<unsigned>((count + 255) / 256)), dim3(256), 0, st, idx->d_vectors,
              idx->n, idx->dim, idx->vectors_bf16_dim, idx->d_vectors_bf16);
    VG_HIP(hipStreamSynchronize(st));
    return VG_OK;
}

VG_API int32_t vg_index_flat_stats(vg_index *idx, int64_t *queries, int64_t *exhaustive, void *stream)
{
    VG_CHECK(idx, VG_ERR_INVALID_ARG, "vg_index_flat_stats: NULL index");
    VG_CHECK(idx->d_flat_stats, VG_ERR_INVALID_ARG, "vg_index_flat_stats: no fp32 vectors attached");
    VG_HIP(hipSetDevice(idx->ctx->device));
    hipStream_t st = vg::pick_stream(idx->ctx, stream);
    unsigned long long h[2] = {0, 0};
    VG_HIP(hipMemcpyAsync(h, idx->d_flat_stats, sizeof h, hipMemcpyDeviceToHost, st));
    VG_HIP(hipStreamSynchronize(st));
    if (queries) *queries = static_cast<int64_t>(h[0]);
    if (exhaustive) *exhaustive = static_cast<int64_t>(h[1]);
    return VG_OK;
}
