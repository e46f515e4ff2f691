// k_kmeans.hip — internal/kmeans (TrainKMeans, AssignPartition, FindClosestCentroids) plus the
// remaining batched L0 seams (SquaredL2Bounded, PqAdcLookup).
#include <algorithm>
#include <numeric>

#include "vg_device.hpp"
#include "vg_exact.hpp"
#include "vg_flat_gemm.hpp"
#include "vg_internal.hpp"

namespace vg {

__host__ __device__ inline uint64_t km_splitmix64(uint64_t x)
{
    x += 0x9e3779b97f4a7c15ULL;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL;
    return x ^ (x >> 31);
}
__host__ __device__ inline uint64_t km_rng(uint64_t seed, uint64_t a, uint64_t b, uint64_t c)
{
    uint64_t h = km_splitmix64(seed);
    h = km_splitmix64(h ^ a);
    h = km_splitmix64(h ^ b);
    h = km_splitmix64(h ^ c);
    return h;
}

// assignment (kmeans.go:54-99): 16 lanes per point, centroids visited in index order;
// SquaredL2Batch / DotBatch order (batch_avx512.c), strict comparison keeps the lowest index
// LIST: the points are list[0 .. *list_count) (the ones the MFMA nomination below could not decide), a block walks
// chunks of 16 list slots
template <bool DOT, bool LIST = false>
__global__ __launch_bounds__(256) void km_assign_kernel(const float *__restrict__ vectors, int64_t n, int dim,
                                                        const float *__restrict__ centroids, int k,
                                                        int32_t *__restrict__ assign, int *__restrict__ changed,
                                                        const int32_t *__restrict__ list = nullptr,
                                                        const int *__restrict__ list_count = nullptr)
{
    const Sub16 sub = Sub16::make(threadIdx.x);
    const int64_t total = LIST ? *list_count : n;
    for (int64_t slot = static_cast<int64_t>(blockIdx.x) * 16 + (threadIdx.x >> 4); slot < total;
         slot += static_cast<int64_t>(gridDim.x) * 16) {
    const int64_t i = LIST ? list[slot] : slot;
    const float *v = vectors + i * dim;
    int best = 0;
    float bd = exact_pair16<DOT, kBatch>(centroids, v, dim, sub);
    for (int c = 1; c < k; c++) {
        const float d = exact_pair16<DOT, kBatch>(centroids + static_cast<int64_t>(c) * dim, v, dim, sub);
        if (DOT ? (d > bd) : (d < bd)) {
            bd = d;
            best = c;
        }
    }
    if ((threadIdx.x & 15) == 0) {
        if (changed && assign[i] != best) *changed = 1;
        assign[i] = best;
    }
    }
}

// member lists in index order: one thread per cluster walks the assignment array
__global__ void km_members_kernel(const int32_t *__restrict__ assign, int64_t n, int k,
                                  const int64_t *__restrict__ offsets, int64_t *__restrict__ members)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= k) return;
    int64_t at = offsets[c];
    for (int64_t i = 0; i < n; i++)
        if (assign[i] == c) members[at++] = i;
}

__global__ void km_count_kernel(const int32_t *__restrict__ assign, int64_t n, int k, int64_t *__restrict__ counts)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= k) return;
    int64_t cnt = 0;
    for (int64_t i = 0; i < n; i++) cnt += assign[i] == c;
    counts[c] = cnt;
}

// ---- the same assignment with the point rows held in registers (dim = 64 * NBLK <= 1024) ----
// A 16-lane group owns kKmRows points (16 float4 per lane and point); the centroids go through LDS
// kKmTile at a time and every LDS read serves all of the group's points.  With dim % 64 == 0 the
// batch kernel has no 16-wide tail, so the order is the 4x16 accumulators + reduce_add tree of
// vg_exact.hpp; the two halves of a float4 go through the packed fp32 ops, each lane the IEEE op of
// its own accumulator chain.
constexpr int kKmRows = 2;
constexpr int kKmTile = 8;
typedef float km_f2 __attribute__((ext_vector_type(2)));
typedef float km_f4 __attribute__((ext_vector_type(4)));  // whole-register copies (HIP's float4 is a union struct)

template <int DIM, int STAGE>
__device__ __forceinline__ void km_fetch_tile(km_f4 (&stage)[STAGE], const float *__restrict__ centroids, int c0, int k,
                                              int tid)
{
    const int cnt4 = (k - c0 < kKmTile ? k - c0 : kKmTile) * (DIM / 4);
    const km_f4 *src = reinterpret_cast<const km_f4 *>(centroids + static_cast<int64_t>(c0) * DIM);
#pragma unroll
    for (int u = 0; u < STAGE; u++) {
        const int t = tid + u * 256;
        stage[u] = src[t < cnt4 ? t : cnt4 - 1];
    }
}

template <bool DOT, int NBLK, bool LIST = false>
__global__ __launch_bounds__(256) void km_assign_regs_kernel(const float *__restrict__ vectors, int64_t n,
                                                             const float *__restrict__ centroids, int k,
                                                             int32_t *__restrict__ assign, int *__restrict__ changed,
                                                             const int32_t *__restrict__ list = nullptr,
                                                             const int *__restrict__ list_count = nullptr,
                                                             float *__restrict__ range_d = nullptr, int32_t *__restrict__ range_i = nullptr,
                                                             int range_len = 0)
{
    constexpr int dim = NBLK * 64;
    __shared__ __attribute__((aligned(16))) float ctile[kKmTile * dim];
    // LIST with range_len > 0: the listed points are few (~1 % of a pass) and the walk over the k centroids is one chain of
    // dependent tile fetches per chunk — 272 chunks on 256 CUs took as long as the chain (187 us at k = 122).  blockIdx.y cuts
    // the centroids into ranges of range_len (a multiple of kKmTile); a workgroup leaves its range's best (value, centroid) per
    // slot in range_d / range_i [slot * gridDim.y + range] and km_list_combine_kernel takes the first best over the ranges in
    // order — the strict comparison of the loop below, so a tie still goes to the lowest centroid.
    const bool ranged = LIST && range_len > 0;
    const int c_lo = ranged ? static_cast<int>(blockIdx.y) * range_len : 0;
    const int c_hi = ranged ? (c_lo + range_len < k ? c_lo + range_len : k) : k;
    if (c_lo >= c_hi) return;
    const int tid = threadIdx.x;
    const Sub16 sub = Sub16::make(tid);
    // LIST: slots of the list instead of rows; a block walks chunks of 16 * kKmRows slots (uniform trip count: the
    // barriers below are reached by every thread)
    const int64_t total = LIST ? *list_count : n;
    for (int64_t chunk = blockIdx.x; chunk * (16 * kKmRows) < total; chunk += gridDim.x) {
    const int64_t i0 = (chunk * 16 + (tid >> 4)) * kKmRows;
    int64_t row_of[kKmRows];
    float4 rr[kKmRows][NBLK];
#pragma unroll
    for (int p = 0; p < kKmRows; p++) {
        const int64_t slot = i0 + p < total ? i0 + p : total - 1;
        const int64_t i = LIST ? list[slot] : slot;
        row_of[p] = i;
        const float4 *r4 = reinterpret_cast<const float4 *>(vectors + i * dim) + sub.f4;
#pragma unroll
        for (int e = 0; e < NBLK; e++) rr[p][e] = r4[e * 16];
    }
    float bd[kKmRows];
    int best[kKmRows];
#pragma unroll
    for (int p = 0; p < kKmRows; p++) {
        bd[p] = 0.0f;
        best[p] = 0;
    }
    constexpr int tile4 = kKmTile * dim / 4;         // float4 per full tile
    constexpr int kStage = (tile4 + 255) / 256;      // ... and per thread
    // the next tile travels through registers while the current one is scored; a ragged last tile
    // re-reads its final float4 instead of branching (those slots are never scored)
    km_f4 stage[kStage];
    km_fetch_tile<dim, kStage>(stage, centroids, c_lo, c_hi, tid);
    for (int c0 = c_lo; c0 < c_hi; c0 += kKmTile) {
        __syncthreads();  // the previous tile is no longer read
#pragma unroll
        for (int u = 0; u < kStage; u++) {
            const int t = tid + u * 256;
            if (tile4 % 256 == 0 || t < tile4) reinterpret_cast<km_f4 *>(ctile)[t] = stage[u];
        }
        __syncthreads();
        km_fetch_tile<dim, kStage>(stage, centroids, c0 + kKmTile < c_hi ? c0 + kKmTile : c0, c_hi, tid);  // last: unused
        const int cnt = c_hi - c0 < kKmTile ? c_hi - c0 : kKmTile;
        for (int cc = 0; cc < cnt; cc++) {
            const float4 *q4 = reinterpret_cast<const float4 *>(ctile + cc * dim) + sub.f4;
            km_f2 acc[kKmRows][2];
#pragma unroll
            for (int p = 0; p < kKmRows; p++) {
                acc[p][0] = km_f2{0.0f, 0.0f};
                acc[p][1] = km_f2{0.0f, 0.0f};
            }
            // the centroid's NBLK float4 are requested together (one LDS round trip per centroid
            // instead of one per 64-float block)
            float4 av4[NBLK];
#pragma unroll
            for (int e = 0; e < NBLK; e++) av4[e] = q4[e * 16];
#pragma unroll
            for (int e = 0; e < NBLK; e++) {
                const float4 a = av4[e];
                const km_f2 alo = {a.x, a.y}, ahi = {a.z, a.w};
#pragma unroll
                for (int p = 0; p < kKmRows; p++) {
                    const km_f2 blo = {rr[p][e].x, rr[p][e].y}, bhi = {rr[p][e].z, rr[p][e].w};
                    if (DOT) {
                        acc[p][0] = __builtin_elementwise_fma(alo, blo, acc[p][0]);
                        acc[p][1] = __builtin_elementwise_fma(ahi, bhi, acc[p][1]);
                    } else {
                        const km_f2 dlo = alo - blo, dhi = ahi - bhi;
                        acc[p][0] = __builtin_elementwise_fma(dlo, dlo, acc[p][0]);
                        acc[p][1] = __builtin_elementwise_fma(dhi, dhi, acc[p][1]);
                    }
                }
            }
            const int c = c0 + cc;
#pragma unroll
            for (int p = 0; p < kKmRows; p++) {
                const float av[4] = {acc[p][0].x, acc[p][0].y, acc[p][1].x, acc[p][1].y};
                float b[4];
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const float h = dpp_partner_add<kDppRowHalfMirror>(av[t]);
                    const float s2 = dpp_partner_add<kDppRowMirror>(h);
                    const float x2 = dpp_partner_add<kDppQuadXor2>(s2);
                    b[t] = dpp_partner_add<kDppQuadXor1>(x2);
                }
                const float total = (b[0] + b[2]) + (b[1] + b[3]);
                if (c == c_lo || (DOT ? (total > bd[p]) : (total < bd[p]))) {
                    bd[p] = total;
                    best[p] = c;
                }
            }
        }
    }
    if ((tid & 15) == 0) {
#pragma unroll
        for (int p = 0; p < kKmRows; p++)
            if (i0 + p < total) {
                if (ranged) {
                    range_d[(i0 + p) * gridDim.y + blockIdx.y] = bd[p];
                    range_i[(i0 + p) * gridDim.y + blockIdx.y] = best[p];
                } else {
                    if (changed && assign[row_of[p]] != best[p]) *changed = 1;
                    assign[row_of[p]] = best[p];
                }
            }
    }
    }
}

// the listed points' assignment from their ranges' bests (km_assign_regs_kernel<.., LIST> with ranges): the first best in
// range order under the loop's own strict comparison
template <bool DOT>
__global__ __launch_bounds__(256) void km_list_combine_kernel(const int32_t *__restrict__ list, const int *__restrict__ list_count,
                                                              const float *__restrict__ range_d, const int32_t *__restrict__ range_i, int ranges,
                                                              int32_t *__restrict__ assign, int *__restrict__ changed)
{
    const int64_t total = *list_count;
    for (int64_t slot = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; slot < total; slot += static_cast<int64_t>(gridDim.x) * 256) {
        float bd = range_d[slot * ranges];
        int best = range_i[slot * ranges];
        for (int r = 1; r < ranges; r++) {
            const float d = range_d[slot * ranges + r];
            if (DOT ? (d > bd) : (d < bd)) {
                bd = d;
                best = range_i[slot * ranges + r];
            }
        }
        const int64_t row = list[slot];
        if (changed && assign[row] != best) *changed = 1;
        assign[row] = best;
    }
}

// ---- member lists in index order: a stable counting sort of the point ids by cluster ----------
// Parts of kKmPartRows consecutive points; (1) per-part histograms, (2) cluster totals, offsets and
// every part's starting rank inside each cluster, (3) one wave per part places its points, ranks
// inside a 64-point chunk from ballots over the bits of the cluster id.
constexpr int kKmLdsK = 4096;       // clusters whose per-part counters fit LDS; above: the walk-all kernels
constexpr int kKmPartRows = 2048;

__global__ __launch_bounds__(256) void km_hist_kernel(const int32_t *__restrict__ assign, int64_t n, int k,
                                                      int32_t *__restrict__ hist)
{
    __shared__ int32_t h[kKmLdsK];
    const int part = blockIdx.x;
    for (int c = threadIdx.x; c < k; c += 256) h[c] = 0;
    __syncthreads();
    const int64_t r0 = static_cast<int64_t>(part) * kKmPartRows;
    const int64_t r1 = r0 + kKmPartRows < n ? r0 + kKmPartRows : n;
    for (int64_t i = r0 + threadIdx.x; i < r1; i += 256) atomicAdd(&h[assign[i]], 1);
    __syncthreads();
    for (int c = threadIdx.x; c < k; c += 256) hist[static_cast<int64_t>(part) * k + c] = h[c];
}

// hist[part][c] becomes the number of cluster-c points in earlier parts, counts[c] the cluster's size: one workgroup per
// cluster scans its column of the histogram 256 parts at a time (the single-workgroup form walked parts x k counters
// through one thread per cluster: 120 us per iteration at 489 parts)
__global__ __launch_bounds__(256) void km_scan_parts_kernel(int32_t *__restrict__ hist, int parts, int k,
                                                            int64_t *__restrict__ counts)
{
    __shared__ int32_t sc[256];
    const int c = blockIdx.x, tid = threadIdx.x;
    int32_t carry = 0;
    for (int p0 = 0; p0 < parts; p0 += 256) {
        const int p = p0 + tid;
        const int32_t v = p < parts ? hist[static_cast<int64_t>(p) * k + c] : 0;
        sc[tid] = v;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {  // inclusive scan
            const int32_t add = tid >= off ? sc[tid - off] : 0;
            __syncthreads();
            sc[tid] += add;
            __syncthreads();
        }
        if (p < parts) hist[static_cast<int64_t>(p) * k + c] = carry + sc[tid] - v;
        const int32_t total = sc[255];
        __syncthreads();
        carry += total;
    }
    if (tid == 0) counts[c] = carry;
}

// offsets[c] = number of points in clusters before c
__global__ __launch_bounds__(256) void km_offsets_kernel(int k, const int64_t *__restrict__ counts, int64_t *__restrict__ offsets)
{
    __shared__ int64_t seg[256];
    const int tid = threadIdx.x;
    const int per = (k + 255) / 256;
    const int cb = tid * per, ce = cb + per < k ? cb + per : k;
    int64_t mine = 0;
    for (int c = cb; c < ce; c++) mine += counts[c];
    seg[tid] = mine;
    __syncthreads();
    if (tid == 0) {
        int64_t run = 0;
        for (int t = 0; t < 256; t++) {
            const int64_t v = seg[t];
            seg[t] = run;
            run += v;
        }
    }
    __syncthreads();
    int64_t run = seg[tid];
    for (int c = cb; c < ce; c++) {
        offsets[c] = run;
        run += counts[c];
    }
}

__global__ __launch_bounds__(64) void km_scatter_kernel(const int32_t *__restrict__ assign, int64_t n, int k, int kbits,
                                                        const int32_t *__restrict__ hist,
                                                        const int64_t *__restrict__ offsets,
                                                        int64_t *__restrict__ members)
{
    __shared__ int32_t base[kKmLdsK];
    const int part = blockIdx.x, lane = threadIdx.x;
    for (int c = lane; c < k; c += 64) base[c] = hist[static_cast<int64_t>(part) * k + c];
    __syncthreads();
    const int64_t r0 = static_cast<int64_t>(part) * kKmPartRows;
    const int64_t r1 = r0 + kKmPartRows < n ? r0 + kKmPartRows : n;
    for (int64_t c0 = r0; c0 < r1; c0 += 64) {
        const int64_t i = c0 + lane;
        const bool active = i < r1;
        const int32_t key = active ? assign[i] : 0;
        uint64_t peers = __ballot(active);
        for (int bit = 0; bit < kbits; bit++) {
            const bool set = (key >> bit) & 1;
            const uint64_t bb = __ballot(set);
            peers &= set ? bb : ~bb;
        }
        const int rank = __popcll(peers & ((1ull << lane) - 1ull));
        if (active) members[offsets[key] + base[key] + rank] = i;
        __syncthreads();
        if (active && rank == 0) base[key] += __popcll(peers);
        __syncthreads();
    }
}

// update (kmeans.go:107-135): per (cluster, coordinate) the sum runs over the members in index
// order (= the reference's single loop over i), then sums * (1/count)
// The loads run kKmAhead members ahead of the additions in a ring of registers refilled kKmGroup at a time (statically
// indexed: the compiler's vmcnt waits then keep kKmAhead - kKmGroup loads in flight at every addition); a member list is
// a chain of dependent loads (member id -> row), so depth, not bandwidth, sets the pace: 16 in flight ran at 2.1 TB/s.
#ifndef VG_KM_AHEAD
#define VG_KM_AHEAD 64
#endif
constexpr int kKmAhead = VG_KM_AHEAD;
constexpr int kKmGroup = 8;
__global__ __launch_bounds__(256) void km_update_kernel(const float *__restrict__ vectors, int64_t n, int dim,
                                                        int k, int iter, uint64_t seed,
                                                        const int64_t *__restrict__ counts,
                                                        const int64_t *__restrict__ offsets,
                                                        const int64_t *__restrict__ members,
                                                        float *__restrict__ centroids)
{
    const int c = blockIdx.y;
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= dim) return;
    const int64_t cnt = counts[c];
    float *dst = centroids + static_cast<int64_t>(c) * dim + d;
    if (cnt > 0) {
        const int64_t *mem = members + offsets[c];
        const float *col = vectors + d;
        float sum = 0.0f;
        int64_t j = 0;
        if (cnt >= kKmAhead) {
            float x[kKmAhead];
#pragma unroll
            for (int u = 0; u < kKmAhead; u++) x[u] = col[mem[u] * dim];
            // members j .. j + kKmAhead - 1 are in x[]; each step adds kKmGroup of them in order and refills their slots
            for (; j + 2 * kKmAhead <= cnt; j += kKmAhead) {
#pragma unroll
                for (int g = 0; g < kKmAhead; g += kKmGroup) {
#pragma unroll
                    for (int u = 0; u < kKmGroup; u++) sum += x[g + u];
#pragma unroll
                    for (int u = 0; u < kKmGroup; u++) x[g + u] = col[mem[j + kKmAhead + g + u] * dim];
                }
            }
#pragma unroll
            for (int u = 0; u < kKmAhead; u++) sum += x[u];
            j += kKmAhead;
        }
        for (; j < cnt; j++) sum += col[mem[j] * dim];
        const float scale = 1.0f / static_cast<float>(cnt);
        *dst = sum * scale;
    } else {
        const int64_t idx = static_cast<int64_t>(km_rng(seed, 0, 2 + static_cast<uint64_t>(iter), c) %
                                                 static_cast<uint64_t>(n));
        *dst = vectors[idx * dim + d];
    }
}

__global__ void km_gather_rows_kernel(const float *__restrict__ vectors, int dim, const int64_t *__restrict__ rows,
                                      int k, float *__restrict__ out)
{
    const int64_t gid = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (gid >= static_cast<int64_t>(k) * dim) return;
    out[gid] = vectors[rows[gid / dim] * dim + gid % dim];
}

__global__ __launch_bounds__(256) void bounded_batch_kernel(const float *__restrict__ query,
                                                            const float *__restrict__ targets, int dim, int64_t n,
                                                            const float *__restrict__ bounds, int64_t n_bounds,
                                                            float *__restrict__ dist, int32_t *__restrict__ exceeded)
{
    const Sub16 sub = Sub16::make(threadIdx.x);
    const int64_t groups = static_cast<int64_t>(gridDim.x) * 16;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * 16 + (threadIdx.x >> 4); i < n; i += groups) {
        float v = exact_pair16<false, kBounded, true>(targets + i * dim, query, dim, sub);  // (read once: nontemporal)
        const float b = bounds[n_bounds == 1 ? 0 : i];
        const bool over = v > b;  // partial sums never decrease: some block's partial > bound <=> the full sum is
        // the reference returns the PARTIAL total of the block where it stopped: replayed for the pairs that exceed
        if (over) v = exact_l2_bounded_partial16(targets + i * dim, query, dim, sub, b, v);
        if ((threadIdx.x & 15) == 0) {
            dist[i] = v;
            exceeded[i] = over ? 1 : 0;
        }
    }
}

// pqAdcLookupAvx512 (floats_avx512.c:135-167): thread per code row
__global__ void adc_lookup_batch_kernel(const float *__restrict__ table, const uint8_t *__restrict__ codes, int m,
                                        int64_t n, float *__restrict__ out)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t *c = codes + i * m;
    float acc[16];
#pragma unroll
    for (int l = 0; l < 16; l++) acc[l] = 0.0f;
    int j = 0;
    for (; j + 16 <= m; j += 16) {
#pragma unroll
        for (int l = 0; l < 16; l++) acc[l] = acc[l] + table[(j + l) * 256 + c[j + l]];
    }
    float total = reduce16_regs(acc);
    for (; j < m; j++) total = total + table[j * 256 + c[j]];
    out[i] = total;
}

// The same sums as a streaming scan (m % 16 == 0, table <= 128 KiB): the kernel above reads a code byte by byte at an
// m-byte stride and every table entry from global memory (0.86 TB/s of codes at m = 96).  Here the table sits in LDS
// (m KiB, one persistent workgroup per CU), a wave takes 64 rows — one contiguous 64 m-byte block, read as whole lines —
// and turns them through its LDS (row stride 16 x odd: conflict-free ds_read_b128); each lane then walks ITS code:
// acc[l] += table[(j + l) * 256 + code[j + l]] for j ascending, the reduce tree — the arithmetic of pqAdcLookupAvx512.
// (The re-tiled index scan, pq_adc_scan_kernel, pre-rotates the codes per lane so that its lookups avoid most bank
// conflicts; a row-major batch cannot, and runs at the LDS rate of random 4-byte reads.)
template <int M16>  // m / 16 when known at compile time, 0 = any
__global__ __launch_bounds__(512) void adc_lookup_batch_lds_kernel(const float *__restrict__ table, const uint8_t *__restrict__ codes,
                                                                   int m_rt, int stride, int64_t n, float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char adc_smem[];
    const int m = M16 ? M16 * 16 : m_rt;
    const int m16 = M16 ? M16 : m_rt >> 4;
    float *lut = reinterpret_cast<float *>(adc_smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, waves = blockDim.x >> 6;
    {
        const float4 *src = reinterpret_cast<const float4 *>(table);
        float4 *dst = reinterpret_cast<float4 *>(lut);
        for (int i = tid; i < m * 64; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();
    unsigned char *stage = adc_smem + static_cast<size_t>(m) * 1024 + static_cast<size_t>(wave) * 64 * stride;
    const int64_t n_tiles = (n + 63) / 64;
    const int units = 64 * m16;  // 16-byte units of a tile
    const int64_t tile_step = static_cast<int64_t>(gridDim.x) * waves;
    // unit e of tile t (past n: the tile's last valid unit again — valid memory, values unused)
    auto get = [&](int64_t t, int e) {
        const int64_t row0 = t * 64;
        const int64_t last_unit = (n - row0 < 64 ? n - row0 : 64) * m16 - 1;
        return load_stream(reinterpret_cast<const uint4 *>(codes + row0 * m + (e <= last_unit ? e : last_unit) * 16));  // read once
    };
    auto put = [&](int e, const uint4 u) {
        const int r = e / m16, part = e - r * m16;
        *reinterpret_cast<uint4 *>(stage + r * stride + part * 16) = u;
    };
    int64_t tile = static_cast<int64_t>(blockIdx.x) * waves + wave;
    if (tile >= n_tiles) return;
    // m = 96: a lane's six units of the NEXT tile are requested before this tile's lookups (8 waves per CU do not hide an
    // HBM round trip per tile by themselves: the waves waited 72 % of their cycles)
    uint4 p0, p1, p2, p3, p4, p5;
    if (M16 == 6) {
        p0 = get(tile, lane);
        p1 = get(tile, lane + 64);
        p2 = get(tile, lane + 128);
        p3 = get(tile, lane + 192);
        p4 = get(tile, lane + 256);
        p5 = get(tile, lane + 320);
    }
    for (; tile < n_tiles; tile += tile_step) {
        const int64_t row0 = tile * 64;
        if (M16 == 6) {
            put(lane, p0);
            put(lane + 64, p1);
            put(lane + 128, p2);
            put(lane + 192, p3);
            put(lane + 256, p4);
            put(lane + 320, p5);
            const int64_t tn = tile + tile_step < n_tiles ? tile + tile_step : tile;
            p0 = get(tn, lane);
            p1 = get(tn, lane + 64);
            p2 = get(tn, lane + 128);
            p3 = get(tn, lane + 192);
            p4 = get(tn, lane + 256);
            p5 = get(tn, lane + 320);
        } else {
            for (int e0 = 0; e0 < units; e0 += 64) {
                const uint4 u = get(tile, e0 + lane);
                if (e0 + lane < units) put(e0 + lane, u);
            }
        }
        float acc[16];
#pragma unroll
        for (int l = 0; l < 16; l++) acc[l] = 0.0f;
        for (int g = 0; g < m16; g++) {
            const uint4 c = *reinterpret_cast<const uint4 *>(stage + lane * stride + g * 16);
            const uint32_t w[4] = {c.x, c.y, c.z, c.w};
            const float *row = lut + g * 16 * 256;
            float t[16];
#pragma unroll
            for (int l = 0; l < 16; l++) t[l] = row[l * 256 + ((w[l >> 2] >> (8 * (l & 3))) & 0xFFu)];
#pragma unroll
            for (int l = 0; l < 16; l++) acc[l] = acc[l] + t[l];
        }
        const float total = reduce16_regs(acc);
        if (row0 + lane < n) out[row0 + lane] = total;
    }
}


// ---- assignment by MFMA nomination + exact decision ------------------------------------------------------------------
// The assignment pass is a [points x centroids x dim] product: n * k * dim fused multiply-adds on the matrix cores
// instead of n * k * dim (sub, fma) pairs plus a 16-lane reduction per pair on the vector ALU.  The matrix cores do
// not add in squaredL2BatchAvx512's order, so their scores only NOMINATE:
//   1. km_gemm_kernel     s~(x, c) = |c|^2 - 2 x.c  (L2; -x.c for Dot / Cosine) for every pair — the pipeline of
//                         flat_gemm_dma_kernel (vg_flat_gemm.hpp: LDS-DMA tiles, swizzled image, v_mfma_f32_32x32x2_f32)
//                         with the centroids as the 128-row A tile and 128 points as the B tile; the epilogue keeps, per
//                         point and centroid tile, the smallest score, its centroid and the second smallest score
//   2. km_decide_kernel   merges the centroid tiles and compares the gap between the two smallest scores with a bound on
//                         |s~ - s| + |D_ref - D| (below).  A gap above the bound PROVES that the reference's own
//                         arithmetic picks the same centroid: the assignment is written.  Everything else — near ties,
//                         duplicate centroids, non-finite input — goes on a list
//   3. km_assign_*_kernel<LIST>  the reference-order kernels above over the listed points (all k centroids each)
// so every assignment is either proven equal to, or computed by, the reference's arithmetic.
//
// The bound.  u = 2^-24, X = |x|^2, C = max_c |c|^2, D = |x - c|^2 = X + s, s = |c|^2 - 2 x.c (exact values).
//   reference:  D_ref = D (1 + t), |t| <= g_r, r = dim/64 + 32: one rounding in x_i - c_i, one per fused multiply-add
//               of the accumulator chain (dim/64 of them + up to 4 16-wide steps), 6 tree additions, < 16 scalar steps
//   matrix:     |s~ - s| <= E = (2 dim + 16) u (2 sqrt(X C) + C): at most two roundings per product of the dot
//               product (the guide: the fp32 MFMA is an fmaf chain), the norm |c|^2 (dim roundings), one final fma
//   c* = the reference's choice, cm = the smallest score:  D_ref(c*) <= D_ref(cm)
//               =>  s(c*) - s(cm) <= g_r (D(c*) + D(cm)) <= 2 g_r (sqrt X + sqrt C)^2
//               =>  s~(c*) <= s~(cm) + 2 E + 2 g_r (sqrt X + sqrt C)^2 =: s~(cm) + margin
//   A second-smallest score above s~(cm) + margin therefore leaves cm as the only possible c*.  Dot / Cosine: the same
//   with D replaced by x.c; the L2 margin is the larger of the two and is used for both.  The margin is evaluated in
//   fp32 with a 5 % allowance for its own rounding and for the rounding of X and C (computed in any order).
struct KmPart {   // per (centroid tile, point): the two smallest scores with their centroids, and the third smallest
    float m1, m2, m3;
    int32_t i1, i2;
    int32_t pad[3];
};
static_assert(sizeof(KmPart) == 32, "two 16-byte stores per point");

// (m1, i1) <= (m2, i2) <= m3 of the union of two such triples (ties between scores: any order — a tie is never decided
// by the scores)
__device__ __forceinline__ void km_merge3(float &m1, int &i1, float &m2, int &i2, float &m3, float o1, int oi1, float o2,
                                          int oi2, float o3)
{
    // the smallest three of {m1, m2, m3, o1, o2, o3}, each list sorted
    const bool a = o1 < m1;                       // which list holds the smallest
    const float f1 = a ? o1 : m1;
    const int fi1 = a ? oi1 : i1;
    const float x2 = a ? o2 : m2, y1 = a ? m1 : o1;   // next of the winner's list, head of the other
    const int xi2 = a ? oi2 : i2, yi1 = a ? i1 : oi1;
    const float x3 = a ? o3 : m3, y2 = a ? m2 : o2;
    const bool b = y1 < x2;                       // second smallest: the other list's head or the winner's next
    const float f2 = b ? y1 : x2;
    const int fi2 = b ? yi1 : xi2;
    const float f3 = b ? fminf(x2, y2) : fminf(x3, y1);
    m1 = f1; i1 = fi1; m2 = f2; i2 = fi2; m3 = f3;
}

// |row|^2 in any order (feeds the bound only): one wave per row, 16-byte loads
__global__ __launch_bounds__(256) void km_norms_kernel(const float *__restrict__ v, int64_t n, int dim, float *__restrict__ out)
{
    const int64_t row = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const int lane = threadIdx.x & 63;
    const float4 *r4 = reinterpret_cast<const float4 *>(v + row * dim);
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    for (int j = lane; j < dim / 4; j += 64) {
        const float4 x = load_stream(r4 + j);
        s0 = __builtin_fmaf(x.x, x.x, s0);
        s1 = __builtin_fmaf(x.y, x.y, s1);
        s2 = __builtin_fmaf(x.z, x.z, s2);
        s3 = __builtin_fmaf(x.w, x.w, s3);
    }
    float s = (s0 + s1) + (s2 + s3);
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) out[row] = s;
}

// rows of fp32 -> rows of bfloat16 splits, 2 dim elements each: [hi | lo] (see km_gemm_kernel<.., BF16>).  hi =
// round-to-nearest-even of the value, lo = the same of the (exact) remainder.
__device__ __forceinline__ uint16_t km_bf16_rne(float x)
{
    const uint32_t b = __float_as_uint(x);
    return static_cast<uint16_t>((b + 0x7FFFu + ((b >> 16) & 1u)) >> 16);
}
template <bool CENTROID>
__global__ __launch_bounds__(256) void km_split_kernel(const float *__restrict__ src, int64_t rows, int dim,
                                                       uint16_t *__restrict__ dst)
{
    // four elements per thread: one 16-byte load, two 8-byte stores (dim % 64 == 0: a group never straddles a row)
    const int64_t g = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const int64_t e = g * 4;
    if (e >= rows * dim) return;
    const int64_t r = e / dim;
    const int j = static_cast<int>(e - r * dim);
    const float4 x = load_stream(reinterpret_cast<const float4 *>(src + e));
    const float xs[4] = {x.x, x.y, x.z, x.w};
    uint16_t hi[4], lo[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        hi[t] = km_bf16_rne(xs[t]);
        lo[t] = km_bf16_rne(xs[t] - __uint_as_float(static_cast<uint32_t>(hi[t]) << 16));
    }
    const uint2 h2 = make_uint2(hi[0] | (static_cast<uint32_t>(hi[1]) << 16), hi[2] | (static_cast<uint32_t>(hi[3]) << 16));
    const uint2 l2 = make_uint2(lo[0] | (static_cast<uint32_t>(lo[1]) << 16), lo[2] | (static_cast<uint32_t>(lo[3]) << 16));
    uint16_t *o = dst + r * 2 * dim + j;
    *reinterpret_cast<uint2 *>(o) = h2;
    *reinterpret_cast<uint2 *>(o + dim) = l2;
}

// the centroids' side of the epilogue: cadd[c] = |c|^2 (L2) or 0 (Dot) for c < k, +Inf for the padding of the last tile;
// *cmax_bits = max |c|^2 as float bits (non-negative floats order like their bits; zeroed by the caller), +Inf when a
// centroid holds a non-finite value (then nothing is decided by the matrix scores).  One wave per centroid.
__global__ __launch_bounds__(256) void km_cent_norms_kernel(const float *__restrict__ cent, int k, int dim, bool dot,
                                                            float *__restrict__ cadd, int kpad, int *__restrict__ cmax_bits)
{
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= kpad) return;
    float s = 0.0f;
    if (c < k)
        for (int j = lane; j < dim; j += 64) {
            const float x = cent[static_cast<int64_t>(c) * dim + j];
            s = __builtin_fmaf(x, x, s);
        }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (!(s <= 3.0e38f)) s = INFINITY;  // NaN or overflow
    if (lane == 0) {
        cadd[c] = c < k ? (dot ? 0.0f : s) : INFINITY;
        atomicMax(cmax_bits, __float_as_int(s));
    }
}

// BF16: both operands are rows of bfloat16 SPLITS [hi | lo] (km_split_kernel: hi = the value rounded to bfloat16, lo = the
// remainder rounded to bfloat16); `dim` then counts the 4-byte words of such a row (= the row's dimension).  The K loop takes
// every 64-element chunk three times — x_hi c_hi, x_hi c_lo, x_lo c_hi — with the 16x faster v_mfma_f32_32x32x16_bf16, so the
// accumulators add up x_hi c_hi + x_hi c_lo + x_lo c_hi.  (r05 first stored points as [hi | lo | hi] and centroids as
// [hi | hi | lo], one pass over 3 dim elements: 6 bytes per element from HBM; now 4 — the second use of a point chunk's hi
// half comes from L2, one K step after the first.)  What the split drops is x_lo c_lo and the second remainders: at most
// 3 * 2^-16 |x_i| |c_i| per element, which enters the bound of km_decide_kernel in place of the fp32 products' roundings.
template <bool DOT, bool BF16 = false>
__global__ __launch_bounds__(kGemmThreads) void km_gemm_kernel(const float *__restrict__ centroids, int k,
                                                              const float *__restrict__ vectors, int64_t n, int dim,
                                                              const float *__restrict__ cadd, KmPart *__restrict__ part)
{
    extern __shared__ float gemm_lds[];
    // block order as flat_gemm_dma_kernel: blocks b, b+8, ... share an XCD; the centroid tiles of one point tile run
    // back to back on one XCD, so the points cross the fabric once
    const int mtiles = (k + kGemmBM - 1) / kGemmBM;
    const int64_t ntiles = (n + kGemmBN - 1) / kGemmBN;
    const int64_t bt = blockIdx.x;
    const int64_t xcd = bt & 7, jx = bt >> 3;
    const int64_t tn = (jx / mtiles) * 8 + xcd;
    const int tm = static_cast<int>(jx % mtiles);
    if (tn >= ntiles) return;
    const int q0 = tm * kGemmBM;
    const int64_t n0 = tn * kGemmBN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    const int drow = wave * 8 + (lane >> 3);
    const int dgl = (lane & 7) ^ ((wave * 4 + (lane >> 4)) & 7);
    const float *const abase = centroids + static_cast<int64_t>(q0) * dim;
    const float *const bbase = vectors + n0 * dim;
    uint32_t aoff[kGemmPasses], boff[kGemmPasses];
#pragma unroll
    for (int p = 0; p < kGemmPasses; p++) {
        int qa = q0 + p * 32 + drow;
        if (qa >= k) qa = k - 1;  // padding rows re-read the last centroid; their scores are +Inf through cadd
        int64_t nb = n0 + p * 32 + drow;
        if (nb >= n) nb = n - 1;
        aoff[p] = static_cast<uint32_t>((static_cast<int64_t>(qa - q0) * dim + dgl * 4) * 4);
        boff[p] = static_cast<uint32_t>(((nb - n0) * dim + dgl * 4) * 4);
    }
    const int half = dim / 2;  // BF16: words of the hi (and of the lo) part of a row
    const int ksteps = BF16 ? 3 * (half / kGemmBK) : (dim + kGemmBK - 1) / kGemmBK;
    const int full_steps = BF16 ? ksteps : dim / kGemmBK;
    const uint32_t lds0 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(
        (__attribute__((address_space(3))) void *)gemm_lds));
    auto piece = [&](int b, int t, int p) {
        return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(
            static_cast<int>(lds0 + ((b * 2 + t) * kDmaTile + (p * 32 + wave * 8) * kGemmBK) * 4)));
    };
    auto dma_tile = [&](int kt) {
        const int k0 = kt * kGemmBK, b = kt & 1;
        if constexpr (BF16) {
            // chunk kc three times: (c_hi, x_hi), (c_lo, x_hi), (c_hi, x_lo) — A = centroids, B = points — and every tile
            // staged ONCE: x_hi stays in B slot 0 for the first two steps, x_lo goes to B slot 1; c_hi sits in A slot kc & 1
            // for steps 0 and 2, c_lo in the other for step 1.  What step kt + 1 needs that is not in LDS yet is requested
            // during step kt, always into a slot step kt does not read (slots: see the loop).
            const int kc = kt / 3, t = kt - 3 * kc;
            const int sa = kc & 1;
            if (t == 0) {
#pragma unroll
                for (int p = 0; p < kGemmPasses; p++) {
                    glds16(abase + kc * kGemmBK, aoff[p], piece(sa, 0, p));
                    glds16(bbase + kc * kGemmBK, boff[p], piece(0, 1, p));
                }
            } else if (t == 1) {
#pragma unroll
                for (int p = 0; p < kGemmPasses; p++) glds16(abase + kc * kGemmBK + half, aoff[p], piece(sa ^ 1, 0, p));
            } else {
#pragma unroll
                for (int p = 0; p < kGemmPasses; p++) glds16(bbase + kc * kGemmBK + half, boff[p], piece(1, 1, p));
            }
        } else if (kt < full_steps) {
#pragma unroll
            for (int p = 0; p < kGemmPasses; p++) {
                glds16(abase + k0, aoff[p], piece(b, 0, p));
                glds16(bbase + k0, boff[p], piece(b, 1, p));
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
    const int h = lane >> 5, f = (lane >> 1) & 7;
    int goff[4];
#pragma unroll
    for (int j = 0; j < 4; j++) goff[j] = ((2 * j + h) ^ f) * 4;
    const int a_row = (wr * 64 + (lane & 31)) * kGemmBK;
    const int b_row = (wc * 64 + (lane & 31)) * kGemmBK;

    const float cadd_reg = tid < kGemmBM ? cadd[q0 + tid] : 0.0f;  // before the K loop: off its tail
    dma_tile(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < ksteps; kt++) {
        const float *As = gemm_lds + (kt & 1) * 2 * kDmaTile, *Bs = As + kDmaTile;
        if constexpr (BF16) {
            const int kc = kt / 3, t = kt - 3 * kc;
            As = gemm_lds + ((kc & 1) ^ (t == 1 ? 1 : 0)) * 2 * kDmaTile;
            Bs = gemm_lds + (t == 2 ? 1 : 0) * 2 * kDmaTile + kDmaTile;
        }
        if (kt + 1 < ksteps) dma_tile(kt + 1);
        float4 fa[2][2], fb[2][2];  // [parity][row block]
        fa[0][0] = *reinterpret_cast<const float4 *>(As + a_row + goff[0]);
        fa[0][1] = *reinterpret_cast<const float4 *>(As + a_row + 32 * kGemmBK + goff[0]);
        fb[0][0] = *reinterpret_cast<const float4 *>(Bs + b_row + goff[0]);
        fb[0][1] = *reinterpret_cast<const float4 *>(Bs + b_row + 32 * kGemmBK + goff[0]);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int c = j & 1, nx = c ^ 1;
            if (j < 3) {
                fa[nx][0] = *reinterpret_cast<const float4 *>(As + a_row + goff[j + 1]);
                fa[nx][1] = *reinterpret_cast<const float4 *>(As + a_row + 32 * kGemmBK + goff[j + 1]);
                fb[nx][0] = *reinterpret_cast<const float4 *>(Bs + b_row + goff[j + 1]);
                fb[nx][1] = *reinterpret_cast<const float4 *>(Bs + b_row + 32 * kGemmBK + goff[j + 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
#define VG_KM_MFMA4(comp)                                                                                  \
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
                VG_KM_MFMA4(x)
                VG_KM_MFMA4(y)
                VG_KM_MFMA4(z)
                VG_KM_MFMA4(w)
            }
#undef VG_KM_MFMA4
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of tile kt+1 have landed
        __syncthreads();                                   // ... everyone's have; tile kt is free
    }

    // Epilogue.  C/D map of the 32x32 MFMA: column = lane & 31 (a point), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    // (a centroid).  A lane scans its 32 centroids of each of its 2 points for (smallest, where, second smallest),
    // merges with lane ^ 32 (the other rows of the same columns), then the two waves of a column block merge in LDS.
    float *lds_add = gemm_lds;                                     // [128]
    float *lds_m = gemm_lds + kGemmBM;                             // [3][2 wr][128 points]
    int *lds_i = reinterpret_cast<int *>(lds_m + 3 * 2 * kGemmBN);  // [2][2 wr][128 points]
    if (tid < kGemmBM) lds_add[tid] = cadd_reg;
    __syncthreads();
    float4 t4[2][4];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int g = 0; g < 4; g++)
            t4[i][g] = *reinterpret_cast<const float4 *>(lds_add + wr * 64 + i * 32 + 8 * g + 4 * h);
    const float coef = DOT ? -1.0f : -2.0f;
#pragma unroll
    for (int j = 0; j < 2; j++) {
        float m1 = INFINITY, m2 = INFINITY, m3 = INFINITY;
        int i1 = 0, i2 = 0;
#pragma unroll
        for (int i = 0; i < 2; i++) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const float4 tv = t4[i][r >> 2];
                const float add = (r & 3) == 0 ? tv.x : (r & 3) == 1 ? tv.y : (r & 3) == 2 ? tv.z : tv.w;
                const float sc = __builtin_fmaf(coef, acc[i][j][r], add);
                const int c = q0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                // m1 <= m2 <= m3: the middle one of (m2, sc, m3) is the new third, of (m1, sc, m2) the new second
                m3 = __builtin_amdgcn_fmed3f(m2, sc, m3);
                const bool lt1 = sc < m1, lt2 = sc < m2;
                i2 = lt1 ? i1 : (lt2 ? c : i2);
                m2 = __builtin_amdgcn_fmed3f(m1, sc, m2);
                i1 = lt1 ? c : i1;
                m1 = fminf(m1, sc);
            }
        }
        km_merge3(m1, i1, m2, i2, m3, __shfl_xor(m1, 32), __shfl_xor(i1, 32), __shfl_xor(m2, 32), __shfl_xor(i2, 32),
                  __shfl_xor(m3, 32));
        if (h == 0) {
            const int at = wr * kGemmBN + wc * 64 + j * 32 + (lane & 31);
            lds_m[at] = m1;
            lds_m[2 * kGemmBN + at] = m2;
            lds_m[4 * kGemmBN + at] = m3;
            lds_i[at] = i1;
            lds_i[2 * kGemmBN + at] = i2;
        }
    }
    __syncthreads();
    if (tid < kGemmBN && n0 + tid < n) {
        float m1 = lds_m[tid], m2 = lds_m[2 * kGemmBN + tid], m3 = lds_m[4 * kGemmBN + tid];
        int i1 = lds_i[tid], i2 = lds_i[2 * kGemmBN + tid];
        km_merge3(m1, i1, m2, i2, m3, lds_m[kGemmBN + tid], lds_i[kGemmBN + tid], lds_m[3 * kGemmBN + tid],
                  lds_i[3 * kGemmBN + tid], lds_m[5 * kGemmBN + tid]);
        float4 *dst = reinterpret_cast<float4 *>(part + static_cast<int64_t>(tm) * n + n0 + tid);
        dst[0] = make_float4(m1, m2, m3, __int_as_float(i1));
        dst[1] = make_float4(__int_as_float(i2), 0.0f, 0.0f, 0.0f);
    }
}

// Three outcomes per point: the gap to the second smallest score exceeds the margin -> decided; only the gap to the THIRD
// does -> the reference's choice is one of two centroids: (point, lo, hi) goes on the pair list; otherwise the point
// goes on the full list.  List positions come from one atomic per workgroup and list (positions inside a workgroup by
// an LDS counter): the order of a list is irrelevant, every entry is handled on its own.
__global__ __launch_bounds__(256) void km_decide_kernel(const KmPart *__restrict__ part, int mtiles, int64_t n,
                                                        const float *__restrict__ xnorm, float coef_e, float coef_r,
                                                        int32_t *__restrict__ assign, int *__restrict__ changed,
                                                        int32_t *__restrict__ list, int32_t *__restrict__ pairs /* [n][3] */,
                                                        int *__restrict__ counts /* [0] full list, [1] pair list, [2] max |c|^2 bits */)
{
    __shared__ int s_cnt[2], s_base[2];
    if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    int kind = 0, pos = 0;  // 0 decided (or past n), 1 full list, 2 pair list
    int i1 = 0, i2 = 0;
    if (i < n) {
        float m1 = INFINITY, m2 = INFINITY, m3 = INFINITY;
        for (int t = 0; t < mtiles; t++) {
            const float4 *src = reinterpret_cast<const float4 *>(part + static_cast<int64_t>(t) * n + i);
            const float4 a = src[0];
            const int bi2 = __float_as_int(src[1].x);
            km_merge3(m1, i1, m2, i2, m3, a.x, __float_as_int(a.w), a.y, bi2, a.z);
        }
        const float X = xnorm[i], C = __int_as_float(counts[2]);
        const float sx = sqrtf(X), sc = sqrtf(C);
        const float cross = 2.0f * sx * sc + C, dmax = (sx + sc) * (sx + sc);
        const float margin = 1.05f * (coef_e * cross + coef_r * dmax) + 1e-30f;
        // every comparison is false on NaN: non-finite rows, centroids or scores fall through to the full list
        const bool finite = X + C < 1e30f;
        if (finite && m2 - m1 > margin) {
            if (changed && assign[i] != i1) *changed = 1;
            assign[i] = i1;
        } else {
            kind = (finite && m3 - m1 > margin) ? 2 : 1;
            pos = atomicAdd(&s_cnt[kind - 1], 1);
        }
    }
    __syncthreads();
    if (threadIdx.x < 2 && s_cnt[threadIdx.x] > 0) s_base[threadIdx.x] = atomicAdd(&counts[threadIdx.x], s_cnt[threadIdx.x]);
    __syncthreads();
    if (kind == 1) {
        list[s_base[0] + pos] = static_cast<int32_t>(i);
    } else if (kind == 2) {
        int32_t *e = pairs + 3 * static_cast<int64_t>(s_base[1] + pos);
        e[0] = static_cast<int32_t>(i);
        e[1] = i1 < i2 ? i1 : i2;
        e[2] = i1 < i2 ? i2 : i1;
    }
}

// the pair list: the reference-order distance to the two candidates; the reference's scan (strict compare, lowest index
// first) picks `hi` only if it is strictly better than `lo`
template <bool DOT>
__global__ __launch_bounds__(256) void km_pairs_kernel(const float *__restrict__ vectors, int dim,
                                                       const float *__restrict__ centroids,
                                                       const int32_t *__restrict__ pairs, const int *__restrict__ counts,
                                                       int32_t *__restrict__ assign, int *__restrict__ changed)
{
    const Sub16 sub = Sub16::make(threadIdx.x);
    const int total = counts[1];
    for (int64_t e = static_cast<int64_t>(blockIdx.x) * 16 + (threadIdx.x >> 4); e < total;
         e += static_cast<int64_t>(gridDim.x) * 16) {
        const int64_t i = pairs[3 * e];
        const int lo = pairs[3 * e + 1], hi = pairs[3 * e + 2];
        const float *v = vectors + i * dim;
        const float dl = exact_pair16<DOT, kBatch>(centroids + static_cast<int64_t>(lo) * dim, v, dim, sub);
        const float dh = exact_pair16<DOT, kBatch>(centroids + static_cast<int64_t>(hi) * dim, v, dim, sub);
        const int best = (DOT ? (dh > dl) : (dh < dl)) ? hi : lo;
        if ((threadIdx.x & 15) == 0) {
            if (changed && assign[i] != best) *changed = 1;
            assign[i] = best;
        }
    }
}

}  // namespace vg

// pick the register-resident assignment when the row shape allows it
// list != nullptr: the points list[0 .. *list_count) (count known to the device only: the grid is sized for a share of n
// and strides)
template <int NBLK>
static int32_t km_launch_regs(bool dot, const float *v, int64_t n, const float *cent, int k, int32_t *assign, int *changed,
                              hipStream_t st, const int32_t *list, const int *list_count, float *range_d = nullptr, int32_t *range_i = nullptr)
{
    const int64_t chunks = (n + 16 * vg::kKmRows - 1) / (16 * vg::kKmRows);
    // (the listed points by centroid ranges of 32 when the caller brought the scratch: see the kernel)
    const int range_len = 4 * vg::kKmTile, ranges = (k + range_len - 1) / range_len;
    if (list && range_d && ranges > 1 && !vg::hook(vg::kHookKmNoRanges)) {
        const unsigned gx = static_cast<unsigned>(std::min<int64_t>(chunks, 2048));
        if (dot) {
            VG_LAUNCH((vg::km_assign_regs_kernel<true, NBLK, true>), dim3(gx, ranges), dim3(256), 0, st, v, n, cent, k, assign, changed,
                      list, list_count, range_d, range_i, range_len);
            VG_LAUNCH(vg::km_list_combine_kernel<true>, dim3(256), dim3(256), 0, st, list, list_count, range_d, range_i, ranges, assign, changed);
        } else {
            VG_LAUNCH((vg::km_assign_regs_kernel<false, NBLK, true>), dim3(gx, ranges), dim3(256), 0, st, v, n, cent, k, assign, changed,
                      list, list_count, range_d, range_i, range_len);
            VG_LAUNCH(vg::km_list_combine_kernel<false>, dim3(256), dim3(256), 0, st, list, list_count, range_d, range_i, ranges, assign, changed);
        }
        return VG_OK;
    }
    if (list) {
        const unsigned gx = static_cast<unsigned>(std::min<int64_t>(chunks, 2048));
        if (dot)
            VG_LAUNCH((vg::km_assign_regs_kernel<true, NBLK, true>), dim3(gx), dim3(256), 0, st, v, n, cent, k, assign, changed,
                      list, list_count);
        else
            VG_LAUNCH((vg::km_assign_regs_kernel<false, NBLK, true>), dim3(gx), dim3(256), 0, st, v, n, cent, k, assign, changed,
                      list, list_count);
        return VG_OK;
    }
    const unsigned gx = static_cast<unsigned>(chunks);
    if (dot)
        VG_LAUNCH((vg::km_assign_regs_kernel<true, NBLK>), dim3(gx), dim3(256), 0, st, v, n, cent, k, assign, changed, nullptr,
                  nullptr);
    else
        VG_LAUNCH((vg::km_assign_regs_kernel<false, NBLK>), dim3(gx), dim3(256), 0, st, v, n, cent, k, assign, changed, nullptr,
                  nullptr);
    return VG_OK;
}

// the reference-order assignment (squaredL2BatchAvx512 / dotBatchAvx512 summation order)
static int32_t km_launch_exact(bool dot, const float *v, int64_t n, int dim, const float *cent, int k, int32_t *assign,
                               int *changed, hipStream_t st, const int32_t *list = nullptr, const int *list_count = nullptr,
                               float *range_d = nullptr, int32_t *range_i = nullptr)
{
    const bool aligned = (reinterpret_cast<uintptr_t>(v) & 15) == 0 && (reinterpret_cast<uintptr_t>(cent) & 15) == 0;
    if (aligned) {
        switch (dim) {
        case 64: return km_launch_regs<1>(dot, v, n, cent, k, assign, changed, st, list, list_count, range_d, range_i);
        case 128: return km_launch_regs<2>(dot, v, n, cent, k, assign, changed, st, list, list_count, range_d, range_i);
        case 256: return km_launch_regs<4>(dot, v, n, cent, k, assign, changed, st, list, list_count, range_d, range_i);
        case 384: return km_launch_regs<6>(dot, v, n, cent, k, assign, changed, st, list, list_count, range_d, range_i);
        case 512: return km_launch_regs<8>(dot, v, n, cent, k, assign, changed, st, list, list_count, range_d, range_i);
        case 768: return km_launch_regs<12>(dot, v, n, cent, k, assign, changed, st, list, list_count, range_d, range_i);
        case 1024: return km_launch_regs<16>(dot, v, n, cent, k, assign, changed, st, list, list_count, range_d, range_i);
        default: break;
        }
    }
    const int64_t groups = (n + 15) / 16;
    if (list) {
        const unsigned gx = static_cast<unsigned>(std::min<int64_t>(groups, 4096));
        if (dot)
            VG_LAUNCH((vg::km_assign_kernel<true, true>), dim3(gx), dim3(256), 0, st, v, n, dim, cent, k, assign, changed, list,
                      list_count);
        else
            VG_LAUNCH((vg::km_assign_kernel<false, true>), dim3(gx), dim3(256), 0, st, v, n, dim, cent, k, assign, changed, list,
                      list_count);
        return VG_OK;
    }
    const unsigned gx = static_cast<unsigned>(groups);
    if (dot)
        VG_LAUNCH((vg::km_assign_kernel<true>), dim3(gx), dim3(256), 0, st, v, n, dim, cent, k, assign, changed, nullptr, nullptr);
    else
        VG_LAUNCH((vg::km_assign_kernel<false>), dim3(gx), dim3(256), 0, st, v, n, dim, cent, k, assign, changed, nullptr, nullptr);
    return VG_OK;
}

// MFMA nomination + exact decision (see km_gemm_kernel): scratch of one vg_kmeans_assign / vg_kmeans_train call
struct KmMfma {
    vg::DevTmp<float> xnorm, cadd;
    vg::DevTmp<uint16_t> xsplit, csplit;  // bf16 splits of the points (made once) and of the centroids (every pass)
    bool bf16 = false;
    vg::DevTmp<vg::KmPart> part;
    vg::DevTmp<int32_t> list, pairs;
    vg::DevTmp<float> range_d;    // the listed points' bests by centroid range (km_assign_regs_kernel<.., LIST>): capacity slots x ranges
    vg::DevTmp<int32_t> range_i;
    vg::DevTmp<int> count;  // [0] full list, [1] pair list, [2] max |c|^2 (float bits)
    int mtiles = 0;
    bool on = false;

    static bool eligible(const float *v, int64_t n, int dim, const float *cent, int k)
    {
        const bool aligned = ((reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(cent)) & 15) == 0;
        // below a few thousand points the passes around the GEMM cost more than the reference-order kernel
        return aligned && dim % 4 == 0 && dim >= 32 && n >= 4096 && n <= INT32_MAX && k >= 2 && !vg::hook(vg::kHookKmNoMfma);
    }
    // allocates and computes |x|^2 (the points do not change between the iterations of a training run).
    // passes: how many assignment passes will follow — with three or more, and rows of whole 64-element blocks, the points
    // are also split into bfloat16 [hi | lo] (3.1 GB per 1M x 768, written once) and the passes run on the 16x faster
    // bf16 matrix instruction, HBM-bound instead of MFMA-bound (1.64 -> 0.69 ms per 1M x 768 x 122: DESIGN.md section 8)
    int32_t init(const float *v, int64_t n, int dim, int k, hipStream_t st, int passes = 1, int64_t hbm_bytes = 0)
    {
        mtiles = (k + vg::kGemmBM - 1) / vg::kGemmBM;
        const int64_t split_bytes = n * 2 * dim * 2;
        bf16 = dim % 64 == 0 && (passes >= 3 || vg::hook(vg::kHookKmBf16)) && !vg::hook(vg::kHookKmNoBf16) &&
               (hbm_bytes == 0 || split_bytes <= hbm_bytes / 8);
        if (bf16) {
            VG_TRY(xsplit.init(static_cast<size_t>(n) * 2 * dim, st));
            VG_TRY(csplit.init(static_cast<size_t>(mtiles) * vg::kGemmBM * 2 * dim, st));
            const int64_t tot = n * dim / 4;
            VG_LAUNCH(vg::km_split_kernel<false>, dim3(static_cast<unsigned>((tot + 255) / 256)), dim3(256), 0, st, v, n, dim,
                      xsplit.ptr);
        }
        VG_TRY(xnorm.init(static_cast<size_t>(n), st));
        VG_TRY(cadd.init(static_cast<size_t>(mtiles) * vg::kGemmBM, st));
        VG_TRY(part.init(static_cast<size_t>(mtiles) * n, st));
        VG_TRY(list.init(static_cast<size_t>(n), st));
        VG_TRY(pairs.init(static_cast<size_t>(n) * 3, st));
        {   // (ranges of 32 centroids for the listed points' exact pass: km_launch_regs — every point can be listed)
            const int64_t ranges = (k + 4 * vg::kKmTile - 1) / (4 * vg::kKmTile);
            if (ranges > 1 && n * ranges * 8 <= (int64_t(256) << 20)) {
                VG_TRY(range_d.init(static_cast<size_t>(n * ranges), st));
                VG_TRY(range_i.init(static_cast<size_t>(n * ranges), st));
            }
        }
        VG_TRY(count.init(3, st));
        VG_LAUNCH(vg::km_norms_kernel, dim3(static_cast<unsigned>((n + 3) / 4)), dim3(256), 0, st, v, n, dim, xnorm.ptr);
        on = true;
        return VG_OK;
    }
    int32_t assign(bool dot, const float *v, int64_t n, int dim, const float *cent, int k, int32_t *out, int *changed,
                   hipStream_t st)
    {
        VG_HIP(hipMemsetAsync(count.ptr, 0, 3 * sizeof(int), st));
        const int kpad = mtiles * vg::kGemmBM;
        VG_LAUNCH(vg::km_cent_norms_kernel, dim3(static_cast<unsigned>((kpad + 3) / 4)), dim3(256), 0, st, cent, k, dim, dot,
                  cadd.ptr, kpad, count.ptr + 2);
        const int64_t ntiles = (n + vg::kGemmBN - 1) / vg::kGemmBN;
        const unsigned blocks = static_cast<unsigned>(((ntiles + 7) / 8) * 8 * mtiles);
        const float u = 5.9604645e-8f;
        float coef_e = 2.0f * (2.0f * dim + 16.0f) * u, coef_r = 2.0f * (dim / 64 + 32.0f) * u;
        if (bf16) {
            const int64_t tot = static_cast<int64_t>(k) * dim / 4;
            VG_LAUNCH(vg::km_split_kernel<true>, dim3(static_cast<unsigned>((tot + 255) / 256)), dim3(256), 0, st, cent, k, dim,
                      csplit.ptr);
            auto kern = dot ? vg::km_gemm_kernel<true, true> : vg::km_gemm_kernel<false, true>;
            VG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(vg::kDmaLdsBytes)));
            VG_LAUNCH(kern, dim3(blocks), dim3(vg::kGemmThreads), vg::kDmaLdsBytes, st, reinterpret_cast<const float *>(csplit.ptr),
                      k, reinterpret_cast<const float *>(xsplit.ptr), n, dim, cadd.ptr, part.ptr);
            // |s~ - s|: the split drops at most 3 * 2^-16 |x_i| |c_i| per element (x = hi + lo + r, |r| <= 2^-16 |x|, the
            // same for c; the dropped terms are lo lo, x r_c, r_x c); the products of bfloat16 values are exact in fp32 and
            // every one of the 3 dim additions into the fp32 accumulator is taken to round (the matrix unit's internal
            // order is not documented: the worst case of any order), + the |c|^2 column and the final fused multiply-add
            coef_e = 2.0f * (3.0f * 1.52587890625e-5f * 1.01f + (3.0f * dim + 3.0f * dim / 16 + 16.0f + dim) * u);
        } else {
            auto kern = dot ? vg::km_gemm_kernel<true> : vg::km_gemm_kernel<false>;
            VG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(vg::kDmaLdsBytes)));
            VG_LAUNCH(kern, dim3(blocks), dim3(vg::kGemmThreads), vg::kDmaLdsBytes, st, cent, k, v, n, dim, cadd.ptr, part.ptr);
        }
        if (vg::hook(vg::kHookKmListAll)) coef_e = INFINITY;
        VG_LAUNCH(vg::km_decide_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, st, part.ptr, mtiles, n,
                  xnorm.ptr, coef_e, coef_r, out, changed, list.ptr, pairs.ptr, count.ptr);
        const unsigned pg = static_cast<unsigned>(std::min<int64_t>((n + 15) / 16, 2048));
        if (dot)
            VG_LAUNCH(vg::km_pairs_kernel<true>, dim3(pg), dim3(256), 0, st, v, dim, cent, pairs.ptr, count.ptr, out, changed);
        else
            VG_LAUNCH(vg::km_pairs_kernel<false>, dim3(pg), dim3(256), 0, st, v, dim, cent, pairs.ptr, count.ptr, out, changed);
        return km_launch_exact(dot, v, n, dim, cent, k, out, changed, st, list.ptr, count.ptr, range_d.ptr, range_i.ptr);
    }
};

static bool km_metric_ok(int32_t metric) { return metric == VG_METRIC_L2 || metric == VG_METRIC_DOT || metric == VG_METRIC_COSINE; }

VG_API int32_t vg_kmeans_assign(vg_ctx *ctx, const float *vectors, int64_t n, int32_t dim, const float *centroids,
                                int32_t k, int32_t metric, int32_t *out, void *stream)
{
    VG_CHECK(ctx, VG_ERR_INVALID_ARG, "vg_kmeans_assign: ctx is NULL");
    VG_CHECK(km_metric_ok(metric), VG_ERR_UNSUPPORTED, "unsupported metric for float32: %d", metric);
    VG_CHECK(n >= 0 && dim > 0 && k > 0, VG_ERR_INVALID_ARG, "vg_kmeans_assign: bad sizes");
    if (n == 0) return VG_OK;
    VG_CHECK(vectors && centroids && out, VG_ERR_INVALID_ARG, "vg_kmeans_assign: NULL buffer");
    VG_HIP(hipSetDevice(ctx->device));
    hipStream_t st = vg::pick_stream(ctx, stream);
    vg::DevIn<float> v, c;
    vg::DevOut<int32_t> o;
    VG_TRY(v.init(vectors, static_cast<size_t>(n) * dim, st));
    VG_TRY(c.init(centroids, static_cast<size_t>(k) * dim, st));
    VG_TRY(o.init(out, static_cast<size_t>(n), st));
    KmMfma mf;
    if (KmMfma::eligible(v.ptr, n, dim, c.ptr, k)) VG_TRY(mf.init(v.ptr, n, dim, k, st, 1, ctx->hbm_bytes));
    {
        vg::ProfScope prof(ctx, "km_assign", st);
        if (mf.on)
            VG_TRY(mf.assign(metric != VG_METRIC_L2, v.ptr, n, dim, c.ptr, k, o.ptr, nullptr, st));
        else
            VG_TRY(km_launch_exact(metric != VG_METRIC_L2, v.ptr, n, dim, c.ptr, k, o.ptr, nullptr, st));
    }
    VG_TRY(o.finish());
    if (o.on_host()) VG_HIP(hipStreamSynchronize(st));
    return VG_OK;
}

VG_API int32_t vg_kmeans_train(vg_ctx *ctx, const float *vectors, int64_t n, int32_t dim, int32_t k, int32_t metric,
                               int32_t max_iter, uint64_t seed, float *centroids, int32_t *produced, void *stream)
{
    VG_CHECK(ctx, VG_ERR_INVALID_ARG, "vg_kmeans_train: ctx is NULL");
    if (produced) *produced = 0;
    VG_CHECK(dim > 0 && k > 0 && n >= 0 && max_iter >= 0, VG_ERR_INVALID_ARG, "vg_kmeans_train: bad sizes");
    if (n < k) return VG_OK;  // kmeans.go:17-20: not enough vectors to cluster
    VG_CHECK(km_metric_ok(metric), VG_ERR_UNSUPPORTED, "unsupported metric for float32: %d", metric);
    VG_CHECK(vectors && centroids, VG_ERR_INVALID_ARG, "vg_kmeans_train: NULL buffer");
    VG_HIP(hipSetDevice(ctx->device));
    hipStream_t st = vg::pick_stream(ctx, stream);
    vg::DevIn<float> v;
    vg::DevOut<float> cent;
    VG_TRY(v.init(vectors, static_cast<size_t>(n) * dim, st));
    VG_TRY(cent.init(centroids, static_cast<size_t>(k) * dim, st));
    // first k entries of the permutation (partial Fisher-Yates over the counter stream)
    std::vector<int64_t> perm(static_cast<size_t>(n));
    std::iota(perm.begin(), perm.end(), int64_t(0));
    for (int64_t i = 0; i < k && i < n - 1; i++) {
        const int64_t j = i + static_cast<int64_t>(vg::km_rng(seed, 0, 1, static_cast<uint64_t>(i)) %
                                                   static_cast<uint64_t>(n - i));
        std::swap(perm[static_cast<size_t>(i)], perm[static_cast<size_t>(j)]);
    }
    vg::DevTmp<int64_t> rows, counts, offsets, members;
    vg::DevTmp<int32_t> assign;
    vg::DevTmp<int> changed;
    VG_TRY(rows.init(static_cast<size_t>(k), st));
    VG_TRY(counts.init(static_cast<size_t>(k), st));
    VG_TRY(offsets.init(static_cast<size_t>(k), st));
    VG_TRY(members.init(static_cast<size_t>(n), st));
    VG_TRY(assign.init(static_cast<size_t>(n), st));
    VG_TRY(changed.init(1, st));
    VG_HIP(hipMemcpyAsync(rows.ptr, perm.data(), sizeof(int64_t) * k, hipMemcpyHostToDevice, st));
    VG_HIP(hipStreamSynchronize(st));  // perm is a local
    const int64_t tot = static_cast<int64_t>(k) * dim;
    VG_LAUNCH(vg::km_gather_rows_kernel, dim3(static_cast<unsigned>((tot + 255) / 256)), dim3(256), 0, st,
                       v.ptr, dim, rows.ptr, k, cent.ptr);
    VG_HIP(hipMemsetAsync(assign.ptr, 0, sizeof(int32_t) * static_cast<size_t>(n), st));
    const unsigned kx = static_cast<unsigned>((k + 63) / 64);
    const bool sorted = k <= vg::kKmLdsK;
    const int parts = static_cast<int>((n + vg::kKmPartRows - 1) / vg::kKmPartRows);
    int kbits = 0;
    while ((1 << kbits) < k) kbits++;
    vg::DevTmp<int32_t> hist;
    if (sorted) VG_TRY(hist.init(static_cast<size_t>(parts) * k, st));
    std::vector<int64_t> hcounts, hoff;
    KmMfma mf;
    if (KmMfma::eligible(v.ptr, n, dim, cent.ptr, k)) VG_TRY(mf.init(v.ptr, n, dim, k, st, max_iter, ctx->hbm_bytes));
    for (int it = 0; it < max_iter; it++) {
        VG_HIP(hipMemsetAsync(changed.ptr, 0, sizeof(int), st));
        {
            vg::ProfScope prof(ctx, "km_assign", st);
            if (mf.on)
                VG_TRY(mf.assign(metric != VG_METRIC_L2, v.ptr, n, dim, cent.ptr, k, assign.ptr, changed.ptr, st));
            else
                VG_TRY(km_launch_exact(metric != VG_METRIC_L2, v.ptr, n, dim, cent.ptr, k, assign.ptr, changed.ptr, st));
        }
        int hchanged = 0;
        VG_HIP(hipMemcpyAsync(&hchanged, changed.ptr, sizeof(int), hipMemcpyDeviceToHost, st));
        VG_HIP(hipStreamSynchronize(st));  // one sync per Lloyd iteration: the convergence test is the host's
        if (!hchanged) break;               // kmeans.go:101-103
        vg::ProfScope prof(ctx, "km_update", st);  // member lists + per-cluster sums in index order
        if (sorted) {
            VG_LAUNCH(vg::km_hist_kernel, dim3(parts), dim3(256), 0, st, assign.ptr, n, k, hist.ptr);
            VG_LAUNCH(vg::km_scan_parts_kernel, dim3(k), dim3(256), 0, st, hist.ptr, parts, k, counts.ptr);
            VG_LAUNCH(vg::km_offsets_kernel, dim3(1), dim3(256), 0, st, k, counts.ptr, offsets.ptr);
            VG_LAUNCH(vg::km_scatter_kernel, dim3(parts), dim3(64), 0, st, assign.ptr, n, k, kbits, hist.ptr, offsets.ptr,
                      members.ptr);
        } else {  // more clusters than LDS counters: every cluster walks the assignment array
            hcounts.resize(static_cast<size_t>(k));
            hoff.resize(static_cast<size_t>(k));
            VG_LAUNCH(vg::km_count_kernel, dim3(kx), dim3(64), 0, st, assign.ptr, n, k, counts.ptr);
            VG_HIP(hipMemcpyAsync(hcounts.data(), counts.ptr, sizeof(int64_t) * k, hipMemcpyDeviceToHost, st));
            VG_HIP(hipStreamSynchronize(st));
            int64_t run = 0;
            for (int c = 0; c < k; c++) {
                hoff[static_cast<size_t>(c)] = run;
                run += hcounts[static_cast<size_t>(c)];
            }
            VG_HIP(hipMemcpyAsync(offsets.ptr, hoff.data(), sizeof(int64_t) * k, hipMemcpyHostToDevice, st));
            VG_LAUNCH(vg::km_members_kernel, dim3(kx), dim3(64), 0, st, assign.ptr, n, k, offsets.ptr, members.ptr);
        }
        // 128 lanes per workgroup: 6 x k workgroups spread over the CUs more evenly than 3 x k (1.08 -> 0.99 ms per iteration at
        // 1M x 768 x 122; 64 lanes: 1.02).  A cluster's members are one sequential chain per dimension, a wave has at most 64
        // loads in flight (vmcnt): a cluster streams at ~100 GB/s however it is cut, which is the first iterations' tail
#ifndef VG_KM_UPD_THREADS
#define VG_KM_UPD_THREADS 128
#endif
        VG_LAUNCH(vg::km_update_kernel, dim3(static_cast<unsigned>((dim + VG_KM_UPD_THREADS - 1) / VG_KM_UPD_THREADS), k), dim3(VG_KM_UPD_THREADS), 0, st,
                           v.ptr, n, dim, k, it, seed, counts.ptr, offsets.ptr, members.ptr, cent.ptr);
        if (!sorted) VG_HIP(hipStreamSynchronize(st));  // hoff is reused next iteration
    }
    VG_TRY(cent.finish());
    VG_HIP(hipStreamSynchronize(st));
    if (produced) *produced = 1;
    return VG_OK;
}

VG_API int32_t vg_find_closest_centroids(vg_ctx *ctx, const float *query, const float *centroids, int32_t dim,
                                         int32_t k, int32_t nprobe, int32_t metric, int32_t *out, int32_t *n_out,
                                         void *stream)
{
    VG_CHECK(ctx, VG_ERR_INVALID_ARG, "vg_find_closest_centroids: ctx is NULL");
    if (n_out) *n_out = 0;
    VG_CHECK(km_metric_ok(metric), VG_ERR_UNSUPPORTED, "unsupported metric for float32: %d", metric);
    VG_CHECK(dim > 0 && k >= 0 && nprobe >= 0, VG_ERR_INVALID_ARG, "vg_find_closest_centroids: bad sizes");
    if (k == 0 || nprobe == 0) return VG_OK;
    VG_CHECK(query && centroids && out, VG_ERR_INVALID_ARG, "vg_find_closest_centroids: NULL buffer");
    int n = nprobe > k ? k : nprobe;
    std::vector<float> d(static_cast<size_t>(k));
    int32_t s = metric == VG_METRIC_L2 ? vg_squared_l2_batch(ctx, query, centroids, dim, k, d.data(), stream)
                                       : vg_dot_batch(ctx, query, centroids, dim, k, d.data(), stream);
    if (s != VG_OK) return s;
    if (metric != VG_METRIC_L2)
        for (auto &x : d) x = -x;  // kmeans.go:238-241
    std::vector<int32_t> id(static_cast<size_t>(k));
    std::iota(id.begin(), id.end(), 0);
    if (n <= k / 4 && n < 16) {  // kmeans.go:254-268 selection
        for (int i = 0; i < n; i++) {
            int mi = i;
            for (int j = i + 1; j < k; j++)
                if (d[static_cast<size_t>(j)] < d[static_cast<size_t>(mi)]) mi = j;
            std::swap(d[static_cast<size_t>(i)], d[static_cast<size_t>(mi)]);
            std::swap(id[static_cast<size_t>(i)], id[static_cast<size_t>(mi)]);
            out[i] = id[static_cast<size_t>(i)];
        }
    } else {  // kmeans.go:271-278 full sort (pdqsort in the reference: ties unpinned; here by index)
        std::vector<int32_t> order(id);
        std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
            return d[static_cast<size_t>(a)] < d[static_cast<size_t>(b)];
        });
        for (int i = 0; i < n; i++) out[i] = order[static_cast<size_t>(i)];
    }
    if (n_out) *n_out = n;
    return VG_OK;
}

VG_API int32_t vg_squared_l2_bounded_batch(vg_ctx *ctx, const float *query, const float *targets, int64_t dim,
                                           int64_t n, const float *bounds, int64_t n_bounds, float *dist,
                                           int32_t *exceeded, void *stream)
{
    VG_CHECK(ctx, VG_ERR_INVALID_ARG, "vg_squared_l2_bounded_batch: ctx is NULL");
    if (n <= 0) return VG_OK;
    VG_CHECK(dim >= 0 && dim < (1 << 30), VG_ERR_INVALID_ARG, "vg_squared_l2_bounded_batch: bad dim");
    VG_CHECK(bounds && dist && exceeded && (n_bounds == 1 || n_bounds == n), VG_ERR_INVALID_ARG,
             "vg_squared_l2_bounded_batch: bounds must have 1 or n entries");
    VG_CHECK(dim == 0 || (query && targets), VG_ERR_INVALID_ARG, "vg_squared_l2_bounded_batch: NULL buffer");
    VG_HIP(hipSetDevice(ctx->device));
    hipStream_t st = vg::pick_stream(ctx, stream);
    vg::DevIn<float> q, t, b;
    vg::DevOut<float> od;
    vg::DevOut<int32_t> oe;
    VG_TRY(q.init(query, static_cast<size_t>(dim), st));
    VG_TRY(t.init(targets, static_cast<size_t>(n) * dim, st));
    VG_TRY(b.init(bounds, static_cast<size_t>(n_bounds), st));
    VG_TRY(od.init(dist, static_cast<size_t>(n), st));
    VG_TRY(oe.init(exceeded, static_cast<size_t>(n), st));
    int64_t blocks = std::min<int64_t>((n + 15) / 16, 4096);
    VG_LAUNCH(vg::bounded_batch_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, st, q.ptr, t.ptr,
                       static_cast<int>(dim), n, b.ptr, n_bounds, od.ptr, oe.ptr);
    VG_TRY(od.finish());
    VG_TRY(oe.finish());
    if (od.on_host() || oe.on_host()) VG_HIP(hipStreamSynchronize(st));
    return VG_OK;
}

VG_API int32_t vg_pq_adc_lookup_batch(vg_ctx *ctx, const float *table, const uint8_t *codes, int64_t m, int64_t n,
                                      float *out, void *stream)
{
    VG_CHECK(ctx, VG_ERR_INVALID_ARG, "vg_pq_adc_lookup_batch: ctx is NULL");
    VG_CHECK(m >= 0 && n >= 0 && m < (1 << 20), VG_ERR_INVALID_ARG, "vg_pq_adc_lookup_batch: bad sizes");
    if (n == 0) return VG_OK;
    VG_CHECK(out && (m == 0 || (table && codes)), VG_ERR_INVALID_ARG, "vg_pq_adc_lookup_batch: NULL buffer");
    VG_HIP(hipSetDevice(ctx->device));
    hipStream_t st = vg::pick_stream(ctx, stream);
    vg::DevIn<float> t;
    vg::DevIn<uint8_t> c;
    vg::DevOut<float> o;
    VG_TRY(t.init(table, static_cast<size_t>(m) * 256, st));
    VG_TRY(c.init(codes, static_cast<size_t>(n) * m, st));
    VG_TRY(o.init(out, static_cast<size_t>(n), st));
    // table in LDS + rows turned through LDS when both fit (m % 16 == 0)
    const int stride = static_cast<int>(16 * ((m >> 4) | 1));
    const int64_t lds_free = 160 * 1024 - m * 1024;
    const int waves = m > 0 && m % 16 == 0 ? static_cast<int>(std::min<int64_t>(8, lds_free / (64 * stride))) : 0;
    const bool aligned = ((reinterpret_cast<uintptr_t>(c.ptr) | reinterpret_cast<uintptr_t>(t.ptr)) & 15) == 0;
    if (waves >= 2 && aligned) {
        const size_t lds = static_cast<size_t>(m) * 1024 + static_cast<size_t>(waves) * 64 * stride;
        const int64_t tiles = (n + 63) / 64;
        const unsigned blocks = static_cast<unsigned>(std::min<int64_t>((tiles + waves - 1) / waves, std::max(ctx->compute_units, 1)));
        auto kern = m == 96 ? vg::adc_lookup_batch_lds_kernel<6> : vg::adc_lookup_batch_lds_kernel<0>;
        VG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   static_cast<int>(lds)));
        VG_LAUNCH(kern, dim3(blocks), dim3(waves * 64), lds, st, t.ptr, c.ptr, static_cast<int>(m), stride, n, o.ptr);
    } else {
        VG_LAUNCH(vg::adc_lookup_batch_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, st,
                           t.ptr, c.ptr, static_cast<int>(m), n, o.ptr);
    }
    VG_TRY(o.finish());
    if (o.on_host()) VG_HIP(hipStreamSynchronize(st));
    return VG_OK;
}
