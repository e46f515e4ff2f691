// k_kmeans.hip — internal/kmeans (TrainKMeans, AssignPartition, FindClosestCentroids) plus the
// remaining batched L0 seams (SquaredL2Bounded, PqAdcLookup).
#include <algorithm>
#include <numeric>

#include "vg_device.hpp"
#include "vg_exact.hpp"
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
template <bool DOT>
__global__ __launch_bounds__(256) void km_assign_kernel(const float *__restrict__ vectors, int64_t n, int dim,
                                                        const float *__restrict__ centroids, int k,
                                                        int32_t *__restrict__ assign, int *__restrict__ changed)
{
    const Sub16 sub = Sub16::make(threadIdx.x);
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 16 + (threadIdx.x >> 4);
    if (i >= n) return;
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

template <bool DOT, int NBLK>
__global__ __launch_bounds__(256) void km_assign_regs_kernel(const float *__restrict__ vectors, int64_t n,
                                                             const float *__restrict__ centroids, int k,
                                                             int32_t *__restrict__ assign, int *__restrict__ changed)
{
    constexpr int dim = NBLK * 64;
    __shared__ __attribute__((aligned(16))) float ctile[kKmTile * dim];
    const int tid = threadIdx.x;
    const Sub16 sub = Sub16::make(tid);
    const int64_t i0 = (static_cast<int64_t>(blockIdx.x) * 16 + (tid >> 4)) * kKmRows;
    float4 rr[kKmRows][NBLK];
#pragma unroll
    for (int p = 0; p < kKmRows; p++) {
        const int64_t i = i0 + p < n ? i0 + p : n - 1;
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
    km_fetch_tile<dim, kStage>(stage, centroids, 0, k, tid);
    for (int c0 = 0; c0 < k; c0 += kKmTile) {
        __syncthreads();  // the previous tile is no longer read
#pragma unroll
        for (int u = 0; u < kStage; u++) {
            const int t = tid + u * 256;
            if (tile4 % 256 == 0 || t < tile4) reinterpret_cast<km_f4 *>(ctile)[t] = stage[u];
        }
        __syncthreads();
        km_fetch_tile<dim, kStage>(stage, centroids, c0 + kKmTile < k ? c0 + kKmTile : c0, k, tid);  // last: unused
        const int cnt = k - c0 < kKmTile ? k - c0 : kKmTile;
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
                if (c == 0 || (DOT ? (total > bd[p]) : (total < bd[p]))) {
                    bd[p] = total;
                    best[p] = c;
                }
            }
        }
    }
    if ((tid & 15) == 0) {
#pragma unroll
        for (int p = 0; p < kKmRows; p++)
            if (i0 + p < n) {
                if (changed && assign[i0 + p] != best[p]) *changed = 1;
                assign[i0 + p] = best[p];
            }
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

// hist[part][c] becomes the number of cluster-c points in earlier parts; counts / offsets as the
// update kernel wants them
__global__ __launch_bounds__(256) void km_scan_kernel(int32_t *__restrict__ hist, int parts, int k,
                                                      int64_t *__restrict__ counts, int64_t *__restrict__ offsets)
{
    __shared__ int64_t seg[256];
    const int tid = threadIdx.x;
    const int per = (k + 255) / 256;
    const int cb = tid * per, ce = cb + per < k ? cb + per : k;
    int64_t mine = 0;
    for (int c = cb; c < ce; c++) {
        int32_t run = 0;
        for (int p = 0; p < parts; p++) {
            const int32_t v = hist[static_cast<int64_t>(p) * k + c];
            hist[static_cast<int64_t>(p) * k + c] = run;
            run += v;
        }
        counts[c] = run;
        mine += run;
    }
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
constexpr int kKmAhead = 16;
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
        float sum = 0.0f;
        int64_t j = 0;
        // kKmAhead gathered loads in flight, added in member order
        for (; j + kKmAhead <= cnt; j += kKmAhead) {
            float x[kKmAhead];
#pragma unroll
            for (int u = 0; u < kKmAhead; u++) x[u] = vectors[mem[j + u] * dim + d];
#pragma unroll
            for (int u = 0; u < kKmAhead; u++) sum += x[u];
        }
        for (; j < cnt; j++) sum += vectors[mem[j] * dim + d];
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

}  // namespace vg

// pick the register-resident assignment when the row shape allows it
template <int NBLK>
static int32_t km_launch_regs(bool dot, const float *v, int64_t n, const float *cent, int k, int32_t *assign, int *changed,
                              hipStream_t st)
{
    const unsigned gx = static_cast<unsigned>((n + 16 * vg::kKmRows - 1) / (16 * vg::kKmRows));
    if (dot)
        VG_LAUNCH((vg::km_assign_regs_kernel<true, NBLK>), dim3(gx), dim3(256), 0, st, v, n, cent, k, assign, changed);
    else
        VG_LAUNCH((vg::km_assign_regs_kernel<false, NBLK>), dim3(gx), dim3(256), 0, st, v, n, cent, k, assign, changed);
    return VG_OK;
}

static int32_t km_launch_assign(bool dot, const float *v, int64_t n, int dim, const float *cent, int k, int32_t *assign,
                                int *changed, hipStream_t st)
{
    const bool aligned = (reinterpret_cast<uintptr_t>(v) & 15) == 0 && (reinterpret_cast<uintptr_t>(cent) & 15) == 0;
    if (aligned) {
        switch (dim) {
        case 64: return km_launch_regs<1>(dot, v, n, cent, k, assign, changed, st);
        case 128: return km_launch_regs<2>(dot, v, n, cent, k, assign, changed, st);
        case 256: return km_launch_regs<4>(dot, v, n, cent, k, assign, changed, st);
        case 384: return km_launch_regs<6>(dot, v, n, cent, k, assign, changed, st);
        case 512: return km_launch_regs<8>(dot, v, n, cent, k, assign, changed, st);
        case 768: return km_launch_regs<12>(dot, v, n, cent, k, assign, changed, st);
        case 1024: return km_launch_regs<16>(dot, v, n, cent, k, assign, changed, st);
        default: break;
        }
    }
    const unsigned gx = static_cast<unsigned>((n + 15) / 16);
    if (dot)
        VG_LAUNCH(vg::km_assign_kernel<true>, dim3(gx), dim3(256), 0, st, v, n, dim, cent, k, assign, changed);
    else
        VG_LAUNCH(vg::km_assign_kernel<false>, dim3(gx), dim3(256), 0, st, v, n, dim, cent, k, assign, changed);
    return VG_OK;
}

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
    {
        vg::ProfScope prof(ctx, "km_assign", st);
        VG_TRY(km_launch_assign(metric != VG_METRIC_L2, v.ptr, n, dim, c.ptr, k, o.ptr, nullptr, st));
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
    for (int it = 0; it < max_iter; it++) {
        VG_HIP(hipMemsetAsync(changed.ptr, 0, sizeof(int), st));
        {
            vg::ProfScope prof(ctx, "km_assign", st);
            VG_TRY(km_launch_assign(metric != VG_METRIC_L2, v.ptr, n, dim, cent.ptr, k, assign.ptr, changed.ptr, st));
        }
        int hchanged = 0;
        VG_HIP(hipMemcpyAsync(&hchanged, changed.ptr, sizeof(int), hipMemcpyDeviceToHost, st));
        VG_HIP(hipStreamSynchronize(st));  // one sync per Lloyd iteration: the convergence test is the host's
        if (!hchanged) break;               // kmeans.go:101-103
        vg::ProfScope prof(ctx, "km_update", st);  // member lists + per-cluster sums in index order
        if (sorted) {
            VG_LAUNCH(vg::km_hist_kernel, dim3(parts), dim3(256), 0, st, assign.ptr, n, k, hist.ptr);
            VG_LAUNCH(vg::km_scan_kernel, dim3(1), dim3(256), 0, st, hist.ptr, parts, k, counts.ptr, offsets.ptr);
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
        VG_LAUNCH(vg::km_update_kernel, dim3(static_cast<unsigned>((dim + 255) / 256), k), dim3(256), 0, st,
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
