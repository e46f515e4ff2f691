// k_pq_train.hip — ProductQuantizer Train / Encode / Decode / ComputeAsymmetricDistance on the
// device (internal/quantization/pq.go:68-260, :275-433).
//
// Determinism: every fp32 operation the reference performs sequentially (k-means++ running
// sums, the per-cluster coordinate sums of updateCentroids, the per-term int8-dequant L2) is
// performed here in the same order, so for a given seed the device result equals the CPU
// oracle's bit for bit.  Parallelism comes from the independent units:
// sub-quantizers x points (assignment), sub-quantizers x clusters x coordinates (update).
#include <algorithm>

#include "vg_device.hpp"
#include "vg_hnsw_layer.hpp"
#include "vg_internal.hpp"

namespace vg {

// ---- counter-based RNG shared with the oracle (vgo_rng_u64) ---------------------------------
__host__ __device__ inline uint64_t splitmix64(uint64_t x)
{
    x += 0x9e3779b97f4a7c15ULL;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL;
    return x ^ (x >> 31);
}
__host__ __device__ inline uint64_t rng_u64(uint64_t seed, uint64_t a, uint64_t b, uint64_t c)
{
    uint64_t h = splitmix64(seed);
    h = splitmix64(h ^ a);
    h = splitmix64(h ^ b);
    h = splitmix64(h ^ c);
    return h;
}
__device__ inline float rng_f32(uint64_t r) { return static_cast<float>(r >> 40) * (1.0f / 16777216.0f); }

// squaredL2Avx512 (floats_avx512.c:69-129) of two short vectors by ONE thread: n < 64 is only the
// FMA-contracted scalar tail; longer sub-vectors emulate the 4x16 accumulators and the tree.
__device__ inline float l2_avx512_thread(const float *a, const float *b, int n)
{
    float total = 0.0f;
    int i = 0;
    if (n >= 64) {
        float acc[64];
        for (int l = 0; l < 64; l++) acc[l] = 0.0f;
        for (; i + 64 <= n; i += 64)
            for (int l = 0; l < 64; l++) {
                const float d = a[i + l] - b[i + l];
                acc[l] = __builtin_fmaf(d, d, acc[l]);
            }
        float s[16];
        for (int l = 0; l < 16; l++) s[l] = (acc[l] + acc[16 + l]) + (acc[32 + l] + acc[48 + l]);
        total = reduce16_regs(s);
    }
    for (; i < n; i++) {
        const float d = a[i] - b[i];
        total = __builtin_fmaf(d, d, total);
    }
    return total;
}

// squaredL2Int8DequantizedGeneric (internal/simd/kernels.go:354-362) against a pre-dequantised
// centroid (v = float32(code)*scale + offset, two separately rounded ops done once)
__device__ inline float l2_deq_thread(const float *q, const float *v, int n)
{
    float sum = 0.0f;
    for (int i = 0; i < n; i++) {
        const float d = q[i] - v[i];
        const float dd = d * d;
        sum = sum + dd;
    }
    return sum;
}

// ---- training layout: the sub-vectors of one sub-quantizer made contiguous ([m][n][sd]) so that
// every training kernel streams its own slab instead of 32-byte slices of 3 KiB rows ----------
__global__ void pq_slab_kernel(const float *__restrict__ vectors, int64_t n, int dim, int sd, int sub0,
                               float *__restrict__ slabs)
{
    const int ls = blockIdx.y;
    const int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (e >= n * sd) return;
    const int64_t i = e / sd;
    const int j = static_cast<int>(e % sd);
    slabs[static_cast<int64_t>(ls) * n * sd + e] = vectors[i * dim + static_cast<int64_t>(sub0 + ls) * sd + j];
}

// squaredL2Avx512 of a slab row against a centroid in LDS.  SD in {4, 8, 16}: the row comes in as
// float4 loads and the whole distance is the FMA-contracted scalar tail (n < 64); SD == 0 is the
// generic path.
template <int SD>
__device__ __forceinline__ float l2_train(const float *__restrict__ row, const float *cen, int sd)
{
    if constexpr (SD == 0) {
        return l2_avx512_thread(row, cen, sd);
    } else {
        float total = 0.0f;
#pragma unroll
        for (int t = 0; t < SD / 4; t++) {
            const float4 x = reinterpret_cast<const float4 *>(row)[t];
            const float4 c = reinterpret_cast<const float4 *>(cen)[t];
            float d = x.x - c.x;
            total = __builtin_fmaf(d, d, total);
            d = x.y - c.y;
            total = __builtin_fmaf(d, d, total);
            d = x.z - c.z;
            total = __builtin_fmaf(d, d, total);
            d = x.w - c.w;
            total = __builtin_fmaf(d, d, total);
        }
        return total;
    }
}

// ---- k-means++ (pq.go:281-338): one workgroup per sub-quantizer -----------------------------
// The reference keeps minDistSq[] and, per new centroid, (1) updates it, (2) adds it up, (3) walks the same running
// sum until it passes rand * sum.  Through r04 (2) was the reference's own chain of n dependent fp32 additions (60 ms of
// the 88 ms Train at 65 536 x 768, and the same 60 ms however few sub-quantizers a GPU of a sharded Train holds).
// Since r05 the running sum is BLOCKED — one definition, the same in the CPU restatement the tests check this kernel
// against, stated and justified as a deviation from pq.go:296-336 in docs/history/rounds_r02_r05.md §12.4 and include/vecgo_hip.h (the reference's random numbers
// are unseeded, so no run of it can be reproduced bit for bit anyway): T_b = the balanced pairwise tree over the 64
// elements of block b (the six shuffle steps of a wave that holds the block one element per lane), P_b = P_(b-1) + T_b
// over the blocks in order, sum = P_last; the pick = first block whose prefix is not below the target, then the
// reference's sequential walk inside that block from P_(b-1) (the block's last element if that walk ends below).
// So per new centroid: every wave updates and sums whole blocks (coalesced: a block is 64 consecutive slab rows), one
// lane scans the n / 64 block totals in LDS, and the pick reads one block.
constexpr int kPPThreads = 1024;
constexpr int kPPBlock = 64;
constexpr int kPPScan = 4096;  // block totals scanned per LDS chunk (n <= 262 144 never leaves LDS)

// tot[0 .. len): block totals -> prefixes continuing `run`; one thread, the LDS reads 16 ahead of the additions
__device__ __forceinline__ float pp_scan_chunk(float *tot, int len, float run)
{
    int t = 0;
    for (; t + 16 <= len; t += 16) {
        float4 x[4];
#pragma unroll
        for (int u = 0; u < 4; u++) x[u] = reinterpret_cast<const float4 *>(tot + t)[u];
        float4 y[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            run = run + x[u].x; y[u].x = run;
            run = run + x[u].y; y[u].y = run;
            run = run + x[u].z; y[u].z = run;
            run = run + x[u].w; y[u].w = run;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) reinterpret_cast<float4 *>(tot + t)[u] = y[u];
    }
    for (; t < len; t++) {
        run = run + tot[t];
        tot[t] = run;
    }
    return run;
}

template <int SD>
__global__ __launch_bounds__(kPPThreads) void pq_kmeanspp_kernel(
    const float *__restrict__ slabs, int64_t n, int sd_rt, int k, uint64_t seed,
    float *__restrict__ mind_all, float *__restrict__ pref_all, float *__restrict__ cent_all, int sub0)
{
    __shared__ __attribute__((aligned(16))) float tot[kPPScan];
    __shared__ __attribute__((aligned(16))) float cur[256];  // current centroid (sd <= 256)
    __shared__ float walk[kPPBlock];
    __shared__ float s_sum;
    __shared__ int s_blk;
    __shared__ long long s_chosen;
    // sub = the sub-quantizer (global index: RNG stream); ls = its slot in the scratch arrays of
    // this call, which may train only the range [sub0, sub0 + gridDim.x)
    const int sd = SD ? SD : sd_rt;
    const int ls = blockIdx.x, sub = sub0 + ls;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t nblk = (n + kPPBlock - 1) / kPPBlock;
    const bool in_lds = nblk <= kPPScan;  // the prefixes live in LDS for the whole kernel
    const float *slab = slabs + static_cast<int64_t>(ls) * n * sd;
    float *mind = mind_all + static_cast<int64_t>(ls) * n;
    float *pref = pref_all + static_cast<int64_t>(ls) * nblk;
    float *cent = cent_all + static_cast<int64_t>(ls) * k * sd;

    if (n < k) {  // pq.go:285-291
        for (int t = tid; t < k * sd; t += kPPThreads) cent[t] = slab[static_cast<int64_t>((t / sd) % n) * sd + (t % sd)];
        return;
    }
    uint64_t ctr = 0;
    long long first = static_cast<long long>(rng_u64(seed, sub, 1, ctr++) % static_cast<uint64_t>(n));
    for (int t = tid; t < sd; t += kPPThreads) {
        cur[t] = slab[first * sd + t];
        cent[t] = cur[t];
    }
    __syncthreads();
    for (int c = 0; c < k; c++) {
        // (c >= 1) choose the next centroid from the running sum; (c == 0) it is `first`
        if (c >= 1) {
            const float sum = s_sum;
            if (sum == 0.0f) {  // pq.go:306-310: random vector, mind/sum untouched
                if (tid == 0) s_chosen = static_cast<long long>(rng_u64(seed, sub, 1, ctr) % static_cast<uint64_t>(n));
                ctr++;
                __syncthreads();
                const long long ch = s_chosen;
                for (int t = tid; t < sd; t += kPPThreads) cent[c * sd + t] = slab[ch * sd + t];
                __syncthreads();
                continue;
            }
            const float target = rng_f32(rng_u64(seed, sub, 1, ctr)) * sum;
            ctr++;
            if (tid == 0) {
                s_blk = 0x7fffffff;
                s_chosen = 0;  // pq.go:317 `chosen := 0`
            }
            __syncthreads();
            // first block whose prefix is not below the target (`!(prefix < target)` also stops at a NaN prefix)
            for (int64_t b = tid; b < nblk; b += kPPThreads)
                if (!((in_lds ? tot[b] : pref[b]) < target)) {
                    atomicMin(&s_blk, static_cast<int>(b));
                    break;
                }
            __syncthreads();
            const int blk = s_blk;
            if (blk != 0x7fffffff) {
                const int64_t e0 = static_cast<int64_t>(blk) * kPPBlock;
                const int len = static_cast<int>(n - e0 < kPPBlock ? n - e0 : kPPBlock);
                if (tid < len) walk[tid] = mind[e0 + tid];
                __syncthreads();
                if (tid == 0) {
                    float cum = blk ? (in_lds ? tot[blk - 1] : pref[blk - 1]) : 0.0f;
                    long long ch = e0 + len - 1;  // the sequential walk may end below the tree's total
                    for (int t = 0; t < len; t++) {
                        cum = cum + walk[t];
                        if (cum >= target) {
                            ch = e0 + t;
                            break;
                        }
                    }
                    s_chosen = ch;
                }
                __syncthreads();
            }
            const long long ch = s_chosen;
            for (int t = tid; t < sd; t += kPPThreads) {
                cur[t] = slab[ch * sd + t];
                cent[c * sd + t] = cur[t];
            }
            __syncthreads();
        }
        if (c == k - 1 && c >= 1) {
            // the reference still updates minDistSq after the last centroid; the values are
            // never read again, skip the pass
            break;
        }
        // (1) minDistSq against the new centroid and the block totals: a wave per block of 64 consecutive points,
        // kPPUnroll blocks per trip with all their loads issued first (one block per trip ran at one L2 round trip per
        // block: 70 us per centroid)
        auto finish_block = [&](int64_t b, float mv) {
            // balanced pairwise tree: ((m0 + m1) + (m2 + m3)) + ... — lane i += lane i + 1, + 2, + 4, + 8 inside its row of 16
            // (DPP row_shl: one full-rate vector instruction each; the ds_bpermute form of __shfl_down was six LDS round trips
            // per 64 points), then the four row sums as (r0 + r16) + (r32 + r48): the same tree
            float x = mv;
            x = x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x101, 0xF, 0xF, false));
            x = x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x102, 0xF, 0xF, false));
            x = x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x104, 0xF, 0xF, false));
            x = x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x108, 0xF, 0xF, false));
            const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 0));
            const float r16 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 16));
            const float r32 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 32));
            const float r48 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 48));
            x = (r0 + r16) + (r32 + r48);
            if (lane == 0) {
                if (in_lds)
                    tot[b] = x;
                else
                    pref[b] = x;
            }
        };
#if defined(VG_PP_PROBE) && VG_PP_PROBE == 1  // stage probe (tools/build_variant.sh): no distance pass after the first
        if (c >= 1) {
            __syncthreads();
            continue;
        }
#endif
        if constexpr (SD != 0) {
            // software-pipelined: the loads of the wave's next kPPUnroll blocks are in flight while the current ones are
            // reduced (r05: one trip at a time — 8 trips of 128 blocks, each a full round trip to the memory-side cache the
            // 2 MB slab lives in — ran a CU at 52 GB/s: 49 us per centroid)
            constexpr int kPPUnroll = SD >= 16 ? 2 : (SD >= 8 ? 3 : 4);  // two sets of these blocks live in registers
            constexpr int64_t kTrip = (kPPThreads / 64) * kPPUnroll;
            float cv[SD];
#pragma unroll
            for (int t = 0; t < SD; t++) cv[t] = cur[t];
            float4 ra[kPPUnroll][SD / 4], rb[kPPUnroll][SD / 4];
            float oa[kPPUnroll], ob[kPPUnroll];
            auto load = [&](int64_t b0, float4 (&r)[kPPUnroll][SD / 4], float (&old)[kPPUnroll]) {
#pragma unroll
                for (int u = 0; u < kPPUnroll; u++) {
                    const int64_t i = (b0 + u) * kPPBlock + lane;
                    const int64_t ic = i < n ? i : n - 1;  // past the end: a valid address, the value is not used
                    const float4 *v4 = reinterpret_cast<const float4 *>(slab + ic * SD);
#pragma unroll
                    for (int t = 0; t < SD / 4; t++) r[u][t] = v4[t];
#if defined(VG_PP_PROBE) && VG_PP_PROBE == 3  // stage probe: no minDistSq traffic
                    old[u] = 1.0f;
#else
                    old[u] = c != 0 ? mind[ic] : 0.0f;
#endif
                }
            };
            auto work = [&](int64_t b0, const float4 (&r)[kPPUnroll][SD / 4], const float (&old)[kPPUnroll]) {
#pragma unroll
                for (int u = 0; u < kPPUnroll; u++) {
                    const int64_t b = b0 + u;
                    if (b >= nblk) break;  // uniform per wave
                    const int64_t i = b * kPPBlock + lane;
                    float total = 0.0f;  // l2_train<SD>: the n < 64 form of squaredL2Avx512, one FMA chain
#pragma unroll
                    for (int t = 0; t < SD / 4; t++) {
                        float d = r[u][t].x - cv[4 * t];
                        total = __builtin_fmaf(d, d, total);
                        d = r[u][t].y - cv[4 * t + 1];
                        total = __builtin_fmaf(d, d, total);
                        d = r[u][t].z - cv[4 * t + 2];
                        total = __builtin_fmaf(d, d, total);
                        d = r[u][t].w - cv[4 * t + 3];
                        total = __builtin_fmaf(d, d, total);
                    }
                    float mv = 0.0f;  // elements past n count as +0
                    if (i < n) {
                        mv = total;
                        if (c != 0) mv = total < old[u] ? total : old[u];
#if !(defined(VG_PP_PROBE) && VG_PP_PROBE == 3)
                        mind[i] = mv;
#endif
                    }
#if defined(VG_PP_PROBE) && VG_PP_PROBE == 4  // stage probe: no block totals
                    if (lane == 0 && mv == 12345.0f) tot[b] = mv;
#else
                    finish_block(b, mv);
#endif
                }
            };
            int64_t b0 = static_cast<int64_t>(wave) * kPPUnroll;
            if (b0 < nblk) load(b0, ra, oa);
            for (; b0 < nblk; b0 += 2 * kTrip) {
                const bool more = b0 + kTrip < nblk;
                if (more) load(b0 + kTrip, rb, ob);
                work(b0, ra, oa);
                if (more) {
                    if (b0 + 2 * kTrip < nblk) load(b0 + 2 * kTrip, ra, oa);
                    work(b0 + kTrip, rb, ob);
                }
            }
        } else {
            for (int64_t b = wave; b < nblk; b += kPPThreads / 64) {
                const int64_t i = b * kPPBlock + lane;
                float mv = 0.0f;
                if (i < n) {
                    const float d = l2_train<SD>(slab + i * sd, cur, sd);
                    mv = d;
                    if (c != 0) {
                        const float oldv = mind[i];
                        mv = d < oldv ? d : oldv;
                    }
                    mind[i] = mv;
                }
                finish_block(b, mv);
            }
        }
        __syncthreads();
        // (2) prefixes of the block totals, in block order, by one thread
#if defined(VG_PP_PROBE) && VG_PP_PROBE == 2  // stage probe: no prefix scan
        if (tid == 0) s_sum = 1.0f;
        __syncthreads();
        continue;
#endif
        if (in_lds) {
            if (tid == 0) s_sum = pp_scan_chunk(tot, static_cast<int>(nblk), 0.0f);
        } else {
            float run = 0.0f;  // live in thread 0
            for (int64_t b0 = 0; b0 < nblk; b0 += kPPScan) {
                const int len = static_cast<int>(nblk - b0 < kPPScan ? nblk - b0 : kPPScan);
                for (int t = tid; t < len; t += kPPThreads) tot[t] = pref[b0 + t];
                __syncthreads();
                if (tid == 0) run = pp_scan_chunk(tot, len, run);
                __syncthreads();
                for (int t = tid; t < len; t += kPPThreads) pref[b0 + t] = tot[t];
                __syncthreads();
            }
            if (tid == 0) s_sum = run;
        }
        __syncthreads();
    }
}

// ---- Lloyd assignment (pq.go:353-386, :416-433): thread per (point, sub-quantizer) ----------
__global__ __launch_bounds__(256) void pq_assign_kernel(const float *__restrict__ slabs, int64_t n, int sd,
                                                        int k, const float *__restrict__ cent_all,
                                                        int32_t *__restrict__ assign_all,
                                                        int *__restrict__ changed,
                                                        const int *__restrict__ done)
{
    extern __shared__ float cent[];  // k*sd
    const int ls = blockIdx.y;
    if (done[ls]) return;
    for (int t = threadIdx.x; t < k * sd; t += blockDim.x)
        cent[t] = cent_all[static_cast<int64_t>(ls) * k * sd + t];
    __syncthreads();
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *v = slabs + (static_cast<int64_t>(ls) * n + i) * sd;
    float best = 3.40282346638528859811704183484516925440e+38f;
    int bi = 0;
    for (int c = 0; c < k; c++) {
        const float d = l2_avx512_thread(v, cent + c * sd, sd);
        if (d < best) {
            best = d;
            bi = c;
        }
    }
    int32_t *a = assign_all + static_cast<int64_t>(ls) * n + i;
    if (*a != bi) {
        *a = bi;
        changed[ls] = 1;
    }
}

// The same for sub-vector dims 4 / 8 / 16: rows in registers (kAssignRows per thread), centroids as
// float4 LDS broadcasts.  Distances are the n < 64 form of squaredL2Avx512: one FMA chain.
constexpr int kAssignRows = 4;
typedef float float2v __attribute__((ext_vector_type(2)));
template <int SD>
__global__ __launch_bounds__(256) void pq_assign_vec_kernel(const float *__restrict__ slabs, int64_t n, int k,
                                                            const float *__restrict__ cent_all,
                                                            int32_t *__restrict__ assign_all,
                                                            int *__restrict__ changed,
                                                            const int *__restrict__ done)
{
    extern __shared__ float cent[];  // k*SD
    const int ls = blockIdx.y;
    if (done[ls]) return;
    for (int t = threadIdx.x; t < k * SD; t += blockDim.x)
        cent[t] = cent_all[static_cast<int64_t>(ls) * k * SD + t];
    __syncthreads();
    const int64_t i0 = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) * kAssignRows;
    if (i0 >= n) return;
    // rows go in pairs through the packed fp32 ALU ops (v_pk_add_f32 / v_pk_fma_f32): each half is the
    // IEEE op of its own row, so the per-row FMA chain is unchanged
    static_assert(kAssignRows % 2 == 0, "rows are processed in pairs");
    float2v q[kAssignRows / 2][SD];
#pragma unroll
    for (int r = 0; r < kAssignRows; r++) {
        const int64_t i = i0 + r < n ? i0 + r : n - 1;
        const float4 *v4 = reinterpret_cast<const float4 *>(slabs + (static_cast<int64_t>(ls) * n + i) * SD);
#pragma unroll
        for (int t = 0; t < SD / 4; t++) {
            const float4 x = v4[t];
            q[r / 2][4 * t][r & 1] = x.x;
            q[r / 2][4 * t + 1][r & 1] = x.y;
            q[r / 2][4 * t + 2][r & 1] = x.z;
            q[r / 2][4 * t + 3][r & 1] = x.w;
        }
    }
    int bi[kAssignRows];
    float best[kAssignRows];
#pragma unroll
    for (int r = 0; r < kAssignRows; r++) {
        bi[r] = 0;
        best[r] = 3.40282346638528859811704183484516925440e+38f;
    }
    for (int c = 0; c < k; c++) {
        float cv[SD];
        const float4 *c4 = reinterpret_cast<const float4 *>(cent + c * SD);
#pragma unroll
        for (int t = 0; t < SD / 4; t++) {
            const float4 x = c4[t];
            cv[4 * t] = x.x; cv[4 * t + 1] = x.y; cv[4 * t + 2] = x.z; cv[4 * t + 3] = x.w;
        }
#pragma unroll
        for (int rp = 0; rp < kAssignRows / 2; rp++) {
            float2v total = {0.0f, 0.0f};
#pragma unroll
            for (int t = 0; t < SD; t++) {
                const float2v cc = {cv[t], cv[t]};
                const float2v d = q[rp][t] - cc;
                total = __builtin_elementwise_fma(d, d, total);
            }
#pragma unroll
            for (int h = 0; h < 2; h++)
                if (total[h] < best[2 * rp + h]) {
                    best[2 * rp + h] = total[h];
                    bi[2 * rp + h] = c;
                }
        }
    }
    bool any = false;
#pragma unroll
    for (int r = 0; r < kAssignRows; r++)
        if (i0 + r < n) {
            int32_t *a = assign_all + static_cast<int64_t>(ls) * n + i0 + r;
            if (*a != bi[r]) {
                *a = bi[r];
                any = true;
            }
        }
    if (any) changed[ls] = 1;
}

// ---- Lloyd update (pq.go:388-414).  The reference sums every cluster's points in index order in a
// single pass; the same order comes out of a stable counting sort of the point ids by cluster
// (one workgroup per sub-quantizer) followed by one thread per (cluster, coordinate) walking its
// own segment ----
constexpr int kBucketThreads = 1024;  // (256: 0.27 ms per iteration at 65 536 points — 256 chunks of three barriers each)

__global__ __launch_bounds__(kBucketThreads) void pq_bucket_kernel(int64_t n, int k,
                                                                  const int32_t *__restrict__ assign_all,
                                                                  int32_t *__restrict__ order_all,
                                                                  int32_t *__restrict__ seg_all,
                                                                  const int *__restrict__ changed,
                                                                  const int *__restrict__ done)
{
    const int ls = blockIdx.x, t = threadIdx.x;
    if (done[ls] || !changed[ls]) return;
    static_assert(kBucketThreads >= 256, "threads 0 .. 255 own one cluster id each in the base update");
    __shared__ int32_t cnt[256];
    __shared__ int32_t base[256];
    __shared__ int32_t wcnt[kBucketThreads / 64][256];  // per wave: members of each cluster in this chunk
    const int32_t *assign = assign_all + static_cast<int64_t>(ls) * n;
    int32_t *order = order_all + static_cast<int64_t>(ls) * n;
    int32_t *seg = seg_all + static_cast<int64_t>(ls) * 2 * k;
    const int w = t >> 6, lane = t & 63;
    if (t < 256) {
        cnt[t] = 0;
        for (int u = 0; u < kBucketThreads / 64; u++) wcnt[u][t] = 0;
    }
    __syncthreads();
    for (int64_t i = t; i < n; i += kBucketThreads) atomicAdd(&cnt[assign[i]], 1);
    __syncthreads();
    if (t == 0) {
        int32_t run = 0;
        for (int c = 0; c < k; c++) {
            base[c] = run;
            seg[2 * c] = run;
            seg[2 * c + 1] = cnt[c];
            run += cnt[c];
        }
    }
    __syncthreads();
    for (int64_t i0 = 0; i0 < n; i0 += kBucketThreads) {
        const int64_t i = i0 + t;
        const bool active = i < n;
        const int32_t key = active ? assign[i] : 0;
        // lanes of this wave holding the same cluster id (ids < 256: eight ballots)
        uint64_t peers = __ballot(active);
#pragma unroll
        for (int bit = 0; bit < 8; bit++) {
            const bool set = (key >> bit) & 1;
            const uint64_t bb = __ballot(set);
            peers &= set ? bb : ~bb;
        }
        const int rank_w = __popcll(peers & ((1ull << lane) - 1ull));
        if (active && rank_w == 0) wcnt[w][key] = __popcll(peers);
        __syncthreads();
        if (active) {
            int r = rank_w;  // stable: earlier waves of the chunk first, then earlier lanes
            for (int u = 0; u < w; u++) r += wcnt[u][key];
            order[base[key] + r] = static_cast<int32_t>(i);
        }
        __syncthreads();
        if (t < 256) {
            int32_t add = 0;
#pragma unroll
            for (int u = 0; u < kBucketThreads / 64; u++) {
                add += wcnt[u][t];
                wcnt[u][t] = 0;
            }
            base[t] += add;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void pq_update_kernel(const float *__restrict__ slabs, int64_t n,
                                                        int sd, int k, int iter, uint64_t seed,
                                                        const int32_t *__restrict__ order_all,
                                                        const int32_t *__restrict__ seg_all,
                                                        float *__restrict__ cent_all,
                                                        const int *__restrict__ changed,
                                                        const int *__restrict__ done, int sub0)
{
    const int ls = blockIdx.y, sub = sub0 + ls;
    if (done[ls] || !changed[ls]) return;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= k * sd) return;
    const int c = t / sd, j = t % sd;
    const int32_t *order = order_all + static_cast<int64_t>(ls) * n;
    const int32_t *seg = seg_all + static_cast<int64_t>(ls) * 2 * k;
    const float *col = slabs + static_cast<int64_t>(ls) * n * sd + j;
    const int32_t start = seg[2 * c], count = seg[2 * c + 1];
    float sum = 0.0f;
    // member id -> row is a chain of two dependent loads: kUpdAhead of them in flight per thread, added in member order
    // (one at a time the kernel ran at two L2 round trips per member: 0.32 ms per iteration at 65 536 x 96)
    constexpr int kUpdAhead = 16;
    int32_t p = 0;
    for (; p + kUpdAhead <= count; p += kUpdAhead) {
        int32_t id[kUpdAhead];
        float x[kUpdAhead];
#pragma unroll
        for (int u = 0; u < kUpdAhead; u++) id[u] = order[start + p + u];
#pragma unroll
        for (int u = 0; u < kUpdAhead; u++) x[u] = col[static_cast<int64_t>(id[u]) * sd];
#pragma unroll
        for (int u = 0; u < kUpdAhead; u++) sum += x[u];
    }
    for (; p < count; p++) sum += col[static_cast<int64_t>(order[start + p]) * sd];
    float *dst = cent_all + (static_cast<int64_t>(ls) * k + c) * sd + j;
    if (count > 0) {
        *dst = sum / static_cast<float>(count);
    } else {  // pq.go:408-411 re-seed an empty cluster with a random vector
        const int64_t idx = static_cast<int64_t>(rng_u64(seed, sub, 2 + static_cast<uint64_t>(iter), c) %
                                                 static_cast<uint64_t>(n));
        *dst = col[idx * sd];
    }
}

// after assign+update of one iteration: a sub-quantizer whose assignments did not change stops
// (pq.go:346-348 `break`); `changed` is cleared for the next iteration
__global__ void pq_iter_end_kernel(int m, int *__restrict__ changed, int *__restrict__ done)
{
    const int sub = blockIdx.x * blockDim.x + threadIdx.x;
    if (sub >= m) return;
    if (!done[sub] && !changed[sub]) done[sub] = 1;
    changed[sub] = 0;
}

// ---- int8 quantisation of the trained centroids (pq.go:97-136) --------------------------------
__global__ __launch_bounds__(256) void pq_quantize_kernel(const float *__restrict__ cent_all, int k,
                                                          int sd, int8_t *__restrict__ codebooks,
                                                          float *__restrict__ scales,
                                                          float *__restrict__ offsets, int sub0)
{
    __shared__ float smin[256], smax[256];
    const int ls = blockIdx.x, sub = sub0 + ls;
    const int tid = threadIdx.x;
    const int cnt = k * sd;
    const float *cent = cent_all + static_cast<int64_t>(ls) * cnt;
    float mn = 3.40282346638528859811704183484516925440e+38f, mx = -mn;
    for (int t = tid; t < cnt; t += 256) {
        const float v = cent[t];
        if (v < mn) mn = v;
        if (v > mx) mx = v;
    }
    smin[tid] = mn;
    smax[tid] = mx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            if (smin[tid + s] < smin[tid]) smin[tid] = smin[tid + s];
            if (smax[tid + s] > smax[tid]) smax[tid] = smax[tid + s];
        }
        __syncthreads();
    }
    mn = smin[0];
    mx = smax[0];
    if (mx == mn) mx = mn + 1e-6f;
    const float scale = (mx - mn) / 255.0f;
    const float offset = mn + 128.0f * scale;
    if (tid == 0) {
        scales[sub] = scale;
        offsets[sub] = offset;
    }
    for (int t = tid; t < cnt; t += 256) {
        const float q = (cent[t] - mn) / scale;
        // math.Round(float64(q)): half away from zero
        int val = static_cast<int>(round(static_cast<double>(q)));
        if (val < 0) val = 0;
        if (val > 255) val = 255;
        codebooks[static_cast<int64_t>(sub) * cnt + t] = static_cast<int8_t>(val - 128);
    }
}

// ---- Encode (pq.go:147-176): workgroup = (sub-quantizer, 256 rows); the sub-quantizer's
// dequantised codebook sits in LDS and is read as a broadcast -----------------------------------
__global__ __launch_bounds__(256) void pq_encode_kernel(const float *__restrict__ vectors, int64_t n,
                                                        int dim, int m, int sd, int k,
                                                        const int8_t *__restrict__ codebooks,
                                                        const float *__restrict__ scales,
                                                        const float *__restrict__ offsets,
                                                        uint8_t *__restrict__ codes)
{
    extern __shared__ float deq[];  // k*sd
    const int sub = blockIdx.y;
    const float scale = scales[sub], offset = offsets[sub];
    for (int t = threadIdx.x; t < k * sd; t += blockDim.x) {
        float v = static_cast<float>(codebooks[static_cast<int64_t>(sub) * k * sd + t]) * scale;
        deq[t] = v + offset;
    }
    __syncthreads();
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *v = vectors + i * dim + static_cast<int64_t>(sub) * sd;
    float q[16];
    const bool small = sd <= 16;
    if (small)
        for (int t = 0; t < sd; t++) q[t] = v[t];
    const float *qp = small ? q : v;
    int best = 0;
    float bd = l2_deq_thread(qp, deq, sd);
    for (int c = 1; c < k; c++) {
        const float d = l2_deq_thread(qp, deq + c * sd, sd);
        if (d < bd) {
            bd = d;
            best = c;
        }
    }
    codes[i * m + sub] = static_cast<uint8_t>(best);
}

// Encode for sub-vector dims that are a multiple of 4 (8 at the BASELINE shape): the dequantised
// centroid is read from LDS as float4 broadcasts (sd/4 reads instead of sd) and every thread scores
// kEncRows rows against it, so the 256 x sd table walk is amortised over kEncRows rows.  Same
// arithmetic as pq_encode_kernel: per dimension d = q - v, dd = d * d, sum = sum + dd, in order.
constexpr int kEncRows = 4;
template <int SD>
__global__ __launch_bounds__(256) void pq_encode_vec_kernel(const float *__restrict__ vectors, int64_t n,
                                                            int dim, int m, int k,
                                                            const int8_t *__restrict__ codebooks,
                                                            const float *__restrict__ scales,
                                                            const float *__restrict__ offsets,
                                                            uint8_t *__restrict__ codes)
{
    extern __shared__ float deq[];  // k*SD
    const int sub = blockIdx.y;
    const float scale = scales[sub], offset = offsets[sub];
    for (int t = threadIdx.x; t < k * SD; t += blockDim.x) {
        float v = static_cast<float>(codebooks[static_cast<int64_t>(sub) * k * SD + t]) * scale;
        deq[t] = v + offset;
    }
    __syncthreads();
    const int64_t i0 = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) * kEncRows;
    if (i0 >= n) return;
    // rows in pairs through the packed fp32 ops (v_pk_add_f32 / v_pk_mul_f32), each half the IEEE op
    // of its own row
    static_assert(kEncRows % 2 == 0, "rows are processed in pairs");
    float2v q[kEncRows / 2][SD];
#pragma unroll
    for (int r = 0; r < kEncRows; r++) {
        const int64_t i = i0 + r < n ? i0 + r : n - 1;
        const float4 *v4 = reinterpret_cast<const float4 *>(vectors + i * dim + static_cast<int64_t>(sub) * SD);
#pragma unroll
        for (int t = 0; t < SD / 4; t++) {
            const float4 x = v4[t];
            q[r / 2][4 * t][r & 1] = x.x;
            q[r / 2][4 * t + 1][r & 1] = x.y;
            q[r / 2][4 * t + 2][r & 1] = x.z;
            q[r / 2][4 * t + 3][r & 1] = x.w;
        }
    }
    int best[kEncRows];
    float bd[kEncRows];
#pragma unroll
    for (int r = 0; r < kEncRows; r++) {
        best[r] = 0;
        bd[r] = 3.40282346638528859811704183484516925440e+38f;
    }
    for (int c = 0; c < k; c++) {
        float cv[SD];
        const float4 *c4 = reinterpret_cast<const float4 *>(deq + c * SD);
#pragma unroll
        for (int t = 0; t < SD / 4; t++) {
            const float4 x = c4[t];
            cv[4 * t] = x.x; cv[4 * t + 1] = x.y; cv[4 * t + 2] = x.z; cv[4 * t + 3] = x.w;
        }
#pragma unroll
        for (int rp = 0; rp < kEncRows / 2; rp++) {
            float2v sum = {0.0f, 0.0f};
#pragma unroll
            for (int t = 0; t < SD; t++) {
                const float2v cc = {cv[t], cv[t]};
                const float2v d = q[rp][t] - cc;
                const float2v dd = d * d;
                sum = sum + dd;
            }
            // FindNearestCentroidInt8 (kernels.go:376-396): centroid 0 seeds the minimum whatever its
            // distance is (NaN included), later ones need a strict '<'
#pragma unroll
            for (int h = 0; h < 2; h++)
                if (c == 0 || sum[h] < bd[2 * rp + h]) {
                    bd[2 * rp + h] = sum[h];
                    best[2 * rp + h] = c;
                }
        }
    }
#pragma unroll
    for (int r = 0; r < kEncRows; r++)
        if (i0 + r < n) codes[(i0 + r) * m + sub] = static_cast<uint8_t>(best[r]);
}

__global__ void pq_decode_kernel(const uint8_t *__restrict__ codes, int64_t n, int dim, int m, int sd,
                                 int k, const int8_t *__restrict__ codebooks,
                                 const float *__restrict__ scales, const float *__restrict__ offsets,
                                 float *__restrict__ out)
{
    const int64_t gid = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (gid >= n * dim) return;
    const int64_t i = gid / dim;
    const int d = static_cast<int>(gid % dim);
    const int sub = d / sd, t = d % sd;
    const int c = codes[i * m + sub];
    const float v = static_cast<float>(codebooks[(static_cast<int64_t>(sub) * k + c) * sd + t]) * scales[sub];
    out[gid] = v + offsets[sub];
}

// sub-dimension 8, 256 centroids: a thread owns one sub-quantizer (its scale and offset loaded once) and walks `rpt`
// rows — no division per element, one 8-byte centroid load, two 16-byte stores (the element kernel above: 1.5 TB/s
// of output)
__global__ __launch_bounds__(256) void pq_decode8_kernel(const uint8_t *__restrict__ codes, int64_t n, int m,
                                                         const int8_t *__restrict__ codebooks, const float *__restrict__ scales,
                                                         const float *__restrict__ offsets, float *__restrict__ out, int rpt)
{
    // threads of the grid: (row block, sub-quantizer), sub-quantizer fastest — every lane of every wave has work
    const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const int64_t rb = t / m;
    const int sub = static_cast<int>(t - rb * m);
    const int64_t r0 = rb * rpt;
    if (r0 >= n) return;
    const float scale = scales[sub], offset = offsets[sub];
    const uint2 *cb = reinterpret_cast<const uint2 *>(codebooks) + static_cast<int64_t>(sub) * 256;
    for (int64_t row = r0; row < r0 + rpt && row < n; row++) {
        const uint2 e = cb[codes[row * m + sub]];
        float o[8];
#pragma unroll
        for (int t = 0; t < 8; t++) {
            const uint32_t w = t < 4 ? e.x : e.y;
            const float v = static_cast<float>(static_cast<int>(static_cast<int8_t>(w >> (8 * (t & 3))))) * scale;
            o[t] = v + offset;
        }
        float4 *dst = reinterpret_cast<float4 *>(out + (row * m + sub) * 8);
        dst[0] = make_float4(o[0], o[1], o[2], o[3]);
        dst[1] = make_float4(o[4], o[5], o[6], o[7]);
    }
}

// ComputeAsymmetricDistance (pq.go:234-260): thread per code row, terms added sequentially over m
__global__ void pq_asym_kernel(const float *__restrict__ query, const uint8_t *__restrict__ codes,
                               int64_t n, int dim, int m, int sd, int k,
                               const int8_t *__restrict__ codebooks, const float *__restrict__ scales,
                               const float *__restrict__ offsets, float *__restrict__ out)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float distance = 0.0f;
    for (int sub = 0; sub < m; sub++) {
        const int c = codes[i * m + sub];
        const int8_t *cb = codebooks + (static_cast<int64_t>(sub) * k + c) * sd;
        const float scale = scales[sub], offset = offsets[sub];
        float sum = 0.0f;
        for (int t = 0; t < sd; t++) {
            float v = static_cast<float>(cb[t]) * scale;
            v = v + offset;
            const float d = query[sub * sd + t] - v;
            const float dd = d * d;
            sum = sum + dd;
        }
        distance = distance + sum;
    }
    out[i] = distance;
}

// The same distances with the graph walks' term code (vg_hnsw_layer.hpp: sub-dimension 8, two sub-quantizers per
// packed-fp32 instruction, the query's constants laid out once per workgroup in LDS, 16 centroid loads in flight per
// lane): 1.12 -> ~0.1 ms per million codes at m = 96.  Bit for bit the loop above (the same five rounded operations per
// dimension, terms added in sub-quantizer order).
__global__ __launch_bounds__(256) void pq_asym_direct_kernel(const float *__restrict__ query, const uint8_t *__restrict__ codes,
                                                             int64_t n, int m, const int8_t *__restrict__ codebooks,
                                                             const float *__restrict__ scales, const float *__restrict__ offsets,
                                                             float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) float asym_qprep[];
    for (int e = threadIdx.x; e < (m >> 1) * kPqPairFloats; e += blockDim.x) {
        const int p = e / kPqPairFloats, r = e - p * kPqPairFloats;
        float v;
        if (r < 16)
            v = query[(2 * p + (r & 1)) * 8 + (r >> 1)];
        else if (r < 18)
            v = scales[2 * p + (r - 16)];
        else
            v = offsets[2 * p + (r - 18)];
        asym_qprep[e] = v;  // pq_direct_prepare's image, by the whole workgroup
    }
    __syncthreads();
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = pq_direct_distance(codes + i * m, codebooks, scales, offsets, query, asym_qprep, m);
}


// ---- nearest centroid of every (row, sub-quantizer) by MFMA nomination + exact decision (sub-dimension 8, K = 256) ----
// Encode (pq.go:147-176, FindNearestCentroidInt8) and the Lloyd assignment pass of Train (pq.go:353-386) are both
// "256 eight-dimensional distances per (row, sub-quantizer), keep the smallest": 3 (Encode: sub, mul, add — Go does not
// fuse) or 2 (Train: sub, fma) vector lane-operations per element when computed as the reference writes them, and the
// r04 kernels ran at 0.66 / 0.58 of that rate.  The matrix cores compute s~(c) = |v_c|^2 - 2 x.v_c with one fused
// multiply-add per element (v_mfma_f32_32x32x2_f32, five of them for a 32 centroids x 32 rows tile: the eight dimensions
// and a ninth k slot that adds |v_c|^2 against a row of ones); the vector ALU only keeps, per row, the
// smallest score, where it was, and the second smallest (4 instructions per score).  As for k-means (k_kmeans.hip: the
// derivation is the same with dim = 8) a gap between the two smallest scores above
//     margin = 2 * 32 u (2 sqrt(X C) + C) + 2 * 12 u (sqrt X + sqrt C)^2      (u = 2^-24, X = |x_sub|^2, C = max_c |v_c|^2)
// proves that the reference's arithmetic picks the same centroid; every other (row, sub-quantizer) pair — one in ~10^4
// on random data — goes on a list and is decided by the reference-order loop (pq_fix_kernel).  Operand layout: the
// MFMA's two k slots of step kk are dimensions kk (lanes 0-31) and 4 + kk (lanes 32-63), so a lane's A and B operands
// are one 16-byte half of a centroid / of a row's sub-vector.
using pq_f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int kNomRowsPerWave = 1024, kNomRowsPerBlock = 4 * kNomRowsPerWave;  // (the centroid set-up of a block is ~2 us)

struct PqNomList {  // (row, sub-quantizer slot) pairs the scores could not decide
    int32_t row, sub;
};

// Block -> (sub-quantizer, row tile) of a 1-D grid of tiles * m workgroups: the m sub-quantizers of one row tile are
// consecutive workgroups, and (m a multiple of 32) the four sub-quantizers whose 32-byte pieces share a 128-byte line of
// a row go to the SAME XCD (workgroups b, b + 8, b + 16, ... share an XCD and its L2), 8 dispatch slots apart — the line
// is then fetched from HBM once instead of four times (grid.x = tiles, grid.y = sub-quantizer: every sub-quantizer
// streamed the rows again, 12 GB of line traffic per 1M x 768 Encode)
__device__ __forceinline__ void pq_nom_block(int m, int &sub, int &tile_x)
{
    const int b = blockIdx.x;
    tile_x = b / m;
    const int bp = b - tile_x * m;
    if ((m & 31) == 0) {
        const int xcd = bp & 7, slot = bp >> 3;             // slot 0 .. m/8 - 1
        sub = 4 * (xcd + 8 * (slot >> 2)) + (slot & 3);     // line group xcd + 8 * (slot / 4), piece slot % 4
    } else {
        sub = bp;
    }
}

// ENC: rows = vectors [n][dim], sub-quantizer blockIdx.y, centroids dequantised from the int8 codebook, out = codes.
// !ENC: rows = the training slabs [ls][n][8], centroids = cent_all [ls][256][8] fp32, out = assign_all (+ changed).
template <bool ENC>
__global__ __launch_bounds__(256) void pq_nominate_kernel(const float *__restrict__ rows, int64_t n, int dim, int m,
                                                          const int8_t *__restrict__ codebooks, const float *__restrict__ scales,
                                                          const float *__restrict__ offsets, const float *__restrict__ cent_all,
                                                          uint8_t *__restrict__ codes, int32_t *__restrict__ assign_all,
                                                          int *__restrict__ changed, const int *__restrict__ done,
                                                          PqNomList *__restrict__ list, int *__restrict__ list_count,
                                                          int64_t row_first /* rows [row_first, n) of `rows` */,
                                                          float extra_margin /* test hook: +Inf lists every pair */)
{
    __shared__ float s_cmax[4];
    int sub, tile_x;
    pq_nom_block(m, sub, tile_x);
    if (!ENC && done[sub]) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, c_in = lane & 31;
    // A operands of the 8 centroid blocks: -2 v (exact), and the ninth k slot: |v|^2 of this lane's centroid against a
    // row of ones (lanes 0-31; the tenth slot, lanes 32-63, multiplies zeros) — a fifth MFMA per tile instead of a
    // 128-register image of the |v|^2 column as the accumulators' starting value (which left one wave per SIMD)
    float4 a[8];
    float acn[8];
    float cmax = 0.0f;
#pragma unroll
    for (int cb = 0; cb < 8; cb++) {
        const int c = cb * 32 + c_in;
        float4 v;
        if (ENC) {
            const uint32_t w = reinterpret_cast<const uint32_t *>(codebooks)[(static_cast<int64_t>(sub) * 256 + c) * 2 + h];
            const float scale = scales[sub], offset = offsets[sub];
            float t0 = static_cast<float>(static_cast<int>(static_cast<int8_t>(w))) * scale;
            float t1 = static_cast<float>(static_cast<int>(static_cast<int8_t>(w >> 8))) * scale;
            float t2 = static_cast<float>(static_cast<int>(static_cast<int8_t>(w >> 16))) * scale;
            float t3 = static_cast<float>(static_cast<int>(static_cast<int8_t>(w >> 24))) * scale;
            v = make_float4(t0 + offset, t1 + offset, t2 + offset, t3 + offset);  // pq.go:205-215's two rounded operations
        } else {
            v = *reinterpret_cast<const float4 *>(cent_all + (static_cast<int64_t>(sub) * 256 + c) * 8 + 4 * h);
        }
        a[cb] = make_float4(-2.0f * v.x, -2.0f * v.y, -2.0f * v.z, -2.0f * v.w);
        float part = v.x * v.x;
        part = __builtin_fmaf(v.y, v.y, part);
        part = __builtin_fmaf(v.z, v.z, part);
        part = __builtin_fmaf(v.w, v.w, part);
        const float cnv = part + __shfl_xor(part, 32);
        acn[cb] = h == 0 ? cnv : 0.0f;
        cmax = fmaxf(cmax, cnv == cnv ? cnv : INFINITY);
    }
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) cmax = fmaxf(cmax, __shfl_xor(cmax, off));
    if (lane == 0) s_cmax[wave] = cmax;   // (every wave holds all 256 centroids: the four values are equal)
    __syncthreads();
    const float C = s_cmax[0], sqc = sqrtf(C);
    const float ones = h == 0 ? 1.0f : 0.0f;
    const float u = 5.9604645e-8f;
    const int64_t row0 = row_first + static_cast<int64_t>(tile_x) * kNomRowsPerBlock + wave * kNomRowsPerWave;
    const int64_t stride = ENC ? dim : 8;
    const float *xbase = ENC ? rows + static_cast<int64_t>(sub) * 8 : rows + static_cast<int64_t>(sub) * n * 8;
    auto load_rows = [&](int pb) {  // this lane's half of the sub-vector of row pb * 32 + (lane & 31) (past the end: the last row)
        const int64_t p = row0 + pb * 32 + c_in;
        const int64_t pc = p < n ? p : n - 1;
        return *reinterpret_cast<const float4 *>(xbase + pc * stride + 4 * h);
    };
    float4 xnext = load_rows(0);
    for (int pb = 0; pb < kNomRowsPerWave / 32; pb++) {
        const int64_t p = row0 + pb * 32 + c_in;
        if (row0 + pb * 32 >= n) break;  // wave-uniform
        const float4 xb = xnext;
        xnext = load_rows(pb + 1 < kNomRowsPerWave / 32 ? pb + 1 : pb);  // in flight under this block's 32 MFMAs
        float xp = xb.x * xb.x;
        xp = __builtin_fmaf(xb.y, xb.y, xp);
        xp = __builtin_fmaf(xb.z, xb.z, xp);
        xp = __builtin_fmaf(xb.w, xb.w, xp);
        const float X = xp + __shfl_xor(xp, 32);
        float m1 = INFINITY, m2 = INFINITY;
        int ir = 0, bcb = 0;
        // two accumulator sets: the four MFMAs of block cb + 1 are issued before the vector ALU reads block cb (one set
        // made every block wait for the previous one's 64 reads: 4400 cycles per 32 rows where the matrix unit needs 2048)
        auto tile = [&](int cb) {
            pq_f32x16 acc = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#if defined(VG_NOM_PROBE) && VG_NOM_PROBE == 2  // stage probe (tools/build_variant.sh): no matrix instructions
            for (int r = 0; r < 16; r++) acc[r] = a[cb].x * xb.y + static_cast<float>(r);
            return acc;
#endif
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(acn[cb], ones, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cb].x, xb.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cb].y, xb.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cb].z, xb.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cb].w, xb.w, acc, 0, 0, 0);
            return acc;
        };
        auto scan = [&](const pq_f32x16 &acc, int cb) {
#if defined(VG_NOM_PROBE) && VG_NOM_PROBE == 1  // stage probe: the matrix results are not scanned
            m1 = fminf(m1, acc[cb & 15]);
            return;
#endif
            const float before = m1;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const float sc = acc[r];
                m2 = __builtin_amdgcn_fmed3f(m1, sc, m2);  // m1 <= m2: the middle one is the new second smallest
                const bool lt = sc < m1;                   // one compare feeds both selects (fminf would add a canonicalising
                ir = lt ? r : ir;                          // v_max per value: the scores come straight from the matrix unit)
                m1 = lt ? sc : m1;
            }
            bcb = m1 < before ? cb : bcb;
        };
        pq_f32x16 acc_a = tile(0), acc_b;
#pragma unroll
        for (int cb = 0; cb < 8; cb += 2) {
            acc_b = tile(cb + 1);
            __builtin_amdgcn_sched_barrier(0);
            scan(acc_a, cb);
            __builtin_amdgcn_sched_barrier(0);
            if (cb + 2 < 8) acc_a = tile(cb + 2);
            __builtin_amdgcn_sched_barrier(0);
            scan(acc_b, cb + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        int idx = bcb * 32 + (ir & 3) + 8 * (ir >> 2) + 4 * h;
        const float om1 = __shfl_xor(m1, 32), om2 = __shfl_xor(m2, 32);
        const int oidx = __shfl_xor(idx, 32);
        m2 = fminf(fmaxf(m1, om1), fminf(m2, om2));
        idx = om1 < m1 ? oidx : idx;
        m1 = fminf(m1, om1);
        if (h == 0 && p < n) {
            const float sx = sqrtf(X);
            const float cross = 2.0f * sx * sqc + C, dmax = (sx + sqc) * (sx + sqc);
            const float margin = 1.05f * (64.0f * u * cross + 24.0f * u * dmax) + 1e-30f + extra_margin;
            // every comparison is false on NaN: non-finite rows, centroids or scores go on the list
            if (X + C < 1e30f && m2 - m1 > margin) {
                if (ENC) {
                    codes[p * m + sub] = static_cast<uint8_t>(idx);
                } else {
                    int32_t *dst = assign_all + static_cast<int64_t>(sub) * n + p;
                    if (*dst != idx) {
                        *dst = idx;
                        changed[sub] = 1;
                    }
                }
            } else {
                const int at = atomicAdd(list_count, 1);
                list[at] = PqNomList{static_cast<int32_t>(p), sub};
            }
        }
    }
}

// ---- the same nomination on the bfloat16 matrix instruction, the scores scanned as packed keys ---------------------------
// v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 rate: pq_nominate_kernel spends 4.3 of its 6.3 ms per 1M x 768 inside
// five fp32 MFMAs per tile, and the other 2 ms in 4 vector instructions per score.  Here
//  * a tile is TWO v_mfma_f32_32x32x16_bf16: with x = x_hi + x_lo and w = -2 v = w_hi + w_lo (hi = the value rounded to
//    bfloat16, lo = the remainder rounded to bfloat16) the 32 k slots hold  w_hi.x_hi (8) + w_lo.x_hi (8) + w_hi.x_lo (8),
//    three slots |v|^2 (a 3-way bfloat16 split, exact) against ones, three slots ones against |x|^2 (1 + 2^-10) (the
//    same): the accumulator is D~ = |x|^2 (1 + 2^-10) + |v|^2 - 2 x.v, POSITIVE (the 2^-10 |x|^2 outweighs every
//    rounding below), so float order = unsigned order of the bit patterns;
//  * a score's low 8 bits are replaced by its centroid's index inside the lane (v_and_or_b32) and the running smallest /
//    second smallest are v_min_u32 / v_med3_u32 on these keys: 3 instructions per score, the index riding along.
// Bound on |D~ - D - 2^-10 X| per score: the split drops w_lo.x_lo and the second remainders, <= 3 * 2^-16 |w| |x| = 3 * 2^-16 * 2 sqrt(X C);
// bfloat16 products are exact in fp32 and each of the 30 additions is taken to round (any order): 32 u (2 sqrt(X C) + C +
// 1.001 X); the key drops 8 bits: 2^-16 D~.  Everything is inside
//     margin = 2 [ (3.1 * 2^-16 + 32 u) (2 sqrt(X C) + C) + (2^-16 * 1.01 + 32 u) * 1.002 (sqrt X + sqrt C)^2 ] + 2 * 12 u (..)^2
// (the last term: the reference's own roundings, as before).
typedef __bf16 pq_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 pq_bf16x2 __attribute__((ext_vector_type(2)));
typedef float pq_f32x2 __attribute__((ext_vector_type(2)));

// (hi, lo) bfloat16 halves of two floats: hi = round-to-nearest-even, lo = the same of the exact remainder
__device__ __forceinline__ void pq_split2(float a, float b, uint32_t &hi, uint32_t &lo)
{
    const pq_bf16x2 h = __builtin_convertvector(pq_f32x2{a, b}, pq_bf16x2);
    hi = __builtin_bit_cast(uint32_t, h);
    const float ra = a - __uint_as_float(hi << 16), rb = b - __uint_as_float(hi & 0xFFFF0000u);
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(pq_f32x2{ra, rb}, pq_bf16x2));
}
// x = p0 + p1 + p2 exactly (three bfloat16: 24 significant bits), as the low halves of three words
__device__ __forceinline__ void pq_split3(float x, uint32_t &p0, uint32_t &p1, uint32_t &p2)
{
    const pq_bf16x2 a = __builtin_convertvector(pq_f32x2{x, 0.0f}, pq_bf16x2);
    p0 = __builtin_bit_cast(uint32_t, a) & 0xFFFFu;
    const float r1 = x - __uint_as_float(p0 << 16);
    const pq_bf16x2 b = __builtin_convertvector(pq_f32x2{r1, 0.0f}, pq_bf16x2);
    p1 = __builtin_bit_cast(uint32_t, b) & 0xFFFFu;
    const float r2 = r1 - __uint_as_float(p1 << 16);
    const pq_bf16x2 c = __builtin_convertvector(pq_f32x2{r2, 0.0f}, pq_bf16x2);
    p2 = __builtin_bit_cast(uint32_t, c) & 0xFFFFu;
}

// v_med3_u32 (no builtin; not volatile: the compiler may schedule it)
__device__ __forceinline__ uint32_t pq_umed3(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

__device__ __forceinline__ uint32_t pq_umin3(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// (134 registers: three waves per SIMD; held to four — 128 registers, 11 spilled — it is slower, 0.276 vs 0.242 ms per
// 65 536 x 96 pairs: the kernel is not short of waves)
#ifndef VG_NOM_WAVES_ATTR
#define VG_NOM_WAVES_ATTR
#endif
template <bool ENC>
__global__ __launch_bounds__(256) VG_NOM_WAVES_ATTR void pq_nominate_bf16_kernel(const float *__restrict__ rows, int64_t n, int dim, int m,
                                                               const int8_t *__restrict__ codebooks,
                                                               const float *__restrict__ scales, const float *__restrict__ offsets,
                                                               const float *__restrict__ cent_all, uint8_t *__restrict__ codes,
                                                               int32_t *__restrict__ assign_all, int *__restrict__ changed,
                                                               const int *__restrict__ done, PqNomList *__restrict__ list,
                                                               int *__restrict__ list_count, int64_t row_first, float extra_margin)
{
    __shared__ float s_cmax[4];
    __shared__ uint16_t s_dec[4][kNomRowsPerWave];  // per wave and row: the nominated centroid | 0x100 = undecided (listed)
    int sub, tile_x;
    pq_nom_block(m, sub, tile_x);
    if (!ENC && done[sub]) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, c_in = lane & 31;
    // A operands of the 8 centroid blocks (uint4 = 8 bfloat16 = this lane's 8 k slots of one instruction)
    uint4 a1[8], a2[8];
    float cmax = 0.0f;
#pragma unroll
    for (int cb = 0; cb < 8; cb++) {
        const int c = cb * 32 + c_in;
        float v[8];
        if (ENC) {
            const uint2 w = reinterpret_cast<const uint2 *>(codebooks)[static_cast<int64_t>(sub) * 256 + c];
            const float scale = scales[sub], offset = offsets[sub];
#pragma unroll
            for (int t = 0; t < 8; t++) {
                const uint32_t ww = t < 4 ? w.x : w.y;
                const float f = static_cast<float>(static_cast<int>(static_cast<int8_t>(ww >> (8 * (t & 3))))) * scale;
                v[t] = f + offset;  // pq.go:205-215's two rounded operations
            }
        } else {
            const float4 lo4 = *reinterpret_cast<const float4 *>(cent_all + (static_cast<int64_t>(sub) * 256 + c) * 8);
            const float4 hi4 = *reinterpret_cast<const float4 *>(cent_all + (static_cast<int64_t>(sub) * 256 + c) * 8 + 4);
            v[0] = lo4.x; v[1] = lo4.y; v[2] = lo4.z; v[3] = lo4.w;
            v[4] = hi4.x; v[5] = hi4.y; v[6] = hi4.z; v[7] = hi4.w;
        }
        float cn = 0.0f;
#pragma unroll
        for (int t = 0; t < 8; t++) cn = __builtin_fmaf(v[t], v[t], cn);
        uint32_t wh[4], wl[4];
#pragma unroll
        for (int t = 0; t < 4; t++) pq_split2(-2.0f * v[2 * t], -2.0f * v[2 * t + 1], wh[t], wl[t]);
        uint32_t c0, c1, c2;
        pq_split3(cn, c0, c1, c2);
        const uint32_t one = 0x3F80u;  // bfloat16 1.0
        // instruction 1: k 0-7 (h = 0) w_hi, k 8-15 (h = 1) w_lo; instruction 2: k 0-7 w_hi, k 8-15 (|v|^2 x 3, 1, 1, 1, 0, 0)
        a1[cb] = h == 0 ? make_uint4(wh[0], wh[1], wh[2], wh[3]) : make_uint4(wl[0], wl[1], wl[2], wl[3]);
        a2[cb] = h == 0 ? make_uint4(wh[0], wh[1], wh[2], wh[3]) : make_uint4(c0 | (c1 << 16), c2 | (one << 16), one | (one << 16), 0u);
        cmax = fmaxf(cmax, cn == cn ? cn : INFINITY);
    }
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) cmax = fmaxf(cmax, __shfl_xor(cmax, off));
    if (lane == 0) s_cmax[wave] = cmax;
    __syncthreads();
    const float C = s_cmax[0], sqc = sqrtf(C);
    const float u = 5.9604645e-8f, t16 = 1.52587890625e-5f;
    const int64_t row0 = row_first + static_cast<int64_t>(tile_x) * kNomRowsPerBlock + wave * kNomRowsPerWave;
    const int64_t stride = ENC ? dim : 8;
    const float *xbase = ENC ? rows + static_cast<int64_t>(sub) * 8 : rows + static_cast<int64_t>(sub) * n * 8;
    // the mask lives in a VGPR the compiler cannot see through: v_and_or_b32 then takes (score, mask, index) with the index
    // as its one scalar operand — a literal mask leaves no room for a literal index and costs a v_and + v_or per score
    uint32_t keep;
    asm volatile("v_mov_b32 %0, 0xffffff00" : "=v"(keep));
    // this wave's rows [row0, row0 + rows_mine); a lane's row of block pb: pb * 32 + (lane & 31), past the end the last one
    // (32-bit offsets from the wave's first row: at most 1024 rows of `stride` floats)
    const int64_t left = n - row0;
    const int rows_mine = left <= 0 ? 0 : (left < kNomRowsPerWave ? static_cast<int>(left) : kNomRowsPerWave);
    constexpr int kNb = kNomRowsPerWave / 32;
    const float *const wbase = xbase + row0 * stride;
    const uint32_t ustride = static_cast<uint32_t>(stride);
    auto load_rows = [&](int pb, float4 &lo4, float4 &hi4) {
        int rel = pb * 32 + c_in;
        rel = rel < rows_mine ? rel : rows_mine - 1;
        const float *src = wbase + static_cast<uint32_t>(rel) * ustride;
        lo4 = *reinterpret_cast<const float4 *>(src);
        hi4 = *reinterpret_cast<const float4 *>(src + 4);
    };
    float4 nlo = make_float4(0.0f, 0.0f, 0.0f, 0.0f), nhi = nlo;
    if (rows_mine > 0) load_rows(0, nlo, nhi);
    for (int pb = 0; pb < kNb; pb++) {
        const int64_t p = row0 + pb * 32 + c_in;
        if (pb * 32 >= rows_mine) break;  // wave-uniform
        const float x[8] = {nlo.x, nlo.y, nlo.z, nlo.w, nhi.x, nhi.y, nhi.z, nhi.w};
#if !(defined(VG_NOM_PROBE) && VG_NOM_PROBE == 4)  // stage probe 4: the first block's rows again and again (no loads in the loop)
        load_rows(pb + 1 < kNb ? pb + 1 : pb, nlo, nhi);  // in flight under this block's work (nothing else of the loop touches memory)
#endif
        float X = 0.0f;
#pragma unroll
        for (int t = 0; t < 8; t++) X = __builtin_fmaf(x[t], x[t], X);
        uint32_t xh[4], xl[4];
#pragma unroll
        for (int t = 0; t < 4; t++) pq_split2(x[2 * t], x[2 * t + 1], xh[t], xl[t]);
        uint32_t x0, x1, x2;
        pq_split3(X * 1.0009765625f, x0, x1, x2);
        const uint32_t one = 0x3F80u;
        const uint4 b1 = make_uint4(xh[0], xh[1], xh[2], xh[3]);
        const uint4 b2 = h == 0 ? make_uint4(xl[0], xl[1], xl[2], xl[3]) : make_uint4(one | (one << 16), one | (x0 << 16), x1 | (x2 << 16), 0u);
        uint32_t m1 = 0xFFFFFFFFu, m2 = 0xFFFFFFFFu;
        auto tile = [&](int cb) {
            pq_f32x16 acc = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#if defined(VG_NOM_PROBE) && VG_NOM_PROBE == 5  // stage probe 5: no matrix instructions (the scan reads operand words)
            for (int r = 0; r < 16; r++) acc[r] = __uint_as_float((&a1[cb].x)[r & 3] ^ (&b1.x)[(r >> 2) & 3]);
            return acc;
#endif
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(pq_bf16x8, a1[cb]), __builtin_bit_cast(pq_bf16x8, b1), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(pq_bf16x8, a2[cb]), __builtin_bit_cast(pq_bf16x8, b2), acc, 0, 0, 0);
            return acc;
        };
        auto scan = [&](const pq_f32x16 &acc, int cb) {
#if defined(VG_NOM_PROBE) && VG_NOM_PROBE == 1  // stage probe: the matrix results are not scanned
            m1 = m1 < __float_as_uint(acc[cb]) ? m1 : __float_as_uint(acc[cb]);
            return;
#endif
            // Four keys at a time.  With m1 <= m2 the two smallest of {m1, m2, a, b} are min3(m1, a, b) and
            // min(m2, med3(m1, a, b)) (the second smallest of {m1, a, b}; m2 can only replace it from above), so two pairs cost
            // 2 med3 + 2 min3 + one min3 that folds both medians into m2: 5 instructions per 4 keys where the one-key-at-a-time
            // form (med3 + min per key) took 8 — and the chain through m1 is half as long.  (tools/ubench/valu_rate.hip: these
            // integer / 3-operand forms all issue at the same rate, ~3 cycles per wave-instruction and SIMD at three waves.)
#pragma unroll
            for (int r = 0; r < 16; r += 4) {
                uint32_t key[4];
#pragma unroll
                for (int e = 0; e < 4; e++)  // the centroid's index inside this lane's 128 (the h bit is added after the scan)
                    key[e] = (__float_as_uint(acc[r + e]) & keep) | static_cast<uint32_t>(cb * 32 + ((r + e) & 3) + 8 * ((r + e) >> 2));
                const uint32_t t1 = pq_umed3(m1, key[0], key[1]);
                m1 = pq_umin3(m1, key[0], key[1]);
                const uint32_t t2 = pq_umed3(m1, key[2], key[3]);
                m1 = pq_umin3(m1, key[2], key[3]);
                m2 = pq_umin3(m2, t1, t2);
            }
        };
#if defined(VG_NOM_PROBE) && VG_NOM_PROBE == 3  // stage probe: loads and splits only
        m1 = b1.x ^ b2.y ^ a1[pb & 7].x;
        m2 = m1 + 512;
#else
        pq_f32x16 acc_a = tile(0), acc_b;
#pragma unroll
        for (int cb = 0; cb < 8; cb += 2) {
            acc_b = tile(cb + 1);
            scan(acc_a, cb);
            if (cb + 2 < 8) acc_a = tile(cb + 2);
            scan(acc_b, cb + 1);
        }
#endif
        m1 |= static_cast<uint32_t>(4 * h);
        m2 |= static_cast<uint32_t>(4 * h);
        const uint32_t o1 = __shfl_xor(m1, 32), o2 = __shfl_xor(m2, 32);
        const uint32_t lo1 = m1 < o1 ? m1 : o1, hi1 = m1 < o1 ? o1 : m1, lo2 = m2 < o2 ? m2 : o2;
        m2 = hi1 < lo2 ? hi1 : lo2;
        m1 = lo1;
        if (h == 0 && p < n) {
            const uint32_t idx = m1 & 0xFFu;
            const float d1 = __uint_as_float(m1 & keep), d2 = __uint_as_float(m2 & keep);
            const float sx = __builtin_amdgcn_sqrtf(X);  // v_sqrt_f32, 1 ulp (the IEEE sqrtf is ~15 instructions): inside the 1.05
            const float cross = 2.0f * sx * sqc + C, dmax = (sx + sqc) * (sx + sqc);
            const float margin = 1.05f * (2.0f * ((3.1f * t16 + 32.0f * u) * cross + (1.01f * t16 + 32.0f * u) * 1.002f * dmax) +
                                          24.0f * u * dmax) + 1e-30f + extra_margin;
            // every comparison is false on NaN: non-finite rows, centroids or scores go on the list (a NaN score's key has
            // every exponent bit set: it can only be the LARGEST key, so a finite smallest key is a real smallest score)
            bool listed = !(X + C < 1e30f && d2 - d1 > margin);
#if defined(VG_NOM_PROBE)  // stage probes compute garbage: nothing goes to pq_fix_kernel, whose time would drown the stage's
            listed = false;
#endif
            // The decision goes to LDS, not to memory: a byte store per row here (and, for Train, a load of the row's previous
            // assignment first) is a vector-memory operation in front of the next block's `s_waitcnt vmcnt(0)` for its rows —
            // every block waited out a store's (or a dependent load's) round trip, and one block in eight a returning atomic for
            // its listed rows: r05's kernel spent half its time there (stage probes: 4.4 ms with, 2.2 ms without the loop's
            // memory traffic).  The wave's 1024 decisions leave together after the loop.
            s_dec[wave][pb * 32 + c_in] = static_cast<uint16_t>(idx | (listed ? 0x100u : 0u));
        }
    }
    // flush: this wave's rows [row0, row0 + rows_mine): codes / assignments of the decided rows, one atomic for the listed ones
    int total = 0;
    for (int i0 = 0; i0 < rows_mine; i0 += 64) {
        const int i = i0 + lane;
        total += __popcll(__ballot(i < rows_mine && (s_dec[wave][i] & 0x100u)));
    }
    int at = 0;
    if (total) {
        if (lane == 0) at = atomicAdd(list_count, total);
        at = __shfl(at, 0);
    }
    bool any_changed = false;
    for (int i0 = 0; i0 < rows_mine; i0 += 64) {
        const int i = i0 + lane;
        const bool in = i < rows_mine;
        const uint32_t v = in ? s_dec[wave][i] : 0u;
        const bool listed = in && (v & 0x100u);
        const int64_t pr = row0 + i;
        if (in && !listed) {
            if (ENC) {
                codes[pr * m + sub] = static_cast<uint8_t>(v);
            } else {
                int32_t *dst = assign_all + static_cast<int64_t>(sub) * n + pr;
                if (*dst != static_cast<int32_t>(v)) {
                    *dst = static_cast<int32_t>(v);
                    any_changed = true;
                }
            }
        }
        const uint64_t lm = __ballot(listed);
        if (listed) list[at + __popcll(lm & ((1ull << lane) - 1ull))] = PqNomList{static_cast<int32_t>(pr), sub};
        at += __popcll(lm);
    }
    if (!ENC && __ballot(any_changed) && lane == 0) changed[sub] = 1;
}

// the listed pairs, decided as the reference writes the loop
template <bool ENC>
__global__ __launch_bounds__(256) void pq_fix_kernel(const float *__restrict__ rows, int64_t n, int dim, int m,
                                                     const int8_t *__restrict__ codebooks, const float *__restrict__ scales,
                                                     const float *__restrict__ offsets, const float *__restrict__ cent_all,
                                                     uint8_t *__restrict__ codes, int32_t *__restrict__ assign_all,
                                                     int *__restrict__ changed, const PqNomList *__restrict__ list,
                                                     const int *__restrict__ list_count)
{
    const int total = *list_count;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int64_t p = list[e].row;
        const int sub = list[e].sub;
        if (ENC) {
            const float *x = rows + p * dim + static_cast<int64_t>(sub) * 8;
            const float scale = scales[sub], offset = offsets[sub];
            float q[8];
            for (int t = 0; t < 8; t++) q[t] = x[t];
            int best = 0;
            float bd = 0.0f;
            for (int c = 0; c < 256; c++) {  // FindNearestCentroidInt8 (kernels.go:376-396)
                const int8_t *cb = codebooks + (static_cast<int64_t>(sub) * 256 + c) * 8;
                float sum = 0.0f;
                for (int t = 0; t < 8; t++) {
                    float v = static_cast<float>(cb[t]) * scale;
                    v = v + offset;
                    const float d = q[t] - v;
                    const float dd = d * d;
                    sum = sum + dd;
                }
                if (c == 0 || sum < bd) {
                    bd = sum;
                    best = c;
                }
            }
            codes[p * m + sub] = static_cast<uint8_t>(best);
        } else {
            const float *x = rows + (static_cast<int64_t>(sub) * n + p) * 8;
            const float *cent = cent_all + static_cast<int64_t>(sub) * 256 * 8;
            float best = 3.40282346638528859811704183484516925440e+38f;
            int bi = 0;
            for (int c = 0; c < 256; c++) {  // findNearestCentroid (pq.go:416-433)
                const float d = l2_train<8>(x, cent + c * 8, 8);
                if (d < best) {
                    best = d;
                    bi = c;
                }
            }
            int32_t *dst = assign_all + static_cast<int64_t>(sub) * n + p;
            if (*dst != bi) {
                *dst = bi;
                changed[sub] = 1;
            }
        }
    }
}

}  // namespace vg

VG_API int32_t vg_pq_train_subset(vg_pq *pq, const float *vectors, int64_t n, int32_t iters, uint64_t seed,
                                  int32_t sub_begin, int32_t sub_count, void *stream)
{
    VG_CHECK(pq, VG_ERR_INVALID_ARG, "vg_pq_train: NULL quantizer");
    VG_CHECK(n > 0 && vectors, VG_ERR_INVALID_ARG, "no vectors provided for training");  // pq.go:69-71
    VG_CHECK(iters >= 0, VG_ERR_INVALID_ARG, "vg_pq_train: iters < 0");
    VG_CHECK(n <= INT32_MAX, VG_ERR_UNSUPPORTED, "vg_pq_train: more than 2^31-1 training vectors");
    VG_CHECK(sub_begin >= 0 && sub_count >= 0 && sub_begin + sub_count <= pq->m, VG_ERR_INVALID_ARG,
             "vg_pq_train_subset: sub-quantizer range [%d, %d) outside [0, %d)", sub_begin, sub_begin + sub_count,
             pq->m);
    VG_CHECK(pq->subdim <= 256, VG_ERR_UNSUPPORTED, "vg_pq_train: sub-vector dim %d > 256", pq->subdim);
    if (sub_count == 0) return VG_OK;
    VG_HIP(hipSetDevice(pq->ctx->device));
    hipStream_t st = vg::pick_stream(pq->ctx, stream);
    const int m = sub_count, k = pq->k, sd = pq->subdim, dim = pq->dim;
    vg::DevIn<float> v;
    VG_TRY(v.init(vectors, static_cast<size_t>(n) * dim, st));
    vg::DevTmp<float> mind, cent, slabs, pref;
    vg::DevTmp<int32_t> assign, order, seg;
    vg::DevTmp<int> flags;
    const size_t nblk = static_cast<size_t>((n + vg::kPPBlock - 1) / vg::kPPBlock);
    VG_TRY(slabs.init(static_cast<size_t>(m) * n * sd, st));
    VG_TRY(mind.init(static_cast<size_t>(m) * n, st));
    VG_TRY(pref.init(static_cast<size_t>(m) * nblk, st));
    VG_TRY(cent.init(static_cast<size_t>(m) * k * sd, st));
    VG_TRY(assign.init(static_cast<size_t>(m) * n, st));
    VG_TRY(order.init(static_cast<size_t>(m) * n, st));
    VG_TRY(seg.init(static_cast<size_t>(m) * 2 * k, st));
    VG_TRY(flags.init(static_cast<size_t>(2 * m), st));
    VG_HIP(hipMemsetAsync(assign.ptr, 0, sizeof(int32_t) * static_cast<size_t>(m) * n, st));
    VG_HIP(hipMemsetAsync(flags.ptr, 0, sizeof(int) * 2 * m, st));
    int *changed = flags.ptr, *done = flags.ptr + m;

    VG_LAUNCH(vg::pq_slab_kernel, dim3(static_cast<unsigned>((n * sd + 255) / 256), m), dim3(256), 0, st, v.ptr, n, dim,
              sd, sub_begin, slabs.ptr);
    const size_t lds = static_cast<size_t>(k) * sd * sizeof(float);
    // one sub-quantizer's centroids live in LDS during assignment: up to 152 KiB of the CU's 160
    VG_CHECK(lds <= 152 * 1024, VG_ERR_UNSUPPORTED, "vg_pq_train: codebook of one sub-quantizer exceeds 152 KiB");
    if (lds > 48 * 1024)
        VG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(vg::pq_assign_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    const unsigned gx = static_cast<unsigned>((n + 255) / 256);
    const unsigned gx_vec = static_cast<unsigned>((n + 256 * vg::kAssignRows - 1) / (256 * vg::kAssignRows));
    // Lloyd assignment on the matrix cores (pq_nominate_kernel) when the shape is the one it is written for
    const bool nominate = sd == 8 && k == 256 && !vg::hook(vg::kHookPqNoMfma);
    const unsigned nom_gx = static_cast<unsigned>((n + vg::kNomRowsPerBlock - 1) / vg::kNomRowsPerBlock);
    const float nom_extra = vg::hook(vg::kHookPqListAll) ? INFINITY : 0.0f;
    auto nom_kern = vg::hook(vg::kHookPqFp32Mfma) ? vg::pq_nominate_kernel<false> : vg::pq_nominate_bf16_kernel<false>;
    vg::DevTmp<vg::PqNomList> nom_list;
    vg::DevTmp<int> nom_count;
    if (nominate) {
        VG_TRY(nom_list.init(static_cast<size_t>(m) * n, st));
        VG_TRY(nom_count.init(1, st));
    }
    const unsigned ux = static_cast<unsigned>((k * sd + 255) / 256);
#define VG_PQ_TRAIN_SD(SD)                                                                                       \
    do {                                                                                                         \
        {                                                                                                        \
            vg::ProfScope prof(pq->ctx, "pq_kmeanspp", st);                                                      \
            VG_LAUNCH(vg::pq_kmeanspp_kernel<SD>, dim3(m), dim3(vg::kPPThreads), 0, st, slabs.ptr, n, sd, k, seed, \
                      mind.ptr, pref.ptr, cent.ptr, sub_begin);                                                  \
        }                                                                                                        \
        for (int it = 0; it < iters; it++) {                                                                     \
            {                                                                                                    \
                vg::ProfScope prof(pq->ctx, "pq_assign", st);                                                    \
                if (nominate) {                                                                                  \
                    VG_HIP(hipMemsetAsync(nom_count.ptr, 0, sizeof(int), st));                                   \
                    VG_LAUNCH(nom_kern, dim3(nom_gx * m), dim3(256), 0, st, slabs.ptr, n, dim, m, nullptr, \
                              nullptr, nullptr, cent.ptr, nullptr, assign.ptr, changed, done, nom_list.ptr, nom_count.ptr,  \
                              int64_t(0), nom_extra);                                                            \
                    VG_LAUNCH(vg::pq_fix_kernel<false>, dim3(2048), dim3(64), 0, st, slabs.ptr, n, dim, m, nullptr, nullptr, \
                              nullptr, cent.ptr, nullptr, assign.ptr, changed, nom_list.ptr, nom_count.ptr);     \
                } else if (SD)                                                                                   \
                    VG_LAUNCH(vg::pq_assign_vec_kernel<(SD ? SD : 4)>, dim3(gx_vec, m), dim3(256), lds, st, slabs.ptr, n, k, \
                              cent.ptr, assign.ptr, changed, done);                                              \
                else                                                                                             \
                    VG_LAUNCH(vg::pq_assign_kernel, dim3(gx, m), dim3(256), lds, st, slabs.ptr, n, sd, k, cent.ptr, \
                              assign.ptr, changed, done);                                                        \
            }                                                                                                    \
            vg::ProfScope prof(pq->ctx, "pq_update", st);                                                        \
            VG_LAUNCH(vg::pq_bucket_kernel, dim3(m), dim3(vg::kBucketThreads), 0, st, n, k, assign.ptr,          \
                      order.ptr, seg.ptr, changed, done);                                                        \
            VG_LAUNCH(vg::pq_update_kernel, dim3(ux, m), dim3(256), 0, st, slabs.ptr, n, sd, k, it, seed,        \
                      order.ptr, seg.ptr, cent.ptr, changed, done, sub_begin);                                   \
            VG_LAUNCH(vg::pq_iter_end_kernel, dim3((m + 63) / 64), dim3(64), 0, st, m, changed, done);           \
        }                                                                                                        \
    } while (0)
    if (sd == 8)
        VG_PQ_TRAIN_SD(8);
    else if (sd == 4)
        VG_PQ_TRAIN_SD(4);
    else if (sd == 16)
        VG_PQ_TRAIN_SD(16);
    else
        VG_PQ_TRAIN_SD(0);
#undef VG_PQ_TRAIN_SD
    VG_LAUNCH(vg::pq_quantize_kernel, dim3(m), dim3(256), 0, st, cent.ptr, k, sd,
                       pq->d_codebooks, pq->d_scales, pq->d_offsets, sub_begin);
    VG_HIP(hipStreamSynchronize(st));
    if (sub_begin == 0 && sub_count == pq->m) pq->trained = true;
    return VG_OK;
}

VG_API int32_t vg_pq_train(vg_pq *pq, const float *vectors, int64_t n, int32_t iters, uint64_t seed,
                           void *stream)
{
    VG_CHECK(pq, VG_ERR_INVALID_ARG, "vg_pq_train: NULL quantizer");
    return vg_pq_train_subset(pq, vectors, n, iters, seed, 0, pq->m, stream);
}

VG_API int32_t vg_pq_encode(vg_pq *pq, const float *vectors, int64_t n, uint8_t *codes, void *stream)
{
    VG_CHECK(pq, VG_ERR_INVALID_ARG, "vg_pq_encode: NULL quantizer");
    VG_CHECK(pq->trained, VG_ERR_NOT_TRAINED, "ProductQuantizer not trained");
    VG_CHECK(n >= 0, VG_ERR_INVALID_ARG, "vg_pq_encode: n < 0");
    if (n == 0) return VG_OK;
    VG_CHECK(vectors && codes, VG_ERR_INVALID_ARG, "vg_pq_encode: NULL buffer");
    VG_HIP(hipSetDevice(pq->ctx->device));
    hipStream_t st = vg::pick_stream(pq->ctx, stream);
    vg::DevIn<float> v;
    vg::DevOut<uint8_t> c;
    VG_TRY(v.init(vectors, static_cast<size_t>(n) * pq->dim, st));
    VG_TRY(c.init(codes, static_cast<size_t>(n) * pq->m, st));
    const size_t lds = static_cast<size_t>(pq->k) * pq->subdim * sizeof(float);
    VG_CHECK(lds <= 152 * 1024, VG_ERR_UNSUPPORTED, "vg_pq_encode: codebook of one sub-quantizer exceeds 152 KiB");
    if (lds > 48 * 1024)
        VG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(vg::pq_encode_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    // grid.y = m <= 65535 is guaranteed by dim limits; grid.x up to 2^31
    const bool vec_ok = pq->dim % 4 == 0 && (reinterpret_cast<uintptr_t>(v.ptr) & 15) == 0;
    const unsigned gx_vec = static_cast<unsigned>((n + 256 * vg::kEncRows - 1) / (256 * vg::kEncRows));
    {
    vg::ProfScope prof(pq->ctx, "pq_encode", st);
    if (vec_ok && pq->subdim == 8 && pq->k == 256 && n <= INT32_MAX && !vg::hook(vg::kHookPqNoMfma)) {
        // nearest centroids on the matrix cores (pq_nominate_kernel), the pairs they cannot decide by the reference's loop;
        // in chunks of rows so that the list of a chunk (worst case: every pair) stays below 256 MB
        const int64_t chunk = std::max<int64_t>(vg::kNomRowsPerBlock, ((int64_t(256) << 20) / (static_cast<int64_t>(pq->m) * 8)) /
                                                                           vg::kNomRowsPerBlock * vg::kNomRowsPerBlock);
        vg::DevTmp<vg::PqNomList> nom_list;
        vg::DevTmp<int> nom_count;
        VG_TRY(nom_list.init(static_cast<size_t>(std::min(chunk, n)) * pq->m, st));
        VG_TRY(nom_count.init(1, st));
        const float extra = vg::hook(vg::kHookPqListAll) ? INFINITY : 0.0f;
        for (int64_t r0 = 0; r0 < n; r0 += chunk) {
            const int64_t r1 = std::min(n, r0 + chunk);
            VG_HIP(hipMemsetAsync(nom_count.ptr, 0, sizeof(int), st));
            const unsigned gxn = static_cast<unsigned>((r1 - r0 + vg::kNomRowsPerBlock - 1) / vg::kNomRowsPerBlock);
            auto nom_kern = vg::hook(vg::kHookPqFp32Mfma) ? vg::pq_nominate_kernel<true> : vg::pq_nominate_bf16_kernel<true>;
            VG_LAUNCH(nom_kern, dim3(gxn * pq->m), dim3(256), 0, st, v.ptr, r1, pq->dim, pq->m, pq->d_codebooks,
                      pq->d_scales, pq->d_offsets, nullptr, c.ptr, nullptr, nullptr, nullptr, nom_list.ptr, nom_count.ptr, r0,
                      extra);
            VG_LAUNCH(vg::pq_fix_kernel<true>, dim3(4096), dim3(64), 0, st, v.ptr, r1, pq->dim, pq->m, pq->d_codebooks,
                      pq->d_scales, pq->d_offsets, nullptr, c.ptr, nullptr, nullptr, nom_list.ptr, nom_count.ptr);
        }
    } else if (vec_ok && pq->subdim == 8)
        VG_LAUNCH(vg::pq_encode_vec_kernel<8>, dim3(gx_vec, pq->m), dim3(256), lds, st, v.ptr, n, pq->dim, pq->m, pq->k,
                  pq->d_codebooks, pq->d_scales, pq->d_offsets, c.ptr);
    else if (vec_ok && pq->subdim == 4)
        VG_LAUNCH(vg::pq_encode_vec_kernel<4>, dim3(gx_vec, pq->m), dim3(256), lds, st, v.ptr, n, pq->dim, pq->m, pq->k,
                  pq->d_codebooks, pq->d_scales, pq->d_offsets, c.ptr);
    else if (vec_ok && pq->subdim == 16)
        VG_LAUNCH(vg::pq_encode_vec_kernel<16>, dim3(gx_vec, pq->m), dim3(256), lds, st, v.ptr, n, pq->dim, pq->m,
                  pq->k, pq->d_codebooks, pq->d_scales, pq->d_offsets, c.ptr);
    else
        VG_LAUNCH(vg::pq_encode_kernel, dim3(static_cast<unsigned>((n + 255) / 256), pq->m), dim3(256),
                  lds, st, v.ptr, n, pq->dim, pq->m, pq->subdim, pq->k, pq->d_codebooks, pq->d_scales,
                  pq->d_offsets, c.ptr);
    }
    VG_TRY(c.finish());
    if (c.on_host()) VG_HIP(hipStreamSynchronize(st));
    return VG_OK;
}

VG_API int32_t vg_pq_decode(vg_pq *pq, const uint8_t *codes, int64_t n, float *out, void *stream)
{
    VG_CHECK(pq, VG_ERR_INVALID_ARG, "vg_pq_decode: NULL quantizer");
    VG_CHECK(pq->trained, VG_ERR_NOT_TRAINED, "ProductQuantizer not trained");
    VG_CHECK(n >= 0, VG_ERR_INVALID_ARG, "vg_pq_decode: n < 0");
    if (n == 0) return VG_OK;
    VG_CHECK(codes && out, VG_ERR_INVALID_ARG, "vg_pq_decode: NULL buffer");
    VG_HIP(hipSetDevice(pq->ctx->device));
    hipStream_t st = vg::pick_stream(pq->ctx, stream);
    vg::DevIn<uint8_t> c;
    vg::DevOut<float> o;
    VG_TRY(c.init(codes, static_cast<size_t>(n) * pq->m, st));
    VG_TRY(o.init(out, static_cast<size_t>(n) * pq->dim, st));
    const int64_t total = n * pq->dim;
    if (pq->subdim == 8 && pq->k == 256 && (reinterpret_cast<uintptr_t>(pq->d_codebooks) & 7) == 0 &&
        (reinterpret_cast<uintptr_t>(o.ptr) & 15) == 0) {
        const int rpt = 16;
        const int64_t threads = ((n + rpt - 1) / rpt) * pq->m;
        VG_LAUNCH(vg::pq_decode8_kernel, dim3(static_cast<unsigned>((threads + 255) / 256)), dim3(256), 0, st,
                  c.ptr, n, pq->m, pq->d_codebooks, pq->d_scales, pq->d_offsets, o.ptr, rpt);
    } else {
        VG_LAUNCH(vg::pq_decode_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0,
                           st, c.ptr, n, pq->dim, pq->m, pq->subdim, pq->k, pq->d_codebooks, pq->d_scales,
                           pq->d_offsets, o.ptr);
    }
    VG_TRY(o.finish());
    if (o.on_host()) VG_HIP(hipStreamSynchronize(st));
    return VG_OK;
}

VG_API int32_t vg_pq_asymmetric_distance_batch(vg_pq *pq, const float *query, const uint8_t *codes,
                                               int64_t n, float *out, void *stream)
{
    VG_CHECK(pq, VG_ERR_INVALID_ARG, "vg_pq_asymmetric_distance_batch: NULL quantizer");
    VG_CHECK(pq->trained, VG_ERR_NOT_TRAINED, "ProductQuantizer not trained");
    VG_CHECK(n >= 0, VG_ERR_INVALID_ARG, "vg_pq_asymmetric_distance_batch: n < 0");
    if (n == 0) return VG_OK;
    VG_CHECK(query && codes && out, VG_ERR_INVALID_ARG, "vg_pq_asymmetric_distance_batch: NULL buffer");
    VG_HIP(hipSetDevice(pq->ctx->device));
    hipStream_t st = vg::pick_stream(pq->ctx, stream);
    vg::DevIn<float> q;
    vg::DevIn<uint8_t> c;
    vg::DevOut<float> o;
    VG_TRY(q.init(query, static_cast<size_t>(pq->dim), st));
    VG_TRY(c.init(codes, static_cast<size_t>(n) * pq->m, st));
    VG_TRY(o.init(out, static_cast<size_t>(n), st));
    if (pq->subdim == 8 && pq->k == 256 && (reinterpret_cast<uintptr_t>(pq->d_codebooks) & 7) == 0 &&
        (reinterpret_cast<uintptr_t>(c.ptr) & 15) == 0)
        VG_LAUNCH(vg::pq_asym_direct_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256),
                  static_cast<size_t>(pq->m >> 1) * vg::kPqPairFloats * sizeof(float), st, q.ptr, c.ptr, n, pq->m,
                  pq->d_codebooks, pq->d_scales, pq->d_offsets, o.ptr);
    else
        VG_LAUNCH(vg::pq_asym_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, st,
                           q.ptr, c.ptr, n, pq->dim, pq->m, pq->subdim, pq->k, pq->d_codebooks, pq->d_scales,
                           pq->d_offsets, o.ptr);
    VG_TRY(o.finish());
    if (o.on_host()) VG_HIP(hipStreamSynchronize(st));
    return VG_OK;
}
