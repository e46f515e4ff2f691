// k_rabitq.hip — RaBitQuantizer (internal/quantization/rabitq.go) and simd.Hamming on the device.
#include <algorithm>

#include "vg_device.hpp"
#include "vg_exact.hpp"
#include "vg_internal.hpp"
#include "vg_cand_replay.hpp"

namespace vg {

int32_t launch_page_patch(int64_t nq, int k, int off, int kk, bool descending, const int *always_one,
                          const uint32_t *fids, const float *fscores, uint32_t *ids, float *scores, uint64_t *min_keys,
                          hipStream_t st);
int32_t launch_topk_merge(const uint64_t *partial, int64_t nq, int lists, int k, bool descending,
                          uint32_t *ids, float *scores, hipStream_t st, const int *only_if = nullptr,
                          const int *always = nullptr);

__host__ __device__ inline int rq_words(int dim) { return (dim + 63) / 64; }

// Encode (rabitq.go:51-78): 16 lanes per vector.  Norm: dotProductAvx512 order, float64 sqrt.
// Bits: lane L of the group owns elements 4L..4L+3 of each 64-element word = one nibble.
__global__ __launch_bounds__(256) void rabitq_encode_kernel(const float *__restrict__ vectors, int64_t n,
                                                            int dim, uint8_t *__restrict__ codes)
{
    const int64_t row = static_cast<int64_t>(blockIdx.x) * 16 + (threadIdx.x >> 4);
    if (row >= n) return;
    const Sub16 sub = Sub16::make(threadIdx.x);
    const int L = threadIdx.x & 15;
    const float *v = vectors + row * dim;
    const int nw = rq_words(dim);
    const int64_t cb = static_cast<int64_t>(nw) * 8 + 4;
    uint8_t *out = codes + row * cb;
    const float sumsq = exact_pair16<true, kPair>(v, v, dim, sub);
    const float norm = static_cast<float>(sqrt(static_cast<double>(sumsq)));
    for (int w = 0; w < nw; w++) {
        uint32_t nib = 0;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int e = w * 64 + 4 * L + t;
            if (e < dim && v[e] >= 0.0f) nib |= 1u << t;
        }
        const uint32_t other = static_cast<uint32_t>(
            __builtin_amdgcn_update_dpp(0, static_cast<int>(nib), kDppQuadXor1, 0xF, 0xF, false));
        if ((L & 1) == 0) out[w * 8 + (L >> 1)] = static_cast<uint8_t>(nib | (other << 4));
    }
    if (L == 0) {
        uint32_t nb = __float_as_uint(norm);
        out[nw * 8 + 0] = nb & 0xFF;
        out[nw * 8 + 1] = (nb >> 8) & 0xFF;
        out[nw * 8 + 2] = (nb >> 16) & 0xFF;
        out[nw * 8 + 3] = (nb >> 24) & 0xFF;
    }
}

__device__ inline float rq_formula(float qn, float yn, float dimf, float hamming)
{
    // rabitq.go:170-175, fp32, left to right, no fusion
    const float t1 = qn - yn;
    const float t1sq = t1 * t1;
    float t2 = 4.0f * qn;
    t2 = t2 * yn;
    t2 = t2 / dimf;
    t2 = t2 * hamming;
    return t1sq + t2;
}

// popcount(a ^ c) over nbytes bytes: dwords, 8 loads in flight at a time, when both are 4-byte aligned
__device__ __forceinline__ int hamming_bytes(const uint8_t *__restrict__ a, const uint8_t *__restrict__ c, int64_t nbytes)
{
    int h = 0;
    int64_t b = 0;
    if (((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(c)) & 3) == 0) {
        const uint32_t *aw = reinterpret_cast<const uint32_t *>(a), *cw = reinterpret_cast<const uint32_t *>(c);
        const int64_t nw = nbytes >> 2;
        int64_t w = 0;
        for (; w + 8 <= nw; w += 8) {
            uint32_t x[8];
#pragma unroll
            for (int u = 0; u < 8; u++) x[u] = cw[w + u];
#pragma unroll
            for (int u = 0; u < 8; u++) h += __popc(x[u] ^ aw[w + u]);
        }
        for (; w < nw; w++) h += __popc(cw[w] ^ aw[w]);
        b = nw << 2;
    }
    for (; b < nbytes; b++) h += __popc(static_cast<unsigned>(a[b] ^ c[b]));
    return h;
}

// Distance of one query code (bits + norm, as produced by Encode) against n reference-layout
// codes: thread per row.
__global__ void rabitq_distance_kernel(const uint8_t *__restrict__ qcode, const uint8_t *__restrict__ codes,
                                       int64_t n, int dim, float *__restrict__ out)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int nb = rq_words(dim) * 8;
    const uint8_t *c = codes + i * (nb + 4);
    const int h = hamming_bytes(qcode, c, nb);
    uint32_t qb = qcode[nb] | (qcode[nb + 1] << 8) | (qcode[nb + 2] << 16) | (static_cast<uint32_t>(qcode[nb + 3]) << 24);
    uint32_t yb = c[nb] | (c[nb + 1] << 8) | (c[nb + 2] << 16) | (static_cast<uint32_t>(c[nb + 3]) << 24);
    out[i] = rq_formula(__uint_as_float(qb), __uint_as_float(yb), static_cast<float>(dim), static_cast<float>(h));
}

__global__ void hamming_batch_kernel(const uint8_t *__restrict__ a, const uint8_t *__restrict__ codes,
                                     int64_t nbytes, int64_t n, int32_t *__restrict__ out)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = hamming_bytes(a, codes + i * nbytes, nbytes);
}

// reference layout -> [tile][group][lane] 16-byte pieces of the sign bits + norms[n]
__global__ void rabitq_retile_kernel(const uint8_t *__restrict__ codes, int64_t n, int nb, int groups,
                                     int64_t n_tiles, uint4 *__restrict__ tiles, float *__restrict__ norms)
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
        const uint8_t *src = codes + row * (nb + 4);
        for (int b = 0; b < 16; b++) {
            const int at = g * 16 + b;
            if (at < nb) w[b >> 2] |= static_cast<uint32_t>(src[at]) << (8 * (b & 3));
        }
        if (g == 0) {
            uint32_t yb = src[nb] | (src[nb + 1] << 8) | (src[nb + 2] << 16) | (static_cast<uint32_t>(src[nb + 3]) << 24);
            norms[row] = __uint_as_float(yb);
        }
    }
    tiles[gid] = make_uint4(w[0], w[1], w[2], w[3]);
}

// Exhaustive RaBitQ scan with fused top-k.  HBM-bound: (16*groups + 4) bytes per row.
constexpr int kRqTiles = 2;  // 64-row tiles per trip of the d = 768 scan (4 is no faster)
constexpr int kRqWaves = 4;
constexpr int kRqThreads = kRqWaves * 64;
__global__ __launch_bounds__(kRqThreads) void rabitq_scan_kernel(
    const uint4 *__restrict__ tiles, const float *__restrict__ norms, int64_t n_rows, int64_t n_tiles,
    int groups, int dim, const uint8_t *__restrict__ qcodes /* nq * (nb+4) */, int nb, int slices, int nq,
    int k, uint64_t *__restrict__ partial, const uint64_t *__restrict__ min_keys)
{
    __shared__ uint4 qbits[64];  // up to 1024 bytes of sign bits (dim <= 8192)
    __shared__ uint64_t lists[kRqWaves * 64];
    __shared__ int valid[kRqWaves];
    const int b = blockIdx.x;
    const int xcd = b & 7;
    const int o = b >> 3;
    const int q = o % nq;
    const int s = (o / nq) * 8 + xcd;
    const int64_t t0 = n_tiles * s / slices, t1 = n_tiles * (s + 1) / slices;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint8_t *qc = qcodes + static_cast<int64_t>(q) * (nb + 4);
    if (tid < groups) {
        uint32_t w[4] = {0, 0, 0, 0};
        for (int bb = 0; bb < 16; bb++) {
            const int at = tid * 16 + bb;
            if (at < nb) w[bb >> 2] |= static_cast<uint32_t>(qc[at]) << (8 * (bb & 3));
        }
        qbits[tid] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    const uint32_t qnb = qc[nb] | (qc[nb + 1] << 8) | (qc[nb + 2] << 16) | (static_cast<uint32_t>(qc[nb + 3]) << 24);
    const float qn = __uint_as_float(qnb);
    const float dimf = static_cast<float>(dim);
    __syncthreads();
    WaveTopK tk;
    tk.init(k);
    // paged results (k > 64): only keys after the previous page's last one (the first page passes no floor)
    const bool paged = min_keys != nullptr;
    const uint64_t floor_key = paged ? min_keys[q] : 0;
    // several tiles per trip: kRqTiles * groups independent 16-byte loads in flight per lane before the
    // first popcount (one tile per trip left the wave idle for a full HBM round trip per 64 rows)
    auto score = [&](int64_t tile, int h) {
        const int64_t row = tile * 64 + lane;
        uint64_t key = kKeyMax;
        if (row < n_rows) {
            const float d = rq_formula(qn, norms[row], dimf, static_cast<float>(h));
            key = make_key(d, static_cast<uint32_t>(row), false);
            if (paged && key <= floor_key) key = kKeyMax;
        }
        tk.offer(key, lane);
    };
    // One-query passes DEAL the tiles round-robin over the workgroups (trip i of workgroup s, wave w = tile
    // (i*slices + s)*waves + w): the chip streams one moving window of the code array instead of `slices` distant
    // ones (the mapping that bought the ADC scan 4 %; here 154.7 -> 153.5 us at 10M rows, within
    // noise).  Several queries keep the slice mapping: consecutive workgroups of an XCD take different queries over
    // the SAME slice and share it in L2.
    const bool dealt = nq == 1;
    const int64_t step = dealt ? static_cast<int64_t>(slices) * kRqWaves : kRqWaves;
    const int64_t end = dealt ? n_tiles : t1;
    int64_t tile = dealt ? static_cast<int64_t>(s) * kRqWaves + wave : t0 + wave;
    if (groups == 6) {  // d = 768: fully unrolled; kRqTiles tiles (6 * kRqTiles 16-byte loads) in flight per lane
        // r04: a RING of tiles instead of trips — a slot is refilled with the tile kRqTiles steps ahead as soon as its
        // popcounts are done, so the wave always has kRqTiles - 1 tiles in flight while it scores one (the trips loaded
        // kRqTiles tiles, scored them all, and started the next loads with nothing in flight)
        uint4 c[kRqTiles][6];
        float y[kRqTiles];
        int64_t tl[kRqTiles];
        auto fill = [&](int t) {  // (t is a compile-time slot after unrolling)
            const uint4 *tp = tiles + (tl[t] * 6) * 64 + lane;
#pragma unroll
            for (int g = 0; g < 6; g++) c[t][g] = load_stream(tp + g * 64);
            const int64_t row = tl[t] * 64 + lane;
            y[t] = row < n_rows ? norms[row] : 0.0f;
        };
#pragma unroll
        for (int t = 0; t < kRqTiles; t++) {
            tl[t] = tile + t * step;
            if (tl[t] < end) fill(t);
        }
        for (bool any = true; any;) {
            any = false;
#pragma unroll
            for (int t = 0; t < kRqTiles; t++) {
                if (tl[t] >= end) continue;  // (uniform)
                any = true;
                int h = 0;
#pragma unroll
                for (int g = 0; g < 6; g++) {
                    const uint4 qq = qbits[g];
                    h += __popc(c[t][g].x ^ qq.x) + __popc(c[t][g].y ^ qq.y) + __popc(c[t][g].z ^ qq.z) + __popc(c[t][g].w ^ qq.w);
                }
                const int64_t row = tl[t] * 64 + lane;
                const float yn = y[t];
                tl[t] += static_cast<int64_t>(kRqTiles) * step;
                if (tl[t] < end) fill(t);
                uint64_t key = kKeyMax;
                if (row < n_rows) {
                    key = make_key(rq_formula(qn, yn, dimf, static_cast<float>(h)), static_cast<uint32_t>(row), false);
                    if (paged && key <= floor_key) key = kKeyMax;
                }
                tk.offer(key, lane);
            }
        }
        tile = end;
    }
    for (; tile < end; tile += step) {
        const uint4 *tp = tiles + (tile * groups) * 64 + lane;
        int h = 0;
        for (int g = 0; g < groups; g++) {
            const uint4 c = load_stream(tp + g * 64);
            const uint4 qq = qbits[g];
            h += __popc(c.x ^ qq.x) + __popc(c.y ^ qq.y) + __popc(c.z ^ qq.z) + __popc(c.w ^ qq.w);
        }
        score(tile, h);
    }
    wg_rank_merge<kRqWaves>(tk, lists, valid, wave, lane, tid, k,
                            partial + (static_cast<int64_t>(q) * slices + s) * k);
}

// Exhaustive RaBitQ scan of a BATCH of queries.  The reference scans one query at a time (flat/segment.go:606-723);
// run that way on the GPU every query streams all the codes again (r02: 1024 queries x 5M rows in 113 ms, the
// codes served from L2).  Here a workgroup takes kRqMq queries over its slice: a wave loads a row's sign bits ONCE
// (kRqMqTiles tiles of 64 rows: groups * kRqMqTiles 16-byte loads in flight per lane) and keeps them in registers
// while the queries' bits arrive by LDS broadcast reads — per (row, query) 2 vector instructions per 32 dimensions
// (v_xor + v_bcnt accumulate) + the distance formula with its true division (rabitq.go:170-175), i.e. bound by the
// vector ALU, not by HBM or L2: ~65 instructions per 64 (row, query) pairs.  Every query keeps its own k best keys
// (one per lane, registers); a float pre-test against the k-th score skips the key construction for rows that cannot
// enter.  Block order as in the one-query kernel: the query blocks of one slice run on one XCD and share the slice
// in its L2.
constexpr int kRqMq = 16;      // queries per workgroup pass
constexpr int kRqMqTiles = 2;  // 64-row tiles per trip
template <int G>               // 16-byte groups per row, compile-time (6 at dim 768); 0 = runtime `groups`
__global__ __launch_bounds__(kRqThreads) void rabitq_scan_mq_kernel(
    const uint4 *__restrict__ tiles, const float *__restrict__ norms, int64_t n_rows, int64_t n_tiles,
    int groups_rt, int dim, const uint8_t *__restrict__ qcodes /* nq * (nb+4) */, int nb, int slices, int nq,
    int k, uint64_t *__restrict__ partial, const uint64_t *__restrict__ min_keys)
{
    extern __shared__ __attribute__((aligned(16))) uint4 rq_smem[];  // kRqMq * groups query bits, then merge scratch
    const int groups = G ? G : groups_rt;
    uint4 *qbits = rq_smem;
    uint64_t *lists = reinterpret_cast<uint64_t *>(rq_smem + kRqMq * groups);
    int *valid = reinterpret_cast<int *>(lists + kRqWaves * 64);
    float *qnorm = reinterpret_cast<float *>(valid + kRqWaves);
    const int ng = (nq + kRqMq - 1) / kRqMq;
    const int b = blockIdx.x;
    const int xcd = b & 7;
    const int o = b >> 3;
    const int qg = o % ng;
    const int s = (o / ng) * 8 + xcd;
    const int q0 = qg * kRqMq;
    const int cnt = nq - q0 < kRqMq ? nq - q0 : kRqMq;
    const int64_t t0 = n_tiles * s / slices, t1 = n_tiles * (s + 1) / slices;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < cnt * groups; e += kRqThreads) {
        const int qi = e / groups, g = e - qi * groups;
        const uint8_t *qc = qcodes + static_cast<int64_t>(q0 + qi) * (nb + 4);
        uint32_t w[4] = {0, 0, 0, 0};
        for (int bb = 0; bb < 16; bb++) {
            const int at = g * 16 + bb;
            if (at < nb) w[bb >> 2] |= static_cast<uint32_t>(qc[at]) << (8 * (bb & 3));
        }
        qbits[e] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    if (tid < cnt) {
        const uint8_t *qc = qcodes + static_cast<int64_t>(q0 + tid) * (nb + 4);
        qnorm[tid] = __uint_as_float(qc[nb] | (qc[nb + 1] << 8) | (qc[nb + 2] << 16) | (static_cast<uint32_t>(qc[nb + 3]) << 24));
    }
    __syncthreads();
    const float dimf = static_cast<float>(dim);
    WaveTopK tk[kRqMq];
    float tau_d[kRqMq];  // score of the k-th key (+inf until k keys were seen): rows above it cannot enter
#pragma unroll
    for (int qi = 0; qi < kRqMq; qi++) {
        tk[qi].init(k);
        tau_d[qi] = INFINITY;
    }
    const bool paged = min_keys != nullptr;
    constexpr int GM = G ? G : 1;
    for (int64_t tile = t0 + wave * kRqMqTiles; tile < t1; tile += kRqWaves * kRqMqTiles) {
        uint4 c[kRqMqTiles][GM];
        float y[kRqMqTiles];
        bool live[kRqMqTiles];
#pragma unroll
        for (int t = 0; t < kRqMqTiles; t++) {
            const int64_t tt = tile + t < t1 ? tile + t : tile;  // a trip's second tile may lie past the slice
            const int64_t row = tt * 64 + lane;
            live[t] = tile + t < t1 && row < n_rows;
            if (G) {
#pragma unroll
                for (int g = 0; g < GM; g++) c[t][g] = tiles[(tt * G + g) * 64 + lane];
            }
            y[t] = live[t] ? norms[row] : 0.0f;
        }
#pragma unroll
        for (int qi = 0; qi < kRqMq; qi++) {
            if (qi < cnt) {
                int h[kRqMqTiles];
#pragma unroll
                for (int t = 0; t < kRqMqTiles; t++) h[t] = 0;
                if (G) {
#pragma unroll
                    for (int g = 0; g < GM; g++) {
                        const uint4 qq = qbits[qi * G + g];
#pragma unroll
                        for (int t = 0; t < kRqMqTiles; t++)
                            h[t] += __popc(c[t][g].x ^ qq.x) + __popc(c[t][g].y ^ qq.y) + __popc(c[t][g].z ^ qq.z) +
                                    __popc(c[t][g].w ^ qq.w);
                    }
                } else {
                    for (int g = 0; g < groups; g++) {
                        const uint4 qq = qbits[qi * groups + g];
#pragma unroll
                        for (int t = 0; t < kRqMqTiles; t++) {
                            const int64_t tt = tile + t < t1 ? tile + t : tile;
                            const uint4 cc = tiles[(tt * groups + g) * 64 + lane];
                            h[t] += __popc(cc.x ^ qq.x) + __popc(cc.y ^ qq.y) + __popc(cc.z ^ qq.z) + __popc(cc.w ^ qq.w);
                        }
                    }
                }
                const float qn = qnorm[qi];
#pragma unroll
                for (int t = 0; t < kRqMqTiles; t++) {
                    const float d = rq_formula(qn, y[t], dimf, static_cast<float>(h[t]));
                    if (__ballot(live[t] && d <= tau_d[qi])) {
                        const int64_t row = (tile + t) * 64 + lane;
                        uint64_t key = live[t] ? make_key(d, static_cast<uint32_t>(row), false) : kKeyMax;
                        if (paged && key <= min_keys[q0 + qi]) key = kKeyMax;
                        tk[qi].offer(key, lane);
                        tau_d[qi] = tk[qi].tau == kKeyMax ? INFINITY : key_score(tk[qi].tau, false);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int qi = 0; qi < kRqMq; qi++) {
        if (qi < cnt) {
            wg_rank_merge<kRqWaves>(tk[qi], lists, valid, wave, lane, tid, k,
                                    partial + (static_cast<int64_t>(q0 + qi) * slices + s) * k);
            __syncthreads();
        }
    }
}

int32_t launch_rabitq_encode(const float *d_vectors, int64_t n, int dim, uint8_t *d_codes, hipStream_t st)
{
    if (n == 0) return VG_OK;
    VG_LAUNCH(rabitq_encode_kernel, dim3(static_cast<unsigned>((n + 15) / 16)), dim3(256), 0, st,
                       d_vectors, n, dim, d_codes);
    return VG_OK;
}

// max |v[i]| (NaN / Inf counted as +Inf) into *out_bits (zeroed by the caller; non-negative floats order like their bits)
__global__ __launch_bounds__(256) void absmax_kernel(const float *__restrict__ v, int64_t n, int *__restrict__ out_bits)
{
    float mx = 0.0f;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * 256) {
        const float a = fabsf(v[i]);
        mx = fmaxf(mx, a == a ? a : INFINITY);
    }
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if ((threadIdx.x & 63) == 0) atomicMax(out_bits, __float_as_int(mx));
}

// vg_cand_replay.hpp's scorer for the RaBitQ scan: rq.Distance of the query's code (sign bits + norm, rabitq.go:119-176) against a
// row's, from the row-major copy of the codes.  A NaN: a non-finite norm on either side, or 4 |q| |y| overflowing to +Inf next
// to a Hamming distance of 0 (rq_formula: t2 = 4 qn yn / dim * hamming).
struct RabitqScorer {
    const uint8_t *rows;     // n * (nb + 4)
    const uint8_t *qcodes;   // nq * (nb + 4), the queries' own codes (rabitq_encode_kernel)
    const float *norm_absmax;
    int dim, nb;
    __device__ const uint8_t *qcode(int64_t qi) const { return qcodes + qi * (nb + 4); }
    __device__ static float norm_of(const uint8_t *c, int nb)
    {
        return __uint_as_float(c[nb] | (c[nb + 1] << 8) | (c[nb + 2] << 16) | (static_cast<uint32_t>(c[nb + 3]) << 24));
    }
    __device__ bool risk(int64_t qi, const float *, int tid) const
    {
        __shared__ int flag;
        const float qn = norm_of(qcode(qi), nb), ma = norm_absmax[0];
        const bool bad = !is_finite_f32(qn) || !is_finite_f32(ma) || !(4.0f * fabsf(qn) * ma < 1e38f) || !(score_bound(fabsf(qn), ma, false) < 1e38f);
        return block_any(bad, &flag, tid);
    }
    __device__ void prepare(int64_t, const float *, int) const {}
    __device__ void score_chunk(int64_t qi, const float *, int64_t row0, int64_t n, int tid, float *out) const
    {
        const int64_t row = row0 + tid;
        if (row >= n) return;
        const uint8_t *qc = qcode(qi), *c = rows + row * (nb + 4);
        const int h = hamming_bytes(qc, c, nb);
        out[tid] = rq_formula(norm_of(qc, nb), norm_of(c, nb), static_cast<float>(dim), static_cast<float>(h));
    }
};

static int rq_slices(int64_t nq, int64_t n_tiles, int cus)
{
    int64_t s = (4 * static_cast<int64_t>(cus) + nq - 1) / nq;  // ~4 workgroups of 256 threads per CU (swept 2..8: 3-4 best)
    s = ((s + 7) / 8) * 8;
    int64_t max_s = (n_tiles / 8) * 8;
    if (max_s < 8) max_s = 8;
    if (s > max_s) s = max_s;
    if (s < 8) s = 8;
    return static_cast<int>(s);
}

}  // namespace vg

VG_API int64_t vg_rabitq_code_bytes(int32_t dim) { return static_cast<int64_t>((dim + 63) / 64) * 8 + 4; }

VG_API int32_t vg_rabitq_encode(vg_ctx *ctx, int32_t dim, const float *vectors, int64_t n, uint8_t *codes,
                                void *stream)
{
    VG_CHECK(ctx, VG_ERR_INVALID_ARG, "vg_rabitq_encode: ctx is NULL");
    VG_CHECK(dim > 0 && n >= 0, VG_ERR_INVALID_ARG, "vg_rabitq_encode: bad dim or n");
    if (n == 0) return VG_OK;
    VG_CHECK(vectors && codes, VG_ERR_INVALID_ARG, "vg_rabitq_encode: NULL buffer");
    VG_HIP(hipSetDevice(ctx->device));
    hipStream_t st = vg::pick_stream(ctx, stream);
    const int64_t cb = vg_rabitq_code_bytes(dim);
    vg::DevIn<float> v;
    vg::DevOut<uint8_t> c;
    VG_TRY(v.init(vectors, static_cast<size_t>(n) * dim, st));
    VG_TRY(c.init(codes, static_cast<size_t>(n) * cb, st));
    VG_LAUNCH(vg::rabitq_encode_kernel, dim3(static_cast<unsigned>((n + 15) / 16)), dim3(256), 0, st,
                       v.ptr, n, dim, c.ptr);
    VG_TRY(c.finish());
    if (c.on_host()) VG_HIP(hipStreamSynchronize(st));
    return VG_OK;
}

VG_API int32_t vg_rabitq_distance_batch(vg_ctx *ctx, int32_t dim, const float *query, const uint8_t *codes,
                                        int64_t n, float *out, void *stream)
{
    VG_CHECK(ctx, VG_ERR_INVALID_ARG, "vg_rabitq_distance_batch: ctx is NULL");
    VG_CHECK(dim > 0 && n >= 0, VG_ERR_INVALID_ARG, "vg_rabitq_distance_batch: bad dim or n");
    if (n == 0) return VG_OK;
    VG_CHECK(query && codes && out, VG_ERR_INVALID_ARG, "vg_rabitq_distance_batch: NULL buffer");
    VG_HIP(hipSetDevice(ctx->device));
    hipStream_t st = vg::pick_stream(ctx, stream);
    const int64_t cb = vg_rabitq_code_bytes(dim);
    vg::DevIn<float> q;
    vg::DevIn<uint8_t> c;
    vg::DevOut<float> o;
    vg::DevTmp<uint8_t> qcode;
    VG_TRY(q.init(query, static_cast<size_t>(dim), st));
    VG_TRY(c.init(codes, static_cast<size_t>(n) * cb, st));
    VG_TRY(o.init(out, static_cast<size_t>(n), st));
    VG_TRY(qcode.init(static_cast<size_t>(cb), st));
    VG_LAUNCH(vg::rabitq_encode_kernel, dim3(1), dim3(256), 0, st, q.ptr, int64_t(1), dim, qcode.ptr);
    VG_LAUNCH(vg::rabitq_distance_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, st,
                       qcode.ptr, c.ptr, n, dim, o.ptr);
    VG_TRY(o.finish());
    if (o.on_host()) VG_HIP(hipStreamSynchronize(st));
    return VG_OK;
}

VG_API int32_t vg_hamming_batch(vg_ctx *ctx, const uint8_t *a, const uint8_t *codes, int64_t nbytes, int64_t n,
                                int32_t *out, void *stream)
{
    VG_CHECK(ctx, VG_ERR_INVALID_ARG, "vg_hamming_batch: ctx is NULL");
    VG_CHECK(nbytes >= 0 && n >= 0, VG_ERR_INVALID_ARG, "vg_hamming_batch: negative size");
    if (n == 0) return VG_OK;
    VG_CHECK(out && (nbytes == 0 || (a && codes)), VG_ERR_INVALID_ARG, "vg_hamming_batch: NULL buffer");
    VG_HIP(hipSetDevice(ctx->device));
    hipStream_t st = vg::pick_stream(ctx, stream);
    vg::DevIn<uint8_t> da, dc;
    vg::DevOut<int32_t> o;
    VG_TRY(da.init(a, static_cast<size_t>(nbytes), st));
    VG_TRY(dc.init(codes, static_cast<size_t>(n) * nbytes, st));
    VG_TRY(o.init(out, static_cast<size_t>(n), st));
    VG_LAUNCH(vg::hamming_batch_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, st,
                       da.ptr, dc.ptr, nbytes, n, o.ptr);
    VG_TRY(o.finish());
    if (o.on_host()) VG_HIP(hipStreamSynchronize(st));
    return VG_OK;
}

VG_API int32_t vg_index_set_rabitq_codes(vg_index *idx, const uint8_t *codes, void *stream)
{
    VG_CHECK(idx, VG_ERR_INVALID_ARG, "vg_index_set_rabitq_codes: NULL index");
    VG_CHECK(idx->n == 0 || codes, VG_ERR_INVALID_ARG, "vg_index_set_rabitq_codes: codes is NULL");
    VG_CHECK(idx->dim <= 8192, VG_ERR_UNSUPPORTED, "vg_index_set_rabitq_codes: dim %d > 8192", idx->dim);
    VG_HIP(hipSetDevice(idx->ctx->device));
    hipStream_t st = vg::pick_stream(idx->ctx, stream);
    if (idx->d_rq_tiles) {
        VG_HIP(hipStreamSynchronize(st));
        VG_HIP(hipFree(idx->d_rq_tiles));
        VG_HIP(hipFree(idx->d_rq_norms));
        idx->d_rq_tiles = nullptr;
        idx->d_rq_norms = nullptr;
    }
    const int nb = vg::rq_words(idx->dim) * 8;
    idx->rq_groups = (nb + 15) / 16;
    idx->n_tiles = (idx->n + 63) / 64;
    if (idx->n == 0) return VG_OK;
    const int64_t total = idx->n_tiles * idx->rq_groups * 64;
    VG_HIP(hipMalloc(reinterpret_cast<void **>(&idx->d_rq_tiles), static_cast<size_t>(total) * 16));
    // (+ one float: the largest |norm|, +Inf when one of them is not finite — vg_cand_replay.hpp's risk test)
    VG_HIP(hipMalloc(reinterpret_cast<void **>(&idx->d_rq_norms), static_cast<size_t>(idx->n + 1) * sizeof(float)));
    VG_HIP(hipMemsetAsync(idx->d_rq_norms + idx->n, 0, sizeof(float), st));
    vg::DevIn<uint8_t> in;
    VG_TRY(in.init(codes, static_cast<size_t>(idx->n) * (nb + 4), st));
    if (idx->d_rq_rows) {
        VG_HIP(hipFree(idx->d_rq_rows));
        idx->d_rq_rows = nullptr;
    }
    VG_HIP(hipMalloc(reinterpret_cast<void **>(&idx->d_rq_rows), static_cast<size_t>(idx->n) * (nb + 4)));
    VG_HIP(hipMemcpyAsync(idx->d_rq_rows, in.ptr, static_cast<size_t>(idx->n) * (nb + 4), hipMemcpyDeviceToDevice, st));
    VG_LAUNCH(vg::rabitq_retile_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, st,
                       in.ptr, idx->n, nb, idx->rq_groups, idx->n_tiles, reinterpret_cast<uint4 *>(idx->d_rq_tiles),
                       idx->d_rq_norms);
    VG_LAUNCH(vg::absmax_kernel, dim3(static_cast<unsigned>(std::min<int64_t>((idx->n + 255) / 256, 1024))), dim3(256), 0, st, idx->d_rq_norms,
              idx->n, reinterpret_cast<int *>(idx->d_rq_norms + idx->n));
    VG_HIP(hipStreamSynchronize(st));
    return VG_OK;
}

VG_API int32_t vg_search_rabitq(vg_index *idx, const float *queries, int64_t nq, int32_t k, uint32_t *ids,
                                float *scores, void *stream)
{
    VG_CHECK(idx, VG_ERR_INVALID_ARG, "vg_search_rabitq: NULL index");
    VG_CHECK(nq >= 0 && k >= 0, VG_ERR_INVALID_ARG, "vg_search_rabitq: negative nq or k");
    if (nq == 0 || k == 0) return VG_OK;
    VG_CHECK(idx->n == 0 || idx->d_rq_tiles, VG_ERR_NOT_READY, "vg_search_rabitq: index has no RaBitQ codes");
    VG_CHECK(queries && ids && scores, VG_ERR_INVALID_ARG, "vg_search_rabitq: NULL buffer");
    VG_CHECK(k <= 512, VG_ERR_UNSUPPORTED, "vg_search_rabitq: k=%d exceeds 512", k);
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
        const int nb = vg::rq_words(idx->dim) * 8;
        // two or more queries: blocks of kRqMq queries share every code load (rabitq_scan_mq_kernel)
        const bool mq = nq >= 2;
        const int64_t units = mq ? (nq + vg::kRqMq - 1) / vg::kRqMq : nq;  // workgroups per slice
        const int slices = vg::rq_slices(units, idx->n_tiles, idx->ctx->compute_units);
        const size_t mq_lds = static_cast<size_t>(vg::kRqMq) * idx->rq_groups * 16 + vg::kRqWaves * 64 * sizeof(uint64_t) +
                              vg::kRqWaves * sizeof(int) + vg::kRqMq * sizeof(float);
        vg::ArenaCall ar(idx->ctx, st);
        const int i_qcodes = ar.add(static_cast<size_t>(nq) * (nb + 4));
        // a wave keeps 64 keys: k > 64 comes in pages of 64, one scan per page
        const bool paged = k > 64;
        const int pk = paged ? 64 : k;
        const int i_partial = ar.add(sizeof(uint64_t) * static_cast<size_t>(nq) * slices * pk);
        const int i_pid = ar.add(paged ? sizeof(uint32_t) * static_cast<size_t>(nq) * pk : 0);
        const int i_psc = ar.add(paged ? sizeof(float) * static_cast<size_t>(nq) * pk : 0);
        const int i_floor = ar.add(paged ? sizeof(uint64_t) * static_cast<size_t>(nq) : 0);
        const int i_one = ar.add(paged ? 256 : 0);
        VG_TRY(ar.commit());
        struct { uint8_t *ptr; } qcodes{ar.get<uint8_t>(i_qcodes)};
        struct { uint64_t *ptr; } partial{ar.get<uint64_t>(i_partial)};
        uint64_t *floor_keys = ar.get<uint64_t>(i_floor);
        uint32_t *pid = ar.get<uint32_t>(i_pid);
        float *psc = ar.get<float>(i_psc);
        int *one = ar.get<int>(i_one);
        if (paged) VG_HIP(hipMemsetAsync(one, 1, sizeof(int), st));
        VG_LAUNCH(vg::rabitq_encode_kernel, dim3(static_cast<unsigned>((nq + 15) / 16)), dim3(256), 0, st,
                           q.ptr, nq, idx->dim, qcodes.ptr);
        const int64_t max_q = (1ll << 30) / slices;
        for (int off = 0; off < k; off += 64) {
            const int kk = paged ? std::min(64, k - off) : k;
            for (int64_t q0 = 0; mq && q0 < nq; q0 += max_q * vg::kRqMq) {  // whole query blocks per launch
                const int64_t cnt = std::min<int64_t>(nq - q0, max_q * vg::kRqMq);
                const int64_t ng = (cnt + vg::kRqMq - 1) / vg::kRqMq;
                auto kern = idx->rq_groups == 6 ? vg::rabitq_scan_mq_kernel<6> : vg::rabitq_scan_mq_kernel<0>;
                vg::ProfScope prof(idx->ctx, "rabitq_scan_mq", st);
                VG_LAUNCH(kern, dim3(static_cast<unsigned>(ng * slices)), dim3(vg::kRqThreads), mq_lds, st,
                          reinterpret_cast<const uint4 *>(idx->d_rq_tiles), idx->d_rq_norms, idx->n, idx->n_tiles,
                          idx->rq_groups, idx->dim, qcodes.ptr + q0 * (nb + 4), nb, slices, static_cast<int>(cnt), kk,
                          partial.ptr + q0 * slices * kk, off ? floor_keys + q0 : nullptr);
            }
            for (int64_t q0 = 0; !mq && q0 < nq; q0 += max_q) {
                const int64_t cnt = nq - q0 < max_q ? nq - q0 : max_q;
                vg::ProfScope prof(idx->ctx, "rabitq_scan", st);
                VG_LAUNCH(vg::rabitq_scan_kernel, dim3(static_cast<unsigned>(cnt * slices)),
                                   dim3(vg::kRqThreads), 0, st, reinterpret_cast<const uint4 *>(idx->d_rq_tiles),
                                   idx->d_rq_norms, idx->n, idx->n_tiles, idx->rq_groups, idx->dim,
                                   qcodes.ptr + q0 * (nb + 4), nb, slices, static_cast<int>(cnt), kk,
                                   partial.ptr + q0 * slices * kk, off ? floor_keys + q0 : nullptr);
            }
            if (!paged) {
                VG_TRY(vg::launch_topk_merge(partial.ptr, nq, slices, k, false, oid.ptr, osc.ptr, st));
            } else {
                VG_TRY(vg::launch_topk_merge(partial.ptr, nq, slices, kk, false, pid, psc, st));
                VG_TRY(vg::launch_page_patch(nq, k, off, kk, false, one, pid, psc, oid.ptr, osc.ptr, floor_keys, st));
            }
        }
        // queries whose distances may hold a NaN: the reference's heap, operation by operation (vg_cand_replay.hpp)
        VG_TRY(vg::launch_cand_replay(vg::RabitqScorer{idx->d_rq_rows, qcodes.ptr, idx->d_rq_norms + idx->n, idx->dim, nb}, q.ptr, idx->dim, idx->n, nq, k,
                                      false, nullptr, 0, oid.ptr, osc.ptr, st));
    }
    VG_TRY(oid.finish());
    VG_TRY(osc.finish());
    if (oid.on_host() || osc.on_host()) VG_HIP(hipStreamSynchronize(st));
    return VG_OK;
}
