// k_adc.hip — PQ-ADC flat scan with fused top-k (flat/segment.go:476-483, :678-689,
// :714-721) for gfx950.
//
// HBM layout of the codes ("tiles"): the reference keeps codes row-major,
// codes[i*m:(i+1)*m].  A lane-per-row scan of that layout issues 16-byte loads at a
// 96-byte stride (48 partially used lines per wave-instruction).  set_pq_codes re-tiles
// once: tile t = rows [64t, 64t+64); within a tile, group g (16 consecutive
// sub-quantizers) of all 64 rows is contiguous:
//     byte address = ((t*G + g)*64 + lane)*16,   G = ceil(m/16)
// so every wave-level load is one fully coalesced 1 KiB request and the scan streams
// exactly N*G*16 bytes (= N*m when 16 | m).
//
// LDS bank conflicts: a 256-entry table row per sub-quantizer puts a wave's 64 random
// lookups on random banks (ds_read_b32: 32 lanes over 32 banks, ~3.4 cycles per 32-lane
// group instead of 1).  Full 16-wide groups therefore use a ROTATED image: lane i
// (r = i & 15) reads, in slot s, sub-quantizer l = (s + r) & 15 of the group, and the LUT is
// stored [g][c][l] so that bank = 16*(c & 1) + l — the 16 rotations of a 32-lane group never
// collide, only the two lanes sharing a rotation can (p = 1/2).  The code bytes are
// pre-rotated per lane at re-tile time so slot s is byte s of the lane's 16-byte word.
// The rotation does not change the arithmetic: slot s accumulates the same
// sub-quantizers (g ascending) as reference lane l = (s+r)&15, and the
// _mm512_reduce_add_ps tree (i,i+8),(i,i+4),(i,i+2),(0,1) is invariant under a cyclic
// rotation of its 16 inputs because fp32 addition is commutative.
//
// Per-row arithmetic = pqAdcLookupAvx512 (internal/simd/src/floats_avx512.c:135-167):
// 16 lane accumulators acc[l] += table[(16g+l)*256 + code[16g+l]] for g ascending, the
// _mm512_reduce_add_ps tree, then the m%16 tail added sequentially.  fp32 adds only.
#include <algorithm>

#include "vg_device.hpp"
#include "vg_internal.hpp"
#include "vg_cand_replay.hpp"

namespace vg {

// with more than two waves per SIMD the per-wave register budget is what counts: keep hipcc from hoisting
// the byte extraction of a whole tile (96 temporaries) above the lookups it feeds
#if defined(VG_ADC_FENCE) && VG_ADC_FENCE
#define VG_ADC_SCHED_FENCE __builtin_amdgcn_sched_barrier(0);
#else
#define VG_ADC_SCHED_FENCE
#endif
#ifndef VG_ADC_DEAL
#define VG_ADC_DEAL 1
#endif
#ifndef VG_ADC_WAVES
#define VG_ADC_WAVES 8
#endif
constexpr int kAdcWaves = VG_ADC_WAVES;       // 512 threads, one workgroup per CU (LDS holds the LUT):
                                              // 2 waves/SIMD leaves each wave 256 VGPRs for deep prefetch
constexpr int kAdcThreads = kAdcWaves * kWave;
constexpr int kAdcSyncEvery = 2;              // iterations between candidate-buffer checks
constexpr int kAdcBuf = 4096;                 // candidate keys in LDS (32 KiB)
constexpr int kAdcMaxK = 1024;

__global__ void pq_retile_kernel(const uint8_t *__restrict__ codes, int64_t n, int m, int groups,
                                 int64_t n_tiles, uint4 *__restrict__ tiles)
{
    int64_t gid = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    int64_t total = n_tiles * groups * 64;
    if (gid >= total) return;
    int lane = static_cast<int>(gid & 63);
    int64_t tg = gid >> 6;
    int g = static_cast<int>(tg % groups);
    int64_t t = tg / groups;
    int64_t row = t * 64 + lane;
    uint32_t w[4] = {0, 0, 0, 0};
    if (row < n) {
        const uint8_t *src = codes + row * m + g * 16;
        int cnt = m - g * 16;
        if (cnt >= 16) {  // full group: slot s holds sub-quantizer (s + lane) & 15
            for (int b = 0; b < 16; b++)
                w[b >> 2] |= static_cast<uint32_t>(src[(b + lane) & 15]) << (8 * (b & 3));
        } else {          // m % 16 tail: natural order (summed sequentially)
            for (int b = 0; b < cnt; b++)
                w[b >> 2] |= static_cast<uint32_t>(src[b]) << (8 * (b & 3));
        }
    }
    tiles[gid] = make_uint4(w[0], w[1], w[2], w[3]);
}

__device__ __forceinline__ uint32_t code_byte(const uint4 &c, int l)
{
    uint32_t w = (l < 4) ? c.x : (l < 8) ? c.y : (l < 12) ? c.z : c.w;
    return (w >> (8 * (l & 3))) & 0xFFu;
}


// ---- hand-pipelined LDS gathers -------------------------------------------------------------
// hipcc waits after every few ds_read_b32 of an unrolled gather, which at 2 waves/SIMD leaves the
// LDS pipe mostly idle.  The scan issues each 16-lookup group as 16 back-to-back ds_read_b32 in
// inline asm and retires it with a COUNTED s_waitcnt while the next group's 16 reads are already
// in flight.  hipcc does not track asm loads: every value is an in/out operand of the wait
// statement, so no consumer can be scheduled above it (cdna_hip_programming.md §5.7).
struct Vals8 {
    float v[8];
};

template <int OFF>
__device__ __forceinline__ float lds_read_off(uint32_t addr)
{
    float v;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}

// wait until at most N (<= 15: lgkmcnt is 4 bits) LGKM operations are outstanding; ties the 8
// values to the wait so no consumer can move above it
template <int N>
__device__ __forceinline__ void lds_wait(Vals8 &x)
{
    asm volatile("s_waitcnt lgkmcnt(%8)"
                 : "+v"(x.v[0]), "+v"(x.v[1]), "+v"(x.v[2]), "+v"(x.v[3]), "+v"(x.v[4]),
                   "+v"(x.v[5]), "+v"(x.v[6]), "+v"(x.v[7])
                 : "n"(N));
}

// issue 8 rotated lookups (half H of group G) for code word c: slot s reads the pair-interleaved
// image at byte (G>>1)*32768 + code*128 + (G&1)*64 + rotoff[s]
template <int G, int H>
__device__ __forceinline__ void issue_half(Vals8 &dst, const uint4 &c, const uint32_t (&rotoff)[16])
{
    // ds offsets are 16-bit: groups 4+ go through a base bumped by 64 KiB
    constexpr int OFF = ((G >> 1) & 1) * 32768 + (G & 1) * 64;
    constexpr uint32_t BUMP = (G >> 2) * 65536u;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int sl = H * 8 + i;
        const uint32_t addr = (code_byte(c, sl) << 7) + rotoff[sl] + BUMP;
#ifdef VG_ADC_PROBE_NO_LDS   // stage probe (tools/build_variant.sh): the address arithmetic without the lookup
        dst.v[i] = __uint_as_float(addr + OFF);
#else
        dst.v[i] = lds_read_off<OFF>(addr);
#endif
    }
}

template <int H>
__device__ __forceinline__ void accumulate_half(float (&acc)[16], const Vals8 &x)
{
#pragma unroll
    for (int i = 0; i < 8; i++) acc[H * 8 + i] = acc[H * 8 + i] + x.v[i];
}

// words of the per-query LUT image (pair-interleaved full groups + natural tail rows)
__host__ __device__ inline int lut_image_words(int m) { return (((m >> 4) + 1) >> 1) * 8192 + (m & 15) * 256; }

__device__ __forceinline__ int64_t min64(int64_t a, int64_t b) { return a < b ? a : b; }
__device__ __forceinline__ bool span_nonempty(int64_t t0, int64_t t1) { return t1 > t0; }

struct AdcShared {
    // dynamic LDS: float lut[m*256]; uint64 buf[kAdcBuf]; then these words
    int cnt;
    int pad;
    uint64_t tau;
};

// GF = number of full 16-wide groups when known at compile time (m = 16*GF exactly),
// or -1 for the generic shape (runtime full groups + tail).
// SMALLK: k <= 64 — each wave keeps its top-k in registers (WaveTopK) and the scan loop has
// no workgroup barrier; otherwise candidates go through the shared LDS buffer.
// ONCE: one query per pass — nobody re-reads the codes, so they are streamed (load_stream); with several
// queries the blocks of an XCD share a slice through L2 and the loads stay plain.
// MASKED (SMALLK only): the filtered search of k_probe.hip — a row the query's filter rejects offers no key; desc: a Dot /
// Cosine segment keeps the LARGEST lookups (flat/segment.go:449).  A template flag: the unfiltered scans stay as they were.
// PROBED (SMALLK only): the partition-probed scan (flat/segment.go:727-744) — workgroup (query q, share s of `slices`) walks the
// tile ranges of probes s, s + slices, ... of its query with the same pipelined loop, rows outside a partition's range masked
// (partition bounds are not tile-aligned); min_keys: paged results.  (r05 ran these through a plain gather loop: 2x the time.)
template <int GF, bool SMALLK, bool ONCE, bool MASKED = false, bool PROBED = false>
__global__ __launch_bounds__(kAdcThreads) void pq_adc_scan_kernel(
    const uint4 *__restrict__ tiles, int64_t n_rows, int64_t n_tiles, int m, int groups,
    const float *__restrict__ tables, int slices, int nq, int k, uint64_t *__restrict__ partial,
    int raw_lists, const int *__restrict__ only_if, const uint8_t *__restrict__ mask, int64_t mask_stride, bool desc,
    const uint32_t *__restrict__ probes = nullptr, const uint32_t *__restrict__ part_off = nullptr, int np = 0,
    const uint64_t *__restrict__ min_keys = nullptr)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *lut = reinterpret_cast<float *>(smem);
    const int lut_words = lut_image_words(m);
    const int lut_tail_word = (((m >> 4) + 1) >> 1) * 8192;
    uint64_t *buf = reinterpret_cast<uint64_t *>(smem + static_cast<size_t>(lut_words) * sizeof(float));
    AdcShared *sh = reinterpret_cast<AdcShared *>(buf + kAdcBuf);

    // XCD-aware mapping: blocks b and b+8 share an XCD (and its L2).  Consecutive blocks of
    // one XCD take different queries over the SAME row slice, so the slice is fetched from
    // HBM once per XCD and served to the other queries from that XCD's L2.
    const int b = blockIdx.x;
    const int xcd = b & 7;
    const int o = b >> 3;
    const int q = PROBED ? b / slices : o % nq;
    const int s = PROBED ? b % slices : (o / nq) * 8 + xcd;
    if (only_if && !only_if[q]) return;  // fallback launch: only the flagged queries run
    int64_t t0 = n_tiles * s / slices;
    int64_t t1 = n_tiles * (s + 1) / slices;
    int64_t R0 = 0, R1 = n_rows;  // PROBED: the probed partition's rows
    const uint8_t *mq = MASKED && mask ? mask + static_cast<int64_t>(q) * mask_stride : nullptr;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int rot = lane & 15;
    const float *lut_rot = lut;  // full groups: [g][c][l]; tail rows follow in [j][c] order

    {   // stage this query's lookup table (m KiB) into LDS with 16-byte loads
        const float4 *src = reinterpret_cast<const float4 *>(tables + static_cast<int64_t>(q) * lut_words);
        float4 *dst = reinterpret_cast<float4 *>(lut);
        // 4 independent 16-byte loads in flight per thread.  No per-load guards: a guarded load
        // makes hipcc branch around it and wait vmcnt(0) each time (serialized L2 round trips).
        const int n4 = lut_words / 4;
        const int rounds = n4 / kAdcThreads;
        int r = 0;
        for (; r + 4 <= rounds; r += 4) {
            float4 tmp[4];
#pragma unroll
            for (int u = 0; u < 4; u++) tmp[u] = src[(r + u) * kAdcThreads + tid];
#pragma unroll
            for (int u = 0; u < 4; u++) dst[(r + u) * kAdcThreads + tid] = tmp[u];
        }
        for (int i = r * kAdcThreads + tid; i < n4; i += kAdcThreads) dst[i] = src[i];
    }
    if (tid == 0) {
        sh->cnt = 0;
        sh->tau = kKeyMax;
    }
    __syncthreads();
    uint64_t tau = kKeyMax;
    WaveTopK wtk;
    wtk.init(raw_lists ? 64 : k);  // raw mode: every wave keeps its 64 best whatever k is

    const int gfull = (GF >= 0) ? GF : (m >> 4);
    const int tail = (GF >= 0) ? 0 : (m & 15);
    // Which tiles a workgroup takes.  Several queries per pass: a contiguous slice [t0, t1), so that the blocks of
    // an XCD re-read one slice through their L2.  One query per pass (ONCE): the tiles are dealt round-robin over
    // the workgroups — trip `it` of workgroup s, wave w is tile (it * slices + s) * waves + w — so that at any
    // moment the whole chip streams ONE moving window of the code array instead of `slices` distant ones (DRAM
    // pages stay open, the stream runs at the rate of a plain sequential read).
    constexpr bool kDealt = ONCE && VG_ADC_DEAL && !PROBED;
    for (int jr = PROBED ? s : 0; jr < (PROBED ? np : 1); jr += PROBED ? slices : 1) {
    if (PROBED) {
        const uint32_t pp = probes[static_cast<int64_t>(q) * np + jr];
        R0 = part_off[pp];
        R1 = part_off[pp + 1];
        t0 = R0 >> 6;
        t1 = (R1 + 63) >> 6;
    }
    const int64_t tstride = kDealt ? static_cast<int64_t>(slices) * kAdcWaves : kAdcWaves;
    const int64_t tbase = kDealt ? static_cast<int64_t>(s) * kAdcWaves : t0;
    const int64_t tend = kDealt ? n_tiles : t1;
    const int64_t span = tend - tbase;
    const int iters = span > 0 ? static_cast<int>((span + tstride - 1) / tstride) : 0;

    // software pipeline: the next tile's 16-byte code words are in flight while this
    // tile's lookups run (GF known at compile time)
    uint4 nxt[GF > 0 ? GF : 1];
    // m = 96 single-query passes keep two tiles in flight per wave when the workgroup has 8 waves (256 VGPRs
    // each); with more waves per SIMD the same bytes are in flight one tile ahead
    constexpr bool kTwoAhead = GF == 6 && ONCE && kAdcWaves <= 8;
    uint4 nxt2[kTwoAhead ? 6 : 1];
    uint32_t rotoff[16];
#pragma unroll
    for (int sl = 0; sl < 16; sl++) rotoff[sl] = static_cast<uint32_t>(((sl + rot) & 15) * 4);
    // Prefetch loads are UNGUARDED (a guarded load makes hipcc wait vmcnt(0) at the join): past
    // the end of the slice the address is clamped to the slice's last tile and the data unused.
    const int64_t tlast = tend - 1;
    if (GF > 0 && span > 0) {
        const int64_t tile0 = min64(tbase + wave, tlast);
        const uint4 *tp0 = tiles + (tile0 * groups) * 64 + lane;
#pragma unroll
        for (int g = 0; g < GF; g++) nxt[g] = ONCE ? load_stream(tp0 + g * 64) : tp0[g * 64];
        if (kTwoAhead) {
            const int64_t tile1 = min64(tbase + wave + tstride, tlast);
            const uint4 *tp1 = tiles + (tile1 * groups) * 64 + lane;
#pragma unroll
            for (int g = 0; g < 6; g++) nxt2[g] = load_stream(tp1 + g * 64);
        }
    }
    for (int it = 0; it < iters; it++) {
        const int64_t tile = tbase + static_cast<int64_t>(it) * tstride + wave;
        if (tile < tend) {
            const uint4 *tp = tiles + (tile * groups) * 64 + lane;
            float acc[16];
#pragma unroll
            for (int l = 0; l < 16; l++) acc[l] = 0.0f;
            if (GF == 6) {
                uint4 c[6];
#pragma unroll
                for (int g = 0; g < 6; g++) c[g] = nxt[g];
                if (kTwoAhead) {  // two tiles ahead: ~96 KiB of code loads in flight per CU instead of 48
                    const int64_t tile2 = min64(tile + 2 * tstride, tlast);
                    const uint4 *tn = tiles + (tile2 * groups) * 64 + lane;
#pragma unroll
                    for (int g = 0; g < 6; g++) {
                        nxt[g] = nxt2[g];
                        nxt2[g] = load_stream(tn + g * 64);
                    }
                } else {
                    const int64_t tile1 = min64(tile + tstride, tlast);
                    const uint4 *tn = tiles + (tile1 * groups) * 64 + lane;
#pragma unroll
                    for (int g = 0; g < 6; g++) nxt[g] = ONCE ? load_stream(tn + g * 64) : tn[g * 64];
                }
#ifdef VG_ADC_PROBE_NO_HBM   // stage probe: every tile re-uses the first tile's code words (no loads in the loop)
#pragma unroll
                for (int g = 0; g < 6; g++) {
                    nxt[g] = c[g];
                    if (kTwoAhead) nxt2[g] = c[g];
                }
#endif
                // two register sets ping-pong: the next 8 lookups are in flight while the
                // previous 8 retire (same order of additions per slot: g ascending)
                Vals8 va, vb;
                issue_half<0, 0>(va, c[0], rotoff);
#define VG_STEP(GA, HA, GB, HB, X, Y)            \
    issue_half<GB, HB>(Y, c[GB], rotoff);        \
    lds_wait<8>(X);                              \
    accumulate_half<HA>(acc, X);                 \
    VG_ADC_SCHED_FENCE
                VG_STEP(0, 0, 0, 1, va, vb)
                VG_STEP(0, 1, 1, 0, vb, va)
                VG_STEP(1, 0, 1, 1, va, vb)
                VG_STEP(1, 1, 2, 0, vb, va)
                VG_STEP(2, 0, 2, 1, va, vb)
                VG_STEP(2, 1, 3, 0, vb, va)
                VG_STEP(3, 0, 3, 1, va, vb)
                VG_STEP(3, 1, 4, 0, vb, va)
                VG_STEP(4, 0, 4, 1, va, vb)
                VG_STEP(4, 1, 5, 0, vb, va)
                VG_STEP(5, 0, 5, 1, va, vb)
#undef VG_STEP
                lds_wait<0>(vb);
                accumulate_half<1>(acc, vb);
            } else if (GF >= 0) {
                uint4 c[GF > 0 ? GF : 1];
#pragma unroll
                for (int g = 0; g < GF; g++) c[g] = nxt[g];
                {
                    const int64_t tile1 = min64(tile + tstride, tlast);
                    const uint4 *tn = tiles + (tile1 * groups) * 64 + lane;
#pragma unroll
                    for (int g = 0; g < GF; g++) nxt[g] = ONCE ? load_stream(tn + g * 64) : tn[g * 64];
                }
#pragma unroll
                for (int g = 0; g < GF; g++) {
#pragma unroll
                    for (int sl = 0; sl < 16; sl++)
                        acc[sl] = acc[sl] + lut_rot[((g >> 1) * 256 + code_byte(c[g], sl)) * 32 + (g & 1) * 16 + ((sl + rot) & 15)];
                }
            } else {
                for (int g = 0; g < gfull; g++) {
                    uint4 c = tp[g * 64];
#pragma unroll
                    for (int sl = 0; sl < 16; sl++)
                        acc[sl] = acc[sl] + lut_rot[((g >> 1) * 256 + code_byte(c, sl)) * 32 + (g & 1) * 16 + ((sl + rot) & 15)];
                }
            }
            float total = reduce16_regs(acc);
            if (tail) {
                uint4 c = tp[gfull * 64];
                for (int l = 0; l < tail; l++)
                    total = total + lut[lut_tail_word + l * 256 + code_byte(c, l)];
            }
            const int64_t row = tile * 64 + lane;
            uint64_t key;
            if constexpr (MASKED || PROBED) {
                const bool live = row < n_rows && (!PROBED || (row >= R0 && row < R1)) && (!MASKED || mask_bit(mq, row));
                key = live ? make_key(total, static_cast<uint32_t>(row), desc) : kKeyMax;
                if (PROBED && min_keys && key <= min_keys[q]) key = kKeyMax;  // paged results (k > 64)
            } else {
                key = (row < n_rows) ? make_key(total, static_cast<uint32_t>(row), false) : kKeyMax;
            }
            if (SMALLK) {
                wtk.offer(key, lane);
            } else if (key < tau) {
                int pos = atomicAdd(&sh->cnt, 1);
                buf[pos] = key;  // pos < kAdcBuf by the sync protocol below
            }
        }
        if (SMALLK) continue;
        // Every kAdcSyncEvery iterations at most kAdcSyncEvery*1024 keys were appended;
        // compact when the next round could overflow the buffer.
        if ((it % kAdcSyncEvery) == kAdcSyncEvery - 1 || it == iters - 1) {
            __syncthreads();
            int c = sh->cnt;
            const bool last = (it == iters - 1);
            if (last || c > kAdcBuf - kAdcSyncEvery * kAdcThreads) {
                for (int i = c + tid; i < kAdcBuf; i += kAdcThreads) buf[i] = kKeyMax;
                __syncthreads();
                bitonic_sort_lds(buf, kAdcBuf, tid, kAdcThreads);
                if (tid == 0) {
                    int keep = c < k ? c : k;
                    sh->cnt = keep;
                    sh->tau = (c >= k) ? buf[k - 1] : kKeyMax;
                }
                __syncthreads();
            }
            tau = sh->tau;
        }
    }
    }  // probed ranges
    if (SMALLK && raw_lists) {
        partial[((static_cast<int64_t>(q) * slices + s) * kAdcWaves + wave) * 64 + lane] = wtk.list;
        return;
    }
    uint64_t *out = partial + (static_cast<int64_t>(q) * slices + s) * k;
    if (SMALLK) {
        wg_rank_merge<kAdcWaves>(wtk, buf, reinterpret_cast<int *>(buf + kAdcWaves * 64), wave, lane,
                                 tid, k, out);
        return;
    }
    const int have = sh->cnt;
    for (int i = tid; i < k; i += kAdcThreads) out[i] = (i < have) ? buf[i] : kKeyMax;
}

// (r01/r02 carried a second m = 96 kernel here, `pq_adc_scan96_kernel`: 16-lane halves of a wave one group apart so
// that no two lanes of a 32-lane LDS group share a bank — 25 lookups per clock and CU in the microbenchmark against
// 15 for the rotated image below.  Bit-exact, but its lane-dependent address selects cost ~7 vector instructions per
// lookup and it never won: 70 k vs 82 k queries/s at 1M rows (r01), and on the LDS-bound batch of r03 — 64 queries x
// 10M rows — 7.0 k vs 8.2 k queries/s.  The one-query scan is bound by the HBM stream, not by LDS (r02 stage probes,
// DESIGN.md section 4), so there was nothing for it to win there either.  Dropped in r03.)

// One workgroup per query: merges `lists` ascending key lists of length k (kKeyMax padded)
// into the k best keys, best first, and decodes them to (id, score).
// Pruning: the k-th key of any single list bounds the global k-th key from above, and so does
// the k-th smallest list head; only keys <= T = min of the two bounds can make the result.
constexpr int kMergeThreads = 256;
constexpr int kMergeBuf = 4096;
__global__ __launch_bounds__(kMergeThreads) void topk_merge_kernel(
    const uint64_t *__restrict__ partial, int lists, int k, bool descending,
    uint32_t *__restrict__ ids, float *__restrict__ scores, const int *__restrict__ only_if,
    const int *__restrict__ always)
{
    __shared__ uint64_t buf[kMergeBuf];
    __shared__ unsigned long long tmin;
    __shared__ int cnt;
    const int q = blockIdx.x;
    if (only_if && !only_if[q] && !(always && always[0])) return;  // query not selected
    const int tid = threadIdx.x;
    const uint64_t *src = partial + static_cast<int64_t>(q) * lists * k;
    const int64_t total = static_cast<int64_t>(lists) * k;
    uint32_t *oid = ids + static_cast<int64_t>(q) * k;
    float *osc = scores + static_cast<int64_t>(q) * k;
    if (tid == 0) {
        tmin = kKeyMax;
        cnt = 0;
    }
    __syncthreads();
    uint64_t t = kKeyMax;
    for (int l = tid; l < lists; l += kMergeThreads) {
        uint64_t v = src[static_cast<int64_t>(l) * k + (k - 1)];
        t = v < t ? v : t;
    }
    if (t != kKeyMax) atomicMin(&tmin, static_cast<unsigned long long>(t));
    // Second bound: the heads of k different lists are k different keys, so the k-th smallest HEAD
    // is >= the global k-th key too.  With many short lists it is far tighter than the first
    // (1024 lists of 10: ~2.5 survivors per list under T1, ~1.5k in total under the head bound).
    if (lists >= k && lists <= kMergeBuf && k >= 1) {
        int n2 = 1;
        while (n2 < lists) n2 <<= 1;
        for (int l = tid; l < n2; l += kMergeThreads) buf[l] = l < lists ? src[static_cast<int64_t>(l) * k] : kKeyMax;
        __syncthreads();
        bitonic_sort_lds(buf, n2, tid, kMergeThreads);
        const uint64_t hk = buf[k - 1];
        __syncthreads();  // everyone has read buf[k-1] before the buffer is reused below
        if (tid == 0 && hk != kKeyMax) atomicMin(&tmin, static_cast<unsigned long long>(hk));
    }
    __syncthreads();
    const uint64_t T = tmin;
    // lists are ascending: walk each from its head while keys stay <= T (usually 0-2 steps)
    for (int l = tid; l < lists; l += kMergeThreads) {
        const uint64_t *lp = src + static_cast<int64_t>(l) * k;
        for (int i = 0; i < k; i++) {
            const uint64_t key = lp[i];
            if (key == kKeyMax || key > T) break;
            int pos = atomicAdd(&cnt, 1);
            if (pos < kMergeBuf) buf[pos] = key;
        }
    }
    __syncthreads();
    const int c = cnt;
    int have;
    if (c <= 1024) {
        // brute-force rank among the survivors (keys are unique)
        for (int i = tid; i < c; i += kMergeThreads) {
            const uint64_t e = buf[i];
            int rank = 0;
            for (int j = 0; j < c; j++) rank += (buf[j] < e) ? 1 : 0;
            if (rank < k) {
                oid[rank] = key_row(e);
                osc[rank] = key_score(e, descending);
            }
        }
        have = c < k ? c : k;
    } else {
        // many survivors (heavy ties / adversarial order): chunked sort of everything
        int kept = 0;
        int64_t pos = 0;
        __syncthreads();
        do {
            int room = kMergeBuf - kept;
            int take = static_cast<int>((total - pos) < room ? (total - pos) : room);
            for (int i = tid; i < take; i += kMergeThreads) buf[kept + i] = src[pos + i];
            for (int i = kept + take + tid; i < kMergeBuf; i += kMergeThreads) buf[i] = kKeyMax;
            __syncthreads();
            bitonic_sort_lds(buf, kMergeBuf, tid, kMergeThreads);
            pos += take;
            kept = k;  // k <= kAdcMaxK < kMergeBuf; slots past the real keys hold kKeyMax
        } while (pos < total);
        have = 0;
        for (int i = tid; i < k; i += kMergeThreads) {
            const uint64_t e = buf[i];
            if (e != kKeyMax) {
                oid[i] = key_row(e);
                osc[i] = key_score(e, descending);
            }
        }
        // count real keys among the first k (uniform across threads)
        int lo = 0, hi = k;
        while (lo < hi) {
            int mid = (lo + hi) >> 1;
            if (buf[mid] != kKeyMax) lo = mid + 1; else hi = mid;
        }
        have = lo;
    }
    for (int i = have + tid; i < k; i += kMergeThreads) {
        oid[i] = VG_INVALID_ID;
        osc[i] = descending ? -INFINITY : INFINITY;
    }
}

// k > 64: per query, the k best keys of the union of `lists` per-wave lists (64 keys each,
// ascending, kKeyMax padded) and a proof that nothing better was dropped: a wave only ever
// discards keys above its own 64th key, so the result is exact iff the k-th selected key is not
// above T = min over FULL lists of their 64th key.  flag[q] = 1 asks for the exhaustive path.
__global__ __launch_bounds__(kMergeThreads) void topk_select_verify_kernel(
    const uint64_t *__restrict__ raw, int lists, int k, bool descending, uint32_t *__restrict__ ids,
    float *__restrict__ scores, int *__restrict__ flags)
{
    __shared__ uint64_t buf[kMergeBuf];
    __shared__ unsigned long long tsafe;
    const int64_t q = blockIdx.x;
    const int tid = threadIdx.x;
    const uint64_t *src = raw + q * lists * 64;
    const int64_t total = static_cast<int64_t>(lists) * 64;
    if (tid == 0) tsafe = kKeyMax;
    __syncthreads();
    uint64_t t = kKeyMax;
    for (int l = tid; l < lists; l += kMergeThreads) {
        const uint64_t v = src[static_cast<int64_t>(l) * 64 + 63];
        t = v < t ? v : t;
    }
    if (t != kKeyMax) atomicMin(&tsafe, static_cast<unsigned long long>(t));
    __shared__ int cnt;
    __shared__ unsigned long long tcut_s;
    bool done = false;
    if (total > kMergeBuf && lists <= kMergeBuf) {
        // Prune before sorting.  An upper bound on the k-th smallest key of the union:
        //   lists >= k : the k-th smallest list HEAD (k distinct keys are <= it)
        //   otherwise  : max_l list_l[j-1], j = ceil(k / lists) (the first j of every list are k keys)
        uint64_t tcut;
        if (lists >= k) {
            for (int i = tid; i < kMergeBuf; i += kMergeThreads)
                buf[i] = i < lists ? src[static_cast<int64_t>(i) * 64] : kKeyMax;
            __syncthreads();
            bitonic_sort_lds(buf, kMergeBuf, tid, kMergeThreads);
            tcut = buf[k - 1];
            __syncthreads();
        } else {
            const int j = (k + lists - 1) / lists;  // <= 64 because k <= lists * 64 is checked by the host
            if (tid == 0) tcut_s = 0;
            __syncthreads();
            uint64_t mx = 0;
            for (int l = tid; l < lists; l += kMergeThreads) {
                const uint64_t v = src[static_cast<int64_t>(l) * 64 + (j - 1)];
                mx = v > mx ? v : mx;
            }
            atomicMax(&tcut_s, static_cast<unsigned long long>(mx));
            __syncthreads();
            tcut = tcut_s;
        }
        if (tid == 0) cnt = 0;
        __syncthreads();
        for (int l = tid; l < lists; l += kMergeThreads) {  // lists are ascending: stop at the cut
            const uint64_t *lp = src + static_cast<int64_t>(l) * 64;
            for (int i = 0; i < 64; i++) {
                const uint64_t key = lp[i];
                if (key == kKeyMax || key > tcut) break;
                const int p = atomicAdd(&cnt, 1);
                if (p < kMergeBuf) buf[p] = key;
            }
        }
        __syncthreads();
        const int c = cnt;
        if (c <= kMergeBuf) {
            for (int i = c + tid; i < kMergeBuf; i += kMergeThreads) buf[i] = kKeyMax;
            __syncthreads();
            bitonic_sort_lds(buf, kMergeBuf, tid, kMergeThreads);
            done = true;
        }
        __syncthreads();
    }
    if (!done) {
        int kept = 0;
        int64_t pos = 0;
        do {
            const int room = kMergeBuf - kept;
            const int take = static_cast<int>((total - pos) < room ? (total - pos) : room);
            for (int i = tid; i < take; i += kMergeThreads) buf[kept + i] = src[pos + i];
            for (int i = kept + take + tid; i < kMergeBuf; i += kMergeThreads) buf[i] = kKeyMax;
            __syncthreads();
            bitonic_sort_lds(buf, kMergeBuf, tid, kMergeThreads);
            pos += take;
            kept = k;
        } while (pos < total);
    }
    const uint64_t T = tsafe;
    const uint64_t kth = buf[k - 1];
    if (tid == 0) flags[q] = (T == kKeyMax || (kth != kKeyMax && kth <= T)) ? 0 : 1;
    for (int i = tid; i < k; i += kMergeThreads) {
        const uint64_t e = buf[i];
        ids[q * k + i] = e == kKeyMax ? VG_INVALID_ID : key_row(e);
        scores[q * k + i] = e == kKeyMax ? (descending ? -INFINITY : INFINITY) : key_score(e, descending);
    }
}

__global__ void patch_results_kernel(const int *__restrict__ flags, int k, const uint32_t *__restrict__ fids,
                                     const float *__restrict__ fscores, uint32_t *__restrict__ ids,
                                     float *__restrict__ scores)
{
    const int64_t q = blockIdx.x;
    if (!flags[q]) return;
    for (int i = threadIdx.x; i < k; i += blockDim.x) {
        ids[q * k + i] = fids[q * k + i];
        scores[q * k + i] = fscores[q * k + i];
    }
}

int32_t launch_topk_merge(const uint64_t *partial, int64_t nq, int lists, int k, bool descending,
                          uint32_t *ids, float *scores, hipStream_t st, const int *only_if = nullptr,
                          const int *always = nullptr)
{
    if (nq == 0 || k == 0) return VG_OK;
    VG_LAUNCH(topk_merge_kernel, dim3(static_cast<unsigned>(nq)), dim3(kMergeThreads), 0,
                       st, partial, lists, k, descending, ids, scores, only_if, always);
    return VG_OK;
}

int32_t launch_pq_build_table(const vg_pq *pq, const float *d_queries, int64_t nq,
                              float *d_tables, bool scan_layout, hipStream_t st);

// ---- partition-probed ADC scan (flat/segment.go:727-744 over the :678-689 branch) -----------------
// One workgroup per (query, share of its probe list): the query's table is staged once, then the
// tiles covering each probed partition's row range [R0, R1) are scanned with the rows outside the
// range masked (partition bounds are not tile-aligned).  A few thousand rows per partition: the
// plain gather loop, no hand pipelining.  k <= 64.
__global__ __launch_bounds__(kAdcThreads) void pq_adc_probe_kernel(
    const uint4 *__restrict__ tiles, int64_t n_rows, int m, int groups, const float *__restrict__ tables,
    const uint32_t *__restrict__ probes, const uint32_t *__restrict__ part_off, int np, int split, int k,
    uint64_t *__restrict__ partial, const uint64_t *__restrict__ min_keys, bool desc, const uint8_t *__restrict__ mask,
    int64_t mask_stride)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *lut = reinterpret_cast<float *>(smem);
    const int lut_words = lut_image_words(m);
    const int lut_tail_word = (((m >> 4) + 1) >> 1) * 8192;
    uint64_t *buf = reinterpret_cast<uint64_t *>(smem + static_cast<size_t>(lut_words) * sizeof(float));
    const int y = blockIdx.x;
    const int64_t q = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rot = lane & 15;
    {
        const float4 *src = reinterpret_cast<const float4 *>(tables + q * lut_words);
        float4 *dst = reinterpret_cast<float4 *>(lut);
        const int n4 = lut_words / 4;
        const int rounds = n4 / kAdcThreads;
        int r = 0;
        for (; r + 4 <= rounds; r += 4) {
            float4 tmp[4];
#pragma unroll
            for (int u = 0; u < 4; u++) tmp[u] = src[(r + u) * kAdcThreads + tid];
#pragma unroll
            for (int u = 0; u < 4; u++) dst[(r + u) * kAdcThreads + tid] = tmp[u];
        }
        for (int i = r * kAdcThreads + tid; i < n4; i += kAdcThreads) dst[i] = src[i];
    }
    __syncthreads();
    const int gfull = m >> 4, tail = m & 15;
    const uint8_t *mq = mask ? mask + q * mask_stride : nullptr;  // filter.Matches (segment.go:631-635)
    WaveTopK wtk;
    wtk.init(k);
    for (int j = y; j < np; j += split) {
        const uint32_t p = probes[q * np + j];
        const int64_t R0 = part_off[p], R1 = part_off[p + 1];
        const int64_t tt0 = R0 >> 6, tt1 = (R1 + 63) >> 6;
        for (int64_t tile = tt0 + wave; tile < tt1; tile += kAdcWaves) {
            const uint4 *tp = tiles + (tile * groups) * 64 + lane;
            const int64_t row = tile * 64 + lane;
            const bool live = row >= R0 && row < R1 && row < n_rows && mask_bit(mq, row);
            if (mq && !__any(live)) continue;  // a tile the filter leaves nothing of: its codes are not read
            float acc[16];
#pragma unroll
            for (int l = 0; l < 16; l++) acc[l] = 0.0f;
            for (int g = 0; g < gfull; g++) {
                const uint4 c = tp[g * 64];
#pragma unroll
                for (int sl = 0; sl < 16; sl++)
                    acc[sl] = acc[sl] + lut[((g >> 1) * 256 + code_byte(c, sl)) * 32 + (g & 1) * 16 + ((sl + rot) & 15)];
            }
            float total = reduce16_regs(acc);
            if (tail) {
                const uint4 c = tp[gfull * 64];
                for (int l = 0; l < tail; l++) total = total + lut[lut_tail_word + l * 256 + code_byte(c, l)];
            }
            uint64_t key = live ? make_key(total, static_cast<uint32_t>(row), desc) : kKeyMax;
            if (min_keys && key <= min_keys[q]) key = kKeyMax;  // paged results (k > 64)
            wtk.offer(key, lane);
        }
    }
    wg_rank_merge<kAdcWaves>(wtk, buf, reinterpret_cast<int *>(buf + kAdcWaves * 64), wave, lane, tid, k,
                             partial + (q * split + y) * k);
}

int32_t launch_probe_scan_adc(const vg_index *idx, const float *tables, const uint32_t *probes, const uint32_t *part_off,
                              int64_t nq, int np, int split, int k, uint64_t *partial, const uint64_t *min_keys, bool desc,
                              const uint8_t *mask, int64_t mask_stride, hipStream_t st)
{
    const vg_pq *pq = idx->pq;
    // a table that fits LDS next to the scan kernel's buffers, k <= 64: the pipelined scan kernel over the probed ranges
    const size_t scan_lds = static_cast<size_t>(lut_image_words(pq->m)) * sizeof(float) + kAdcBuf * sizeof(uint64_t) + sizeof(AdcShared);
    if (k <= 64 && scan_lds <= 160 * 1024 && !hook(kHookProbeNoGroup)) {
        auto pick = [&](auto masked_tag) {
            constexpr bool M = decltype(masked_tag)::value;
            return pq->m == 96 ? pq_adc_scan_kernel<6, true, false, M, true> : pq_adc_scan_kernel<-1, true, false, M, true>;
        };
        auto kern = mask ? pick(std::true_type{}) : pick(std::false_type{});
        VG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   static_cast<int>(scan_lds)));
        const int64_t max_q = (int64_t(1) << 30) / split;
        for (int64_t q0 = 0; q0 < nq; q0 += max_q) {
            const int64_t cnt = nq - q0 < max_q ? nq - q0 : max_q;
            ProfScope prof(idx->ctx, "pq_adc_probe", st);
            VG_LAUNCH(kern, dim3(static_cast<unsigned>(cnt * split)), dim3(kAdcThreads), scan_lds, st,
                      reinterpret_cast<const uint4 *>(idx->d_pq_tiles), idx->n, idx->n_tiles, pq->m, idx->pq_groups,
                      tables + q0 * lut_image_words(pq->m), split, static_cast<int>(cnt), k, partial + q0 * split * k, 0, nullptr,
                      mask ? mask + q0 * mask_stride : nullptr, mask_stride, desc, probes + q0 * np, part_off, np,
                      min_keys ? min_keys + q0 : nullptr);
        }
        return VG_OK;
    }
    const size_t lds = static_cast<size_t>(lut_image_words(pq->m)) * sizeof(float) + kAdcWaves * 64 * sizeof(uint64_t) + 64;
    VG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(pq_adc_probe_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    for (int64_t q0 = 0; q0 < nq; q0 += 65535) {
        const int64_t cnt = nq - q0 < 65535 ? nq - q0 : 65535;
        ProfScope prof(idx->ctx, "pq_adc_probe", st);
        VG_LAUNCH(pq_adc_probe_kernel, dim3(static_cast<unsigned>(split), static_cast<unsigned>(cnt)), dim3(kAdcThreads),
                  lds, st, reinterpret_cast<const uint4 *>(idx->d_pq_tiles), idx->n, pq->m, idx->pq_groups,
                  tables + q0 * lut_image_words(pq->m), probes + q0 * np, part_off, np, split, k,
                  partial + q0 * split * k, min_keys ? min_keys + q0 : nullptr, desc, mask ? mask + q0 * mask_stride : nullptr,
                  mask_stride);
    }
    return VG_OK;
}

static int adc_slices(int64_t nq, int64_t n_tiles, int cus)
{
    // smallest multiple of 8 (one group per XCD) with slices*nq >= #CUs, at least one
    // workgroup-iteration of tiles per slice
    int64_t s = (cus + nq - 1) / nq;
    s = ((s + 7) / 8) * 8;
    int64_t max_s = (n_tiles + kAdcWaves - 1) / kAdcWaves;
    max_s = (max_s / 8) * 8;
    if (max_s < 8) max_s = 8;
    if (s > max_s) s = max_s;
    if (s < 8) s = 8;
    return static_cast<int>(s);
}

template <int GF, bool SMALLK>
static int32_t launch_scan(const vg_index *idx, const float *tables, int64_t nq, int k,
                           int slices, uint64_t *partial, hipStream_t st, int raw_lists = 0,
                           const int *only_if = nullptr, const uint8_t *mask = nullptr, int64_t mask_stride = 0,
                           bool desc = false)
{
    const vg_pq *pq = idx->pq;
    size_t lds = static_cast<size_t>(lut_image_words(pq->m)) * sizeof(float) + kAdcBuf * sizeof(uint64_t) +
                 sizeof(AdcShared);
    auto kern = nq == 1 ? pq_adc_scan_kernel<GF, SMALLK, true> : pq_adc_scan_kernel<GF, SMALLK, false>;
    if (SMALLK && (mask || desc)) kern = nq == 1 ? pq_adc_scan_kernel<GF, SMALLK, true, SMALLK> : pq_adc_scan_kernel<GF, SMALLK, false, SMALLK>;
    VG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                               hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    // grid.x limit is 2^31-1; chunk the queries if needed
    const int64_t max_q = (1ll << 30) / slices;
    for (int64_t q0 = 0; q0 < nq; q0 += max_q) {
        int64_t cnt = nq - q0 < max_q ? nq - q0 : max_q;
        ProfScope prof(idx->ctx, "pq_adc_scan", st);
        VG_LAUNCH(kern, dim3(static_cast<unsigned>(cnt * slices)), dim3(kAdcThreads), lds,
                           st, reinterpret_cast<const uint4 *>(idx->d_pq_tiles), idx->n,
                           idx->n_tiles, pq->m, idx->pq_groups,
                           tables + q0 * lut_image_words(pq->m), slices, static_cast<int>(cnt), k,
                           partial + q0 * slices * (raw_lists ? kAdcWaves * 64 : k), raw_lists,
                           only_if ? only_if + q0 : nullptr, mask ? mask + q0 * mask_stride : nullptr, mask_stride, desc,
                           static_cast<const uint32_t *>(nullptr), static_cast<const uint32_t *>(nullptr), 0,
                           static_cast<const uint64_t *>(nullptr));
    }
    return VG_OK;
}

// ---------------------------------------------------------------------------------------------
// m above what one LDS image holds (fp32 table of m KiB: m > 96 with the top-k buffers; d = 1536 at 8 dims per
// sub-quantizer is m = 192).  The table is walked in CHUNKS of up to 6 full groups (96 KiB): a workgroup takes a
// batch of kWideTiles tiles per wave, keeps their 16 slot accumulators in registers, and for each chunk stages
// that chunk's table rows into LDS and adds the chunk's groups for every tile of the batch.  Every row still
// adds its groups in ascending order into the same 16 accumulators, so the sum is pqAdcLookupAvx512's.  The
// table is re-staged once per batch (m KiB from L2 against kAdcWaves * kWideTiles * 64 * m bytes of codes: a
// quarter of the code traffic at 4 tiles per wave).  k <= 64.
// ---------------------------------------------------------------------------------------------
constexpr int kWideTiles = 4;
constexpr int kWideChunkGroups = 6;
constexpr int kWideChunkWords = (kWideChunkGroups / 2) * 8192;

__global__ __launch_bounds__(kAdcThreads) void pq_adc_scan_wide_kernel(
    const uint4 *__restrict__ tiles, int64_t n_rows, int64_t n_tiles, int m, int groups,
    const float *__restrict__ tables, int slices, int nq, int k, uint64_t *__restrict__ partial)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *lut = reinterpret_cast<float *>(smem);                  // one chunk: [pair][c][32]
    float *lut_tail = lut + kWideChunkWords;                       // m % 16 natural rows [j][c]
    uint64_t *buf = reinterpret_cast<uint64_t *>(lut_tail + 15 * 256);
    const int b = blockIdx.x;
    const int xcd = b & 7, o = b >> 3;
    const int q = o % nq;
    const int s = (o / nq) * 8 + xcd;
    const int64_t t0 = n_tiles * s / slices, t1 = n_tiles * (s + 1) / slices;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, rot = lane & 15;
    const int gfull = m >> 4, tail = m & 15;
    const int nchunks = gfull ? (gfull + kWideChunkGroups - 1) / kWideChunkGroups : 1;
    const float *image = tables + static_cast<int64_t>(q) * lut_image_words(m);
    const int tail_word = ((gfull + 1) >> 1) * 8192;
    WaveTopK wtk;
    wtk.init(k);
    for (int64_t b0 = t0; b0 < t1; b0 += static_cast<int64_t>(kAdcWaves) * kWideTiles) {
        float acc[kWideTiles][16];
#pragma unroll
        for (int tb = 0; tb < kWideTiles; tb++)
#pragma unroll
            for (int l = 0; l < 16; l++) acc[tb][l] = 0.0f;
        for (int c = 0; c < nchunks; c++) {
            const int g0 = c * kWideChunkGroups;
            const int gc = gfull - g0 < kWideChunkGroups ? gfull - g0 : kWideChunkGroups;
            __syncthreads();  // the previous chunk's lookups are done
            {
                const int words = ((gc + 1) >> 1) * 8192;
                const float4 *src = reinterpret_cast<const float4 *>(image + (g0 >> 1) * 8192);
                float4 *dst = reinterpret_cast<float4 *>(lut);
                for (int i = tid; i < words / 4; i += kAdcThreads) dst[i] = src[i];
                if (c == nchunks - 1 && tail) {
                    const float4 *ts = reinterpret_cast<const float4 *>(image + tail_word);
                    float4 *td = reinterpret_cast<float4 *>(lut_tail);
                    for (int i = tid; i < tail * 64; i += kAdcThreads) td[i] = ts[i];
                }
            }
            __syncthreads();
#pragma unroll
            for (int tb = 0; tb < kWideTiles; tb++) {
                const int64_t tile = b0 + static_cast<int64_t>(tb) * kAdcWaves + wave;
                if (tile < t1) {
                    const uint4 *tp = tiles + (tile * groups) * 64 + lane;
                    for (int g = 0; g < gc; g++) {
                        const uint4 cw = tp[(g0 + g) * 64];
#pragma unroll
                        for (int sl = 0; sl < 16; sl++)
                            acc[tb][sl] = acc[tb][sl] + lut[((g >> 1) * 256 + code_byte(cw, sl)) * 32 + (g & 1) * 16 + ((sl + rot) & 15)];
                    }
                }
            }
        }
#pragma unroll
        for (int tb = 0; tb < kWideTiles; tb++) {
            const int64_t tile = b0 + static_cast<int64_t>(tb) * kAdcWaves + wave;
            uint64_t key = kKeyMax;
            if (tile < t1) {
                float total = reduce16_regs(acc[tb]);
                if (tail) {
                    const uint4 cw = tiles[(tile * groups + gfull) * 64 + lane];
                    for (int l = 0; l < tail; l++) total = total + lut_tail[l * 256 + code_byte(cw, l)];
                }
                const int64_t row = tile * 64 + lane;
                if (row < n_rows) key = make_key(total, static_cast<uint32_t>(row), false);
            }
            wtk.offer(key, lane);
        }
    }
    __syncthreads();
    uint64_t *out = partial + (static_cast<int64_t>(q) * slices + s) * k;
    wg_rank_merge<kAdcWaves>(wtk, buf, reinterpret_cast<int *>(buf + kAdcWaves * 64), wave, lane, tid, k, out);
}

static int32_t launch_scan_wide(const vg_index *idx, const float *tables, int64_t nq, int k, int slices,
                                uint64_t *partial, hipStream_t st)
{
    const size_t lds = (kWideChunkWords + 15 * 256) * sizeof(float) + kAdcWaves * 64 * sizeof(uint64_t) + 64;
    VG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(pq_adc_scan_wide_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    const int64_t max_q = (1ll << 30) / slices;
    for (int64_t q0 = 0; q0 < nq; q0 += max_q) {
        const int64_t cnt = nq - q0 < max_q ? nq - q0 : max_q;
        ProfScope prof(idx->ctx, "pq_adc_scan", st);
        VG_LAUNCH(pq_adc_scan_wide_kernel, dim3(static_cast<unsigned>(cnt * slices)), dim3(kAdcThreads), lds, st,
                  reinterpret_cast<const uint4 *>(idx->d_pq_tiles), idx->n, idx->n_tiles, idx->pq->m, idx->pq_groups,
                  tables + q0 * lut_image_words(idx->pq->m), slices, static_cast<int>(cnt), k, partial + q0 * slices * k);
    }
    return VG_OK;
}

}  // namespace vg

namespace vg {
// (id, score) lists [lists][nq][k] -> keys [nq][lists][k] (ascending per list is preserved)
// list_stride: elements between the starts of two lists in ids[] / scores[] (nq*k when the lists are dense;
// 2*nq*k for the all-gathered [list][ids | score bits][nq][k] image of vg_comm.hip)
__global__ void pack_keys_kernel(const uint32_t *__restrict__ ids, const float *__restrict__ scores,
                                 int lists, int64_t nq, int k, int64_t list_stride, bool descending,
                                 const uint32_t *__restrict__ offsets, uint64_t *__restrict__ keys)
{
    const int64_t gid = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const int64_t total = static_cast<int64_t>(lists) * nq * k;
    if (gid >= total) return;
    const int i = static_cast<int>(gid % k);
    const int64_t q = (gid / k) % nq;
    const int l = static_cast<int>(gid / (static_cast<int64_t>(k) * nq));
    const int64_t src = l * list_stride + q * k + i;
    const uint32_t id = ids[src];
    uint64_t key = kKeyMax;
    if (id != VG_INVALID_ID) key = make_key(scores[src], id + (offsets ? offsets[l] : 0u), descending);
    keys[(q * lists + l) * k + i] = key;
}
}  // namespace vg

namespace vg {
// The engine's fan-in when a list holds a NaN score (include/vecgo_hip.h "NaN scores"): engine/search.go:904-908 pushes every
// segment's candidates — in the order they were popped from the segment's heap, worst first, i.e. each list here from its last
// valid entry to its first — into ONE CandidateHeap with TryPushBounded(k), then pops it (:915-918).  The key merge above is that
// outcome while no score is a NaN; a query with one is replayed here by one wave and overwrites the key merge's answer.  Ids are
// the offset (global) ones: offsets ascend with the segment, so (SegmentID, RowID) orders like the global id.
__global__ __launch_bounds__(64) void merge_nan_replay_kernel(const uint32_t *__restrict__ ids_in, const float *__restrict__ scores_in, int lists,
                                                              int64_t nq, int k, int64_t list_stride, bool desc,
                                                              const uint32_t *__restrict__ offsets, uint32_t *__restrict__ ids,
                                                              float *__restrict__ scores)
{
    extern __shared__ uint64_t merge_lds[];
    CItem *heap = reinterpret_cast<CItem *>(merge_lds);
    const int64_t q = blockIdx.x;
    const int lane = threadIdx.x;
    bool nan = false;
    for (int t = lane; t < lists * k; t += 64) {
        const int64_t src = static_cast<int64_t>(t / k) * list_stride + q * k + t % k;
        if (ids_in[src] != VG_INVALID_ID) {
            const float v = scores_in[src];
            nan = nan || v != v;
        }
    }
    if (!__any(nan)) return;
    int len = 0;
    for (int l = 0; l < lists; l++) {
        const int64_t base = static_cast<int64_t>(l) * list_stride + q * k;
        int cnt = 0;  // valid entries are a prefix of the list
        for (int i0 = 0; i0 < k; i0 += 64) {
            const bool valid = i0 + lane < k && ids_in[base + i0 + lane] != VG_INVALID_ID;
            cnt += __popcll(__ballot(valid));
        }
        const uint32_t off = offsets ? offsets[l] : 0u;
        for (int i = cnt - 1; i >= 0; i--) {
            const CItem x{scores_in[base + i], ids_in[base + i] + off};
            if (len < k) {
                cand_up(heap, len, x, desc);
                len++;
            } else if (cand_better(x, cand_load(heap, 0), desc)) {
                cand_down(heap, 0, len, x, desc);
            }
        }
    }
    const int nres = len;
    for (int i = nres - 1; i >= 0; i--) {
        const CItem it = cand_pop(heap, len, desc);
        if (lane == 0) {
            ids[q * k + i] = it.row;
            scores[q * k + i] = it.score;
        }
    }
    for (int i = nres + lane; i < k; i += 64) {
        ids[q * k + i] = VG_INVALID_ID;
        scores[q * k + i] = desc ? -INFINITY : INFINITY;
    }
}
}  // namespace vg

static int32_t merge_topk_impl(vg_ctx *ctx, const uint32_t *ids_in, const float *scores_in, int64_t list_stride,
                               int32_t lists, int64_t nq, int32_t k, int32_t metric, const uint32_t *id_offsets,
                               uint32_t *ids, float *scores, void *stream);

VG_API int32_t vg_merge_topk(vg_ctx *ctx, const uint32_t *ids_in, const float *scores_in, int32_t lists,
                             int64_t nq, int32_t k, int32_t metric, const uint32_t *id_offsets,
                             uint32_t *ids, float *scores, void *stream)
{
    return merge_topk_impl(ctx, ids_in, scores_in, nq * k, lists, nq, k, metric, id_offsets, ids, scores, stream);
}

VG_API int32_t vg_merge_topk_packed(vg_ctx *ctx, const uint32_t *packed, int32_t lists, int64_t nq, int32_t k,
                                    int32_t metric, const uint32_t *id_offsets, uint32_t *ids, float *scores,
                                    void *stream)
{
    VG_CHECK(lists == 0 || nq == 0 || k == 0 || (packed && vg::is_device_ptr(packed)), VG_ERR_INVALID_ARG,
             "vg_merge_topk_packed: packed must be device memory");
    return merge_topk_impl(ctx, packed, reinterpret_cast<const float *>(packed) + nq * k, 2 * nq * k, lists, nq, k,
                           metric, id_offsets, ids, scores, stream);
}

static int32_t merge_topk_impl(vg_ctx *ctx, const uint32_t *ids_in, const float *scores_in, int64_t list_stride,
                               int32_t lists, int64_t nq, int32_t k, int32_t metric, const uint32_t *id_offsets,
                               uint32_t *ids, float *scores, void *stream)
{
    VG_CHECK(ctx, VG_ERR_INVALID_ARG, "vg_merge_topk: ctx is NULL");
    VG_CHECK(lists >= 0 && nq >= 0 && k >= 0, VG_ERR_INVALID_ARG, "vg_merge_topk: negative count");
    if (nq == 0 || k == 0) return VG_OK;
    VG_CHECK(ids && scores, VG_ERR_INVALID_ARG, "vg_merge_topk: NULL output");
    VG_CHECK(lists == 0 || (ids_in && scores_in), VG_ERR_INVALID_ARG, "vg_merge_topk: NULL input");
    VG_CHECK(k <= vg::kAdcMaxK, VG_ERR_UNSUPPORTED, "vg_merge_topk: k=%d exceeds %d", k, vg::kAdcMaxK);
    VG_CHECK(metric >= VG_METRIC_L2 && metric <= VG_METRIC_DOT, VG_ERR_UNSUPPORTED,
             "vg_merge_topk: unsupported metric %d", metric);
    VG_HIP(hipSetDevice(ctx->device));
    hipStream_t st = vg::pick_stream(ctx, stream);
    const bool desc = metric != VG_METRIC_L2;
    const size_t total = static_cast<size_t>(lists) * nq * k;
    const bool dense = list_stride == nq * k;  // the packed image is device memory already (checked by the caller)
    vg::DevIn<uint32_t> i_in, offs;
    vg::DevIn<float> s_in;
    vg::DevOut<uint32_t> oid;
    vg::DevOut<float> osc;
    if (dense) {
        VG_TRY(i_in.init(ids_in, total, st));
        VG_TRY(s_in.init(scores_in, total, st));
    } else {
        i_in.ptr = ids_in;
        s_in.ptr = scores_in;
    }
    VG_TRY(offs.init(id_offsets, id_offsets ? static_cast<size_t>(lists) : 0, st));
    VG_TRY(oid.init(ids, static_cast<size_t>(nq) * k, st));
    VG_TRY(osc.init(scores, static_cast<size_t>(nq) * k, st));
    const int nl = lists > 0 ? lists : 1;
    vg::ArenaCall ar(ctx, st);
    const int i_keys = ar.add(sizeof(uint64_t) * static_cast<size_t>(nl) * nq * k);
    VG_TRY(ar.commit());
    struct { uint64_t *ptr; } keys{ar.get<uint64_t>(i_keys)};
    if (lists == 0) {
        VG_HIP(hipMemsetAsync(keys.ptr, 0xFF, static_cast<size_t>(nq) * k * 8, st));
    } else {
        VG_LAUNCH(vg::pack_keys_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256),
                           0, st, i_in.ptr, s_in.ptr, lists, nq, k, list_stride, desc, offs.ptr, keys.ptr);
    }
    VG_TRY(vg::launch_topk_merge(keys.ptr, nq, nl, k, desc, oid.ptr, osc.ptr, st));
    if (lists > 0 && !vg::hook(vg::kHookNoCandReplay))  // queries with a NaN score in a list: the engine's heap, operation by operation
        VG_LAUNCH(vg::merge_nan_replay_kernel, dim3(static_cast<unsigned>(nq)), dim3(64), sizeof(uint64_t) * (static_cast<size_t>(k) + 4), st,
                  i_in.ptr, s_in.ptr, lists, nq, k, list_stride, desc, offs.ptr, oid.ptr, osc.ptr);
    VG_TRY(oid.finish());
    VG_TRY(osc.finish());
    if (oid.on_host() || osc.on_host()) VG_HIP(hipStreamSynchronize(st));
    return VG_OK;
}

VG_API int32_t vg_index_set_pq_codes(vg_index *idx, vg_pq *pq, const uint8_t *codes, void *stream)
{
    VG_CHECK(idx && pq, VG_ERR_INVALID_ARG, "vg_index_set_pq_codes: NULL handle");
    VG_CHECK(pq->trained, VG_ERR_NOT_TRAINED, "ProductQuantizer not trained");
    VG_CHECK(pq->dim == idx->dim, VG_ERR_DIM_MISMATCH, "vector dimension mismatch");
    VG_CHECK(idx->n == 0 || codes, VG_ERR_INVALID_ARG, "vg_index_set_pq_codes: codes is NULL");
    VG_HIP(hipSetDevice(idx->ctx->device));
    hipStream_t st = vg::pick_stream(idx->ctx, stream);
    if (idx->d_pq_tiles) {
        VG_HIP(hipStreamSynchronize(st));
        VG_HIP(hipFree(idx->d_pq_tiles));
        idx->d_pq_tiles = nullptr;
    }
    if (idx->d_pq_bf16) {  // the old codes' nomination image (vg_index_enable_pq_nomination again after new codes)
        VG_HIP(hipStreamSynchronize(st));
        VG_HIP(hipFree(idx->d_pq_bf16));
        VG_HIP(hipFree(idx->d_pq_norms));
        VG_HIP(hipFree(idx->d_pq_norm_max));
        idx->d_pq_bf16 = nullptr;
        idx->d_pq_norms = idx->d_pq_norm_max = nullptr;
    }
    idx->pq = pq;
    idx->pq_groups = (pq->m + 15) / 16;
    idx->n_tiles = (idx->n + 63) / 64;
    if (idx->n == 0) return VG_OK;
    size_t bytes = static_cast<size_t>(idx->n_tiles) * idx->pq_groups * 64 * 16;
    VG_HIP(hipMalloc(reinterpret_cast<void **>(&idx->d_pq_tiles), bytes));
    vg::DevIn<uint8_t> in;
    VG_TRY(in.init(codes, static_cast<size_t>(idx->n) * pq->m, st));
    // row-major copy for random access by node id (Vamana search)
    if (idx->d_pq_rows) {
        VG_HIP(hipFree(idx->d_pq_rows));
        idx->d_pq_rows = nullptr;
    }
    VG_HIP(hipMalloc(reinterpret_cast<void **>(&idx->d_pq_rows), static_cast<size_t>(idx->n) * pq->m));
    VG_HIP(hipMemcpyAsync(idx->d_pq_rows, in.ptr, static_cast<size_t>(idx->n) * pq->m, hipMemcpyDeviceToDevice, st));
    int64_t total = idx->n_tiles * idx->pq_groups * 64;
    VG_LAUNCH(vg::pq_retile_kernel, dim3(static_cast<unsigned>((total + 255) / 256)),
                       dim3(256), 0, st, in.ptr, idx->n, pq->m, idx->pq_groups, idx->n_tiles,
                       reinterpret_cast<uint4 *>(idx->d_pq_tiles));
    VG_HIP(hipStreamSynchronize(st));
    return VG_OK;
}

// ---- batched search through a bfloat16 nomination (vg_index_enable_pq_nomination) ----------------------------------------------
// One table scan per query runs at the LDS gather rate: 12.5 ms for 1024 queries x 1M x m = 96.  A row's table sum IS a squared
// distance — sum_j |q_j - C_j[code_j]|^2 = |q - x^|^2 for the DECODED row x^ (pq.go:185-229; the table entries are computed from
// the same fp32 centroid values, pq.go:468-491) — so with the opt-in image (x^ rounded to bfloat16, 2 bytes per dimension) the fused
// flat search's nomination runs on it (k_flat.hip flat_nominate_bf16: threshold from a row sample, bf16 MFMA GEMM, the 64 best per
// query), and this file re-scores those with the reference's own arithmetic — BuildDistanceTable + pqAdcLookupAvx512 on the CODES
// — and proves that no row outside them can enter the k best: outside rows have GEMM score >= tau, and
// |GEMM score + |q|^2 - table sum| <= eps (bfloat16 rounding of both operands, fp32 accumulation on both sides).  A query whose
// proof fails is scanned as before.  The structure is the SQ8 batch search's (k_sq8.hip).
namespace vg {
size_t flat_nominate_bf16_scratch(int64_t cnt, int64_t n, int dim, int sel_k);
int32_t flat_nominate_bf16(vg_ctx *ctx, const uint16_t *rows_bf16, const float *norms, int64_t n, int dim, int dim_pad, const float *queries,
                           int64_t cnt, char *scratch, float *thr, int *counts, uint32_t *cand_id, float *cand_sc, hipStream_t st,
                           bool dot, const uint8_t *mask, int64_t mask_stride, int sel_k, bool pick, const uint64_t **cand_keys, int *cap,
                           const float *norm_max);
constexpr int kPqPickMaxK = 48, kPqNomMaxK = 256;  // as kSq8PickMaxK / kSq8NomMaxK
// the batches the nomination takes: 1M x 768, m = 96, k = 10: 16 queries 0.24 ms scanned / 0.39 nominated, 32: 0.46 / 0.41, 64: 0.86 /
// 0.43, 1024: 12.4 / 1.7 (tools/pq_nominate_time.py) — the scan costs ~12 us per query and 1M rows, the nomination ~0.4 ms up to
// 128 queries; 100k rows: 128 queries 0.27 / 0.31, 256: 0.50 / 0.28 — from 24M (query, row) pairs up.  Test hook VG_PQ_NOM_ALWAYS:
// every batch.
constexpr int64_t kPqNomMinPairs = 24000000;
static int pq_nominate_sel_k(int k) { return k <= kPqPickMaxK ? 8 : k <= 128 ? 16 : 32; }

// one wave per row: lane l decodes sub-quantizers l, l + 64, ... — v = float32(int8) * scale ; v = v + offset, the two rounded
// operations of the reference's dequantisation (pq.go:204-214) — rounds to bfloat16 (nearest even), writes the image row
// (dim_pad elements, zeros from dim on) and the row's norm
__global__ __launch_bounds__(256) void pq_decode_bf16_kernel(const uint8_t *__restrict__ codes, int64_t n, int m, int k, int sd,
                                                             const int8_t *__restrict__ codebooks, const float *__restrict__ scales,
                                                             const float *__restrict__ offsets, uint16_t *__restrict__ out, int dim_pad,
                                                             float *__restrict__ norms, int *__restrict__ norm_max_bits)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (row >= n) return;  // (whole waves)
    uint16_t *dst = out + row * dim_pad;
    float nrm = 0.0f;
    for (int j = lane; j < m; j += 64) {
        const int c = codes[row * m + j];
        const int8_t *cb = codebooks + (static_cast<int64_t>(j) * k + c) * sd;
        const float sc = scales[j], of = offsets[j];
        for (int d = 0; d < sd; d++) {
            float v = static_cast<float>(cb[d]) * sc;
            v = v + of;
            nrm = __builtin_fmaf(v, v, nrm);
            const uint32_t b = __float_as_uint(v);
            dst[j * sd + d] = static_cast<uint16_t>((b + 0x7FFFu + ((b >> 16) & 1u)) >> 16);
        }
    }
    for (int j = m * sd + lane; j < dim_pad; j += 64) dst[j] = 0;
    for (int off = 32; off > 0; off >>= 1) nrm += __shfl_xor(nrm, off);
    if (lane == 0) {
        norms[row] = nrm;
        // a NaN norm (a NaN scale / offset) must reach norm_max: NaN -> +Inf, the proof's comparisons then fail and the scan answers
        const int nb = __float_as_int(nrm == nrm ? nrm : INFINITY);  // non-negative floats order like their bits
        if (nb > *reinterpret_cast<volatile int *>(norm_max_bits)) atomicMax(norm_max_bits, nb);  // (a wave per row: only where it raises the value)
    }
}

// pqAdcLookupAvx512 (internal/simd/src/floats_avx512.c:135-167) of one row's code against a query's table in the reference's
// own layout (m rows of 256): 16 lane sums over the full groups, the _mm512_reduce_add_ps tree, the tail added in order — what
// pq_adc_scan_kernel computes from its LDS image
__device__ __forceinline__ float pq_adc_row_score(const float *__restrict__ lut, const uint8_t *__restrict__ code, int m)
{
    float s[16];
#pragma unroll
    for (int l = 0; l < 16; l++) s[l] = 0.0f;
    int i = 0;
    for (; i + 16 <= m; i += 16) {
        float t[16];
#pragma unroll
        for (int l = 0; l < 16; l++) t[l] = lut[(i + l) * 256 + code[i + l]];
#pragma unroll
        for (int l = 0; l < 16; l++) s[l] = s[l] + t[l];
    }
    float total = reduce16_regs(s);
    for (; i < m; i++) total = total + lut[i * 256 + code[i]];
    return total;
}

// the proof's margin: |s~ + |q|^2 - table sum| for any row.  bfloat16 rounding of q and x^: (2^-7 + 2^-16)(|q|^2 + |x^|^2); the GEMM's
// fp32 accumulation and the norm's: 2 dim u of the same; the table's entries ((sd + 2) u each, relative) and their 16-lane sum
// ((m / 16 + 4 + 15) u): 2 (m + sd + 6) u of the same (|q - x^|^2 <= 2 (|q|^2 + |x^|^2)).
__device__ __forceinline__ float pq_nominate_eps(int dim, int m, int sd, float qn, float norm_max)
{
    return ((4.0f * static_cast<float>(dim) + 2.0f * static_cast<float>(m + sd + 6)) * 5.9604645e-8f + 0.0078125f * 1.02f) * (qn + norm_max) + 1e-30f;
}

// per query: the table sum of its 64 nominated rows from the codes, the k best by (score, row id), and the proof (sq8_verify_kernel's)
__global__ __launch_bounds__(64) void pq_verify_kernel(const uint8_t *__restrict__ codes, int m, int sd, const float *__restrict__ tables,
                                                       const float *__restrict__ queries, const float *__restrict__ norm_max,
                                                       const uint32_t *__restrict__ cand_ids, const float *__restrict__ cand_scores, int k,
                                                       uint32_t *__restrict__ ids, float *__restrict__ scores, int *__restrict__ fail,
                                                       const float *__restrict__ thr, const int *__restrict__ counts, int cap, int thr_stride)
{
    constexpr int kc = 64;
    const int64_t q = blockIdx.x;
    const int lane = threadIdx.x;
    const int dim = m * sd;
    const float *qv = queries + q * dim;
    const uint32_t id = cand_ids[q * kc + lane];
    uint64_t key = kKeyMax;
    if (id != VG_INVALID_ID) key = make_key(pq_adc_row_score(tables + q * m * 256, codes + static_cast<int64_t>(id) * m, m), id, false);
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
        const float eps = pq_nominate_eps(dim, m, sd, qn, norm_max[0]);
        ok = kth != kKeyMax && key_score(kth, false) < (tau + qn) - eps;
    }
    if (lane < k) {
        const uint64_t e = tk.list;
        ids[q * k + lane] = e == kKeyMax ? VG_INVALID_ID : key_row(e);
        scores[q * k + lane] = e == kKeyMax ? INFINITY : key_score(e, false);
    }
    if (lane == 0) fail[q] = ok ? 0 : 1;
}

// the same for k beyond the 64-candidate budget: EVERY appended row is re-scored — one lane per row — and sorted; the proof
// compares the k-th exact score with the threshold itself (sq8_verify_sort_kernel's).  Dynamic LDS: cap keys.
__global__ __launch_bounds__(256) void pq_verify_sort_kernel(const uint8_t *__restrict__ codes, int m, int sd, const float *__restrict__ tables,
                                                             const float *__restrict__ queries, const float *__restrict__ norm_max,
                                                             const uint64_t *__restrict__ cand, const int *__restrict__ counts, int cap, int k,
                                                             uint32_t *__restrict__ ids, float *__restrict__ scores, int *__restrict__ fail,
                                                             const float *__restrict__ thr, int thr_stride)
{
    extern __shared__ uint64_t sortbuf[];
    const int64_t q = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int dim = m * sd;
    const float *qv = queries + q * dim;
    const int total = counts[q];
    const int cnt = total < cap ? total : cap;
    int n2 = 64;
    while (n2 < cnt) n2 <<= 1;
    for (int c = tid; c < cnt; c += 256) {
        const uint32_t id = key_row(cand[q * cap + c]);
        sortbuf[c] = make_key(pq_adc_row_score(tables + q * m * 256, codes + static_cast<int64_t>(id) * m, m), id, false);
    }
    for (int i = cnt + tid; i < n2; i += 256) sortbuf[i] = kKeyMax;
    __syncthreads();
    bitonic_sort_lds(sortbuf, n2, tid, 256);
    for (int i = tid; i < k; i += 256) {
        const uint64_t e = i < n2 ? sortbuf[i] : kKeyMax;
        ids[q * k + i] = e == kKeyMax ? VG_INVALID_ID : key_row(e);
        scores[q * k + i] = e == kKeyMax ? INFINITY : key_score(e, false);
    }
    if (tid >= 64) return;
    float qn = 0.0f;
    for (int j = lane; j < dim; j += 64) qn = __builtin_fmaf(qv[j], qv[j], qn);
    for (int off = 32; off > 0; off >>= 1) qn += __shfl_xor(qn, off);
    const uint64_t kth = k - 1 < n2 ? sortbuf[k - 1] : kKeyMax;
    const float tau = thr[q * thr_stride + (thr_stride - 1)];
    bool ok = total <= cap;
    if (ok && tau != INFINITY)  // (tau == +Inf: no threshold was set, every row was appended)
        ok = kth != kKeyMax && key_score(kth, false) < (tau + qn) - pq_nominate_eps(dim, m, sd, qn, norm_max[0]);
    if (lane == 0) fail[q] = ok ? 0 : 1;
}

// whether a batch takes the nomination (device queries; ascending table sums over the whole segment, no row filter)
static bool pq_nomination_applies(const vg_index *idx, const float *d_queries, int64_t nq, int k, const uint8_t *mask, bool desc)
{
    return idx->d_pq_bf16 && !mask && !desc && (nq * idx->n >= kPqNomMinPairs || hook(kHookPqNomAlways)) && k <= kPqNomMaxK && idx->n > k && idx->pq->k == 256 &&
           (reinterpret_cast<uintptr_t>(d_queries) & 15) == 0;
}

// The nomination + table sums + proof for a batch (device buffers): writes every query's k results and lists the queries whose
// proof failed — the caller scans those.  4096 queries a pass.
static int32_t pq_nominated_pass(vg_index *idx, const float *q, int64_t nq, int k, uint32_t *oid, float *osc, hipStream_t st,
                                 std::vector<int> &failed)
{
    const vg_pq *pq = idx->pq;
    for (int64_t q0 = 0; q0 < nq; q0 += 4096) {
        const int64_t cnt = std::min<int64_t>(4096, nq - q0);
        std::vector<int> h(static_cast<size_t>(cnt));
        {
            ArenaCall ar(idx->ctx, st);
            const int sel_k = pq_nominate_sel_k(k);
            const int i_scr = ar.add(flat_nominate_bf16_scratch(cnt, idx->n, idx->pq_bf16_dim, sel_k));
            const int i_thr = ar.add(sizeof(float) * static_cast<size_t>(cnt) * sel_k);
            const int i_cnt = ar.add(sizeof(int) * static_cast<size_t>(cnt));
            const int i_cid = ar.add(sizeof(uint32_t) * static_cast<size_t>(cnt) * 64);
            const int i_csc = ar.add(sizeof(float) * static_cast<size_t>(cnt) * 64);
            const int i_fail = ar.add(sizeof(int) * static_cast<size_t>(cnt));
            const int i_tab = ar.add(sizeof(float) * static_cast<size_t>(cnt) * pq->m * 256);
            VG_TRY(ar.commit());
            float *thr = ar.get<float>(i_thr), *csc = ar.get<float>(i_csc), *tables = ar.get<float>(i_tab);
            int *counts = ar.get<int>(i_cnt), *fail = ar.get<int>(i_fail);
            uint32_t *cid = ar.get<uint32_t>(i_cid);
            const float *qq = q + q0 * idx->dim;
            const uint64_t *cand = nullptr;
            int cap = 0;
            VG_TRY(launch_pq_build_table(pq, qq, cnt, tables, false, st));
            VG_TRY(flat_nominate_bf16(idx->ctx, idx->d_pq_bf16, idx->d_pq_norms, idx->n, idx->dim, idx->pq_bf16_dim, qq, cnt, ar.get<char>(i_scr),
                                      thr, counts, cid, csc, st, false, nullptr, 0, sel_k, k <= kPqPickMaxK, &cand, &cap, idx->d_pq_norm_max));
            if (k <= kPqPickMaxK)
                VG_LAUNCH(pq_verify_kernel, dim3(static_cast<unsigned>(cnt)), dim3(64), 0, st, idx->d_pq_rows, pq->m, pq->subdim, tables, qq,
                          idx->d_pq_norm_max, cid, csc, k, oid + q0 * k, osc + q0 * k, fail, thr, counts, cap, sel_k);
            else
                VG_LAUNCH(pq_verify_sort_kernel, dim3(static_cast<unsigned>(cnt)), dim3(256), sizeof(uint64_t) * static_cast<size_t>(cap), st,
                          idx->d_pq_rows, pq->m, pq->subdim, tables, qq, idx->d_pq_norm_max, cand, counts, cap, k, oid + q0 * k, osc + q0 * k,
                          fail, thr, sel_k);
            VG_HIP(hipMemcpyAsync(h.data(), fail, sizeof(int) * static_cast<size_t>(cnt), hipMemcpyDeviceToHost, st));
            VG_HIP(hipStreamSynchronize(st));
        }
        for (int64_t i = 0; i < cnt; i++)
            if (h[static_cast<size_t>(i)]) failed.push_back(static_cast<int>(q0 + i));
    }
    return VG_OK;
}
}  // namespace vg

VG_API int32_t vg_index_enable_pq_nomination(vg_index *idx, int32_t on, void *stream)
{
    VG_CHECK(idx, VG_ERR_INVALID_ARG, "vg_index_enable_pq_nomination: NULL index");
    VG_HIP(hipSetDevice(idx->ctx->device));
    hipStream_t st = vg::pick_stream(idx->ctx, stream);
    if (idx->d_pq_bf16) {
        VG_HIP(hipStreamSynchronize(st));
        VG_HIP(hipFree(idx->d_pq_bf16));
        VG_HIP(hipFree(idx->d_pq_norms));
        VG_HIP(hipFree(idx->d_pq_norm_max));
        idx->d_pq_bf16 = nullptr;
        idx->d_pq_norms = idx->d_pq_norm_max = nullptr;
    }
    if (!on) return VG_OK;
    VG_CHECK(idx->pq && idx->d_pq_rows, VG_ERR_NOT_READY, "vg_index_enable_pq_nomination: index has no PQ codes");
    const vg_pq *pq = idx->pq;
    VG_CHECK(pq->trained, VG_ERR_NOT_TRAINED, "ProductQuantizer not trained");
    VG_CHECK(pq->k == 256, VG_ERR_UNSUPPORTED, "vg_index_enable_pq_nomination: needs numCentroids == 256 (got %d), as the table scan does", pq->k);
    const int bdim = (idx->dim + 63) & ~63;  // whole K steps of the bf16 GEMM; the padding is zeros
    // the three arrays are published together, after the image is built: a failure half way leaves the index as it was
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
    hipLaunchKernelGGL(vg::pq_decode_bf16_kernel, dim3(static_cast<unsigned>((idx->n + 3) / 4)), dim3(256), 0, st, idx->d_pq_rows, idx->n, pq->m,
                       pq->k, pq->subdim, pq->d_codebooks, pq->d_scales, pq->d_offsets, img, bdim, norms, reinterpret_cast<int *>(norm_max));
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) VG_HIP(give_up(e));
    idx->pq_bf16_dim = bdim;
    idx->d_pq_bf16 = img;
    idx->d_pq_norms = norms;
    idx->d_pq_norm_max = norm_max;
    return VG_OK;
}

namespace vg {
int32_t pq_adc_search_masked(vg_index *idx, const float *queries, int64_t nq, int32_t k, const uint8_t *mask, int64_t mask_stride,
                             bool desc, uint32_t *ids, float *scores, void *stream);
}

VG_API int32_t vg_search_pq_adc(vg_index *idx, const float *queries, int64_t nq, int32_t k,
                                uint32_t *ids, float *scores, void *stream)
{
    return vg::pq_adc_search_masked(idx, queries, nq, k, nullptr, 0, false, ids, scores, stream);
}

// vg_search_pq_adc, and — with `mask` (a DEVICE pointer, bit per row, query q's at mask + q * mask_stride) — the whole-segment
// PQ leg of vg_search_flat_filtered (k_probe.hip; k <= 64 and a table that fits LDS, `desc` by the segment's metric)
namespace vg {
// vg_cand_replay.hpp's scorer for the PQ table scan: pq.AdcDistance of a row's code (flat/segment.go:678-689) — the table's entries
// (BuildDistanceTable, pq.go:468-491: five separately rounded operations per dimension) computed where they are used, summed in
// pqAdcLookupAvx512 order; one lane per row of the row-major codes.  At risk: a non-finite query value, scale or offset, or
// magnitudes whose squares could overflow.
struct PqScorer {
    const uint8_t *codes;     // n * m
    const int8_t *codebooks;  // m * 256 * sd
    const float *scales, *offsets;
    int m, sd;
    __device__ bool risk(int64_t, const float *q, int tid) const
    {
        __shared__ int flag;
        __shared__ float vmax;
        if (tid == 0) vmax = 0.0f;
        __syncthreads();
        bool bad = false;
        float b = 0.0f;
        for (int j = tid; j < m; j += kReplayThreads) {
            bad = bad || !is_finite_f32(scales[j]) || !is_finite_f32(offsets[j]);
            b = fmaxf(b, 128.0f * fabsf(scales[j]) + fabsf(offsets[j]));  // |centroid value| <= 128 |scale| + |offset|
        }
        for (int off = 32; off > 0; off >>= 1) b = fmaxf(b, __shfl_xor(b, off));
        if ((tid & 63) == 0) atomicMax(reinterpret_cast<int *>(&vmax), __float_as_int(b));  // non-negative floats order like their bits
        __syncthreads();
        const float vm = vmax;
        for (int j = tid; j < m * sd; j += kReplayThreads)
            bad = bad || !is_finite_f32(q[j]) || !(score_bound(fabsf(q[j]), vm, false) * static_cast<float>(m * sd) < 1e38f);
        return block_any(bad, &flag, tid);
    }
    __device__ void prepare(int64_t, const float *, int) const {}
    __device__ float entry(const float *q, int j, int c) const
    {
        const int8_t *cb = codebooks + (static_cast<int64_t>(j) * 256 + c) * sd;
        const float scale = scales[j], offset = offsets[j];
        const float *qs = q + j * sd;
        float sum = 0.0f;
        for (int i = 0; i < sd; i++) {
            float v = static_cast<float>(cb[i]) * scale;
            v = v + offset;
            const float d = qs[i] - v;
            const float dd = d * d;
            sum = sum + dd;
        }
        return sum;
    }
    __device__ void score_chunk(int64_t, const float *q, int64_t row0, int64_t n, int tid, float *out) const
    {
        const int64_t row = row0 + tid;
        if (row >= n) return;
        const uint8_t *code = codes + row * m;
        float s[16];
#pragma unroll
        for (int l = 0; l < 16; l++) s[l] = 0.0f;
        int i = 0;
        for (; i + 16 <= m; i += 16)
#pragma unroll
            for (int l = 0; l < 16; l++) s[l] = s[l] + entry(q, i + l, code[i + l]);
        float total = reduce16_regs(s);
        for (; i < m; i++) total = total + entry(q, i, code[i]);
        out[tid] = total;
    }
};
}  // namespace vg

namespace vg {
// the replay for device buffers: the whole segment, or (probes: nq * np partition ids, part_off) the probed partitions; mask: a
// device row filter per query / for the batch, or null; desc: a Dot / Cosine segment keeps the LARGEST sums (flat/segment.go:449)
int32_t pq_nan_replay(vg_index *idx, const float *d_queries, int64_t nq, int k, bool desc, const uint8_t *d_mask, int64_t mask_stride,
                      const uint32_t *d_probes, int np, const uint32_t *d_part_off, uint32_t *d_ids, float *d_scores, hipStream_t st)
{
    if (idx->n == 0) return VG_OK;
    const vg_pq *pq = idx->pq;
    return launch_cand_replay(PqScorer{idx->d_pq_rows, pq->d_codebooks, pq->d_scales, pq->d_offsets, pq->m, pq->subdim}, d_queries, idx->dim, idx->n, nq,
                              k, desc, d_mask, mask_stride, d_ids, d_scores, st, nullptr, d_probes, np, d_part_off);
}
}  // namespace vg

static int32_t pq_adc_search_impl(vg_index *idx, const float *queries, int64_t nq, int32_t k, const uint8_t *mask, int64_t mask_stride,
                                  bool desc, uint32_t *ids, float *scores, void *stream, bool allow_nomination);
int32_t vg::pq_adc_search_masked(vg_index *idx, const float *queries, int64_t nq, int32_t k, const uint8_t *mask,
                                 int64_t mask_stride, bool desc, uint32_t *ids, float *scores, void *stream)
{
    return pq_adc_search_impl(idx, queries, nq, k, mask, mask_stride, desc, ids, scores, stream, true);
}
static int32_t pq_adc_search_impl(vg_index *idx, const float *queries, int64_t nq, int32_t k, const uint8_t *mask, int64_t mask_stride,
                                  bool desc, uint32_t *ids, float *scores, void *stream, bool allow_nomination)
{
    VG_CHECK(idx, VG_ERR_INVALID_ARG, "vg_search_pq_adc: NULL index");
    VG_CHECK(idx->pq != nullptr, VG_ERR_NOT_READY, "vg_search_pq_adc: index has no PQ codes");
    const vg_pq *pq = idx->pq;
    VG_CHECK(pq->trained, VG_ERR_NOT_TRAINED, "ProductQuantizer not trained");
    VG_CHECK(nq >= 0 && k >= 0, VG_ERR_INVALID_ARG, "vg_search_pq_adc: negative nq or k");
    if (nq == 0 || k == 0) return VG_OK;
    VG_CHECK(queries && ids && scores, VG_ERR_INVALID_ARG, "vg_search_pq_adc: NULL buffer");
    VG_CHECK(k <= vg::kAdcMaxK, VG_ERR_UNSUPPORTED, "vg_search_pq_adc: k=%d exceeds %d", k,
             vg::kAdcMaxK);
    // The reference builds the table with stride K and looks it up with stride 256
    // (pq.go:474 vs internal/simd/kernels.go:249); only K == 256 is self-consistent.
    VG_CHECK(pq->k == 256, VG_ERR_UNSUPPORTED,
             "vg_search_pq_adc: LUT scan needs numCentroids == 256 (got %d)", pq->k);
    size_t lds = static_cast<size_t>(vg::lut_image_words(pq->m)) * 4 + vg::kAdcBuf * 8 + sizeof(vg::AdcShared);
    // a table beyond one LDS image (m > 96 at fp32) is walked in 96 KiB chunks: pq_adc_scan_wide_kernel, k <= 64
    const bool wide = lds > 160 * 1024;
    VG_CHECK(mask == nullptr || (!wide && k <= 64), VG_ERR_UNSUPPORTED, "pq_adc_search_masked: k=%d / m=%d take the probe kernels", k, pq->m);
    VG_CHECK(!wide || k <= 64, VG_ERR_UNSUPPORTED,
             "vg_search_pq_adc: m=%d lookup table does not fit the 160 KiB LDS; the chunked scan takes k <= 64", pq->m);
    VG_HIP(hipSetDevice(idx->ctx->device));
    hipStream_t st = vg::pick_stream(idx->ctx, stream);

    vg::DevIn<float> q;
    vg::DevOut<uint32_t> oid;
    vg::DevOut<float> osc;
    VG_TRY(q.init(queries, static_cast<size_t>(nq) * idx->dim, st));
    VG_TRY(oid.init(ids, static_cast<size_t>(nq) * k, st));
    VG_TRY(osc.init(scores, static_cast<size_t>(nq) * k, st));

    // queries whose table sums may hold a NaN: the reference's heap, operation by operation (vg_cand_replay.hpp; not for the queries
    // this function sends to itself after a failed proof: the caller's pass covers them)
    auto nan_replay = [&]() -> int32_t {
        if (!allow_nomination) return VG_OK;
        return vg::pq_nan_replay(idx, q.ptr, nq, k, desc, mask, mask_stride, nullptr, 0, nullptr, oid.ptr, osc.ptr, st);
    };
    if (idx->n == 0) {
        vg::DevTmp<uint64_t> none;
        VG_TRY(none.init(static_cast<size_t>(nq) * k, st));
        VG_HIP(hipMemsetAsync(none.ptr, 0xFF, static_cast<size_t>(nq) * k * 8, st));
        VG_TRY(vg::launch_topk_merge(none.ptr, nq, 1, k, false, oid.ptr, osc.ptr, st));
    } else if (allow_nomination && vg::pq_nomination_applies(idx, q.ptr, nq, k, mask, desc)) {
        std::vector<int> failed;
        VG_TRY(vg::pq_nominated_pass(idx, q.ptr, nq, k, oid.ptr, osc.ptr, st, failed));
        if (!failed.empty()) {  // the table scan for the queries whose proof failed (ties at the k-th score, thresholds too tight)
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
            VG_TRY(pq_adc_search_impl(idx, fq.ptr, nf, k, nullptr, 0, false, fid.ptr, fsc.ptr, st, false));
            for (int64_t i = 0; i < nf; i++) {
                const int64_t at = static_cast<int64_t>(failed[static_cast<size_t>(i)]) * k;
                VG_HIP(hipMemcpyAsync(oid.ptr + at, fid.ptr + i * k, sizeof(uint32_t) * k, hipMemcpyDeviceToDevice, st));
                VG_HIP(hipMemcpyAsync(osc.ptr + at, fsc.ptr + i * k, sizeof(float) * k, hipMemcpyDeviceToDevice, st));
            }
        }
    } else {
        int slices = vg::adc_slices(nq, idx->n_tiles, idx->ctx->compute_units);
        const bool bigk = k > 64;
        const bool bigk_fast = bigk && !vg::hook(vg::kHookAdcBigkExhaustive);  // test hook: LDS-buffer path only
        if (bigk_fast) {
            // enough waves that a wave expects <= ~8 of the k best rows (capacity 64 each)
            int64_t want = ((static_cast<int64_t>(k) / 64 + 7) / 8) * 8;
            int64_t max_s = (((idx->n_tiles + vg::kAdcWaves - 1) / vg::kAdcWaves) / 8) * 8;
            if (max_s < 8) max_s = 8;
            if (want > max_s) want = max_s;
            if (want > slices) slices = static_cast<int>(want);
        }
        vg::ArenaCall ar(idx->ctx, st);
        const int i_tables = ar.add(sizeof(float) * static_cast<size_t>(nq) * vg::lut_image_words(pq->m));
        const int i_partial = ar.add(sizeof(uint64_t) * static_cast<size_t>(nq) * slices *
                                     (bigk_fast ? std::max(k, vg::kAdcWaves * 64) : k));
        const int i_flags = ar.add(bigk_fast ? sizeof(int) * static_cast<size_t>(nq) : 0);
        const int i_fid = ar.add(bigk_fast ? sizeof(uint32_t) * static_cast<size_t>(nq) * k : 0);
        const int i_fsc = ar.add(bigk_fast ? sizeof(float) * static_cast<size_t>(nq) * k : 0);
        VG_TRY(ar.commit());
        struct { float *ptr; } tables{ar.get<float>(i_tables)};
        struct { uint64_t *ptr; } partial{ar.get<uint64_t>(i_partial)};
        VG_TRY(vg::launch_pq_build_table(pq, q.ptr, nq, tables.ptr, true, st));
        if (bigk_fast) {
            int *flags = ar.get<int>(i_flags);
            uint32_t *fid = ar.get<uint32_t>(i_fid);
            float *fsc = ar.get<float>(i_fsc);
            // (1) every wave keeps its 64 best; (2) select k of the union and prove it; (3) the
            // exhaustive LDS-buffer scan re-runs only for queries whose proof failed
            if (pq->m == 96)
                VG_TRY((vg::launch_scan<6, true>(idx, tables.ptr, nq, k, slices, partial.ptr, st, 1)));
            else
                VG_TRY((vg::launch_scan<-1, true>(idx, tables.ptr, nq, k, slices, partial.ptr, st, 1)));
            VG_LAUNCH(vg::topk_select_verify_kernel, dim3(static_cast<unsigned>(nq)), dim3(vg::kMergeThreads), 0, st,
                      partial.ptr, slices * vg::kAdcWaves, k, false, oid.ptr, osc.ptr, flags);
            if (pq->m == 96)
                VG_TRY((vg::launch_scan<6, false>(idx, tables.ptr, nq, k, slices, partial.ptr, st, 0, flags)));
            else
                VG_TRY((vg::launch_scan<-1, false>(idx, tables.ptr, nq, k, slices, partial.ptr, st, 0, flags)));
            VG_TRY(vg::launch_topk_merge(partial.ptr, nq, slices, k, false, fid, fsc, st, flags));
            VG_LAUNCH(vg::patch_results_kernel, dim3(static_cast<unsigned>(nq)), dim3(64), 0, st, flags, k, fid, fsc,
                      oid.ptr, osc.ptr);
            VG_TRY(nan_replay());
            VG_TRY(oid.finish());
            VG_TRY(osc.finish());
            if (oid.on_host() || osc.on_host()) VG_HIP(hipStreamSynchronize(st));
            return VG_OK;
        }
        if (wide)
            VG_TRY(vg::launch_scan_wide(idx, tables.ptr, nq, k, slices, partial.ptr, st));
        else if (pq->m == 96 && k <= 64)
            VG_TRY((vg::launch_scan<6, true>(idx, tables.ptr, nq, k, slices, partial.ptr, st, 0, nullptr, mask, mask_stride, desc)));
        else if (pq->m == 96)
            VG_TRY((vg::launch_scan<6, false>(idx, tables.ptr, nq, k, slices, partial.ptr, st)));
        else if (k <= 64)
            VG_TRY((vg::launch_scan<-1, true>(idx, tables.ptr, nq, k, slices, partial.ptr, st, 0, nullptr, mask, mask_stride, desc)));
        else
            VG_TRY((vg::launch_scan<-1, false>(idx, tables.ptr, nq, k, slices, partial.ptr, st)));
        VG_TRY(vg::launch_topk_merge(partial.ptr, nq, slices, k, desc, oid.ptr, osc.ptr, st));
    }
    VG_TRY(nan_replay());
    VG_TRY(oid.finish());
    VG_TRY(osc.finish());
    if (oid.on_host() || osc.on_host()) VG_HIP(hipStreamSynchronize(st));
    return VG_OK;
}
