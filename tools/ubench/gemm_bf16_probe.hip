// The bfloat16 nomination GEMM (MODE 2: append what falls below each query's threshold), 1024 queries x N rows x 768 elements:
// the 128 x 128 tile (flat_gemm_dma_kernel<.., true>) against the persistent 256 x 256 tile (flat_gemm_bf16_big_kernel, NB = 2 / 3) and
// its stage probes, on REAL bfloat16 data (uniform [-1, 1): what the operands are sets the clock the chip holds —
// MI355X_MICROARCH.md 'DVFS give-back').  Checks the appended (query, row) sets of the two tiles against each other.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -I../../vecgo_amd/csrc -I../../include \
//         gemm_bf16_probe.hip -o gemm_bf16_probe && ./gemm_bf16_probe [N] [nq]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

#include "vg_flat_gemm.hpp"

#define CK(x)                                                        \
    do {                                                             \
        hipError_t e = (x);                                          \
        if (e != hipSuccess) {                                       \
            printf("%s: %s\n", #x, hipGetErrorString(e));            \
            exit(1);                                                 \
        }                                                            \
    } while (0)

__device__ __forceinline__ float rnd(uint64_t i, uint32_t seed)
{
    uint32_t x = uint32_t(i) * 2654435761u + seed + uint32_t(i >> 32) * 40503u;
    x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
    return (float(x >> 8) * (1.0f / 16777216.0f) - 0.5f) * 2.0f;
}
__device__ __forceinline__ uint16_t to_bf16(float v)
{
    const uint32_t u = __float_as_uint(v);
    return uint16_t((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
__global__ void fill_bf16(uint16_t *p, size_t n, uint32_t seed)
{
    size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
    if (i < n) p[i] = to_bf16(rnd(i, seed));
}
__global__ void row_norms(const uint16_t *rows, int64_t n, int dim, float *norms)
{
    int64_t r = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    if (r >= n) return;
    float s = 0.0f;
    for (int j = 0; j < dim; j++) {
        const float v = __uint_as_float(uint32_t(rows[r * dim + j]) << 16);
        s += v * v;
    }
    norms[r] = s;
}

struct Bufs {
    const float *q, *base, *norms, *thr;
    int64_t nq, n;
    int dw, cap;
    int *counts;
    uint64_t *cand;
};

template <typename F>
static float time_ms(const Bufs &b, F launch, int reps = 5)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < reps; rep++) {
        CK(hipMemset(b.counts, 0, sizeof(int) * (b.nq + 1)));
        CK(hipEventRecord(e0));
        launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    return best;
}

static void report(const char *name, const Bufs &b, float ms)
{
    const double tf = 2.0 * double(b.nq) * double(b.n) * (2.0 * b.dw) / (ms * 1e-3) / 1e12;
    printf("%-58s %8.3f ms  %7.1f TFLOP/s  (%.3f of 2500)\n", name, ms, tf, tf / 2500.0);
    fflush(stdout);
}

template <int PROBE>
static void run_old(const char *name, const Bufs &b)
{
    auto kern = vg::flat_gemm_dma_kernel<false, 2, PROBE, true>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(vg::kDmaLdsBytes)));
    const int64_t mt = (b.nq + vg::kGemmBM - 1) / vg::kGemmBM, nt = (b.n + vg::kGemmBN - 1) / vg::kGemmBN;
    const float ms = time_ms(b, [&] {
        hipLaunchKernelGGL(kern, dim3(unsigned(mt * ((nt + 7) / 8) * 8)), dim3(vg::kGemmThreads), vg::kDmaLdsBytes, 0, b.q, b.nq, b.base, b.n,
                           b.dw, b.norms, (float *)nullptr, 1, int64_t(0), b.thr, 1, 0, b.counts, b.cand, b.cap, (const uint8_t *)nullptr,
                           int64_t(0));
    });
    report(name, b, ms);
}

template <int NB, int PROBE>
static void run_big(const char *name, const Bufs &b)
{
    auto kern = vg::flat_gemm_bf16_big_kernel<false, NB, PROBE>;
    const size_t lds = vg::big_lds_bytes<NB>();
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    const int64_t mt = (b.nq + vg::kBigBM - 1) / vg::kBigBM, nt = (b.n + vg::kBigBN - 1) / vg::kBigBN;
    const int64_t slots = mt * ((nt + 7) / 8) * 8;  // persistent: one workgroup per CU
    const unsigned grid = unsigned(slots < 256 ? slots : 256);
    const float ms = time_ms(b, [&] {
        hipLaunchKernelGGL(kern, dim3(grid), dim3(vg::kBigThreads), lds, 0, b.q, b.nq, b.base, b.n, b.dw, b.norms,
                           b.thr, 1, 0, b.counts, b.cand, b.cap, (const uint8_t *)nullptr, int64_t(0), 1);
    });
    report(name, b, ms);
}

// the appended (row id, score) pairs of every query, sorted by row id
static std::vector<std::vector<std::pair<uint32_t, float>>> fetch(const Bufs &b)
{
    std::vector<int> counts(b.nq);
    std::vector<uint64_t> keys(size_t(b.nq) * b.cap);
    CK(hipMemcpy(counts.data(), b.counts, sizeof(int) * b.nq, hipMemcpyDeviceToHost));
    CK(hipMemcpy(keys.data(), b.cand, sizeof(uint64_t) * keys.size(), hipMemcpyDeviceToHost));
    std::vector<std::vector<std::pair<uint32_t, float>>> out(b.nq);
    for (int64_t q = 0; q < b.nq; q++) {
        const int c = std::min(counts[q], b.cap);
        for (int i = 0; i < c; i++) {
            const uint64_t k = keys[q * b.cap + i];
            uint32_t sb = uint32_t(k >> 32);  // make_key: order-preserving bits of the score, ascending (vg_device.hpp ordered_f32)
            sb = (sb & 0x80000000u) ? (sb & 0x7FFFFFFFu) : ~sb;
            float sc;
            std::memcpy(&sc, &sb, 4);
            out[q].push_back({uint32_t(k & 0xFFFFFFFFu), sc});
        }
        std::sort(out[q].begin(), out[q].end());
    }
    return out;
}

int main(int argc, char **argv)
{
    const int64_t n = argc > 1 ? atoll(argv[1]) : 1000000, nq = argc > 2 ? atoll(argv[2]) : 1024;
    const int dim = 768, cap = 4096, dw = dim / 2;
    uint16_t *q, *base;
    float *norms, *thr;
    Bufs b;
    CK(hipMalloc(&q, nq * dim * 2));
    CK(hipMalloc(&base, n * dim * 2));
    CK(hipMalloc(&norms, n * 4));
    CK(hipMalloc(&thr, nq * 4));
    CK(hipMalloc(&b.counts, (nq + 1) * 4));
    CK(hipMalloc(&b.cand, nq * cap * 8));
    fill_bf16<<<unsigned((nq * dim + 255) / 256), 256>>>(q, nq * dim, 1);
    fill_bf16<<<unsigned((n * dim + 255) / 256), 256>>>(base, n * dim, 2);
    row_norms<<<unsigned((n + 255) / 256), 256>>>(base, n, dim, norms);
    // scores |x|^2 - 2 q.x: mean 256, sigma ~ 20 on this data; 190 lets ~0.05 % of the rows through (the library's thresholds: ~512 rows)
    std::vector<float> h(nq, 190.0f);
    CK(hipMemcpy(thr, h.data(), nq * 4, hipMemcpyHostToDevice));
    CK(hipDeviceSynchronize());
    b.q = reinterpret_cast<const float *>(q);
    b.base = reinterpret_cast<const float *>(base);
    b.norms = norms;
    b.thr = thr;
    b.nq = nq;
    b.n = n;
    b.dw = dw;
    b.cap = cap;

    printf("%lld queries x %lld rows x %d bfloat16\n", (long long)nq, (long long)n, dim);
    run_old<0>("128 x 128 tile: full kernel", b);
    const auto ref = fetch(b);
    run_old<1>("128 x 128 tile: - epilogue", b);
    for (int nb = 2; nb <= 3; nb++) {
        if (nb == 2)
            run_big<2, 0>("256 x 256 tile, NB = 2: full kernel", b);
        else
            run_big<3, 0>("256 x 256 tile, NB = 3: full kernel", b);
        const auto got = fetch(b);
        size_t total = 0, only_ref = 0, only_got = 0, score_off = 0;
        for (int64_t qi = 0; qi < nq; qi++) {
            total += got[qi].size();
            size_t i = 0, j = 0;
            while (i < ref[qi].size() || j < got[qi].size()) {
                if (j == got[qi].size() || (i < ref[qi].size() && ref[qi][i].first < got[qi][j].first)) {
                    only_ref++, i++;
                } else if (i == ref[qi].size() || got[qi][j].first < ref[qi][i].first) {
                    only_got++, j++;
                } else {
                    if (fabsf(ref[qi][i].second - got[qi][j].second) > 1e-3f * fabsf(ref[qi][i].second)) score_off++;
                    i++, j++;
                }
            }
        }
        printf("    appended pairs %zu; only in the 128-tile set %zu, only in this set %zu (scores next to the threshold), scores off by > 1e-3: %zu\n",
               total, only_ref, only_got, score_off);
    }
    run_big<3, 32768>("NB = 3: full kernel, counting the flushes", b);
    {
        int fl = 0;
        CK(hipMemcpy(&fl, b.counts + nq, 4, hipMemcpyDeviceToHost));
        printf("    flushes of parked elements: %d over 2048 waves x 61 tiles (%.2f per wave)\n", fl, fl / 2048.0);
    }
    run_big<3, 512>("NB = 3: full kernel, row fills behind the FIRST group", b);
    run_big<3, 4096>("NB = 3: keys stored without atomics (slots 0 / 1 of a list)", b);
    run_big<3, 2048>("NB = 3: passing elements collected and dropped", b);
    run_big<3, 256>("NB = 3: the block maxima only, nothing appended", b);
    run_big<3, 1 | 1024>("NB = 3: - epilogue, accumulators start at 0", b);
    run_big<2, 1>("NB = 2: - epilogue", b);
    run_big<3, 1>("NB = 3: - epilogue", b);
    run_big<3, 1 | 64>("NB = 3: - epilogue, query tiles only by DMA", b);
    run_big<3, 1 | 2>("NB = 3: - epilogue - DMA after the prologue", b);
    run_big<3, 1 | 2 | 8>("NB = 3: - epilogue - DMA - barrier", b);
    run_big<3, 1 | 16>("NB = 3: - epilogue - LDS operand reads", b);
    run_big<3, 1 | 2 | 8 | 16>("NB = 3: matrix instructions only", b);
    return 0;
}
