// Prices the stages of flat_gemm_kernel (vecgo_amd/csrc/vg_flat_gemm.hpp) by timing PROBE variants
// of the same code on 1024 queries x N rows x 768 dims.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -I../../vecgo_amd/csrc \
//         gemm_probe.hip -o gemm_probe && ./gemm_probe [N]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
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

__global__ void fill(float *p, size_t n, uint32_t seed)
{
    size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
    if (i >= n) return;
    uint32_t x = uint32_t(i) * 2654435761u + seed;
    x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
    p[i] = (float(x >> 8) * (1.0f / 16777216.0f) - 0.5f) * 2.0f;
}

template <int MODE, int PROBE, bool DMA = false, bool BF16 = false>
static void run(const char *name, const float *q, int64_t nq, const float *base, int64_t n, int dim,
                const float *norms, float *scores, const float *thr, int *counts, uint64_t *cand, int cap)
{
    auto kern = BF16 ? vg::flat_gemm_dma_kernel<false, MODE, PROBE, true>
                     : (DMA ? vg::flat_gemm_dma_kernel<false, MODE, PROBE> : vg::flat_gemm_kernel<false, MODE, PROBE>);
    const size_t lds = DMA ? vg::kDmaLdsBytes : vg::kGemmLdsBytes;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    const int64_t mt = (nq + vg::kGemmBM - 1) / vg::kGemmBM, nt = (n + vg::kGemmBN - 1) / vg::kGemmBN;
    dim3 grid(unsigned(mt * ((nt + 7) / 8) * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        CK(hipMemset(counts, 0, sizeof(int) * nq));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, grid, dim3(vg::kGemmThreads), lds, 0, q, nq, base, n, dim, norms,
                           scores, 1, n, thr, 1, 0, counts, cand, cap);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    const double tf = 2.0 * double(nq) * double(n) * (BF16 ? 2 * dim : dim) / (best * 1e-3) / 1e12;
    printf("%-44s %8.3f ms  %6.1f TFLOP/s  (%.1f %% of 157.3)\n", name, best, tf, tf / 157.3 * 100);
}

int main(int argc, char **argv)
{
    const int64_t n = argc > 1 ? atoll(argv[1]) : 1000000, nq = 1024;
    const int dim = 768, cap = 4096;
    float *q, *base, *norms, *thr, *scores;
    int *counts;
    uint64_t *cand;
    CK(hipMalloc(&q, nq * dim * 4));
    CK(hipMalloc(&base, n * dim * 4));
    CK(hipMalloc(&norms, n * 4));
    CK(hipMalloc(&thr, nq * 4));
    CK(hipMalloc(&scores, 1 << 20));
    CK(hipMalloc(&counts, nq * 4));
    CK(hipMalloc(&cand, nq * cap * 8));
    fill<<<unsigned((nq * dim + 255) / 256), 256>>>(q, nq * dim, 1);
    fill<<<unsigned((n * dim + 255) / 256), 256>>>(base, n * dim, 2);
    fill<<<unsigned((n + 255) / 256), 256>>>(norms, n, 3);
    std::vector<float> h(nq, -1e30f);  // nothing passes the threshold
    CK(hipMemcpy(thr, h.data(), nq * 4, hipMemcpyHostToDevice));
    CK(hipDeviceSynchronize());
    run<2, 0>("full kernel (MODE 2)", q, nq, base, n, dim, norms, scores, thr, counts, cand, cap);
    run<2, 1>("- epilogue", q, nq, base, n, dim, norms, scores, thr, counts, cand, cap);
    run<2, 1 | 2>("- epilogue - global loads", q, nq, base, n, dim, norms, scores, thr, counts, cand, cap);
    run<2, 1 | 2 | 4>("- epilogue - loads - LDS stores", q, nq, base, n, dim, norms, scores, thr, counts, cand, cap);
    run<2, 1 | 2 | 4 | 8>("- epilogue - loads - stores - barrier", q, nq, base, n, dim, norms, scores, thr, counts, cand, cap);
    run<2, 1 | 2 | 4 | 8 | 16>("MFMA only (no LDS reads either)", q, nq, base, n, dim, norms, scores, thr, counts, cand, cap);
    run<2, 1 | 16>("- epilogue - LDS reads", q, nq, base, n, dim, norms, scores, thr, counts, cand, cap);
    run<2, 1 | 8>("- epilogue - barrier (racy)", q, nq, base, n, dim, norms, scores, thr, counts, cand, cap);
    printf("LDS-DMA kernel:\n");
    run<2, 0, true>("full kernel (MODE 2)", q, nq, base, n, dim, norms, scores, thr, counts, cand, cap);
    run<2, 1, true>("- epilogue", q, nq, base, n, dim, norms, scores, thr, counts, cand, cap);
    run<2, 1 | 32, true>("- epilogue, DMA re-fetches tile 0 (hot)", q, nq, base, n, dim, norms, scores, thr, counts, cand, cap);
    run<2, 1 | 64, true>("- epilogue, DMA of A tiles only", q, nq, base, n, dim, norms, scores, thr, counts, cand, cap);
    run<2, 1 | 128, true>("- epilogue, no vmcnt wait (racy)", q, nq, base, n, dim, norms, scores, thr, counts, cand, cap);
    run<2, 1 | 128 | 8, true>("- epilogue, no vmcnt wait, no barrier", q, nq, base, n, dim, norms, scores, thr, counts, cand, cap);
    run<2, 1 | 2, true>("- epilogue - DMA", q, nq, base, n, dim, norms, scores, thr, counts, cand, cap);
    run<2, 1 | 2 | 8, true>("- epilogue - DMA - barrier", q, nq, base, n, dim, norms, scores, thr, counts, cand, cap);
    run<2, 1 | 2 | 8 | 16, true>("MFMA only", q, nq, base, n, dim, norms, scores, thr, counts, cand, cap);
    // the same buffers read as bfloat16 rows of 768 elements = 384 words (what the values are does not matter here)
    printf("bfloat16 LDS-DMA kernel (768 elements per row; %% of 157.3 is meaningless here):\n");
    const int dw = dim / 2;
    run<2, 0, true, true>("full kernel (MODE 2)", q, nq, base, n, dw, norms, scores, thr, counts, cand, cap);
    run<2, 1, true, true>("- epilogue", q, nq, base, n, dw, norms, scores, thr, counts, cand, cap);
    run<2, 1 | 32, true, true>("- epilogue, DMA re-fetches tile 0 (hot)", q, nq, base, n, dw, norms, scores, thr, counts, cand, cap);
    run<2, 1 | 64, true, true>("- epilogue, DMA of A tiles only", q, nq, base, n, dw, norms, scores, thr, counts, cand, cap);
    run<2, 1 | 128, true, true>("- epilogue, no vmcnt wait (racy)", q, nq, base, n, dw, norms, scores, thr, counts, cand, cap);
    run<2, 1 | 128 | 8, true, true>("- epilogue, no vmcnt wait, no barrier", q, nq, base, n, dw, norms, scores, thr, counts, cand, cap);
    run<2, 1 | 2, true, true>("- epilogue - DMA", q, nq, base, n, dw, norms, scores, thr, counts, cand, cap);
    run<2, 1 | 2 | 8, true, true>("- epilogue - DMA - barrier", q, nq, base, n, dw, norms, scores, thr, counts, cand, cap);
    run<2, 1 | 16, true, true>("- epilogue - LDS reads", q, nq, base, n, dw, norms, scores, thr, counts, cand, cap);
    run<2, 1 | 2 | 8 | 16, true, true>("MFMA only", q, nq, base, n, dw, norms, scores, thr, counts, cand, cap);
    return 0;
}
