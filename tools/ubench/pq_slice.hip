// PQ node scoring: where should a wave's 96 centroid lookups per 64 nodes come from?  
//   MODE 0  the shipped form: 96 x (64-lane gather of 8 bytes from the 2 KiB slice of the 192 KiB int8 codebook, texture path)
//   MODE 1  per-wave LDS ring: each slice brought in by two coalesced global_load_dwordx4 per lane -> ds_write_b128 ->
//           the 64 lookups are ds_read_b64
//   MODE 2  the same with LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write)
//   MODE 3  no lookups at all (the terms' arithmetic alone)
// ARITH = 1 adds the real term arithmetic of vg_hnsw_layer.hpp (pq_term8_quad), 0 just consumes the centroid words.
// One wave per workgroup like hnsw_search_kernel, WPS waves per SIMD through amdgpu_waves_per_eu + LDS padding: the walk
// kernel's heaps leave ~2.6 KiB of LDS per wave at 16 waves per CU (ef 128), i.e. a ring of ONE slice; RING > 1 models
// fewer waves per CU.  Prints ns per 64-node batch and G node scores/s.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../vecgo_amd/csrc -I../../include pq_slice.hip -o pq_slice
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "vg_hnsw_layer.hpp"

#define CK(x)                                                        \
    do {                                                             \
        hipError_t e = (x);                                          \
        if (e != hipSuccess) {                                       \
            printf("%s: %s\n", #x, hipGetErrorString(e));            \
            exit(1);                                                 \
        }                                                            \
    } while (0)

constexpr int M = 96;

__device__ __forceinline__ void glds16_at(const void *src, uint32_t lds_base)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(lds_base)
                 : "memory");
}

template <int MODE, int RING, int ARITH>
__global__ __launch_bounds__(64) void score(const int8_t *__restrict__ cbk, const uint32_t *__restrict__ seeds,
                                            const float *__restrict__ qv, const float *__restrict__ scales,
                                            const float *__restrict__ offsets, float *__restrict__ out, int iters, int pad_words)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *qprep = reinterpret_cast<float *>(smem);                     // 48 * 20 floats
    unsigned char *ring = smem + (M / 2) * vg::kPqPairFloats * 4;       // RING * 2 KiB (16-byte aligned: 3840)
    const int lane = threadIdx.x;
    vg::pq_direct_prepare(qprep, qv, scales, offsets, M, lane);
    __syncthreads();
    const uint2 *cb = reinterpret_cast<const uint2 *>(cbk);
    uint32_t w = seeds[blockIdx.x * 64 + lane];
    float total = 0.0f;
    const uint32_t ring0 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) void *)ring));
    for (int it = 0; it < iters; it++) {
        float distance = 0.0f;
        if (MODE == 1 || MODE == 2) {
            // prologue: slices 0 .. RING-2 in flight
#pragma unroll
            for (int s = 0; s < RING - 1; s++) {
                const unsigned char *src = reinterpret_cast<const unsigned char *>(cb + s * 256);
                if (MODE == 2) {
                    glds16_at(src + lane * 16, ring0 + s * 2048);
                    glds16_at(src + 1024 + lane * 16, ring0 + s * 2048 + 1024);
                }
            }
        }
        for (int s0 = 0; s0 < M; s0 += 4) {
            uint2 e[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int s = s0 + u;
                const uint32_t code = (w >> (8 * u)) & 0xFFu;
                if (MODE == 0) {
                    e[u] = cb[s * 256 + code];
                } else if (MODE == 1) {
                    const uint4 *src = reinterpret_cast<const uint4 *>(cb + s * 256);
                    const uint4 a = src[lane], b = src[64 + lane];
                    uint4 *dst = reinterpret_cast<uint4 *>(ring + (s % RING) * 2048);
                    dst[lane] = a;
                    dst[64 + lane] = b;
                    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's writes are in LDS
                    e[u] = *reinterpret_cast<const uint2 *>(ring + (s % RING) * 2048 + code * 8);
                } else if (MODE == 2) {
                    const int sn = s + RING - 1;  // keep RING - 1 slices ahead
                    if (sn < M) {
                        const unsigned char *src = reinterpret_cast<const unsigned char *>(cb + sn * 256);
                        glds16_at(src + lane * 16, ring0 + (sn % RING) * 2048);
                        glds16_at(src + 1024 + lane * 16, ring0 + (sn % RING) * 2048 + 1024);
                    }
                    // slice s has landed when at most 2 * (RING - 1) DMA instructions are outstanding
                    if (RING == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    else if (RING == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                    else if (RING == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                    e[u] = *reinterpret_cast<const uint2 *>(ring + (s % RING) * 2048 + code * 8);
                    if (RING == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // read before the slot is refilled
                } else {
                    e[u] = make_uint2(code * 0x01010101u + s, code + it);
                }
            }
            if (ARITH) {
                vg::vg_f2 ta, tb;
                vg::pq_term8_quad(e[0], e[1], e[2], e[3], qprep + (s0 >> 1) * vg::kPqPairFloats, ta, tb);
                distance = distance + ta.x;
                distance = distance + ta.y;
                distance = distance + tb.x;
                distance = distance + tb.y;
            } else {
                distance += __uint_as_float((e[0].x ^ e[1].y ^ e[2].x ^ e[3].y) & 0x3fffffffu);
            }
            w = w * 1664525u + 1013904223u;
        }
        total += distance;
    }
    out[blockIdx.x * 64 + lane] = total + (pad_words ? reinterpret_cast<float *>(smem)[pad_words] : 0.0f);
}

template <int MODE, int RING, int ARITH>
static void run(const char *name, int waves_per_cu, const int8_t *cb, const uint32_t *seeds, const float *qv, const float *sc,
                const float *of, float *out, int blocks)
{
    // LDS per workgroup sized so that exactly waves_per_cu workgroups fit a CU's 160 KiB
    const size_t need = (M / 2) * vg::kPqPairFloats * 4 + (MODE == 1 || MODE == 2 ? RING * 2048 : 0);
    size_t lds = (160 * 1024 / waves_per_cu) & ~size_t(255);
    if (lds < need) {
        printf("%-44s needs %zu B of LDS per wave: %d waves per CU do not fit\n", name, need, waves_per_cu);
        return;
    }
    if (lds > 64 * 1024) lds = 64 * 1024;
    auto kern = score<MODE, RING, ARITH>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    const int iters = 400;
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), lds, 0, cb, seeds, qv, sc, of, out, 20, 0);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), lds, 0, cb, seeds, qv, sc, of, out, iters, 0);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    const double batches = static_cast<double>(blocks) * iters;
    printf("%-44s %2d waves/CU  %8.3f ms  %7.1f ns per 64-node batch and CU-wave slot  %6.2f G node scores/s\n", name, waves_per_cu,
           ms, ms * 1e6 / (batches / (256.0 * waves_per_cu)), batches * 64 / (ms * 1e-3) / 1e9);
}

int main()
{
    std::vector<int8_t> hcb(M * 256 * 8);
    for (auto &x : hcb) x = static_cast<int8_t>(rand());
    const int blocks = 256 * 16 * 2;  // two rounds of 16 waves per CU
    std::vector<uint32_t> hs(blocks * 64);
    for (auto &x : hs) x = static_cast<uint32_t>(rand()) * 2654435761u + rand();
    std::vector<float> hq(768), hsc(M), hof(M);
    for (auto &x : hq) x = rand() / float(RAND_MAX);
    for (auto &x : hsc) x = 0.01f;
    for (auto &x : hof) x = 0.5f;
    int8_t *cb;
    uint32_t *seeds;
    float *qv, *sc, *of, *out;
    CK(hipMalloc(&cb, hcb.size()));
    CK(hipMalloc(&seeds, hs.size() * 4));
    CK(hipMalloc(&qv, 768 * 4));
    CK(hipMalloc(&sc, M * 4));
    CK(hipMalloc(&of, M * 4));
    CK(hipMalloc(&out, blocks * 64 * 4));
    CK(hipMemcpy(cb, hcb.data(), hcb.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(seeds, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(qv, hq.data(), 768 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(sc, hsc.data(), M * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(of, hof.data(), M * 4, hipMemcpyHostToDevice));
    for (int wpc : {16, 8}) {
        run<3, 1, 1>("arithmetic only", wpc, cb, seeds, qv, sc, of, out, blocks);
        run<0, 1, 0>("gather (texture path), no arithmetic", wpc, cb, seeds, qv, sc, of, out, blocks);
        run<0, 1, 1>("gather (texture path) + arithmetic", wpc, cb, seeds, qv, sc, of, out, blocks);
        run<1, 1, 0>("slice via registers -> LDS, ring 1, no arith", wpc, cb, seeds, qv, sc, of, out, blocks);
        run<1, 1, 1>("slice via registers -> LDS, ring 1 + arith", wpc, cb, seeds, qv, sc, of, out, blocks);
        run<2, 1, 0>("slice via LDS-DMA, ring 1, no arith", wpc, cb, seeds, qv, sc, of, out, blocks);
        run<2, 1, 1>("slice via LDS-DMA, ring 1 + arith", wpc, cb, seeds, qv, sc, of, out, blocks);
        run<2, 2, 1>("slice via LDS-DMA, ring 2 + arith", wpc, cb, seeds, qv, sc, of, out, blocks);
        run<2, 4, 0>("slice via LDS-DMA, ring 4, no arith", wpc, cb, seeds, qv, sc, of, out, blocks);
        run<2, 4, 1>("slice via LDS-DMA, ring 4 + arith", wpc, cb, seeds, qv, sc, of, out, blocks);
    }
    return 0;
}
