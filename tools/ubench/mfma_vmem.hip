// What does one vector-memory instruction cost a wave that is otherwise issuing fp32 MFMAs
// back to back?  Each wave runs ITERS x { 16 x v_mfma_f32_32x32x2_f32 ; NV x <memory op> } on
// cache-hot addresses; the table prints MFMA-pipe cycles lost per memory op.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 mfma_vmem.hip -o mfma_vmem
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                             \
    do {                                                  \
        hipError_t e = (x);                               \
        if (e != hipSuccess) {                            \
            printf("%s: %s\n", #x, hipGetErrorString(e)); \
            exit(1);                                      \
        }                                                 \
    } while (0)

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

enum Kind { kNone, kLoadX4, kLoadX4Saddr, kLoadX1, kGldsX4, kGldsX4Saddr, kGldsX1, kDsRead128, kDsWrite128, kBufferX4 };

template <int KIND, int NV, int SPREAD>
__global__ __launch_bounds__(256) void probe(const float *__restrict__ src, float *__restrict__ out, int iters)
{
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 acc[4];
    for (int i = 0; i < 4; i++)
        for (int r = 0; r < 16; r++) acc[i][r] = 0.0f;
    float a = src[threadIdx.x], b = src[threadIdx.x + 256];
    const float *p = src + (blockIdx.x & 63) * 4096 + threadIdx.x * 4;
    const uint32_t voff = ((blockIdx.x & 63) * 4096 + threadIdx.x * 4) * 4;
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane(wave * 1024 * 8);
    f32x4 sink = {0, 0, 0, 0};
    lds[threadIdx.x * 4] = a;
    __syncthreads();
    auto memop = [&](int i) {
        if (KIND == kLoadX4) {
            f32x4 v;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p + i * 1024) : "memory");
            sink = v;  // not waited for: the asm result is never consumed before the final wait
        } else if (KIND == kLoadX4Saddr) {
            f32x4 v;
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(voff + i * 4096), "s"(src) : "memory");
            sink = v;
        } else if (KIND == kLoadX1) {
            float v;
            asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p + i * 1024) : "memory");
            sink[0] = v;
        } else if (KIND == kGldsX4) {
            uint32_t keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(p + i * 1024), "s"(lds_base + (i & 7) * 1024) : "memory");
        } else if (KIND == kGldsX4Saddr) {
            uint32_t keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(voff + i * 4096), "s"(lds_base + (i & 7) * 1024), "s"(src) : "memory");
        } else if (KIND == kGldsX1) {
            uint32_t keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(p + i * 1024), "s"(lds_base + (i & 7) * 1024) : "memory");
        } else if (KIND == kDsRead128) {
            f32x4 v;
            asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(uint32_t(threadIdx.x * 16 + (i & 7) * 4096)) : "memory");
            sink = v;
        } else if (KIND == kDsWrite128) {
            asm volatile("ds_write_b128 %0, %1" ::"v"(uint32_t(threadIdx.x * 16 + (i & 7) * 4096)), "v"(sink) : "memory");
        }
    };
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int g = 0; g < 4; g++) {
            if (SPREAD) {
#pragma unroll
                for (int i = g * NV / 4; i < (g + 1) * NV / 4; i++) memop(i);
            } else if (g == 0) {
#pragma unroll
                for (int i = 0; i < NV; i++) memop(i);
            }
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[3], 0, 0, 0);
        }
        if (KIND != kNone) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    float t = sink[0] + sink[1] + sink[2] + sink[3];
    for (int i = 0; i < 4; i++)
        for (int r = 0; r < 16; r++) t += acc[i][r];
    if (t == 123.456f) out[0] = t + lds[lane];
}

template <int KIND, int NV, int SPREAD>
static double run(const char *name, const float *src, float *out, int blocks_per_cu, double base_ms)
{
    const int iters = 2000, blocks = 256 * blocks_per_cu;
    auto k = probe<KIND, NV, SPREAD>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 65536, 0, src, out, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    // per SIMD: blocks_per_cu waves, each iters x 16 MFMAs of 64 cycles
    const double mfma_cycles = double(blocks_per_cu) * iters * 16 * 64;
    const double cycles = best * 1e-3 * 2.4e9;
    const double per_op = NV ? (best - base_ms) * 1e-3 * 2.4e9 / (double(blocks_per_cu) * iters * NV) : 0.0;
    printf("%-34s NV=%2d spread=%d waves/SIMD=%d  %7.3f ms  pipe %5.1f %%  lost/op %6.1f cyc\n", name, NV, SPREAD,
           blocks_per_cu, best, mfma_cycles / cycles * 100, per_op);
    return best;
}

int main()
{
    float *src, *out;
    CK(hipMalloc(&src, 64 * 4096 * 4 * 4 + (1 << 20)));
    CK(hipMemset(src, 0, 64 * 4096 * 4 * 4 + (1 << 20)));
    CK(hipMalloc(&out, 4096));
    for (int w = 1; w <= 2; w++) {
        double b = run<kNone, 0, 0>("MFMA only", src, out, w, 0);
        run<kLoadX4, 8, 0>("global_load_dwordx4 vaddr", src, out, w, b);
        run<kLoadX4, 8, 1>("global_load_dwordx4 vaddr", src, out, w, b);
        run<kLoadX4Saddr, 8, 0>("global_load_dwordx4 saddr+voff", src, out, w, b);
        run<kLoadX1, 8, 0>("global_load_dword vaddr", src, out, w, b);
        run<kGldsX4, 8, 0>("global_load_lds_dwordx4 vaddr", src, out, w, b);
        run<kGldsX4, 8, 1>("global_load_lds_dwordx4 vaddr", src, out, w, b);
        run<kGldsX4Saddr, 8, 0>("global_load_lds_dwordx4 saddr+voff", src, out, w, b);
        run<kGldsX4Saddr, 8, 1>("global_load_lds_dwordx4 saddr+voff", src, out, w, b);
        run<kGldsX1, 8, 0>("global_load_lds_dword vaddr", src, out, w, b);
        run<kDsRead128, 8, 0>("ds_read_b128", src, out, w, b);
        run<kDsWrite128, 8, 0>("ds_write_b128", src, out, w, b);
    }
    return 0;
}
