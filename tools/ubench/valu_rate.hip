// What does one vector instruction of the nomination scans cost on a SIMD, by waves per SIMD?  
// Per opcode: CH independent chains per lane (8: no dependency stall; 1: the dependent-chain latency), 4096 instructions per
// chain per wave, W waves per SIMD (1, 2, 3, 4), one workgroup of 4 * W waves per CU on all 256 CUs.  Prints cycles per
// wave-instruction per SIMD from the shader clock (s_memtime around the loop, median-free: max over the CU's waves is what the
// SIMD needed; the median over all waves is printed) — 2.0 = the SIMD-32 rate of MI355X_MICROARCH.md's 'v_fma_f32 (wave64) 2 cyc', 4.0 = half of it.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                        \
    do {                                                             \
        hipError_t e = (x);                                          \
        if (e != hipSuccess) {                                       \
            printf("%s: %s\n", #x, hipGetErrorString(e));            \
            exit(1);                                                 \
        }                                                            \
    } while (0)

// one instruction of kind OP on chain register x (operands y, z: loop-invariant VGPRs)
template <int OP>
__device__ __forceinline__ void one(uint32_t &x, uint32_t y, uint32_t z)
{
    if (OP == 0) asm volatile("v_min_u32 %0, %0, %1" : "+v"(x) : "v"(y));
    if (OP == 1) asm volatile("v_med3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
    if (OP == 2) asm volatile("v_and_or_b32 %0, %0, %1, 37" : "+v"(x) : "v"(y));
    if (OP == 3) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
    if (OP == 4) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(x) : "v"(y));
    if (OP == 5) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
    if (OP == 6) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(y));
    if (OP == 7) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x) : "v"(y));
    if (OP == 8) asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(x) : "v"(y));
    if (OP == 9) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
    if (OP == 10) asm volatile("v_min_f32 %0, %0, %1" : "+v"(x) : "v"(y));
    if (OP == 11) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
    if (OP == 12) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
    if (OP == 13) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(y) : "vcc");  // 2 instructions
    if (OP == 14) asm volatile("v_pk_min_f16 %0, %0, %1" : "+v"(x) : "v"(y));
    if (OP == 15) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(x) : "v"(y));
    if (OP == 16) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(y));  // needs register pairs: see the kernel
}
static const char *kNames[] = {"v_min_u32", "v_med3_u32", "v_and_or_b32", "v_min3_u32", "v_pk_min_u16", "v_fma_f32", "v_add_f32",
                               "v_cvt_pk_bf16_f32", "v_pk_max_u16", "v_perm_b32", "v_min_f32", "v_med3_f32", "v_min3_f32",
                               "v_cmp_lt_u32+v_cndmask (2 instr)", "v_pk_min_f16", "v_cvt_pkrtz_f16_f32"};
constexpr int kOps = 16;
constexpr int kIters = 512;  // x 8 instructions per chain per iteration

template <int OP, int CH>
__global__ void rate(uint32_t *out, long long *cycles, uint32_t seed)
{
    uint32_t x[8];
    for (int i = 0; i < 8; i++) x[i] = seed * (threadIdx.x + 1) + i * 77u;
    uint32_t y = seed ^ 0x3f801234u, z = seed + 0x40001234u;
    asm volatile("" : "+v"(y), "+v"(z));
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < kIters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int c = 0; c < CH; c++) one<OP>(x[c], y, z);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    uint32_t r = 0;
    for (int i = 0; i < CH; i++) r ^= x[i];
    if (r == 0x12345u) out[0] = r;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int OP, int CH>
static void run(int w, uint32_t *out, long long *dcyc, std::vector<long long> &h)
{
    const int threads = 256 * w, blocks = 256;
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL((rate<OP, CH>), dim3(blocks), dim3(threads), 0, 0, out, dcyc, 12345u + rep);
        CK(hipDeviceSynchronize());
    }
    const int waves = blocks * threads / 64;
    CK(hipMemcpy(h.data(), dcyc, waves * sizeof(long long), hipMemcpyDeviceToHost));
    std::vector<long long> v(h.begin(), h.begin() + waves);
    std::sort(v.begin(), v.end());
    const double med = static_cast<double>(v[waves / 2]);
    const double cyc = med;  // s_memtime counts shader-clock cycles (MI355X_MICROARCH.md, 'DVFS give-back' item 6)
    const double instr = static_cast<double>(kIters) * 8 * CH * (OP == 13 ? 2 : 1);
    printf("  W=%d CH=%d: %.2f cycles per wave-instruction per SIMD (per wave: %.2f)\n", w, CH, cyc / (instr * w), cyc / instr);
}

template <int OP>
static void both(uint32_t *out, long long *dcyc, std::vector<long long> &h)
{
    printf("%s\n", kNames[OP]);
    for (int w : {1, 2, 3, 4}) run<OP, 8>(w, out, dcyc, h);
    run<OP, 1>(1, out, dcyc, h);
    run<OP, 2>(1, out, dcyc, h);
}

int main()
{
    uint32_t *out;
    long long *dcyc;
    CK(hipMalloc(&out, 64));
    CK(hipMalloc(&dcyc, 256 * 16 * sizeof(long long)));
    std::vector<long long> h(256 * 16);
    both<0>(out, dcyc, h);
    both<1>(out, dcyc, h);
    both<2>(out, dcyc, h);
    both<3>(out, dcyc, h);
    both<4>(out, dcyc, h);
    both<5>(out, dcyc, h);
    both<6>(out, dcyc, h);
    both<7>(out, dcyc, h);
    both<8>(out, dcyc, h);
    both<9>(out, dcyc, h);
    both<10>(out, dcyc, h);
    both<11>(out, dcyc, h);
    both<12>(out, dcyc, h);
    both<13>(out, dcyc, h);
    both<14>(out, dcyc, h);
    both<15>(out, dcyc, h);
    return 0;
}
