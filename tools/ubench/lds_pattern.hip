// LDS ds_read_b32 throughput for arbitrary per-lane address patterns (host-generated).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <functional>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
constexpr int WORDS = 24576;  // 96 KiB
__global__ __launch_bounds__(1024) void gather(const int *idx0, float *out, int iters, int delta)
{
    extern __shared__ float lut[];
    for (int i = threadIdx.x; i < WORDS; i += blockDim.x) lut[i] = (float)(i & 1023);
    __syncthreads();
    int idx[16];
    for (int s = 0; s < 16; s++) idx[s] = idx0[(blockIdx.x * blockDim.x + threadIdx.x) * 16 + s];
    float acc[16], vals[16];
    for (int i = 0; i < 16; i++) acc[i] = 0.f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int s = 0; s < 16; s++) {
            float v;
            asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(idx[s] * 4));
            vals[s] = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int s = 0; s < 16; s++) acc[s] += vals[s];
    }
    float t = 0;
    for (int i = 0; i < 16; i++) t += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}
void run(const char *name, int threads, std::function<int(int lane, int s)> f)
{
    int blocks = 256, iters = 2000;
    std::vector<int> h((size_t)blocks * threads * 16);
    for (int b = 0; b < blocks; b++) for (int t = 0; t < threads; t++) for (int s = 0; s < 16; s++)
        h[((size_t)b * threads + t) * 16 + s] = f(t & 63, s) & (WORDS / 2 - 1);
    int *d; float *o;
    CK(hipMalloc(&d, h.size() * 4)); CK(hipMalloc(&o, blocks * threads * 4));
    CK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute((const void *)gather, hipFuncAttributeMaxDynamicSharedMemorySize, WORDS * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    gather<<<blocks, threads, WORDS * 4>>>(d, o, iters, 64);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    gather<<<blocks, threads, WORDS * 4>>>(d, o, iters, 64);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    double lookups = (double)blocks * threads * iters * 16;
    printf("%-28s threads=%4d: %7.3f ms  %6.2f lookups/clk/CU @2.4GHz\n", name, threads, ms,
           lookups / (ms * 1e-3) / 256 / 2.4e9);
    CK(hipFree(d)); CK(hipFree(o));
}
int R() { return rand() & 0x7fffffff; }
int main()
{
    for (int threads : {512, 1024}) {
        run("linear idx=lane", threads, [](int l, int s) { return l + 64 * s; });
        run("broadcast", threads, [](int l, int s) { return 7 + s; });
        run("random word", threads, [](int l, int s) { return R(); });
        run("natural c*1 (row=s)", threads, [](int l, int s) { return s * 256 + (R() & 255); });
        run("rand*32 + lane&31", threads, [](int l, int s) { return R() * 32 + (l & 31); });
        run("rand*64 + lane", threads, [](int l, int s) { return R() * 64 + l; });
        run("rand*16 + lane&15", threads, [](int l, int s) { return R() * 16 + (l & 15); });
        run("rot-ideal (par=lane>>4&1)", threads, [](int l, int s) { return ((R() & ~1) | ((l >> 4) & 1)) * 16 + ((l + s) & 15); });
        run("rand*32 + (lane>>1)&31", threads, [](int l, int s) { return R() * 32 + ((l >> 1) & 31); });
        run("rand*32+(l&15)+16*(l>>5)", threads, [](int l, int s) { return R() * 32 + (l & 15) + 16 * (l >> 5); });
        run("rand*64+(l&15)+16*(l>>4)", threads, [](int l, int s) { return R() * 64 + (l & 15) + 16 * (l >> 4); });
        run("rand*64+(l&31)+32*(l>>5)^..", threads, [](int l, int s) { return R() * 64 + (l & 31) + 32 * (((l >> 5) ^ R()) & 1); });
    }
}
