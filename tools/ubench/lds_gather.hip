// LDS random-gather microbenchmark: which LUT layouts are bank-conflict free for ds_read_b32?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)

// MODE 0: natural  idx = j*256 + c                 (j = subquantizer 0..95)
// MODE 1: rotated  idx = (g*256 + c)*16 + ((s+r)&15)
// MODE 2: same address for all lanes (broadcast, conflict free lower bound)
// MODE 3: idx = (g*256+c)*16 + (lane&15)  with lanes i, i+16 forced different c parity (ideal 32 banks)
// MODE 4: linear idx = lane (+ 64*s)  (perfectly conflict free)
template <int MODE>
__global__ __launch_bounds__(1024) void gather(const uint32_t *codes, float *out, int iters, int zero)
{
    extern __shared__ float lut[];
    for (int i = threadIdx.x; i < 96 * 256; i += blockDim.x) lut[i] = (float)(i & 1023);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int r = lane & 15;
    uint32_t w[4];
    for (int i = 0; i < 4; i++) w[i] = codes[(blockIdx.x * blockDim.x + threadIdx.x) * 4 + i];
    float acc[16];
    for (int i = 0; i < 16; i++) acc[i] = 0.f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int g = 0; g < 6; g++) {
#pragma unroll
            for (int s = 0; s < 16; s++) {
                uint32_t c = (w[s >> 2] >> (8 * (s & 3))) & 0xFF;
                int idx;
                if (MODE == 0) idx = (g * 16 + s) * 256 + c;
                else if (MODE == 1) idx = (g * 256 + c) * 16 + ((s + r) & 15);
                else if (MODE == 2) idx = (g * 16 + s) * 256 + 7 + c * zero;
                else if (MODE == 3) idx = (g * 256 + ((c & 0xFE) | ((lane >> 4) & 1))) * 16 + ((s + r) & 15);
                else if (MODE == 4) idx = (((g * 16 + s) * 64 + lane) & (96 * 256 - 1)) + c * zero;
                else if (MODE == 5) idx = (g * 256 + c) * 16 + ((s + r) & 15);
                else idx = ((g*256 + c) * 32 + (lane & 31)) & (96*256-1);
                if (MODE == 5) acc[s] += __int_as_float(idx); else acc[s] += lut[idx];
            }
        }
        // evolve codes so the compiler cannot hoist
        for (int i = 0; i < 4; i++) w[i] = w[i] * 1664525u + 1013904223u;
    }
    float t = 0;
    for (int i = 0; i < 16; i++) t += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}

template <int MODE>
void run(const char *name, int threads)
{
    int blocks = 256, iters = 200;
    std::vector<uint32_t> h(blocks * threads * 4);
    for (auto &x : h) x = (uint32_t)rand() * 2654435761u + rand();
    uint32_t *d; float *o;
    CK(hipMalloc(&d, h.size() * 4)); CK(hipMalloc(&o, blocks * threads * 4));
    CK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute((const void *)gather<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    gather<MODE><<<blocks, threads, 96 * 1024>>>(d, o, iters, 0);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    gather<MODE><<<blocks, threads, 96 * 1024>>>(d, o, iters, 0);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    double lookups = (double)blocks * threads * iters * 96;
    printf("%-10s threads=%4d: %8.3f ms  %6.2f T lookups/s  (%.2f lookups/clk/CU @2.4GHz)\n", name, threads, ms,
           lookups / ms / 1e9, lookups / (ms * 1e-3) / 256 / 2.4e9);
    CK(hipFree(d)); CK(hipFree(o));
}
int main()
{
    for (int threads : {512, 1024}) {
        run<0>("natural", threads);
        run<1>("rotated", threads);
        run<2>("broadcast", threads);
        run<3>("rot-ideal", threads);
        run<4>("linear", threads);
        run<5>("valu-only", threads);
        run<6>("lane-bank", threads);
    }
}
