// Do LDS lookups (ds_read_b32, 64 lanes on 16 consecutive dwords: the INT4 table pattern) and vector-ALU work overlap
// on a CU, or do they share an issue resource?  Three kernels per (waves, V): L lookups per iteration alone, V packed
// FMAs + V/2.. perms alone, and both interleaved the way int4_scan_tab_kernel interleaves them (8 lookups in flight
// behind a counted wait).  Prints clocks per iteration per CU (one workgroup per CU) at 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
typedef float f2 __attribute__((ext_vector_type(2)));

template <int OFF>
__device__ __forceinline__ float lds_read(uint32_t addr)
{
    float v;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
struct V8 { float v[8]; };
template <int N>
__device__ __forceinline__ void lds_wait(V8 &x)
{
    asm volatile("s_waitcnt lgkmcnt(%8)"
                 : "+v"(x.v[0]), "+v"(x.v[1]), "+v"(x.v[2]), "+v"(x.v[3]), "+v"(x.v[4]), "+v"(x.v[5]), "+v"(x.v[6]), "+v"(x.v[7])
                 : "n"(N));
}
__device__ __forceinline__ void issue8(V8 &x, uint32_t w, uint32_t base)
{
    const uint32_t hi4 = (w >> 2) & 0x3C3C3C3Cu, lo4 = (w << 2) & 0x3C3C3C3Cu;
#define ONE(K) x.v[2*K] = lds_read<(2*K)*64>(__builtin_amdgcn_perm(base, hi4, 0x07060500u | K)); \
               x.v[2*K+1] = lds_read<(2*K+1)*64>(__builtin_amdgcn_perm(base, lo4, 0x07060500u | K));
    ONE(0) ONE(1) ONE(2) ONE(3)
#undef ONE
}

// MODE 0: lookups only, 1: ALU only (the same perms + fmas on register data), 2: both; DEPTH = words in flight (1..4)
template <int MODE, int EXTRA, int DEPTH>
__global__ __launch_bounds__(1024) void probe(const uint32_t *words, float *out, int iters)
{
    extern __shared__ float lut[];
    for (int i = threadIdx.x; i < 12288; i += blockDim.x) lut[i] = (float)(i & 255);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    uint32_t w = words[threadIdx.x];
    f2 acc[8];
    for (int i = 0; i < 8; i++) acc[i] = f2{0.f, 0.f};
    f2 ex[8];
    for (int i = 0; i < 8; i++) ex[i] = f2{1.0f + lane, 0.5f};
    V8 q[DEPTH];
    uint32_t base = 0;
    if (MODE != 1) {
        for (int d = 0; d < DEPTH; d++) issue8(q[d], w + d, base);
    } else {
        for (int d = 0; d < DEPTH; d++) for (int k = 0; k < 8; k++) q[d].v[k] = lane * 0.25f + k;
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            if (MODE != 1) {
                if (DEPTH == 1) lds_wait<0>(q[d]);
                else if (DEPTH == 2) lds_wait<8>(q[d]);
                else lds_wait<15>(q[d]);  // lgkmcnt is 4 bits: <= 15 outstanding of 24 (32) in flight = the oldest 9 (17) are back
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const f2 dd = {q[d].v[2 * k], q[d].v[2 * k + 1]};
                acc[(4 * d + k) & 7] = __builtin_elementwise_fma(dd, dd, acc[(4 * d + k) & 7]);
            }
#pragma unroll
            for (int e = 0; e < EXTRA; e++) ex[e & 7] = __builtin_elementwise_fma(ex[e & 7], ex[(e + 1) & 7], ex[e & 7]);
            w = w * 1664525u + 1013904223u;
            base = (base + 2048u) & 0x7FFFu;
            if (MODE != 1) {
                issue8(q[d], w, base);
            } else {  // the address work without the reads
                const uint32_t hi4 = (w >> 2) & 0x3C3C3C3Cu, lo4 = (w << 2) & 0x3C3C3C3Cu;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    q[d].v[2 * k] = __uint_as_float(__builtin_amdgcn_perm(base, hi4, 0x07060500u | k) | 0x3f000000u);
                    q[d].v[2 * k + 1] = __uint_as_float(__builtin_amdgcn_perm(base, lo4, 0x07060500u | k) | 0x3f000000u);
                }
            }
        }
    }
    if (MODE != 1) for (int d = 0; d < DEPTH; d++) lds_wait<0>(q[d]);
    float t = 0;
    for (int i = 0; i < 8; i++) t += acc[i].x + acc[i].y + ex[i].x + ex[i].y;
    for (int d = 0; d < DEPTH; d++) for (int k = 0; k < 8; k++) t += q[d].v[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}

template <int MODE, int EXTRA, int DEPTH>
double run(int threads, const uint32_t *d, float *o)
{
    const int blocks = 256, iters = 4000;
    CK(hipFuncSetAttribute((const void *)probe<MODE, EXTRA, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    probe<MODE, EXTRA, DEPTH><<<blocks, threads, 100 * 1024>>>(d, o, iters);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    probe<MODE, EXTRA, DEPTH><<<blocks, threads, 100 * 1024>>>(d, o, iters);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    // clocks per 8 lookups (one word) per wave
    return ms * 1e-3 * 2.4e9 / ((double)iters * DEPTH);
}

template <int EXTRA, int DEPTH>
void row(int threads, const uint32_t *d, float *o)
{
    const double l = run<0, EXTRA, DEPTH>(threads, d, o), v = run<1, EXTRA, DEPTH>(threads, d, o), b = run<2, EXTRA, DEPTH>(threads, d, o);
    const int waves = threads / 64;
    printf("waves=%2d depth=%d extra_pk_fma=%2d | clk per word (8 lookups) per WAVE: lds %6.1f  alu %6.1f  both %6.1f | per CU per wave-lookup: lds %5.2f alu %5.2f both %5.2f  (sum %5.2f max %5.2f)\n",
           waves, DEPTH, EXTRA, l, v, b, l / 8 / waves, v / 8 / waves, b / 8 / waves, (l + v) / 8 / waves, (l > v ? l : v) / 8 / waves);
}

int main()
{
    uint32_t *d; float *o;
    CK(hipMalloc(&d, 1024 * 4)); CK(hipMalloc(&o, 256 * 1024 * 4));
    uint32_t h[1024];
    for (int i = 0; i < 1024; i++) h[i] = rand() * 2654435761u;
    CK(hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice));
    for (int threads : {768, 1024}) {
        row<0, 2>(threads, d, o);
        row<0, 4>(threads, d, o);
        row<4, 2>(threads, d, o);
        row<4, 4>(threads, d, o);
        row<8, 2>(threads, d, o);
        row<8, 4>(threads, d, o);
        row<16, 4>(threads, d, o);
    }
    return 0;
}
