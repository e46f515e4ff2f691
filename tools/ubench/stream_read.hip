// stream_read.hip — how fast can gfx950 stream a buffer that does not fit any cache, by load flavour
// (plain, nontemporal builtin, explicit cache-policy bits) and bytes in flight per lane?
// hipcc --offload-arch=gfx950 -O3 -std=c++17 stream_read.hip -o stream_read && ./stream_read
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>
__device__ __forceinline__ f4 ld(const f4 *p)
{
    if (MODE == 1) return __builtin_nontemporal_load(p);
    if (MODE == 2) {
        f4 v;
        asm volatile("global_load_dwordx4 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        return v;
    }
    if (MODE == 3) {
        f4 v;
        asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        return v;
    }
    return *p;
}

// every wave streams whole 1 KiB lines; UNROLL independent loads in flight per lane
template <int MODE, int UNROLL>
__global__ __launch_bounds__(256) void stream_kernel(const f4 *__restrict__ src, size_t n4, float *__restrict__ out)
{
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    f4 acc = {0, 0, 0, 0};
    for (; i + (UNROLL - 1) * stride < n4; i += UNROLL * stride) {
        f4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) v[u] = ld<MODE>(src + i + u * stride);
#pragma unroll
        for (int u = 0; u < UNROLL; u++) acc += v[u];
    }
    for (; i < n4; i += stride) acc += ld<MODE>(src + i);
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = 1.0f;
}

template <int MODE, int UNROLL>
static void run(const f4 *src, size_t n4, float *out, int blocks, const char *name)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 2; w++) stream_kernel<MODE, UNROLL><<<blocks, 256>>>(src, n4, out);
    hipEventRecord(e0);
    const int reps = 10;
    for (int r = 0; r < reps; r++) stream_kernel<MODE, UNROLL><<<blocks, 256>>>(src, n4, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-14s unroll %d blocks %5d: %7.1f us  %6.2f TB/s\n", name, UNROLL, blocks, ms / reps * 1e3,
           n4 * 16.0 / (ms / reps * 1e-3) / 1e12);
}

int main()
{
    const size_t bytes = size_t(3) << 30;  // 3 GiB: the 1M x 768 fp32 corpus
    const size_t n4 = bytes / 16;
    f4 *src;
    float *out;
    hipMalloc(&src, bytes);
    hipMalloc(&out, 4);
    hipMemset(src, 0, bytes);
    for (int blocks : {1024, 2048, 4096, 8192}) {
        run<0, 4>(src, n4, out, blocks, "plain");
        run<0, 8>(src, n4, out, blocks, "plain");
        run<1, 4>(src, n4, out, blocks, "nontemporal");
        run<1, 8>(src, n4, out, blocks, "nontemporal");
    }
    run<2, 1>(src, n4, out, 8192, "asm nt (1)");
    run<3, 1>(src, n4, out, 8192, "asm sc0 sc1");
    return 0;
}
