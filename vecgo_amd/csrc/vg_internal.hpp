// vg_internal.hpp — host-side plumbing shared by the C-ABI translation units.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/vecgo_hip.h"

#define VG_API extern "C" __attribute__((visibility("default")))

namespace vg {

void set_error(const char *fmt, ...);

#define VG_HIP(expr)                                                                          \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            ::vg::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,   \
                            __LINE__);                                                        \
            (void)hipGetLastError();                                                          \
            return _e == hipErrorOutOfMemory ? VG_ERR_OUT_OF_MEMORY : VG_ERR_HIP;             \
        }                                                                                     \
    } while (0)

#define VG_CHECK(cond, status, ...)          \
    do {                                     \
        if (!(cond)) {                       \
            ::vg::set_error(__VA_ARGS__);    \
            return (status);                 \
        }                                    \
    } while (0)

#define VG_TRY(expr)                   \
    do {                               \
        int32_t _s = (expr);           \
        if (_s != VG_OK) return _s;    \
    } while (0)

bool is_device_ptr(const void *p);

}  // namespace vg

struct vg_prof_record {
    const char *name;
    hipEvent_t start, stop;
};

struct vg_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    int compute_units = 0;
    int64_t hbm_bytes = 0;
    char arch[64] = {0};
    bool profiling = false;
    std::mutex prof_mu;
    std::vector<vg_prof_record> prof;
};

namespace vg {
// RAII bracket around a kernel launch: records an event pair on `st` when profiling is on
struct ProfScope {
    vg_ctx *ctx;
    hipStream_t st;
    vg_prof_record rec{};
    bool active = false;
    ProfScope(vg_ctx *c, const char *name, hipStream_t s) : ctx(c), st(s)
    {
        if (!c->profiling) return;
        if (hipEventCreate(&rec.start) != hipSuccess) return;
        if (hipEventCreate(&rec.stop) != hipSuccess) {
            (void)hipEventDestroy(rec.start);
            return;
        }
        rec.name = name;
        (void)hipEventRecord(rec.start, st);
        active = true;
    }
    ~ProfScope()
    {
        if (!active) return;
        (void)hipEventRecord(rec.stop, st);
        std::lock_guard<std::mutex> g(ctx->prof_mu);
        ctx->prof.push_back(rec);
    }
};
}  // namespace vg

namespace vg {

inline hipStream_t pick_stream(vg_ctx *ctx, void *stream)
{
    return stream ? reinterpret_cast<hipStream_t>(stream) : ctx->stream;
}

// Read-only input that may live on the host: staged into HBM for the call.
template <typename T>
struct DevIn {
    const T *ptr = nullptr;
    T *owned = nullptr;
    hipStream_t st = nullptr;
    int32_t init(const T *p, size_t count, hipStream_t s)
    {
        st = s;
        if (count == 0 || p == nullptr) {
            ptr = p;
            return VG_OK;
        }
        if (is_device_ptr(p)) {
            ptr = p;
            return VG_OK;
        }
        VG_HIP(hipMallocAsync(reinterpret_cast<void **>(&owned), count * sizeof(T), s));
        VG_HIP(hipMemcpyAsync(owned, p, count * sizeof(T), hipMemcpyHostToDevice, s));
        ptr = owned;
        return VG_OK;
    }
    ~DevIn()
    {
        if (owned) (void)hipFreeAsync(owned, st);
    }
};

// Output that may live on the host: produced in HBM, copied back by finish().
template <typename T>
struct DevOut {
    T *ptr = nullptr;
    T *owned = nullptr;
    T *host = nullptr;
    size_t count = 0;
    hipStream_t st = nullptr;
    int32_t init(T *p, size_t n, hipStream_t s)
    {
        st = s;
        count = n;
        if (n == 0 || p == nullptr) {
            ptr = p;
            return VG_OK;
        }
        if (is_device_ptr(p)) {
            ptr = p;
            return VG_OK;
        }
        host = p;
        VG_HIP(hipMallocAsync(reinterpret_cast<void **>(&owned), n * sizeof(T), s));
        ptr = owned;
        return VG_OK;
    }
    bool on_host() const { return host != nullptr; }
    int32_t finish()
    {
        if (host && count) {
            VG_HIP(hipMemcpyAsync(host, owned, count * sizeof(T), hipMemcpyDeviceToHost, st));
        }
        return VG_OK;
    }
    ~DevOut()
    {
        if (owned) (void)hipFreeAsync(owned, st);
    }
};

// Scratch in HBM for the duration of a call (stream-ordered pool allocation).
template <typename T>
struct DevTmp {
    T *ptr = nullptr;
    hipStream_t st = nullptr;
    int32_t init(size_t count, hipStream_t s)
    {
        st = s;
        if (count == 0) return VG_OK;
        VG_HIP(hipMallocAsync(reinterpret_cast<void **>(&ptr), count * sizeof(T), s));
        return VG_OK;
    }
    ~DevTmp()
    {
        if (ptr) (void)hipFreeAsync(ptr, st);
    }
};

}  // namespace vg

struct vg_pq {
    vg_ctx *ctx = nullptr;
    int32_t dim = 0, m = 0, k = 0, subdim = 0;
    bool trained = false;
    int8_t *d_codebooks = nullptr;  // m*k*subdim
    float *d_scales = nullptr;      // m
    float *d_offsets = nullptr;     // m
};

struct vg_index {
    vg_ctx *ctx = nullptr;
    int64_t n = 0;
    int32_t dim = 0;
    int32_t metric = 0;
    // PQ codes, re-tiled: [tile][group][lane][16 B]; see k_adc.hip
    vg_pq *pq = nullptr;
    uint8_t *d_pq_tiles = nullptr;
    int64_t n_tiles = 0;
    int32_t pq_groups = 0;  // ceil(m/16)
    // fp32 rows, row-major n*dim (reference layout), plus ||x||^2 for the GEMM path
    float *d_vectors = nullptr;
    float *d_norms = nullptr;
    // RaBitQ: sign bits re-tiled [tile][group][lane][16 B] and the stored norms
    uint8_t *d_rq_tiles = nullptr;
    float *d_rq_norms = nullptr;
    int32_t rq_groups = 0;  // ceil(((dim+63)/64*8) / 16)
    // HNSW adjacency
    uint32_t *d_hnsw_l0 = nullptr;     // n*m0
    uint32_t *d_hnsw_slot = nullptr;   // max_level*n
    uint32_t *d_hnsw_adj = nullptr;    // concatenated level tables
    int64_t *d_hnsw_level_off = nullptr;  // max_level+1 row offsets into d_hnsw_adj (in rows)
    int32_t hnsw_m0 = 0, hnsw_m = 0, hnsw_max_level = 0;
    uint32_t hnsw_entry = 0;
    // Vamana adjacency
    uint32_t *d_vamana = nullptr;      // n*r
    int32_t vamana_r = 0;
    uint32_t vamana_entry = 0;
    // PQ codes in the reference's row-major layout (random access by node id in graph search)
    uint8_t *d_pq_rows = nullptr;
    uint8_t *d_rq_rows = nullptr;
};
