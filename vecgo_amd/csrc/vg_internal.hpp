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

// Kernel launch with error check.  HIP keeps a sticky "last error" that some successful calls
// latch (e.g. pointer-attribute probes of plain host memory inside hipMemcpyDefault), so the
// state is cleared right before the launch and read right after it.
#define VG_LAUNCH(...)                                                                              \
    do {                                                                                            \
        (void)hipGetLastError();                                                                    \
        hipLaunchKernelGGL(__VA_ARGS__);                                                            \
        hipError_t _le = hipGetLastError();                                                         \
        if (_le != hipSuccess) {                                                                    \
            ::vg::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_le), __FILE__,   \
                            __LINE__);                                                              \
            return VG_ERR_HIP;                                                                      \
        }                                                                                           \
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

// Test hooks: switches that force a slower or alternative path so the tests can compare the paths.  Read from
// the environment ONCE (first use) and changed afterwards only through vg_debug_set_hook — the entry points
// test one relaxed atomic, never getenv.
enum Hook {
    kHookFlatNoSmallTile,     // VG_FLAT_NO_SMALL_TILE   always the 128-query GEMM tile
    kHookFlatUnfused,         // VG_FLAT_UNFUSED         materialise the score matrix
    kHookFlatNoScan,          // VG_FLAT_NO_SCAN         no small-batch register scan
    kHookFlatForceExact,      // VG_FLAT_FORCE_EXACT     exhaustive exact kernel for every query
    kHookFlatNoDma,           // VG_FLAT_NO_DMA          register-staged GEMM
    kHookFlatDebug,           // VG_FLAT_DEBUG           print per-call counters
    kHookProbeNoGroup,        // VG_PROBE_NO_GROUP       one pass per (query, probe) pair
    kHookAdcBigkExhaustive,   // VG_ADC_BIGK_EXHAUSTIVE  LDS-buffer path for k > 64
    kHookBuildDebug,          // VG_BUILD_DEBUG          vg_hnsw_build prints its back-link totals
    kHookKmNoMfma,            // VG_KM_NO_MFMA           k-means assignment by the reference-order kernels only
    kHookKmListAll,           // VG_KM_LIST_ALL          k-means: the matrix scores decide nothing, every point is listed
    kHookKmBf16,              // VG_KM_BF16              k-means: the bfloat16-split passes even for a single assignment
    kHookKmNoBf16,            // VG_KM_NO_BF16           k-means: fp32 matrix passes only
    kHookBruteNoFlat,         // VG_BRUTE_NO_FLAT        vg_search_hnsw_brute: always the distance matrix + replay
    kHookPqNoMfma,            // VG_PQ_NO_MFMA           PQ Encode / Lloyd assignment by the reference-order kernels only
    kHookPqListAll,           // VG_PQ_LIST_ALL          PQ Encode / assignment: every (row, sub-quantizer) pair is listed
    kHookPqFp32Mfma,          // VG_PQ_FP32_MFMA         PQ Encode / assignment: the fp32 matrix form (pq_nominate_kernel)
    kHookProbeNoGemm,         // VG_PROBE_NO_GEMM        probed fp32 scan: the exact kernels only, no matrix-core nomination
    kHookFlatNoBigTile,       // VG_FLAT_NO_BIG_TILE     bf16 nomination: the 128 x 128 tile even above 128 queries
    kHookFlatBigTile2,        // VG_FLAT_BIG_TILE_2      bf16 256 x 256 tile with two row-tile buffers (128 KiB) instead of three
    kHookFlatBigEarlyB,       // VG_FLAT_BIG_EARLY_B     bf16 256 x 256 tile: row-tile fills behind the first matrix group of a step
    kHookPqNomAlways,         // VG_PQ_NOM_ALWAYS        vg_index_enable_pq_nomination: batches of any size take the nomination
    kHookKmNoRanges,          // VG_KM_NO_RANGES         k-means: the listed points' exact pass in one walk over the centroids (no ranges)
    kHookNoCandReplay,        // VG_NO_CAND_REPLAY       no heap replay for queries whose scores may hold a NaN (vg_cand_replay.hpp): what the fast paths alone answer
    kHookCount
};
bool hook(Hook h);

// Per-call HBM scratch.  Blocks come from a process-wide cache keyed by (device, stream): a block
// released by one call is handed to the next call on the SAME stream, where stream order makes
// the reuse safe without waiting.  (The runtime's own stream-ordered pool, hipMallocAsync, was
// dropped: on ROCm 7.2 / gfx950 a block it recycled after a pool trim came back zero-filled
// after the copy into it had completed — see DESIGN.md "Scratch memory".)
int32_t scratch_alloc(void **out, size_t bytes, hipStream_t s);
void scratch_free(void *p);
void scratch_trim(int device);

}  // namespace vg

struct vg_prof_record {
    const char *name;
    hipEvent_t start, stop;
};

// Grow-only scratch arena, one per (context, stream): per-call hipMallocAsync of multi-GB
// buffers costs milliseconds of host time; work enqueued on one stream is ordered, so the next
// call on that stream may reuse the same bytes.
struct vg_arena {
    char *base = nullptr;
    size_t cap = 0;
};

struct vg_ctx {
    int device = 0;
    std::mutex arena_mu;
    std::vector<std::pair<hipStream_t, vg_arena>> arenas;
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
        // hipEventReleaseToDevice: the event's release is device scope.  A default event makes the kernel in front
        // of it end with a SYSTEM-scope release (L2 write-back walk), which a kernel pays only because it is being
        // timed: measured +6..10 us on a 160 us scan that writes 20 KB (HIP: "useful to obtain more precise
        // timings of commands between events").
        if (hipEventCreateWithFlags(&rec.start, hipEventReleaseToDevice) != hipSuccess) return;
        if (hipEventCreateWithFlags(&rec.stop, hipEventReleaseToDevice) != hipSuccess) {
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
        // Host buffers are the slow path (cgo, tests): staged through a cached HBM block
        VG_TRY(scratch_alloc(reinterpret_cast<void **>(&owned), count * sizeof(T), s));
        VG_HIP(hipMemcpyAsync(owned, p, count * sizeof(T), hipMemcpyHostToDevice, s));
        ptr = owned;
        return VG_OK;
    }
    ~DevIn()
    {
        if (owned) scratch_free(owned);
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
        VG_TRY(scratch_alloc(reinterpret_cast<void **>(&owned), n * sizeof(T), s));
        ptr = owned;
        return VG_OK;
    }
    bool on_host() const { return host != nullptr; }
    int32_t finish()
    {
        if (host && count) {
            VG_HIP(hipMemcpyAsync(host, owned, count * sizeof(T), hipMemcpyDeviceToHost, st));
            VG_HIP(hipStreamSynchronize(st));
        }
        return VG_OK;
    }
    ~DevOut()
    {
        if (owned) scratch_free(owned);
    }
};

// Carves 256-byte aligned pieces out of the (context, stream) arena.  Usage: add() every piece,
// then commit() once (grows the arena if needed: synchronises the stream only when it grows),
// then get<T>(i).  The arena mutex is held until the ArenaCall is destroyed, i.e. for the
// duration of the enqueue, which serialises concurrent callers of one (context, stream).
struct ArenaCall {
    vg_ctx *ctx;
    hipStream_t st;
    std::vector<size_t> offs;
    size_t total = 0;
    char *base = nullptr;
    std::unique_lock<std::mutex> lock;
    ArenaCall(vg_ctx *c, hipStream_t s) : ctx(c), st(s), lock(c->arena_mu) {}
    int add(size_t bytes)
    {
        offs.push_back(total);
        total += (bytes + 255) & ~size_t(255);
        return static_cast<int>(offs.size()) - 1;
    }
    int32_t commit()
    {
        vg_arena *a = nullptr;
        for (auto &p : ctx->arenas)
            if (p.first == st) a = &p.second;
        if (!a) {
            ctx->arenas.emplace_back(st, vg_arena{});
            a = &ctx->arenas.back().second;
        }
        if (a->cap < total) {
            if (a->base) {
                VG_HIP(hipStreamSynchronize(st));
                VG_HIP(hipFree(a->base));
                a->base = nullptr;
                a->cap = 0;
            }
            size_t want = total + total / 4;
            VG_HIP(hipMalloc(reinterpret_cast<void **>(&a->base), want));
            a->cap = want;
        }
        base = a->base;
        return VG_OK;
    }
    template <typename T>
    T *get(int i) const
    {
        return reinterpret_cast<T *>(base + offs[static_cast<size_t>(i)]);
    }
};

// Scratch in HBM for the duration of a call (cached block, see scratch_alloc).
template <typename T>
struct DevTmp {
    T *ptr = nullptr;
    hipStream_t st = nullptr;
    int32_t init(size_t count, hipStream_t s)
    {
        st = s;
        if (count == 0) return VG_OK;
        VG_TRY(scratch_alloc(reinterpret_cast<void **>(&ptr), count * sizeof(T), s));
        return VG_OK;
    }
    ~DevTmp()
    {
        if (ptr) scratch_free(ptr);
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

struct vg_sq8;
struct vg_int4;

namespace vg {
// Partition-probed scans (k_probe.hip): (query, probe) pairs bucketed by partition and cut into groups
// of up to kProbeQB pairs; one workgroup scores a slice of the partition's rows against a whole group.
constexpr int kProbeQB = 8;
struct ProbeGroup {
    uint32_t part;   // partition
    uint32_t first;  // first entry of the group in pair_of[]
    uint32_t count;  // 1..kProbeQB
};
// The grouped nomination of the probed scans (flat_probe_gemm, k_flat.hip), where the caller re-scores with its own exact
// distance (SQ8: launch_sq8_verify, k_sq8.hip): per (query, probe) pair its thresholds, how many rows fell below the last one,
// and those rows.
struct ProbeNominated {
    float *thr;            // [pairs, sel_k]: the pair's threshold is entry sel_k - 1
    int *counts;           // [pairs]
    uint32_t *cand_id;     // [pairs, 64]: the 64 best appended rows ascending by nomination score (k <= 48 only)
    float *cand_sc;
    int cap;               // appended keys kept per pair (counts above it: rows were dropped)
    int sel_k;
    const uint64_t *cand;  // [pairs, cap]: every appended key (k > 48: all of them are re-scored)
};
constexpr int kProbeGemmMaxK = 160;  // deepest threshold: the 60th best of the 1/8 row sample, ~480 rows pass it
}  // namespace vg

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
    float *d_norm_max = nullptr;  // [1] max ||x||^2: error bound of the GEMM-form scores
    uint16_t *d_vectors_bf16 = nullptr;  // optional bfloat16 copy of the rows: vg_index_enable_bf16_filter
    int32_t vectors_bf16_dim = 0;        // its row length: dim padded with zeros to whole 64-element K steps of the bf16 GEMM
    unsigned long long *d_flat_stats = nullptr;  // [2] queries searched, queries sent to the exhaustive kernel
    // RaBitQ: sign bits re-tiled [tile][group][lane][16 B] and the stored norms
    uint8_t *d_rq_tiles = nullptr;
    float *d_rq_norms = nullptr;
    int32_t rq_groups = 0;  // ceil(((dim+63)/64*8) / 16)
    // HNSW adjacency
    uint32_t *d_hnsw_l0 = nullptr;     // n*m0
    float *d_hnsw_l0_dist = nullptr;   // n*m0 cached edge distances (Neighbor.Dist) for the predicate-aware walk, or null
    uint8_t *d_hnsw_tomb = nullptr;    // g.tombstones as a bitmap (ceil(n/8) bytes), or null: no deleted node
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
    uint16_t *d_pq_bf16 = nullptr;     // vg_index_enable_pq_nomination: the DECODED rows (pq.go:185-229) rounded to bfloat16, n*pq_bf16_dim, or null
    int32_t pq_bf16_dim = 0;           // its row length: dim padded with zeros to whole 64-element K steps of the bf16 GEMM
    float *d_pq_norms = nullptr;       // ... their |x^|^2 (fp32 of the unrounded values), n, and the largest of them
    float *d_pq_norm_max = nullptr;
    uint8_t *d_rq_rows = nullptr;
    // SQ8 codes, re-tiled like the PQ codes: [tile][group of 16 dims][lane][16 B]; see k_sq8.hip
    vg_sq8 *sq = nullptr;
    uint16_t *d_sq_bf16 = nullptr;     // vg_index_enable_sq8_nomination: the dequantised codes rounded to bfloat16, n*sq_bf16_dim, or null
    int32_t sq_bf16_dim = 0;           // its row length: dim padded with zeros to whole 64-element K steps of the bf16 GEMM
    float *d_sq_norms = nullptr;       // ... their |x^|^2 (fp32 of the unrounded values), n, and the largest of them
    float *d_sq_norm_max = nullptr;
    uint8_t *d_sq_tiles = nullptr;
    int32_t sq_groups = 0;  // ceil(dim/16)
    // INT4 codes of a DiskANN segment, row-major n * ceil(dim/2), and the quantizer's lookup table
    uint8_t *d_int4_rows = nullptr;
    const float *int4_table = nullptr;  // borrowed from the vg_int4
    const float *int4_min = nullptr, *int4_diff = nullptr;  // (same)
    // IVF partitions of a flat segment: centroids [P*dim], first row of every partition [P+1]
    float *d_centroids = nullptr;
    uint32_t *d_part_off = nullptr;
    std::vector<uint32_t> h_part_off;  // the same on the host (launch bounds of the grouped GEMM, k_probe.hip)
    int32_t num_partitions = 0;
};
