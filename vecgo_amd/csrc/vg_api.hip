// vg_api.hip — context, error plumbing and handle lifetime of the C ABI.
#include "vg_internal.hpp"

namespace vg {

static thread_local std::string g_last_error;

void set_error(const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

bool is_device_ptr(const void *p)
{
    if (!p) return false;
    hipPointerAttribute_t a;
    hipError_t e = hipPointerGetAttributes(&a, p);
    // plain malloc'd host memory: some ROCm versions return an error, others succeed with
    // hipMemoryTypeUnregistered but still latch a sticky "last error" — clear it either way
    (void)hipGetLastError();
    if (e != hipSuccess) return false;
    return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

namespace {
struct ScratchBlock {
    void *p;
    size_t cap;
    int device;
    hipStream_t st;
    bool in_use;
};
std::mutex g_scratch_mu;
std::vector<ScratchBlock> g_scratch;

size_t scratch_round(size_t bytes)
{
    if (bytes <= (size_t(1) << 20)) {  // powers of two from 256 B to 1 MiB
        size_t c = 256;
        while (c < bytes) c <<= 1;
        return c;
    }
    const size_t mib = size_t(1) << 20;
    return (bytes + bytes / 8 + mib - 1) / mib * mib;  // 12.5 % slack, whole MiB
}
}  // namespace

int32_t scratch_alloc(void **out, size_t bytes, hipStream_t s)
{
    *out = nullptr;
    if (bytes == 0) return VG_OK;
    int dev = 0;
    VG_HIP(hipGetDevice(&dev));
    const size_t want = scratch_round(bytes);
    {
        std::lock_guard<std::mutex> g(g_scratch_mu);
        ScratchBlock *best = nullptr;
        for (auto &b : g_scratch) {
            if (b.in_use || b.device != dev || b.st != s || b.cap < bytes || b.cap > 2 * want) continue;
            if (!best || b.cap < best->cap) best = &b;
        }
        if (best) {
            best->in_use = true;
            *out = best->p;
            return VG_OK;
        }
    }
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {  // out of HBM: give the cached, idle blocks back and retry once
        (void)hipGetLastError();
        scratch_trim(dev);
        e = hipMalloc(&p, want);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        set_error("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        return VG_ERR_HIP;
    }
    std::lock_guard<std::mutex> g(g_scratch_mu);
    g_scratch.push_back(ScratchBlock{p, want, dev, s, true});
    *out = p;
    return VG_OK;
}

void scratch_free(void *p)
{
    if (!p) return;
    std::lock_guard<std::mutex> g(g_scratch_mu);
    for (auto &b : g_scratch)
        if (b.p == p) {
            b.in_use = false;
            return;
        }
}

// Returns every idle block of `device` to the driver.  Work that last used them may still be in
// flight, so the device is drained first.
void scratch_trim(int device)
{
    std::vector<void *> victims;
    {
        std::lock_guard<std::mutex> g(g_scratch_mu);
        size_t w = 0;
        for (size_t i = 0; i < g_scratch.size(); i++) {
            if (!g_scratch[i].in_use && g_scratch[i].device == device)
                victims.push_back(g_scratch[i].p);
            else
                g_scratch[w++] = g_scratch[i];
        }
        g_scratch.resize(w);
    }
    if (victims.empty()) return;
    (void)hipDeviceSynchronize();
    for (void *p : victims) (void)hipFree(p);
}

}  // namespace vg

VG_API int32_t vg_abi_version(void) { return VG_ABI_VERSION; }
VG_API int32_t vg_abi_minor(void) { return VG_ABI_MINOR; }

VG_API const char *vg_last_error(void) { return vg::g_last_error.c_str(); }

VG_API const char *vg_status_string(int32_t s)
{
    switch (s) {
    case VG_OK: return "ok";
    case VG_ERR_INVALID_ARG: return "invalid argument";
    case VG_ERR_DIM_MISMATCH: return "vector dimension mismatch";
    case VG_ERR_NOT_TRAINED: return "ProductQuantizer not trained";
    case VG_ERR_CODE_LENGTH: return "codes length mismatch";
    case VG_ERR_UNSUPPORTED: return "unsupported";
    case VG_ERR_OUT_OF_MEMORY: return "out of device memory";
    case VG_ERR_HIP: return "HIP runtime error";
    case VG_ERR_NO_DEVICE: return "no usable gfx950 device";
    case VG_ERR_NOT_READY: return "index lacks the data this search needs";
    case VG_ERR_FORMAT: return "malformed segment image";
    case VG_ERR_CHECKSUM: return "checksum mismatch";
    default: return "unknown status";
    }
}

VG_API int32_t vg_ctx_create(int32_t device, vg_ctx **out)
{
    VG_CHECK(out != nullptr, VG_ERR_INVALID_ARG, "vg_ctx_create: out is NULL");
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        vg::set_error("vg_ctx_create: no HIP device visible (the HIP path has no CPU fallback)");
        return VG_ERR_NO_DEVICE;
    }
    VG_CHECK(device >= 0 && device < count, VG_ERR_INVALID_ARG,
             "vg_ctx_create: device %d out of range (%d visible)", device, count);
    VG_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    VG_HIP(hipGetDeviceProperties(&prop, device));
    vg_ctx *ctx = new vg_ctx();
    ctx->device = device;
    ctx->compute_units = prop.multiProcessorCount;
    ctx->hbm_bytes = static_cast<int64_t>(prop.totalGlobalMem);
    snprintf(ctx->arch, sizeof ctx->arch, "%s", prop.gcnArchName);
    if (strncmp(ctx->arch, "gfx950", 6) != 0) {
        vg::set_error("vg_ctx_create: device %d is %s; this library carries gfx950 code only",
                      device, ctx->arch);
        delete ctx;
        return VG_ERR_NO_DEVICE;
    }
    hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        vg::set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
        delete ctx;
        return VG_ERR_HIP;
    }
    *out = ctx;
    return VG_OK;
}

VG_API int32_t vg_ctx_destroy(vg_ctx *ctx)
{
    if (!ctx) return VG_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    for (auto &p : ctx->arenas)
        if (p.second.base) (void)hipFree(p.second.base);
    vg::scratch_trim(ctx->device);
    if (ctx->stream) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamDestroy(ctx->stream);
    }
    delete ctx;
    return VG_OK;
}

VG_API int32_t vg_ctx_synchronize(vg_ctx *ctx, void *stream)
{
    VG_CHECK(ctx != nullptr, VG_ERR_INVALID_ARG, "vg_ctx_synchronize: ctx is NULL");
    VG_HIP(hipSetDevice(ctx->device));
    VG_HIP(hipStreamSynchronize(vg::pick_stream(ctx, stream)));
    return VG_OK;
}

VG_API int32_t vg_ctx_device_info(vg_ctx *ctx, char *arch, int32_t arch_len,
                                  int32_t *compute_units, int64_t *hbm_bytes)
{
    VG_CHECK(ctx != nullptr, VG_ERR_INVALID_ARG, "vg_ctx_device_info: ctx is NULL");
    if (arch && arch_len > 0) snprintf(arch, static_cast<size_t>(arch_len), "%s", ctx->arch);
    if (compute_units) *compute_units = ctx->compute_units;
    if (hbm_bytes) *hbm_bytes = ctx->hbm_bytes;
    return VG_OK;
}

VG_API int32_t vg_profile_enable(vg_ctx *ctx, int32_t on)
{
    VG_CHECK(ctx != nullptr, VG_ERR_INVALID_ARG, "vg_profile_enable: ctx is NULL");
    ctx->profiling = on != 0;
    return VG_OK;
}

VG_API int32_t vg_profile_read(vg_ctx *ctx, const char *kernel, int64_t *launches, double *total_ms)
{
    VG_CHECK(ctx && kernel && launches && total_ms, VG_ERR_INVALID_ARG, "vg_profile_read: NULL argument");
    VG_HIP(hipSetDevice(ctx->device));
    std::lock_guard<std::mutex> g(ctx->prof_mu);
    int64_t n = 0;
    double ms = 0.0;
    std::vector<vg_prof_record> keep;
    for (auto &r : ctx->prof) {
        if (strcmp(r.name, kernel) != 0) {
            keep.push_back(r);
            continue;
        }
        float t = 0.0f;
        if (hipEventSynchronize(r.stop) == hipSuccess && hipEventElapsedTime(&t, r.start, r.stop) == hipSuccess) {
            n++;
            ms += t;
        }
        (void)hipEventDestroy(r.start);
        (void)hipEventDestroy(r.stop);
    }
    ctx->prof.swap(keep);
    *launches = n;
    *total_ms = ms;
    return VG_OK;
}

VG_API int32_t vg_index_create(vg_ctx *ctx, int64_t n, int32_t dim, int32_t metric,
                               vg_index **out)
{
    VG_CHECK(ctx && out, VG_ERR_INVALID_ARG, "vg_index_create: NULL argument");
    *out = nullptr;
    VG_CHECK(n >= 0 && n < 0xFFFFFFFFll, VG_ERR_INVALID_ARG,
             "vg_index_create: n=%lld out of range (row ids are uint32)", (long long)n);
    VG_CHECK(dim > 0, VG_ERR_INVALID_ARG, "vg_index_create: dim must be positive");
    VG_CHECK(metric >= VG_METRIC_L2 && metric <= VG_METRIC_HAMMING, VG_ERR_UNSUPPORTED,
             "vg_index_create: unknown metric %d", metric);
    vg_index *idx = new vg_index();
    idx->ctx = ctx;
    idx->n = n;
    idx->dim = dim;
    idx->metric = metric;
    *out = idx;
    return VG_OK;
}

VG_API int32_t vg_index_destroy(vg_index *idx)
{
    if (!idx) return VG_OK;
    (void)hipSetDevice(idx->ctx->device);
    if (idx->d_pq_tiles) (void)hipFree(idx->d_pq_tiles);
    if (idx->d_vectors) (void)hipFree(idx->d_vectors);
    if (idx->d_norms) (void)hipFree(idx->d_norms);
    if (idx->d_norm_max) (void)hipFree(idx->d_norm_max);
    if (idx->d_flat_stats) (void)hipFree(idx->d_flat_stats);
    if (idx->d_vectors_bf16) (void)hipFree(idx->d_vectors_bf16);
    if (idx->d_rq_tiles) (void)hipFree(idx->d_rq_tiles);
    if (idx->d_rq_norms) (void)hipFree(idx->d_rq_norms);
    if (idx->d_hnsw_l0) (void)hipFree(idx->d_hnsw_l0);
    if (idx->d_hnsw_l0_dist) (void)hipFree(idx->d_hnsw_l0_dist);
    if (idx->d_hnsw_tomb) (void)hipFree(idx->d_hnsw_tomb);
    if (idx->d_hnsw_slot) (void)hipFree(idx->d_hnsw_slot);
    if (idx->d_hnsw_adj) (void)hipFree(idx->d_hnsw_adj);
    if (idx->d_hnsw_level_off) (void)hipFree(idx->d_hnsw_level_off);
    if (idx->d_vamana) (void)hipFree(idx->d_vamana);
    if (idx->d_pq_rows) (void)hipFree(idx->d_pq_rows);
    if (idx->d_rq_rows) (void)hipFree(idx->d_rq_rows);
    if (idx->d_sq_tiles) (void)hipFree(idx->d_sq_tiles);
    if (idx->d_sq_bf16) (void)hipFree(idx->d_sq_bf16);
    if (idx->d_pq_bf16) (void)hipFree(idx->d_pq_bf16);
    if (idx->d_pq_norms) (void)hipFree(idx->d_pq_norms);
    if (idx->d_pq_norm_max) (void)hipFree(idx->d_pq_norm_max);
    if (idx->d_sq_norms) (void)hipFree(idx->d_sq_norms);
    if (idx->d_sq_norm_max) (void)hipFree(idx->d_sq_norm_max);
    if (idx->d_int4_rows) (void)hipFree(idx->d_int4_rows);
    if (idx->d_centroids) (void)hipFree(idx->d_centroids);
    if (idx->d_part_off) (void)hipFree(idx->d_part_off);
    delete idx;
    return VG_OK;
}

// ---- test hooks ------------------------------------------------------------------------------------
#include <atomic>
namespace vg {
static std::atomic<int> g_hooks[kHookCount];
static std::once_flag g_hooks_once;
static const char *const kHookNames[kHookCount] = {"VG_FLAT_NO_SMALL_TILE", "VG_FLAT_UNFUSED", "VG_FLAT_NO_SCAN",
                                                   "VG_FLAT_FORCE_EXACT", "VG_FLAT_NO_DMA", "VG_FLAT_DEBUG",
                                                   "VG_PROBE_NO_GROUP", "VG_ADC_BIGK_EXHAUSTIVE", "VG_BUILD_DEBUG",
                                                   "VG_KM_NO_MFMA", "VG_KM_LIST_ALL", "VG_KM_BF16", "VG_KM_NO_BF16", "VG_BRUTE_NO_FLAT", "VG_PQ_NO_MFMA",
                                                   "VG_PQ_LIST_ALL", "VG_PQ_FP32_MFMA", "VG_PROBE_NO_GEMM", "VG_FLAT_NO_BIG_TILE",
                                                   "VG_FLAT_BIG_TILE_2", "VG_FLAT_BIG_EARLY_B", "VG_PQ_NOM_ALWAYS", "VG_KM_NO_RANGES", "VG_NO_CAND_REPLAY"};
static void hooks_from_env()
{
    for (int h = 0; h < kHookCount; h++) {
        const char *e = getenv(kHookNames[h]);
        g_hooks[h].store(e && e[0] == '1' ? 1 : 0, std::memory_order_relaxed);
    }
}
bool hook(Hook h)
{
    std::call_once(g_hooks_once, hooks_from_env);
    return g_hooks[h].load(std::memory_order_relaxed) != 0;
}
}  // namespace vg

VG_API int32_t vg_debug_set_hook(const char *name, int32_t on)
{
    VG_CHECK(name, VG_ERR_INVALID_ARG, "vg_debug_set_hook: NULL name");
    std::call_once(vg::g_hooks_once, vg::hooks_from_env);
    for (int h = 0; h < vg::kHookCount; h++)
        if (std::strcmp(name, vg::kHookNames[h]) == 0) {
            vg::g_hooks[h].store(on ? 1 : 0, std::memory_order_relaxed);
            return VG_OK;
        }
    vg::set_error("vg_debug_set_hook: unknown hook %s", name);
    return VG_ERR_INVALID_ARG;
}
