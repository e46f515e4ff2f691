// vg_comm.hip — the one exchange step of the sharded searches, behind the C ABI: per-shard top-k lists are
// all-gathered over RCCL (xGMI inside a node) on the caller's stream and merged with the reference's
// tie-break (searcher/candidate_queue.go:12-23), the way the engine merges per-segment candidate lists into
// one bounded heap (engine/search.go:904-908).  A Go host can therefore shard a corpus
// across the GPUs of a node with one process (or one OS thread) per GPU and no Python / torch in between.
//
// RCCL is loaded at run time (dlopen of librccl.so.1), so the library has no link-time dependency on it and
// single-GPU users never load it.  One collective per search: ids and score bits of a rank travel as one
// [2][nq][k] int32 block; the gathered [world][2][nq][k] image goes straight into the merge kernel.
#include <dlfcn.h>

#include <cstdio>
#include <mutex>
#include <string>

#include "vg_device.hpp"
#include "vg_internal.hpp"

// The few RCCL declarations this file needs, so that the library builds without RCCL's headers (it never links
// against RCCL either): rccl.h's ABI for them has been the same since NCCL 2.
typedef struct ncclComm *ncclComm_t;
typedef struct {
    char internal[128];
} ncclUniqueId;
typedef int ncclResult_t;
typedef int ncclDataType_t;
enum { ncclSuccess = 0, ncclInt8 = 0 };

namespace vg {

struct RcclApi {
    void *handle = nullptr;
    std::string path;  // the file the symbols came from (dladdr)
    bool reused = false;  // an RCCL that was already mapped into the process (PyTorch's) rather than a second copy
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommCuDevice)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

static RcclApi g_rccl;
static std::mutex g_rccl_mu;

// A process must not run two RCCLs: PyTorch maps its own copy (torch/lib/librccl.so) as soon as torch.distributed
// creates an RCCL group.  Resolution order: (1) whatever RCCL is ALREADY mapped — by soname (RTLD_NOLOAD), then by
// the path /proc/self/maps shows for a file named librccl*; (2) only if none is mapped, the system's librccl.so.1.
// a handle counts only if it really is RCCL (a mapped plugin such as librccl-net.so has "librccl" in its name too)
static bool is_rccl(void *h) { return h && dlsym(h, "ncclGetUniqueId") && dlsym(h, "ncclAllGather"); }

static void *find_mapped_rccl(std::string &path)
{
    for (const char *name : {"librccl.so.1", "librccl.so"}) {
        if (void *h = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL)) {
            if (is_rccl(h)) {
                path = name;
                return h;
            }
            dlclose(h);
        }
    }
    FILE *f = fopen("/proc/self/maps", "r");
    if (!f) return nullptr;
    char line[4096];
    void *h = nullptr;
    while (!h && fgets(line, sizeof(line), f)) {
        const char *start = strchr(line, '/');
        if (!start) continue;
        std::string file(start);
        while (!file.empty() && (file.back() == '\n' || file.back() == ' ')) file.pop_back();
        const std::string deleted = " (deleted)";
        if (file.size() > deleted.size() && file.compare(file.size() - deleted.size(), deleted.size(), deleted) == 0)
            continue;  // the file behind the mapping is gone: nothing dlopen could name
        const size_t slash = file.rfind('/');
        const std::string base = slash == std::string::npos ? file : file.substr(slash + 1);
        // basename librccl.so, librccl.so.1, librccl.so.1.0.xxxx — not librccl-net.so, librccl-anp.so
        if (base.compare(0, 10, "librccl.so") != 0 || (base.size() > 10 && base[10] != '.')) continue;
        void *cand = dlopen(file.c_str(), RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
        if (is_rccl(cand)) {
            h = cand;
            path = file;
        } else if (cand) {
            dlclose(cand);
        }
    }
    fclose(f);
    return h;
}

static int32_t load_rccl()
{
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.handle) return VG_OK;
    RcclApi a;
    void *h = find_mapped_rccl(a.path);
    a.reused = h != nullptr;
    std::string why;
    if (!h) {
        (void)dlerror();  // the message below must be this dlopen's, not an earlier NOLOAD probe's
        h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) {
            const char *e = dlerror();
            why = e ? e : "";
            h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        }
    }
    VG_CHECK(h, VG_ERR_UNSUPPORTED, "vg_comm: cannot load librccl.so.1: %s", why.c_str());
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    a.CommCount = reinterpret_cast<decltype(a.CommCount)>(dlsym(h, "ncclCommCount"));
    a.CommCuDevice = reinterpret_cast<decltype(a.CommCuDevice)>(dlsym(h, "ncclCommCuDevice"));
    a.CommUserRank = reinterpret_cast<decltype(a.CommUserRank)>(dlsym(h, "ncclCommUserRank"));
    a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(h, "ncclAllGather"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    VG_CHECK(a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllGather && a.GetErrorString && a.CommCount &&
                 a.CommCuDevice && a.CommUserRank,
             VG_ERR_UNSUPPORTED, "vg_comm: librccl.so.1 lacks a required symbol");
    Dl_info info;
    if (dladdr(reinterpret_cast<void *>(a.AllGather), &info) && info.dli_fname) a.path = info.dli_fname;
    a.handle = h;
    g_rccl = a;
    return VG_OK;
}

#define VG_RCCL(expr)                                                                                   \
    do {                                                                                                \
        ncclResult_t _r = (expr);                                                                       \
        if (_r != ncclSuccess) {                                                                        \
            ::vg::set_error("%s failed: %s (%s:%d)", #expr, ::vg::g_rccl.GetErrorString(_r), __FILE__,   \
                            __LINE__);                                                                  \
            return VG_ERR_HIP;                                                                          \
        }                                                                                               \
    } while (0)

// local (ids, scores) [nq][k] -> one [2][nq][k] int32 block
__global__ void pack_topk_kernel(const uint32_t *__restrict__ ids, const float *__restrict__ scores, int64_t count,
                                 uint32_t *__restrict__ block)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= count) return;
    block[i] = ids[i];
    block[count + i] = __float_as_uint(scores[i]);
}

}  // namespace vg

struct vg_comm {
    vg_ctx *ctx = nullptr;
    ncclComm_t comm = nullptr;
    int32_t world = 1, rank = 0;
};

VG_API int32_t vg_comm_unique_id(uint8_t *id)
{
    VG_CHECK(id, VG_ERR_INVALID_ARG, "vg_comm_unique_id: NULL buffer");
    VG_TRY(vg::load_rccl());
    ncclUniqueId u;
    VG_RCCL(vg::g_rccl.GetUniqueId(&u));
    static_assert(sizeof(u) == VG_COMM_ID_BYTES, "ncclUniqueId size");
    std::memcpy(id, &u, sizeof(u));
    return VG_OK;
}

VG_API int32_t vg_comm_create(vg_ctx *ctx, int32_t world, int32_t rank, const uint8_t *id, vg_comm **out)
{
    VG_CHECK(ctx && out, VG_ERR_INVALID_ARG, "vg_comm_create: NULL argument");
    VG_CHECK(world >= 1 && rank >= 0 && rank < world, VG_ERR_INVALID_ARG, "vg_comm_create: rank %d of %d", rank, world);
    VG_CHECK(id, VG_ERR_INVALID_ARG, "vg_comm_create: NULL unique id");
    VG_TRY(vg::load_rccl());
    VG_HIP(hipSetDevice(ctx->device));
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    vg_comm *c = new vg_comm;
    c->ctx = ctx;
    c->world = world;
    c->rank = rank;
    ncclResult_t r = vg::g_rccl.CommInitRank(&c->comm, world, u, rank);
    if (r != ncclSuccess) {
        vg::set_error("ncclCommInitRank failed: %s", vg::g_rccl.GetErrorString(r));
        delete c;
        return VG_ERR_HIP;
    }
    *out = c;
    return VG_OK;
}

VG_API int32_t vg_comm_destroy(vg_comm *comm)
{
    if (!comm) return VG_OK;
    if (comm->comm) (void)vg::g_rccl.CommDestroy(comm->comm);
    delete comm;
    return VG_OK;
}

VG_API int32_t vg_comm_info(const vg_comm *comm, int32_t *world, int32_t *rank)
{
    VG_CHECK(comm, VG_ERR_INVALID_ARG, "vg_comm_info: NULL communicator");
    if (world) *world = comm->world;
    if (rank) *rank = comm->rank;
    return VG_OK;
}

VG_API int32_t vg_comm_probe(char *rccl_path, int32_t len)
{
    VG_TRY(vg::load_rccl());
    if (rccl_path && len > 0) std::snprintf(rccl_path, static_cast<size_t>(len), "%s", vg::g_rccl.path.c_str());
    return VG_OK;
}

VG_API int32_t vg_comm_describe(const vg_comm *comm, int32_t *rccl_ranks, int32_t *rccl_rank, int32_t *rccl_device,
                                int32_t *reused_mapped_rccl, char *rccl_path, int32_t len)
{
    VG_CHECK(comm && comm->comm, VG_ERR_INVALID_ARG, "vg_comm_describe: NULL communicator");
    int n = 0, r = 0, d = 0;
    VG_RCCL(vg::g_rccl.CommCount(comm->comm, &n));
    VG_RCCL(vg::g_rccl.CommUserRank(comm->comm, &r));
    VG_RCCL(vg::g_rccl.CommCuDevice(comm->comm, &d));
    if (rccl_ranks) *rccl_ranks = n;
    if (rccl_rank) *rccl_rank = r;
    if (rccl_device) *rccl_device = d;
    if (reused_mapped_rccl) *reused_mapped_rccl = vg::g_rccl.reused ? 1 : 0;
    if (rccl_path && len > 0) std::snprintf(rccl_path, static_cast<size_t>(len), "%s", vg::g_rccl.path.c_str());
    return VG_OK;
}

VG_API int32_t vg_comm_all_gather(vg_comm *comm, const void *send, void *recv, int64_t bytes_per_rank, void *stream)
{
    VG_CHECK(comm, VG_ERR_INVALID_ARG, "vg_comm_all_gather: NULL communicator");
    VG_CHECK(bytes_per_rank >= 0, VG_ERR_INVALID_ARG, "vg_comm_all_gather: negative size");
    if (bytes_per_rank == 0) return VG_OK;
    VG_CHECK(send && recv, VG_ERR_INVALID_ARG, "vg_comm_all_gather: NULL buffer");
    VG_CHECK(vg::is_device_ptr(send) && vg::is_device_ptr(recv), VG_ERR_INVALID_ARG,
             "vg_comm_all_gather: buffers must be device memory");
    VG_HIP(hipSetDevice(comm->ctx->device));
    hipStream_t st = vg::pick_stream(comm->ctx, stream);
    VG_RCCL(vg::g_rccl.AllGather(send, recv, static_cast<size_t>(bytes_per_rank), ncclInt8, comm->comm, st));
    return VG_OK;
}

VG_API int32_t vg_comm_all_gather_topk(vg_comm *comm, const uint32_t *local_ids, const float *local_scores,
                                       int64_t nq, int32_t k, int32_t metric, const uint32_t *id_offsets,
                                       uint32_t *ids, float *scores, void *stream)
{
    VG_CHECK(comm, VG_ERR_INVALID_ARG, "vg_comm_all_gather_topk: NULL communicator");
    VG_CHECK(nq >= 0 && k >= 0, VG_ERR_INVALID_ARG, "vg_comm_all_gather_topk: negative count");
    if (nq == 0 || k == 0) return VG_OK;
    VG_CHECK(local_ids && local_scores && ids && scores, VG_ERR_INVALID_ARG, "vg_comm_all_gather_topk: NULL buffer");
    vg_ctx *ctx = comm->ctx;
    VG_HIP(hipSetDevice(ctx->device));
    hipStream_t st = vg::pick_stream(ctx, stream);
    const size_t count = static_cast<size_t>(nq) * k;
    vg::DevIn<uint32_t> li;
    vg::DevIn<float> ls;
    VG_TRY(li.init(local_ids, count, st));
    VG_TRY(ls.init(local_scores, count, st));
    vg::DevTmp<uint32_t> mine, all;
    VG_TRY(mine.init(2 * count, st));
    VG_TRY(all.init(2 * count * comm->world, st));
    VG_LAUNCH(vg::pack_topk_kernel, dim3(static_cast<unsigned>((count + 255) / 256)), dim3(256), 0, st, li.ptr, ls.ptr,
              static_cast<int64_t>(count), mine.ptr);
    {
        vg::ProfScope prof(ctx, "comm_all_gather", st);
        VG_RCCL(vg::g_rccl.AllGather(mine.ptr, all.ptr, 2 * count * sizeof(uint32_t), ncclInt8, comm->comm, st));
    }
    return vg_merge_topk_packed(ctx, all.ptr, comm->world, nq, k, metric, id_offsets, ids, scores, stream);
}
