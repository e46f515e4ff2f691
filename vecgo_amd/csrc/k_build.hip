// k_build.hip — SURVEY.md §8(f) rank 4: the neighbour-selection steps of graph CONSTRUCTION, i.e. the
// build-time loops that are pairwise distance evaluations:
//   Vamana robustPrune                 internal/segment/diskann/writer.go:571-625
//   HNSW   selectNeighborsHeuristic    internal/hnsw/hnsw.go:1009-1106 (applyHeuristic, fillUpNeighbors)
// One workgroup per node.  Every distance is the reference's pair kernel in its summation order
// (vg_exact.hpp), so the kept set equals what the reference computes from the same candidate order.
// The reference's own builders are not reproducible run to run (rand.Perm, map iteration order,
// unstable sort among equal distances): where it leaves the order open, ties go by ascending id.
#include "vg_device.hpp"
#include "vg_exact.hpp"
#include "vg_internal.hpp"

namespace vg {

constexpr int kBuildThreads = 256;
constexpr int kBuildMaxCand = 1024;  // candidates per node (power of two: bitonic sort)
constexpr int kBuildMaxKeep = 256;   // r / m

struct BuildShared {
    uint64_t keys[kBuildMaxCand];
    uint32_t kept[kBuildMaxKeep];
    float dists[kBuildMaxCand];  // hnsw: the caller's candidate distances
    int flag;
    int nkept;
};

// distance.Provider(metric)(a, b): SquaredL2 or Dot (distance.go:91-106), all 16 lanes of a group
__device__ __forceinline__ float provider_pair(const float *a, const float *b, int dim, bool dot, Sub16 sub)
{
    return dot ? exact_pair16<true, kPair>(a, b, dim, sub) : exact_pair16<false, kPair>(a, b, dim, sub);
}

__global__ __launch_bounds__(kBuildThreads) void robust_prune_kernel(
    const float *__restrict__ base, int64_t n, int dim, int metric, const uint32_t *__restrict__ nodes,
    const uint32_t *__restrict__ cands, int nc, int r, float alpha, uint32_t *__restrict__ out,
    int32_t *__restrict__ counts)
{
    __shared__ BuildShared sh;
    const int64_t b = blockIdx.x;
    const int tid = threadIdx.x, grp = tid >> 4;
    const Sub16 sub = Sub16::make(tid);
    const bool dot = metric != VG_METRIC_L2;
    const uint32_t node = nodes[b];
    const float *nv = base + static_cast<int64_t>(node) * dim;
    int np2 = 1;
    while (np2 < nc) np2 <<= 1;
    // (dist, id) keys; the node itself, invalid ids and (after the sort) duplicates drop out
    for (int c0 = 0; c0 < np2; c0 += kBuildThreads / 16) {
        const int c = c0 + grp;
        uint64_t key = kKeyMax;
        if (c < nc) {
            const uint32_t id = cands[b * nc + c];
            if (id != VG_INVALID_ID && id < n && id != node) {
                const float d = provider_pair(base + static_cast<int64_t>(id) * dim, nv, dim, dot, sub);
                key = make_key(d, id, false);
            }
        }
        if ((tid & 15) == 0 && c < np2) sh.keys[c] = key;
    }
    if (tid == 0) sh.nkept = 0;
    __syncthreads();
    bitonic_sort_lds(sh.keys, np2, tid, kBuildThreads);
    // greedy selection in sorted order (writer.go:598-617)
    for (int i = 0; i < np2; i++) {
        const uint64_t key = sh.keys[i];
        const int nk = sh.nkept;
        if (key == kKeyMax || nk >= r) break;           // uniform: the keys are sorted, kKeyMax last
        if (i > 0 && sh.keys[i - 1] == key) continue;   // duplicate candidate id
        const uint32_t id = key_row(key);
        const float dist = key_score(key, false);
        const float *cv = base + static_cast<int64_t>(id) * dim;
        if (tid == 0) sh.flag = 0;
        __syncthreads();
        for (int s0 = 0; s0 < nk; s0 += kBuildThreads / 16) {
            const int s = s0 + grp;
            if (s < nk) {
                const float dcs = provider_pair(cv, base + static_cast<int64_t>(sh.kept[s]) * dim, dim, dot, sub);
                const float lhs = alpha * dcs;
                if ((tid & 15) == 0 && lhs < dist) sh.flag = 1;  // not diverse
            }
        }
        __syncthreads();
        if (tid == 0 && !sh.flag) {
            sh.kept[nk] = id;
            sh.nkept = nk + 1;
        }
        __syncthreads();
    }
    const int nk = sh.nkept;
    for (int i = tid; i < r; i += kBuildThreads) out[b * r + i] = i < nk ? sh.kept[i] : VG_INVALID_ID;
    if (tid == 0) counts[b] = nk;
}

__global__ __launch_bounds__(kBuildThreads) void hnsw_select_kernel(
    const float *__restrict__ base, int64_t n, int dim, int metric, const uint32_t *__restrict__ cand_ids,
    const float *__restrict__ cand_dists, int nc, int m, uint32_t *__restrict__ out, int32_t *__restrict__ counts)
{
    __shared__ BuildShared sh;
    const int64_t b = blockIdx.x;
    const int tid = threadIdx.x, grp = tid >> 4;
    const Sub16 sub = Sub16::make(tid);
    const uint32_t *ids = cand_ids + b * nc;
    // the candidate list may be padded with VG_INVALID_ID at its end
    int len = 0;
    for (int i = tid; i < nc; i += kBuildThreads) sh.dists[i] = cand_dists[b * nc + i];
    if (tid == 0) {
        while (len < nc && ids[len] != VG_INVALID_ID) len++;
        sh.flag = len;
        sh.nkept = 0;
    }
    __syncthreads();
    len = sh.flag;
    __syncthreads();
    if (len <= m) {  // selectNeighborsSimple: everything, best first (hnsw.go:1014-1016)
        for (int i = tid; i < m; i += kBuildThreads) out[b * m + i] = i < len ? ids[i] : VG_INVALID_ID;
        if (tid == 0) counts[b] = len;
        return;
    }
    for (int i = 0; i < len; i++) {  // applyHeuristic (hnsw.go:1048-1085)
        const int nk = sh.nkept;
        if (nk >= m) break;
        const uint32_t id = ids[i];
        const float cd = sh.dists[i];
        const float *cv = base + static_cast<int64_t>(id) * dim;
        if (tid == 0) sh.flag = 0;
        __syncthreads();
        for (int s0 = 0; s0 < nk; s0 += kBuildThreads / 16) {
            const int s = s0 + grp;
            if (s < nk) {
                const float *rv = base + static_cast<int64_t>(sh.kept[s]) * dim;
                float d;
                if (metric == VG_METRIC_DOT) {
                    d = -exact_pair16<true, kPair>(cv, rv, dim, sub);
                } else {
                    d = exact_pair16<false, kPair>(cv, rv, dim, sub);
                    if (metric == VG_METRIC_COSINE) d = 0.5f * d;
                }
                if ((tid & 15) == 0 && d < cd) sh.flag = 1;
            }
        }
        __syncthreads();
        if (tid == 0 && !sh.flag) {
            sh.kept[nk] = id;
            sh.nkept = nk + 1;
        }
        __syncthreads();
    }
    if (tid == 0) {  // fillUpNeighbors (hnsw.go:1087-1106)
        int nk = sh.nkept;
        for (int i = 0; i < len && nk < m; i++) {
            bool found = false;
            for (int s = 0; s < nk; s++) found |= sh.kept[s] == ids[i];
            if (!found) sh.kept[nk++] = ids[i];
        }
        sh.nkept = nk;
    }
    __syncthreads();
    const int nk = sh.nkept;
    for (int i = tid; i < m; i += kBuildThreads) out[b * m + i] = i < nk ? sh.kept[i] : VG_INVALID_ID;
    if (tid == 0) counts[b] = nk;
}

}  // namespace vg

VG_API int32_t vg_robust_prune(vg_index *idx, const uint32_t *nodes, int64_t n_nodes, const uint32_t *cands,
                               int32_t nc, int32_t r, float alpha, uint32_t *out, int32_t *counts, void *stream)
{
    VG_CHECK(idx, VG_ERR_INVALID_ARG, "vg_robust_prune: NULL index");
    VG_CHECK(n_nodes >= 0 && nc >= 0 && r > 0, VG_ERR_INVALID_ARG, "vg_robust_prune: bad counts");
    if (n_nodes == 0) return VG_OK;
    VG_CHECK(idx->d_vectors, VG_ERR_NOT_READY, "vg_robust_prune: index has no fp32 vectors");
    VG_CHECK(idx->metric != VG_METRIC_HAMMING, VG_ERR_UNSUPPORTED, "unsupported metric for float32: Hamming");
    VG_CHECK(nodes && out && counts && (nc == 0 || cands), VG_ERR_INVALID_ARG, "vg_robust_prune: NULL buffer");
    VG_CHECK(nc <= vg::kBuildMaxCand && r <= vg::kBuildMaxKeep, VG_ERR_UNSUPPORTED,
             "vg_robust_prune: at most %d candidates and %d kept per node", vg::kBuildMaxCand, vg::kBuildMaxKeep);
    VG_HIP(hipSetDevice(idx->ctx->device));
    hipStream_t st = vg::pick_stream(idx->ctx, stream);
    vg::DevIn<uint32_t> nd, cd;
    vg::DevOut<uint32_t> o;
    vg::DevOut<int32_t> cn;
    VG_TRY(nd.init(nodes, static_cast<size_t>(n_nodes), st));
    VG_TRY(cd.init(cands, static_cast<size_t>(n_nodes) * nc, st));
    VG_TRY(o.init(out, static_cast<size_t>(n_nodes) * r, st));
    VG_TRY(cn.init(counts, static_cast<size_t>(n_nodes), st));
    VG_LAUNCH(vg::robust_prune_kernel, dim3(static_cast<unsigned>(n_nodes)), dim3(vg::kBuildThreads), 0, st,
              idx->d_vectors, idx->n, idx->dim, idx->metric, nd.ptr, cd.ptr, nc, r, alpha, o.ptr, cn.ptr);
    VG_TRY(o.finish());
    VG_TRY(cn.finish());
    return VG_OK;
}

VG_API int32_t vg_hnsw_select_neighbors(vg_index *idx, int64_t n_nodes, const uint32_t *cand_ids,
                                        const float *cand_dists, int32_t nc, int32_t m, uint32_t *out,
                                        int32_t *counts, void *stream)
{
    VG_CHECK(idx, VG_ERR_INVALID_ARG, "vg_hnsw_select_neighbors: NULL index");
    VG_CHECK(n_nodes >= 0 && nc >= 0 && m > 0, VG_ERR_INVALID_ARG, "vg_hnsw_select_neighbors: bad counts");
    if (n_nodes == 0) return VG_OK;
    VG_CHECK(idx->d_vectors, VG_ERR_NOT_READY, "vg_hnsw_select_neighbors: index has no fp32 vectors");
    VG_CHECK(idx->metric != VG_METRIC_HAMMING, VG_ERR_UNSUPPORTED, "unsupported metric for float32: Hamming");
    VG_CHECK(out && counts && (nc == 0 || (cand_ids && cand_dists)), VG_ERR_INVALID_ARG,
             "vg_hnsw_select_neighbors: NULL buffer");
    VG_CHECK(nc <= vg::kBuildMaxCand && m <= vg::kBuildMaxKeep, VG_ERR_UNSUPPORTED,
             "vg_hnsw_select_neighbors: at most %d candidates and %d kept per node", vg::kBuildMaxCand,
             vg::kBuildMaxKeep);
    VG_HIP(hipSetDevice(idx->ctx->device));
    hipStream_t st = vg::pick_stream(idx->ctx, stream);
    vg::DevIn<uint32_t> ci;
    vg::DevIn<float> cdist;
    vg::DevOut<uint32_t> o;
    vg::DevOut<int32_t> cn;
    VG_TRY(ci.init(cand_ids, static_cast<size_t>(n_nodes) * nc, st));
    VG_TRY(cdist.init(cand_dists, static_cast<size_t>(n_nodes) * nc, st));
    VG_TRY(o.init(out, static_cast<size_t>(n_nodes) * m, st));
    VG_TRY(cn.init(counts, static_cast<size_t>(n_nodes), st));
    VG_LAUNCH(vg::hnsw_select_kernel, dim3(static_cast<unsigned>(n_nodes)), dim3(vg::kBuildThreads), 0, st,
              idx->d_vectors, idx->n, idx->dim, idx->metric, ci.ptr, cdist.ptr, nc, m, o.ptr, cn.ptr);
    VG_TRY(o.finish());
    VG_TRY(cn.finish());
    return VG_OK;
}
