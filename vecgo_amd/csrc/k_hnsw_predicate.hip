// k_hnsw_predicate.hip — searchExecute with a filter whose selectivity hint is at or below highSelectivityThreshold (0.3) or
// unknown (internal/hnsw/hnsw.go:1107-1146): searchLayerPredicateAware (hnsw.go:1406-1558) on layer 0 after the usual greedy
// descent, then knnSearchInternal's extraction (:1732-1751).
//
// The reference's loop decides per neighbour, IN LIST ORDER, from state the earlier neighbours changed (results.Len(), the worst
// result, consecutiveFilterMisses): whether the node's distance is computed, taken from the cached edge distance (Neighbor.Dist,
// node.go:62-80) or the node skipped.  One wave per query walks it the same way: the passing live neighbours of a popped node are
// scored together, then the per-neighbour decisions are replayed in order, wave-uniform; a rejected node is scored only when the
// replay reaches a branch that computes its distance (then together with the list's remaining rejected nodes); the counters
// count what the reference's branches count.  Two queues as written: the navigation min-heap is unbounded (PushItem; LDS for the
// first items, HBM scratch beyond), the results max-heap is bounded by ef (PushItemBounded; LDS).
#include <algorithm>

#include "vg_device.hpp"
#include "vg_exact.hpp"
#include "vg_hnsw_layer.hpp"
#include "vg_internal.hpp"

namespace vg {

constexpr int kPredLdsCand = 1024;  // navigation-heap items kept in LDS (512 / 4096 measured the same: a walk is bound by the
                                    // chain list -> visited bits -> filter bits -> rows of each pop, not by the heap's HBM levels)
constexpr int kPredMaxEf = 4096;   // results heap in LDS: (ef + 1) * 8 bytes

// cached edge distances of the layer-0 lists, recomputed from the rows: Neighbor.Dist is the distance the insert computed between
// the two nodes (hnsw.go:516, :550, :964 — distFunc of one node's vector against the other's row, symmetric in every kernel)
__global__ __launch_bounds__(64) void hnsw_edge_dist_kernel(const float *__restrict__ base, int64_t n, int dim, int metric,
                                                            const uint32_t *__restrict__ l0, int m0, float *__restrict__ out)
{
    __shared__ float nb_pair[64], nb_bnd[64];
    const int lane = threadIdx.x;
    for (int64_t node = blockIdx.x; node < n; node += gridDim.x) {
        F32ScorerT<false> sc;
        sc.base = base;
        sc.qv = base + node * dim;
        sc.dim = dim;
        sc.metric = metric;
        sc.sub = Sub16::make(lane);
        const uint32_t id_lane = lane < m0 ? l0[node * m0 + lane] : VG_INVALID_ID;
        const uint64_t inval = __ballot(id_lane == VG_INVALID_ID);
        const int count = inval ? __builtin_ctzll(inval) : 64;
        const uint64_t mask = count >= 64 ? ~0ull : ((1ull << count) - 1);
        sc.many(mask, id_lane, lane, nb_pair, nb_bnd);
        __syncthreads();
        if (lane < m0) out[node * m0 + lane] = lane < count ? nb_pair[lane] : 0.0f;
        __syncthreads();
    }
}

__global__ __launch_bounds__(64) void hnsw_predicate_kernel(
    const float *__restrict__ base, int64_t n, int dim, int metric, const uint32_t *__restrict__ l0, const float *__restrict__ l0_dist,
    int m0, int max_level, int m, const uint32_t *__restrict__ slots, const uint32_t *__restrict__ adj,
    const int64_t *__restrict__ level_off, uint32_t entry, const float *__restrict__ queries, int k, int ef,
    const uint8_t *__restrict__ mask, int64_t mask_stride, const uint8_t *__restrict__ deleted, uint32_t *__restrict__ visited_ws,
    int64_t vis_words, HItem *__restrict__ cand_ws, int64_t cand_cap, uint32_t *__restrict__ ids, float *__restrict__ scores,
    vg_search_stats *__restrict__ stats, uint8_t *__restrict__ redo, int64_t redo_first /* < 0: first pass — a walk whose
    navigation queue outgrows cand_cap stops and sets redo[q]; >= 0: second pass over the queries [redo_first, redo_first +
    gridDim.x) of the chunk, only the marked ones, with a queue of one slot per row */)
{
    extern __shared__ __attribute__((aligned(8))) unsigned char smem[];
    float *nb_pair = reinterpret_cast<float *>(smem);
    float *nb_bnd = nb_pair + 64;
    HItem *cand_lo = reinterpret_cast<HItem *>(nb_bnd + 64);
    HItem *res = cand_lo + kPredLdsCand;  // ef + 1 items
    const int64_t q = redo_first < 0 ? blockIdx.x : redo_first + blockIdx.x;
    const int lane = threadIdx.x;
    uint32_t *vis = visited_ws + q * vis_words;
    if (redo_first >= 0) {
        if (!redo[q]) return;
        for (int64_t w = lane; w < vis_words; w += 64) vis[w] = 0;  // the first pass left its marks
        __threadfence();
        __syncthreads();
    }
    HItem *cand_lo_flat = cand_lo;  // (see vamana_search_kernel: the flat LDS address has to pass through a register)
    asm volatile("" : "+s"(cand_lo_flat));
    const SplitHeap cand{cand_lo_flat, cand_ws + static_cast<int64_t>(blockIdx.x) * cand_cap, kPredLdsCand};
    const uint8_t *mq = mask + q * mask_stride;
    F32ScorerT<false> sc;
    sc.base = base;
    sc.qv = queries + q * dim;
    sc.dim = dim;
    sc.metric = metric;
    sc.sub = Sub16::make(lane);

    // greedySearch through the upper layers (hnsw.go:1897-1934)
    uint32_t cur = entry;
    float cur_d = sc.one(cur);
    int64_t st_descent = 1;
    for (int level = max_level; level > 0; level--) {
        auto row_of = [&](uint32_t node) -> const uint32_t * {
            const uint32_t slot = slots[static_cast<int64_t>(level - 1) * n + node];
            return slot == VG_INVALID_ID ? nullptr : adj + (level_off[level - 1] + slot) * m;
        };
        greedy_layer(sc, lane, row_of, m, nb_pair, nb_bnd, cur, cur_d, &st_descent);
    }

    // initializeSearch + processEntryPoint (hnsw.go:1574-1583)
    int cand_len = 0, res_len = 0;
    int64_t st_visited = 0, st_dc = 0, st_skipped = 0, st_pops = 0, st_dropped = 0;
    if (lane == 0) atomicOr(&vis[cur >> 5], 1u << (cur & 31));
    heap_push<false>(cand, cand_len, HItem{cur, cur_d});
    if (mask_bit(mq, cur) && !(deleted && mask_bit(deleted, cur))) heap_push<true>(res, res_len, HItem{cur, cur_d});
    int misses = 0;  // consecutiveFilterMisses
    __syncthreads();

    while (cand_len > 0) {
        const HItem c = heap_pop<false>(cand, cand_len);
        st_pops++;
        if (res_len >= ef && c.dist > heap_get(res, 0).dist) break;
        const int64_t row = static_cast<int64_t>(c.node) * m0;
        const uint32_t id_lane = lane < m0 ? l0[row + lane] : VG_INVALID_ID;
        const float edge_lane = lane < m0 ? l0_dist[row + lane] : 0.0f;
        const uint64_t inval = __ballot(id_lane == VG_INVALID_ID);
        const int count = inval ? __builtin_ctzll(inval) : 64;
        bool fresh = false;
        if (lane < count) {  // CheckAndVisit for the whole list (a node's neighbour ids are distinct)
            const uint32_t bit = 1u << (id_lane & 31);
            fresh = (atomicOr(&vis[id_lane >> 5], bit) & bit) == 0;
        }
        const uint64_t newmask = __ballot(fresh);
        if (!newmask) continue;
        st_visited += __popcll(newmask);
        const bool passes = fresh && mask_bit(mq, id_lane);
        const bool dead = fresh && deleted && mask_bit(deleted, id_lane);
        const uint64_t passmask = __ballot(passes), livemask = __ballot(passes && !dead);
        // every passing live node is scored: together, ahead of the replay.  A rejected node is scored only where the replay asks
        // for it (no cached edge distance while results < ef/2; past both gates while results < ef) — at the first such node the
        // rest of the list's rejected nodes are scored together (a selective filter navigates on edge distances alone: at 1 %
        // selectivity 2.4 k of 181 k visited nodes are scored)
        sc.many(livemask, id_lane, lane, nb_pair, nb_bnd);
        __syncthreads();
        float my_d = nb_pair[lane];
        bool rej_scored = false;
        auto rejected_dist = [&](int j) {
            if (!rej_scored) {
                sc.many(newmask & ~livemask & ~((1ull << j) - 1), id_lane, lane, nb_pair, nb_bnd);
                __syncthreads();
                my_d = nb_pair[lane];
                rej_scored = true;
            }
            return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_d), j));
        };
        uint64_t todo = newmask;
        while (todo) {
            const int j = __builtin_ctzll(todo);
            todo &= todo - 1;
            const bool p = (passmask >> j) & 1, live = (livemask >> j) & 1;
            misses = p ? 0 : misses + 1;
            const uint32_t id = static_cast<uint32_t>(__builtin_amdgcn_readlane(id_lane, j));
            float nd;
            if (live) {
                nd = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_d), j));
                st_dc++;
            } else if (res_len < ef / 2) {
                const float edge = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(edge_lane), j));
                if (edge > 0.0f) {
                    nd = edge;
                } else {
                    nd = rejected_dist(j);
                    st_dc++;
                }
            } else if (res_len < ef) {
                if (misses > 10) {  // filterMissGateThreshold
                    st_skipped++;
                    continue;
                }
                const float edge = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(edge_lane), j));
                if (edge > 0.0f && res_len > 0 && edge > heap_get(res, 0).dist * 1.5f) {
                    st_skipped++;
                    continue;
                }
                nd = rejected_dist(j);
                st_dc++;
            } else {
                st_skipped++;
                continue;
            }
            if (res_len >= ef && nd > heap_get(res, 0).dist) continue;  // shouldExplore
            if (cand_len >= cand_cap) {  // first pass only (the second has a slot per row: a node is pushed at most once)
                if (redo_first < 0) {
                    if (lane == 0) redo[q] = 1;
                    return;
                }
                st_dropped++;
                continue;
            }
            heap_push<false>(cand, cand_len, HItem{id, nd});
            if (live) res_push_bounded<false>(res, res_len, HItem{id, nd}, ef);
        }
        __syncthreads();
    }

    // knnSearchInternal extraction (hnsw.go:1732-1751)
    while (res_len > k) (void)heap_pop<true>(res, res_len);
    const int nres = res_len;
    for (int i = nres - 1; i >= 0; i--) {
        const HItem it = heap_pop<true>(res, res_len);
        if (lane == 0) {
            ids[q * k + i] = it.node;
            scores[q * k + i] = it.dist;
        }
    }
    for (int i = nres + lane; i < k; i += 64) {
        ids[q * k + i] = VG_INVALID_ID;
        scores[q * k + i] = INFINITY;
    }
    if (stats && lane == 0) {
        stats[q].nodes_visited = st_visited;
        stats[q].distance_computations = st_dc;
        stats[q].distance_short_circuits = st_skipped + (st_dropped << 40);
        stats[q].pops = st_pops;
        stats[q].descent_distance_computations = st_descent;
    }
}

int32_t hnsw_edge_distances(vg_index *idx, const float *l0_dist, hipStream_t st)
{
    const size_t count = static_cast<size_t>(idx->n) * idx->hnsw_m0;
    if (idx->d_hnsw_l0_dist) {
        VG_HIP(hipStreamSynchronize(st));  // earlier searches may still read the old array
        VG_HIP(hipFree(idx->d_hnsw_l0_dist));
        idx->d_hnsw_l0_dist = nullptr;
    }
    if (count == 0) return VG_OK;
    float *d = nullptr;
    VG_HIP(hipMalloc(reinterpret_cast<void **>(&d), count * sizeof(float)));
    if (l0_dist) {
        const hipError_t e = hipMemcpyAsync(d, l0_dist, count * sizeof(float), hipMemcpyDefault, st);
        if (e != hipSuccess) {
            (void)hipFree(d);
            VG_HIP(e);
        }
        VG_HIP(hipStreamSynchronize(st));
    } else {
        const unsigned blocks = static_cast<unsigned>(std::min<int64_t>(idx->n, int64_t(idx->ctx->compute_units) * 64));
        ProfScope prof(idx->ctx, "hnsw_edge_dist", st);
        hipLaunchKernelGGL(hnsw_edge_dist_kernel, dim3(blocks), dim3(64), 0, st, idx->d_vectors, idx->n, idx->dim, idx->metric,
                           idx->d_hnsw_l0, idx->hnsw_m0, d);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) {
            (void)hipFree(d);
            VG_HIP(e);
        }
    }
    idx->d_hnsw_l0_dist = d;
    return VG_OK;
}

}  // namespace vg

VG_API int32_t vg_index_set_hnsw_edge_distances(vg_index *idx, const float *l0_dist, void *stream)
{
    VG_CHECK(idx, VG_ERR_INVALID_ARG, "vg_index_set_hnsw_edge_distances: NULL index");
    VG_CHECK(idx->d_hnsw_l0, VG_ERR_NOT_READY, "vg_index_set_hnsw_edge_distances: index has no HNSW graph");
    VG_CHECK(l0_dist || idx->d_vectors, VG_ERR_NOT_READY, "vg_index_set_hnsw_edge_distances: no distances given and no fp32 vectors to compute them from");
    VG_HIP(hipSetDevice(idx->ctx->device));
    return vg::hnsw_edge_distances(idx, l0_dist, vg::pick_stream(idx->ctx, stream));
}

VG_API int32_t vg_index_set_hnsw_tombstones(vg_index *idx, const uint8_t *deleted, void *stream)
{
    VG_CHECK(idx, VG_ERR_INVALID_ARG, "vg_index_set_hnsw_tombstones: NULL index");
    VG_HIP(hipSetDevice(idx->ctx->device));
    hipStream_t st = vg::pick_stream(idx->ctx, stream);
    VG_HIP(hipStreamSynchronize(st));  // earlier searches may still read the old bitmap
    if (idx->d_hnsw_tomb) {
        VG_HIP(hipFree(idx->d_hnsw_tomb));
        idx->d_hnsw_tomb = nullptr;
    }
    const size_t bytes = static_cast<size_t>((idx->n + 7) / 8);
    if (deleted == nullptr || bytes == 0) return VG_OK;
    uint8_t *d = nullptr;
    VG_HIP(hipMalloc(reinterpret_cast<void **>(&d), bytes));
    const hipError_t e = hipMemcpyAsync(d, deleted, bytes, hipMemcpyDefault, st);
    if (e != hipSuccess) {
        (void)hipFree(d);
        VG_HIP(e);
    }
    VG_HIP(hipStreamSynchronize(st));
    idx->d_hnsw_tomb = d;
    return VG_OK;
}

VG_API int32_t vg_search_hnsw_predicate(vg_index *idx, const float *queries, int64_t nq, int32_t k, int32_t ef, const uint8_t *mask,
                                        int64_t mask_stride, const uint8_t *deleted, uint32_t *ids, float *scores,
                                        vg_search_stats *stats, void *stream)
{
    const char *fn = "vg_search_hnsw_predicate";
    VG_CHECK(idx, VG_ERR_INVALID_ARG, "%s: NULL index", fn);
    VG_CHECK(nq >= 0 && k >= 0, VG_ERR_INVALID_ARG, "%s: negative nq or k", fn);
    if (nq == 0 || k == 0) return VG_OK;
    VG_CHECK(mask, VG_ERR_INVALID_ARG, "%s: NULL mask (vg_search_hnsw is the unfiltered walk)", fn);
    VG_CHECK(idx->d_hnsw_l0, VG_ERR_NOT_READY, "%s: index has no HNSW graph", fn);
    VG_CHECK(idx->d_vectors, VG_ERR_NOT_READY, "%s: index has no fp32 vectors", fn);
    VG_CHECK(idx->metric != VG_METRIC_HAMMING, VG_ERR_UNSUPPORTED, "unsupported metric for float32: Hamming");
    VG_CHECK(queries && ids && scores, VG_ERR_INVALID_ARG, "%s: NULL buffer", fn);
    const int64_t mask_bytes = (idx->n + 7) / 8;
    VG_CHECK(mask_stride == 0 || mask_stride >= mask_bytes, VG_ERR_INVALID_ARG, "%s: mask_stride %lld is shorter than a mask (%lld bytes)",
             fn, static_cast<long long>(mask_stride), static_cast<long long>(mask_bytes));
    if (ef < k) ef = k;  // determineEF hnsw.go:1891-1894
    VG_CHECK(ef <= vg::kPredMaxEf, VG_ERR_UNSUPPORTED, "%s: ef=%d exceeds %d (the results heap lives in LDS)", fn, ef, vg::kPredMaxEf);
    VG_HIP(hipSetDevice(idx->ctx->device));
    hipStream_t st = vg::pick_stream(idx->ctx, stream);
    if (!idx->d_hnsw_l0_dist) VG_TRY(vg::hnsw_edge_distances(idx, nullptr, st));  // recomputed from the rows, once
    vg::DevIn<float> q;
    vg::DevIn<uint8_t> mk, dl;
    vg::DevOut<uint32_t> oid;
    vg::DevOut<float> osc;
    vg::DevOut<vg_search_stats> ost;
    VG_TRY(q.init(queries, static_cast<size_t>(nq) * idx->dim, st));
    VG_TRY(mk.init(mask, static_cast<size_t>(mask_stride ? (nq - 1) * mask_stride + mask_bytes : mask_bytes), st));
    VG_TRY(dl.init(deleted, deleted ? static_cast<size_t>(mask_bytes) : 0, st));
    VG_TRY(oid.init(ids, static_cast<size_t>(nq) * k, st));
    VG_TRY(osc.init(scores, static_cast<size_t>(nq) * k, st));
    VG_TRY(ost.init(stats, stats ? static_cast<size_t>(nq) : 0, st));
    const int64_t vis_words = (idx->n + 31) / 32;
    // The navigation queue is unbounded in the reference and can take every row (a filter nothing passes walks the whole
    // component on edge distances).  First pass: 128 k slots per query (an ordinary walk queues a few thousand nodes); a walk that
    // outgrows them stops and is run again with a slot per row, a few queries per launch (2 GiB of scratch, returned after the call)
    const int64_t cand_cap = std::min<int64_t>(idx->n, int64_t(1) << 17);
    const int64_t per_query = vis_words * 4 + cand_cap * 8 + 1;
    const int64_t gib = int64_t(1) << 30;
    const int64_t scratch = std::min<int64_t>(16 * gib, std::max<int64_t>(gib, idx->ctx->hbm_bytes / 16));
    int64_t chunk = std::max<int64_t>(1, scratch / per_query);
    chunk = std::min(chunk, nq);
    vg::ArenaCall ar(idx->ctx, st);
    const int i_vis = ar.add(sizeof(uint32_t) * static_cast<size_t>(chunk) * vis_words);
    const int i_cand = ar.add(sizeof(vg::HItem) * static_cast<size_t>(chunk) * cand_cap);
    const int i_redo = ar.add(static_cast<size_t>(chunk));
    VG_TRY(ar.commit());
    uint8_t *redo = ar.get<uint8_t>(i_redo);
    const bool second_pass = cand_cap < idx->n;
    const int64_t big_chunk = second_pass ? std::max<int64_t>(1, (int64_t(2) << 30) / (idx->n * 8)) : 0;
    vg::DevTmp<vg::HItem> big;
    if (second_pass) VG_TRY(big.init(static_cast<size_t>(std::min(big_chunk, chunk)) * idx->n, st));
    uint32_t *vis = ar.get<uint32_t>(i_vis);
    vg::HItem *cand = ar.get<vg::HItem>(i_cand);
    const size_t lds = 128 * sizeof(float) + sizeof(vg::HItem) * (vg::kPredLdsCand + static_cast<size_t>(ef) + 1);
    VG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(vg::hnsw_predicate_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               static_cast<int>(lds)));
    for (int64_t q0 = 0; q0 < nq; q0 += chunk) {
        const int64_t cnt = std::min(chunk, nq - q0);
        VG_HIP(hipMemsetAsync(vis, 0, static_cast<size_t>(cnt) * vis_words * 4, st));
        VG_HIP(hipMemsetAsync(redo, 0, static_cast<size_t>(cnt), st));
        vg::ProfScope prof(idx->ctx, "hnsw_predicate", st);
        VG_LAUNCH(vg::hnsw_predicate_kernel, dim3(static_cast<unsigned>(cnt)), dim3(64), lds, st, idx->d_vectors, idx->n, idx->dim,
                  idx->metric, idx->d_hnsw_l0, idx->d_hnsw_l0_dist, idx->hnsw_m0, idx->hnsw_max_level, idx->hnsw_m, idx->d_hnsw_slot,
                  idx->d_hnsw_adj, idx->d_hnsw_level_off, idx->hnsw_entry, q.ptr + q0 * idx->dim, k, ef, mk.ptr + q0 * mask_stride,
                  mask_stride, deleted ? dl.ptr : idx->d_hnsw_tomb, vis, vis_words, cand, cand_cap, oid.ptr + q0 * k, osc.ptr + q0 * k,
                  ost.ptr ? ost.ptr + q0 : nullptr, redo, int64_t(-1));
        for (int64_t r0 = 0; second_pass && r0 < cnt; r0 += big_chunk) {  // (every workgroup of an ordinary batch leaves at once)
            const int64_t rc = std::min(big_chunk, cnt - r0);
            VG_LAUNCH(vg::hnsw_predicate_kernel, dim3(static_cast<unsigned>(rc)), dim3(64), lds, st, idx->d_vectors, idx->n, idx->dim,
                      idx->metric, idx->d_hnsw_l0, idx->d_hnsw_l0_dist, idx->hnsw_m0, idx->hnsw_max_level, idx->hnsw_m,
                      idx->d_hnsw_slot, idx->d_hnsw_adj, idx->d_hnsw_level_off, idx->hnsw_entry, q.ptr + q0 * idx->dim, k, ef,
                      mk.ptr + q0 * mask_stride, mask_stride, deleted ? dl.ptr : idx->d_hnsw_tomb, vis, vis_words, big.ptr, idx->n,
                      oid.ptr + q0 * k, osc.ptr + q0 * k, ost.ptr ? ost.ptr + q0 : nullptr, redo, r0);
        }
    }
    VG_TRY(oid.finish());
    VG_TRY(osc.finish());
    VG_TRY(ost.finish());
    if (oid.on_host() || osc.on_host() || ost.on_host()) VG_HIP(hipStreamSynchronize(st));
    return VG_OK;
}
