// vg_hnsw_layer.hpp — one wavefront's walk over one HNSW layer, shared by the query kernel (k_graph.hip) and
// the builder (k_hnsw_build.hip):
//   greedy_layer   the `for changed` loop of greedySearch / insertNode    (hnsw.go:1897-1934, :918-934)
//   search_layer   searchLayerUnfiltered                                 (hnsw.go:1220-1396)
// The heaps are the reference's (vg_heap.hpp), executed uniformly by the wave on arrays the caller owns
// (LDS; for large ef SPLIT between LDS and HBM scratch, vg_heap.hpp); the <= 64 neighbours of a popped node are visited-tested, gathered
// and scored in parallel (16 lanes per fp32 row, 4 rows at a time, reference summation order), then fed
// to the heaps in the node's stored neighbour order.
#pragma once

#include "vg_device.hpp"
#include "vg_exact.hpp"
#include "../../include/vecgo_hip.h"
#include "vg_heap.hpp"

namespace vg {

struct LayerStats {
    int64_t visited = 0, dc = 0, sc = 0, pops = 0;
};

// distance of one node as hnsw wraps it (vectorstore/columnar.go:37-44), all lanes of the 16-lane group
__device__ __forceinline__ float hnsw_node_dist(const float *__restrict__ base, int dim, int metric,
                                                const float *__restrict__ qv, uint32_t id, Sub16 sub)
{
    const float *row = base + static_cast<int64_t>(id) * dim;
    if (metric == kMetricDot) return -exact_pair16<true, kPair>(row, qv, dim, sub);
    const float d = exact_pair16<false, kPair>(row, qv, dim, sub);
    return metric == kMetricCos ? 0.5f * d : d;
}

// ---- node scorers ------------------------------------------------------------------------------------
// A scorer fills nb_pair[j] (what distFunc returns) and nb_bnd[j] (what SquaredL2Bounded run to completion
// returns) for every lane j set in `mask`, lane j holding node id_lane.
struct F32Scorer {  // fp32 rows: 16 lanes per row, 4 rows at a time, reference summation order
    const float *base;
    const float *qv;
    int dim, metric;
    Sub16 sub;
    static constexpr bool kBounded = true;  // SquaredL2Bounded exists for the L2 metric (hnsw.go:1353-1366)
    __device__ __forceinline__ float one(uint32_t id) const { return hnsw_node_dist(base, dim, metric, qv, id, sub); }
    __device__ __forceinline__ void many(uint64_t mask, uint32_t id_lane, int lane, float *nb_pair, float *nb_bnd) const
    {
        while (mask) {
            const int mine = take4(mask, lane);
            const uint32_t id = __shfl(id_lane, mine < 0 ? 0 : mine);
            if (mine >= 0) {
                const float *row = base + static_cast<int64_t>(id) * dim;
                float dp, db;
                if (metric == kMetricDot) {
                    dp = -exact_pair16<true, kPair>(row, qv, dim, sub);
                    db = dp;
                } else {
                    exact_l2_both16(row, qv, dim, sub, dp, db);
                    if (metric == kMetricCos) dp = 0.5f * dp;
                }
                if ((lane & 15) == 0) {
                    nb_pair[mine] = dp;
                    nb_bnd[mine] = db;
                }
            }
        }
    }
};

// ComputeAsymmetricDistance's sum over NG groups of 16 sub-quantizers starting at s0: the NG 16-byte code loads are
// issued together, then the 16*NG table reads (their addresses depend on the code bytes) together — two dependent
// round trips for the whole chunk — and the terms are added in sub-quantizer order, as the reference adds them.
template <int NG>
__device__ __forceinline__ float pq_asym_chunk(const uint8_t *__restrict__ code, const float *__restrict__ lut, int s0,
                                               float distance)
{
    uint4 c[NG];
#pragma unroll
    for (int g = 0; g < NG; g++) c[g] = *reinterpret_cast<const uint4 *>(code + s0 + 16 * g);
    float t[NG * 16];
#pragma unroll
    for (int g = 0; g < NG; g++) {
        const uint32_t w[4] = {c[g].x, c[g].y, c[g].z, c[g].w};
#pragma unroll
        for (int u = 0; u < 16; u++) t[g * 16 + u] = lut[(s0 + g * 16 + u) * 256 + ((w[u >> 2] >> (8 * (u & 3))) & 0xFFu)];
    }
#pragma unroll
    for (int i = 0; i < NG * 16; i++) distance = distance + t[i];
    return distance;
}

// pq.ComputeAsymmetricDistance (pq.go:234-260) of one node's code from the query's table (m * 256 floats in HBM/L2):
// term(s) = the BuildDistanceTable entry of code byte s, summed sequentially over the sub-quantizers.
__device__ __forceinline__ float pq_asym_distance(const uint8_t *__restrict__ code, const float *__restrict__ lut, int m)
{
    float distance = 0.0f;
    int s0 = 0;
    if ((m & 15) == 0) {  // rows of 16-byte multiples are 16-byte aligned (the code array is)
        for (; s0 + 96 <= m; s0 += 96) distance = pq_asym_chunk<6>(code, lut, s0, distance);
        for (; s0 + 32 <= m; s0 += 32) distance = pq_asym_chunk<2>(code, lut, s0, distance);
        for (; s0 + 16 <= m; s0 += 16) distance = pq_asym_chunk<1>(code, lut, s0, distance);
    }
    for (int s = s0; s < m; s++) distance = distance + lut[s * 256 + code[s]];
    return distance;
}

// PQ codes: pq.ComputeAsymmetricDistance (pq.go:234-260) — the way the reference scores graph nodes from PQ
// codes (diskann/segment.go:536-557): term(s) = the BuildDistanceTable entry of the node's code byte, summed
// sequentially over the sub-quantizers.  One node per lane; the query's table (m * 256 floats) is in HBM/L2.
struct PqScorer {
    const uint8_t *rows;  // n * m code bytes
    const float *lut;     // m * 256
    int m;
    static constexpr bool kBounded = false;
    __device__ __forceinline__ float lane_score(uint32_t id) const
    {
        return pq_asym_distance(rows + static_cast<int64_t>(id) * m, lut, m);
    }
    __device__ __forceinline__ float one(uint32_t id) const { return lane_score(id); }
    __device__ __forceinline__ void many(uint64_t mask, uint32_t id_lane, int lane, float *nb_pair, float *nb_bnd) const
    {
        if ((mask >> lane) & 1) {
            const float d = lane_score(id_lane);
            nb_pair[lane] = d;
            nb_bnd[lane] = d;
        }
    }
};

// greedy descent on one layer: repeat { for each neighbour in list order: if nextDist < currDist take it }
// until a pass changes nothing.  row_of(node) -> the node's list on this layer (deg ids, 0xFFFFFFFF ends
// it) or nullptr.  nb_pair: 64 floats of LDS.
template <typename Scorer, typename RowFn>
__device__ __forceinline__ void greedy_layer(const Scorer &sc, int lane, RowFn row_of, int deg, float *nb_pair,
                                             float *nb_bnd, uint32_t &cur, float &cur_d, int64_t *scored = nullptr)
{
    bool changed = true;
    while (changed) {
        changed = false;
        const uint32_t *nbp = row_of(cur);
        if (nbp == nullptr) break;
        const uint32_t id_lane = lane < deg ? nbp[lane] : VG_INVALID_ID;
        const uint64_t inval = __ballot(id_lane == VG_INVALID_ID);
        const int count = inval ? __builtin_ctzll(inval) : 64;
        uint64_t mask = count >= 64 ? ~0ull : ((1ull << count) - 1);
        if (scored) *scored += count;
        sc.many(mask, id_lane, lane, nb_pair, nb_bnd);
        __syncthreads();
        // sequential `if nextDist < currDist` over the list == first strict minimum
        float best_d = cur_d;
        uint32_t best_id = cur;
        for (int i = 0; i < count; i++) {
            const float d = nb_pair[i];
            if (d < best_d) {
                best_d = d;
                best_id = static_cast<uint32_t>(__builtin_amdgcn_readlane(id_lane, i));
                changed = true;
            }
        }
        cur = best_id;
        cur_d = best_d;
        __syncthreads();
    }
}

// searchLayerUnfiltered from (ep, ep_d).  `vis`: this wave's visited bitmap, already clear.  On return
// res[0..res_len) is the results max-heap exactly as the reference's search leaves it.
template <typename Scorer, typename RowFn, typename Heap>
__device__ __forceinline__ void search_layer(const Scorer &sc, bool l2_metric, int lane, RowFn row_of, int deg,
                                             uint32_t ep, float ep_d, int ef, Heap cand, Heap res,
                                             float *nb_pair, float *nb_bnd, uint32_t *vis, int &res_len_out,
                                             LayerStats &st)
{
    int cand_len = 0, res_len = 0;
    if (lane == 0) atomicOr(&vis[ep >> 5], 1u << (ep & 31));
    heap_push<false>(cand, cand_len, HItem{ep, ep_d});
    heap_push<true>(res, res_len, HItem{ep, ep_d});
    const bool use_sc = Scorer::kBounded && l2_metric;
    int cap = ef * 2;
    int stagnant = 0;
    float last_best = 3.40282346638528859811704183484516925440e+38f;
    const int min_cap = ef + ef * 3 / 4;
    __syncthreads();

    while (cand_len > 0) {
        const HItem c = heap_pop<false>(cand, cand_len);
        st.pops++;
        if (res_len > 0) {
            const float worst = heap_get(res, 0).dist;
            if (c.dist > worst && res_len >= ef) break;
            if (worst < last_best * 0.999f) {
                last_best = worst;
                stagnant = 0;
            } else if (res_len >= ef) {
                stagnant++;
                if (stagnant >= 8 && cap > min_cap) {
                    cap -= ef / 8;
                    if (cap < min_cap) cap = min_cap;
                    stagnant = 0;
                }
            }
        }
        const uint32_t *nbp = row_of(c.node);
        const uint32_t id_lane = (nbp != nullptr && lane < deg) ? nbp[lane] : VG_INVALID_ID;
        const uint64_t inval = __ballot(id_lane == VG_INVALID_ID);
        const int count = inval ? __builtin_ctzll(inval) : 64;
        // CheckAndVisit for the whole list at once (neighbour ids of a node are distinct)
        // (a returning L2 atomic: a plain load could hit a stale L1 line of this very bitmap)
        bool fresh = false;
        if (lane < count) {
            const uint32_t bit = 1u << (id_lane & 31);
            fresh = (atomicOr(&vis[id_lane >> 5], bit) & bit) == 0;
        }
        const uint64_t newmask = __ballot(fresh);
        st.visited += __popcll(newmask);
        sc.many(newmask, id_lane, lane, nb_pair, nb_bnd);
        __syncthreads();
        bool has_bound = res_len >= ef;
        float bound = has_bound ? heap_get(res, 0).dist : 0.0f;
        uint64_t todo = newmask;
        while (todo) {
            const int j = __builtin_ctzll(todo);
            todo &= todo - 1;
            const uint32_t id = static_cast<uint32_t>(__builtin_amdgcn_readlane(id_lane, j));
            float nd;
            st.dc++;
            if (use_sc && has_bound) {
                nd = nb_bnd[j];
                if (nd > bound) {  // SquaredL2Bounded reported exceeded
                    st.sc++;
                    continue;
                }
            } else {
                nd = nb_pair[j];
            }
            if (has_bound && nd > bound) continue;
            cand_try_push_bounded(cand, cand_len, HItem{id, nd}, cap);
            res_push_bounded(res, res_len, HItem{id, nd}, ef);
            if (res_len >= ef) {
                bound = heap_get(res, 0).dist;
                has_bound = true;
            }
        }
        __syncthreads();
    }
    res_len_out = res_len;
}

}  // namespace vg
