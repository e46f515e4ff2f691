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
#ifdef VG_WALK_TIMING  // stage probe (tools/build_variant.sh): s_memtime cycles per phase, reported in the stats columns
    int64_t t_pop = 0, t_adj = 0, t_score = 0, t_push = 0, t_cand = 0, t_res = 0, n_push = 0;
#endif
};
#ifdef VG_WALK_TIMING
#define VG_T(var) const int64_t var = static_cast<int64_t>(__builtin_readcyclecounter())
#define VG_TACC(acc, a, b) (acc) += (b) - (a)
#else
#define VG_T(var)
#define VG_TACC(acc, a, b)
#endif

// distance of one node as hnsw wraps it (vectorstore/columnar.go:37-44), all lanes of the 16-lane group
template <bool QLDS = false>
__device__ __forceinline__ float hnsw_node_dist(const float *__restrict__ base, int dim, int metric,
                                                const float *__restrict__ qv, uint32_t id, Sub16 sub)
{
    const float *row = base + static_cast<int64_t>(id) * dim;
    if (metric == kMetricDot) return -exact_pair16<true, kPair, false, QLDS>(row, qv, dim, sub);
    const float d = exact_pair16<false, kPair, false, QLDS>(row, qv, dim, sub);
    return metric == kMetricCos ? 0.5f * d : d;
}

// ---- node scorers ------------------------------------------------------------------------------------
// A scorer fills nb_pair[j] (what distFunc returns) and nb_bnd[j] (what SquaredL2Bounded run to completion
// returns) for every lane j set in `mask`, lane j holding node id_lane.
template <bool QLDS>
struct F32ScorerT {  // fp32 rows: 16 lanes per row, 4 rows at a time, reference summation order; QLDS: qv points into LDS
    const float *base;
    const float *qv;
    int dim, metric;
    Sub16 sub;
    static constexpr bool kBounded = true;  // SquaredL2Bounded exists for the L2 metric (hnsw.go:1353-1366)
    __device__ __forceinline__ static void sync() { __syncthreads(); }  // one wave per workgroup
    __device__ __forceinline__ float one(uint32_t id) const { return hnsw_node_dist<QLDS>(base, dim, metric, qv, id, sub); }
    __device__ __forceinline__ void many(uint64_t mask, uint32_t id_lane, int lane, float *nb_pair, float *nb_bnd) const
    {
        while (mask) {
            const int mine = take4(mask, lane);
            const uint32_t id = __shfl(id_lane, mine < 0 ? 0 : mine);
            if (mine >= 0) {
                const float *row = base + static_cast<int64_t>(id) * dim;
                float dp, db;
                if (metric == kMetricDot) {
                    dp = -exact_pair16<true, kPair, false, QLDS>(row, qv, dim, sub);
                    db = dp;
                } else {
                    exact_l2_both16<false, QLDS>(row, qv, dim, sub, dp, db);
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

using F32Scorer = F32ScorerT<false>;

// ---- ComputeAsymmetricDistance without a table (sub-dimension 8) ------------------------------------------------
// term(s) = squaredL2Int8DequantizedGeneric (kernels.go:354-362) of the node's centroid of sub-quantizer s: per
// dimension  v = float32(int8)*scale ; v = v + offset ; d = q - v ; dd = d*d ; sum = sum + dd  — five separately
// rounded fp32 operations, summed in order: bit for bit the BuildDistanceTable entry of that centroid
// (pq.go:468-491), computed from the quantizer's own int8 codebook (m * 256 * 8 bytes, shared by every query: L2 /
// L1 hits) instead of being looked up in a per-query table.
// One node per lane, TWO sub-quantizers at a time: every operation of the pair (s, s+1) is one packed-fp32
// instruction on the register pair (term s, term s+1) — v_pk_mul_f32 / v_pk_add_f32 round each half exactly like
// the scalar instruction — so a term costs 8 conversions + 20 half-instructions instead of 48.  The pair's constants
// (its 2 x 8 query floats interleaved, the two scales, the two offsets: 20 floats) are laid out once per query in
// the wave's LDS (pq_direct_prepare) and arrive by broadcast reads: no scalar-load round trip per term (the first
// version waited ~96 of them per call), no vector instruction spent on constants.
typedef float vg_f2 __attribute__((ext_vector_type(2)));
constexpr int kPqPairFloats = 20;  // per pair of sub-quantizers: q interleaved [16], scales [2], offsets [2]

// the wave lays out its query's constants: qprep[(s/2)*20 + ...]; m even
__device__ __forceinline__ void pq_direct_prepare(float *qprep, const float *__restrict__ qv,
                                                  const float *__restrict__ scales, const float *__restrict__ offsets,
                                                  int m, int lane)
{
    for (int e = lane; e < (m >> 1) * kPqPairFloats; e += 64) {
        const int p = e / kPqPairFloats, r = e - p * kPqPairFloats;
        float v;
        if (r < 16)
            v = qv[(2 * p + (r & 1)) * 8 + (r >> 1)];
        else if (r < 18)
            v = scales[2 * p + (r - 16)];
        else
            v = offsets[2 * p + (r - 18)];
        qprep[e] = v;
    }
}

// terms of sub-quantizers (s, s+1): centroids ea / eb, constants at `pc` (LDS); returns (term s, term s+1)
__device__ __forceinline__ vg_f2 pq_term8_pair(uint2 ea, uint2 eb, const float *pc)
{
    const float4 *c4 = reinterpret_cast<const float4 *>(pc);
    const float4 k0 = c4[0], k1 = c4[1], k2 = c4[2], k3 = c4[3], k4 = c4[4];
    const vg_f2 q[8] = {{k0.x, k0.y}, {k0.z, k0.w}, {k1.x, k1.y}, {k1.z, k1.w},
                        {k2.x, k2.y}, {k2.z, k2.w}, {k3.x, k3.y}, {k3.z, k3.w}};
    const vg_f2 scale = {k4.x, k4.y}, offset = {k4.z, k4.w};
    vg_f2 sum = {0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint32_t wa = j < 4 ? ea.x : ea.y, wb = j < 4 ? eb.x : eb.y;
        const vg_f2 f = {static_cast<float>(static_cast<int>(static_cast<int8_t>(wa >> (8 * (j & 3))))),
                         static_cast<float>(static_cast<int>(static_cast<int8_t>(wb >> (8 * (j & 3)))))};
        vg_f2 v = f * scale;
        v = v + offset;
        const vg_f2 d = q[j] - v;
        const vg_f2 dd = d * d;
        sum = sum + dd;
    }
    return sum;
}

// FOUR sub-quantizers (s .. s+3) as two packed pairs whose dependent chains are interleaved statement by statement: a
// packed-fp32 result cannot feed the next instruction (the compiler pads every link of a lone chain with an s_nop:
// 0.7 per v_pk op in the r03 walk), so a lane that carries one chain issues at half rate; two chains fill each other's
// gaps.  Every operation and its operands are pq_term8_pair's: (term s, term s+1) -> a, (term s+2, term s+3) -> b.
__device__ __forceinline__ void pq_term8_quad(uint2 e0, uint2 e1, uint2 e2, uint2 e3, const float *pc, vg_f2 &a, vg_f2 &b)
{
    const float4 *c4 = reinterpret_cast<const float4 *>(pc);
    const float4 k0 = c4[0], k1 = c4[1], k2 = c4[2], k3 = c4[3], k4 = c4[4];
    const float4 m0 = c4[5], m1 = c4[6], m2 = c4[7], m3 = c4[8], m4 = c4[9];  // the next pair's constants: + kPqPairFloats floats
    const vg_f2 qa[8] = {{k0.x, k0.y}, {k0.z, k0.w}, {k1.x, k1.y}, {k1.z, k1.w},
                         {k2.x, k2.y}, {k2.z, k2.w}, {k3.x, k3.y}, {k3.z, k3.w}};
    const vg_f2 qb[8] = {{m0.x, m0.y}, {m0.z, m0.w}, {m1.x, m1.y}, {m1.z, m1.w},
                         {m2.x, m2.y}, {m2.z, m2.w}, {m3.x, m3.y}, {m3.z, m3.w}};
    const vg_f2 sa = {k4.x, k4.y}, oa = {k4.z, k4.w}, sb = {m4.x, m4.y}, ob = {m4.z, m4.w};
    vg_f2 suma = {0.0f, 0.0f}, sumb = {0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int sh = 8 * (j & 3);
        const uint32_t w0 = j < 4 ? e0.x : e0.y, w1 = j < 4 ? e1.x : e1.y, w2 = j < 4 ? e2.x : e2.y, w3 = j < 4 ? e3.x : e3.y;
        const vg_f2 fa = {static_cast<float>(static_cast<int>(static_cast<int8_t>(w0 >> sh))),
                          static_cast<float>(static_cast<int>(static_cast<int8_t>(w1 >> sh)))};
        const vg_f2 fb = {static_cast<float>(static_cast<int>(static_cast<int8_t>(w2 >> sh))),
                          static_cast<float>(static_cast<int>(static_cast<int8_t>(w3 >> sh)))};
        vg_f2 va = fa * sa;
        vg_f2 vb = fb * sb;
        va = va + oa;
        vb = vb + ob;
        const vg_f2 da = qa[j] - va;
        const vg_f2 db = qb[j] - vb;
        const vg_f2 dda = da * da;
        const vg_f2 ddb = db * db;
        suma = suma + dda;
        sumb = sumb + ddb;
    }
    a = suma;
    b = sumb;
}

// scalar form (an odd last sub-quantizer; constants from global memory)
__device__ __forceinline__ float pq_term8(uint2 e, const float *__restrict__ q, float scale, float offset)
{
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint32_t w = j < 4 ? e.x : e.y;
        float v = static_cast<float>(static_cast<int>(static_cast<int8_t>(w >> (8 * (j & 3))))) * scale;
        v = v + offset;
        const float d = q[j] - v;
        const float dd = d * d;
        sum = sum + dd;
    }
    return sum;
}

#ifndef VG_PQ_DIRECT_NG
#define VG_PQ_DIRECT_NG 2  // groups of 16 centroid loads in flight per lane (32 registers per group)
#endif
// NG groups of 16 sub-quantizers from s0: the NG code loads, then the 16 * NG centroid loads, are issued together;
// the terms are added in sub-quantizer order
template <int NG>
__device__ __forceinline__ float pq_direct_chunk(const uint8_t *__restrict__ code, const uint2 *__restrict__ cb,
                                                 const float *qprep, int s0, float distance)
{
    uint4 c[NG];
#pragma unroll
    for (int g = 0; g < NG; g++) c[g] = *reinterpret_cast<const uint4 *>(code + s0 + 16 * g);
    uint2 e[NG * 16];
#pragma unroll
    for (int g = 0; g < NG; g++) {
        const uint32_t w[4] = {c[g].x, c[g].y, c[g].z, c[g].w};
#pragma unroll
        for (int u = 0; u < 16; u++) e[g * 16 + u] = cb[(s0 + g * 16 + u) * 256 + ((w[u >> 2] >> (8 * (u & 3))) & 0xFFu)];
    }
#if defined(VG_PQ_LOAD_PROBE) && VG_PQ_LOAD_PROBE == 1  // stage probe (tools/build_variant.sh): every centroid gather twice
#pragma unroll
    for (int g = 0; g < NG; g++) {
        const uint32_t w[4] = {c[g].x, c[g].y, c[g].z, c[g].w};
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const uint2 x = cb[(s0 + g * 16 + u) * 256 + (((w[u >> 2] >> (8 * (u & 3))) & 0xFFu) ^ 0x55u)];
            asm volatile("" ::"v"(x.x), "v"(x.y));
        }
    }
#endif
#pragma unroll
    for (int i = 0; i < NG * 16; i += 4) {  // s0 is a multiple of 16: whole quads, pairs (s0 + i) / 2 and the next one
        vg_f2 ta, tb;
#if defined(VG_PQ_LOAD_PROBE) && VG_PQ_LOAD_PROBE == 2  // stage probe: every term's arithmetic twice
        {
            uint2 a0 = e[i], a1 = e[i + 1], a2 = e[i + 2], a3 = e[i + 3];
            asm volatile("" : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x));
            vg_f2 ua, ub;
            pq_term8_quad(a0, a1, a2, a3, qprep + ((s0 + i) >> 1) * kPqPairFloats, ua, ub);
            asm volatile("" ::"v"(ua.x), "v"(ua.y), "v"(ub.x), "v"(ub.y));
        }
#endif
        pq_term8_quad(e[i], e[i + 1], e[i + 2], e[i + 3], qprep + ((s0 + i) >> 1) * kPqPairFloats, ta, tb);
        distance = distance + ta.x;  // the terms join the sum in sub-quantizer order (pq.go:242-257)
        distance = distance + ta.y;
        distance = distance + tb.x;
        distance = distance + tb.y;
    }
    return distance;
}

__device__ __forceinline__ float pq_direct_distance(const uint8_t *__restrict__ code, const int8_t *__restrict__ codebooks,
                                                    const float *__restrict__ scales, const float *__restrict__ offsets,
                                                    const float *__restrict__ qv, const float *qprep, int m)
{
    const uint2 *cb = reinterpret_cast<const uint2 *>(codebooks);
    float distance = 0.0f;
    int s0 = 0;
    if ((m & 15) == 0) {  // rows of 16-byte multiples are 16-byte aligned (the code array is)
        for (; s0 + 16 * VG_PQ_DIRECT_NG <= m; s0 += 16 * VG_PQ_DIRECT_NG)
            distance = pq_direct_chunk<VG_PQ_DIRECT_NG>(code, cb, qprep, s0, distance);
        for (; s0 + 16 <= m; s0 += 16) distance = pq_direct_chunk<1>(code, cb, qprep, s0, distance);
    }
    for (; s0 + 2 <= m; s0 += 2) {
        const vg_f2 t = pq_term8_pair(cb[s0 * 256 + code[s0]], cb[(s0 + 1) * 256 + code[s0 + 1]],
                                      qprep + (s0 >> 1) * kPqPairFloats);
        distance = distance + t.x;
        distance = distance + t.y;
    }
    if (s0 < m) distance = distance + pq_term8(cb[s0 * 256 + code[s0]], qv + s0 * 8, scales[s0], offsets[s0]);
    return distance;
}

// PQ codes: pq.ComputeAsymmetricDistance (pq.go:234-260) — the way the reference scores graph nodes from PQ
// codes (diskann/segment.go:536-557): term(s) = squaredL2Int8Dequantized of the node's centroid, summed
// sequentially over the sub-quantizers.  One node per lane.
// r02 read every term from the query's BuildDistanceTable image (m * 256 floats = 96 KiB) in global memory: a popped
// node's ~58 fresh neighbours look up random centroids of every sub-quantizer, so one pop pulls nearly the whole
// table through L1, and thousands of resident queries' tables (805 MB for 8192 queries) live in neither L2 nor the
// 256 MB memory-side cache — the walk ran at the HBM rate of its TABLE traffic (18.9 GB per launch against 0.93 GB
// of codes, profiles/r02_traffic.json).  `cb` != nullptr (sub-dimension 8): the terms are computed from the shared
// codebook instead, as the reference computes them; nothing per query is built or kept in memory.
// DIRECT: one kernel instance per form (the table form's loads beside the direct form's set a higher register budget)
template <bool DIRECT>
struct PqScorerT {
    const uint8_t *rows;  // n * m code bytes
    const float *lut;     // m * 256 (table form: sub-dimensions other than 8)
    const int8_t *cb;     // m * 256 * 8 int8 (direct form) or nullptr
    const float *scales, *offsets, *qv;
    const float *qprep;   // LDS: pq_direct_prepare's image of this query (direct form)
    int m;
    static constexpr bool kBounded = false;
    __device__ __forceinline__ static void sync() { __syncthreads(); }
    __device__ __forceinline__ float lane_score(uint32_t id) const
    {
#if defined(VG_PQ_PROBE) && VG_PQ_PROBE == 2  // stage probe (tools/build_variant.sh): no scoring work at all
        return __uint_as_float(0x40000000u | (((id * 2654435761u) ^ static_cast<uint32_t>(reinterpret_cast<uintptr_t>(qv) >> 4)) >> 9));
#endif
        const uint8_t *code = rows + static_cast<int64_t>(id) * m;
        if constexpr (DIRECT)
            return pq_direct_distance(code, cb, scales, offsets, qv, qprep, m);
        else
            return pq_asym_distance(code, lut, m);
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
        Scorer::sync();
        // sequential `if nextDist < currDist` over the list == the smallest distance, first of its equals, if it
        // is below currDist: a wave minimum and one ballot instead of `count` dependent LDS reads
        // (lanes past the list, and NaN distances, stand as +Inf: neither can ever be `< currDist`.  r04 padded with
        // MaxFloat32: against a list of +Inf distances — an Inf in the query — the minimum was then the PADDING, no lane
        // held it, and the lane index taken from an empty ballot read a wild neighbour id: a memory fault)
        const float raw_d = lane < count ? nb_pair[lane] : INFINITY;
        const float my_d = raw_d == raw_d ? raw_d : INFINITY;
        float mn = my_d;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float other = __shfl_xor(mn, o);
            mn = other < mn ? other : mn;
        }
        float best_d = cur_d;
        uint32_t best_id = cur;
        if (count > 0 && mn < cur_d) {  // mn is finite here: some lane of the list holds it
            const uint64_t at = __ballot(lane < count && my_d == mn);
            const int first = __builtin_ctzll(at);
            best_d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_d), first));  // its own bits (-0 == +0)
            best_id = static_cast<uint32_t>(__builtin_amdgcn_readlane(id_lane, first));
            changed = true;
        }
        cur = best_id;
        cur_d = best_d;
        Scorer::sync();
    }
}

// searchLayerUnfiltered from (ep, ep_d).  `vis`: this wave's visited bitmap, already clear.  On return
// res[0..res_len) is the results max-heap exactly as the reference's search leaves it.
// UK: every distance is >= +0 (any metric but Dot), see heap_sift_down_uk (vg_heap.hpp)
// NaN distances (a NaN or an Inf in a row or in the query).  Every comparison with a NaN is FALSE in the reference
// (queue.go:75-82,199-203; hnsw.go:1376 `nextDist > bound`), which this walk relies on NOT happening in two places: the
// unsigned-key sifts order a NaN's bit pattern as the largest key, and the neighbour loop rejects every lane above the
// bound at once because "the bound only falls" — with a NaN inside the results heap a sift-down can bring it to the
// root, and against a NaN bound nothing is rejected any more.  So:
//   `odd_out` != nullptr  one ballot over each scored neighbour list watches for a distance that is a NaN (UK: any bit
//                         pattern above +Inf's, which covers negative values too); at the first one the walk stops with
//                         *odd_out = true, and the caller runs the query again, with UK = false and
//   STRICT                every lane tested against the bound at ITS turn, as the reference's loop is written.
// nullptr / false: the fast form (the builder, and every query that meets no NaN).
template <bool UK = false, bool STRICT = false, typename Scorer, typename RowFn, typename Heap>
__device__ __forceinline__ void search_layer(const Scorer &sc, bool l2_metric, int lane, RowFn row_of, int deg,
                                             uint32_t ep, float ep_d, int ef, Heap cand, Heap res,
                                             float *nb_pair, float *nb_bnd, uint32_t *vis, int &res_len_out,
                                             LayerStats &st, bool *odd_out = nullptr, const uint8_t *dead = nullptr)
{
    // dead (STRICT instances only): g.tombstones as a bitmap — a deleted node goes to the exploration queue like any other but
    // never to the results, and so never moves the bound (hnsw.go:1381-1390; entry point :1559-1565)
    constexpr bool strict = STRICT;
    auto is_odd = [](float d) { return UK ? __float_as_uint(d) > 0x7F800000u : d != d; };
    if (odd_out && is_odd(ep_d)) {  // wave-uniform
        *odd_out = true;
        res_len_out = 0;
        return;
    }
    int cand_len = 0, res_len = 0;
    if (lane == 0) atomicOr(&vis[ep >> 5], 1u << (ep & 31));
    heap_push<false>(cand, cand_len, HItem{ep, ep_d});
    if (!(STRICT && dead && mask_bit(dead, ep))) heap_push<true>(res, res_len, HItem{ep, ep_d});
    const bool use_sc = Scorer::kBounded && l2_metric;
    int cap = ef * 2;
    int stagnant = 0;
    float last_best = 3.40282346638528859811704183484516925440e+38f;
    const int min_cap = ef + ef * 3 / 4;
    Scorer::sync();

    while (cand_len > 0) {
        VG_T(t0);
        const HItem c = heap_pop<false, UK>(cand, cand_len);
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
        VG_T(t1);
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
        uint64_t deadmask = 0;
        if constexpr (STRICT) {
            if (dead) deadmask = __ballot(fresh && mask_bit(dead, id_lane));
        }
        st.visited += __popcll(newmask);
        VG_T(t2);
        sc.many(newmask, id_lane, lane, nb_pair, nb_bnd);
        Scorer::sync();
        // lane j keeps node j's two distances; the loop below reads them with readlane, not from LDS
        const float my_pair = nb_pair[lane];
        const float my_nd = use_sc ? nb_bnd[lane] : my_pair;  // what the reference compares once a bound exists
        if (odd_out && __ballot(((newmask >> lane) & 1) && (is_odd(my_pair) || is_odd(my_nd)))) {
            *odd_out = true;
            res_len_out = 0;
            return;
        }
        VG_T(t3);
        bool has_bound = res_len >= ef;
        float bound = has_bound ? heap_get(res, 0).dist : 0.0f;
        uint64_t todo = newmask;
        uint64_t accepted = 0, pre_bound = 0;
        st.dc += __popcll(newmask);
        // the heap updates are one wave's dependent chain: let it issue ahead of the waves that are scoring (3-4 % on
        // the PQ walk at ef <= 512, nothing elsewhere)
        __builtin_amdgcn_s_setprio(3);
        while (todo) {
            if (has_bound && !strict) {
                // the bound only falls: a node above it now is above it at its turn (SquaredL2Bounded reports
                // exceeded, or `nd > bound` skips it)
                const uint64_t rej = __ballot(my_nd > bound) & todo;
                if (use_sc) st.sc += __popcll(rej);
                todo &= ~rej;
                if (!todo) break;
            }
            const int j = __builtin_ctzll(todo);
            todo &= todo - 1;
            const uint32_t id = static_cast<uint32_t>(__builtin_amdgcn_readlane(id_lane, j));
            const float nd = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(has_bound ? my_nd : my_pair), j));
            if (strict && has_bound && nd > bound) {  // at its turn, against the bound as it stands now
                if (use_sc) st.sc++;
                continue;
            }
            VG_T(tc0);
            accepted |= 1ull << j;  // its candidates-heap push happens after the loop (the two heaps are independent)
            if (!has_bound) pre_bound |= 1ull << j;  // pushed with the SquaredL2 value, not the bounded kernel's
            if (STRICT && ((deadmask >> j) & 1)) continue;  // tombstoned: the exploration queue only
            VG_T(tc1);
            if (has_bound) {  // results heap full: its top is `bound`
                res_replace_top<UK>(res, res_len, HItem{id, nd}, bound);
            } else {
                res_push_bounded<UK>(res, res_len, HItem{id, nd}, ef);
                if (res_len >= ef) {
                    bound = heap_get(res, 0).dist;
                    has_bound = true;
                }
            }
            VG_T(tc2);
            VG_TACC(st.t_cand, tc0, tc1);
            VG_TACC(st.t_res, tc1, tc2);
#ifdef VG_WALK_TIMING
            st.n_push++;
#endif
        }
        // TryPushBounded of the accepted nodes, in neighbour order (what the reference interleaves with the results
        // heap's updates; neither heap reads the other).  While the heap is below its cap they are plain pushes: one run
        // (heap_push_run_min); at the cap each one is the replace-the-closest path.
        if (accepted) {
            const float acc_d = ((pre_bound >> lane) & 1) ? my_pair : my_nd;  // the value the node was accepted with
            int fit = cap - cand_len;
            fit = fit < 0 ? 0 : fit;
            uint64_t run = accepted, rest = 0;
            if (__popcll(accepted) > fit) {
                run = 0;
                uint64_t a = accepted;
                for (int c = 0; c < fit; c++) {
                    run |= a & (~a + 1);
                    a &= a - 1;
                }
                rest = a;
            }
            heap_push_run_min(cand, cand_len, run, id_lane, acc_d);
            while (rest) {
                const int j = __builtin_ctzll(rest);
                rest &= rest - 1;
                cand_try_push_bounded<UK>(cand, cand_len,
                                          HItem{static_cast<uint32_t>(__builtin_amdgcn_readlane(id_lane, j)),
                                                __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc_d), j))}, cap);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        Scorer::sync();
        VG_T(t4);
        VG_TACC(st.t_pop, t0, t1);
        VG_TACC(st.t_adj, t1, t2);
        VG_TACC(st.t_score, t2, t3);
        VG_TACC(st.t_push, t3, t4);
    }
    res_len_out = res_len;
}

// The results heap's items in ascending order without popping them: a sort of a copy (`keys`: the finished candidates
// heap's LDS, 2 * ef items >= the next power of two of res_len; an item's 8 bytes are its key, distance bits above the
// node id — distances >= +0, UK).  Popping the heap gives the same order exactly when no two of the `check` closest
// distances are equal (ties pop in the order the heap's layout dictates) and none is a NaN: returns false then, and
// the caller pops.
__device__ __forceinline__ bool results_sorted_lds(const HItem *res, int res_len, uint64_t *keys, int check, int lane)
{
    const uint64_t *items = reinterpret_cast<const uint64_t *>(res);
    int n2 = 1;
    while (n2 < res_len) n2 <<= 1;
    bool bad = false;
    __syncthreads();
    for (int i = lane; i < n2; i += 64) {
        uint64_t key = ~0ull;
        if (i < res_len) {
            key = items[i];
            bad |= static_cast<uint32_t>(key >> 32) > 0x7F800000u;
        }
        keys[i] = key;
    }
    __syncthreads();
    bitonic_sort_lds(keys, n2, lane, 64);
    const int have = res_len < check ? res_len : check;
    for (int i = lane; i + 1 < have; i += 64) bad |= (keys[i] >> 32) == (keys[i + 1] >> 32);
    return __ballot(bad) == 0;
}

}  // namespace vg
