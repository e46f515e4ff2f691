// k_graph.hip — graph traversal with the reference's exact heap semantics:
//   hnsw.KNNSearch: greedySearch (hnsw.go:1897-1934) + searchLayerUnfiltered (hnsw.go:1220-1396)
//   diskann.Segment.searchInternal (diskann/segment.go:503-706)
//
// One wavefront per query.  The traversal itself is sequentially dependent (each pop depends on
// the previous one), so throughput comes from many queries in flight while the per-query result
// is the sequential reference's: the 4-ary PriorityQueue (internal/searcher/queue.go) is
// restated operation by operation (same sift loops, same strict comparisons) on a per-wave
// array, executed uniformly by the wave; the <=64 neighbours of a popped node are visited-tested,
// gathered and scored in parallel (16 lanes per fp32 row, 4 rows at a time, reference summation
// order), then fed to the heaps in the node's stored neighbour order.
#include <algorithm>
#include <type_traits>

#include "vg_device.hpp"
#include "vg_exact.hpp"
#include "vg_heap.hpp"
#include "vg_hnsw_layer.hpp"
#include "vg_internal.hpp"
#include "vg_cand_replay.hpp"

namespace vg {

int32_t launch_pq_build_table(const vg_pq *pq, const float *d_queries, int64_t nq, float *d_tables,
                              bool scan_layout, hipStream_t st);


// ---- HNSW ----------------------------------------------------------------------------------------
// SPLIT = false: both heaps of the query live in LDS (3 * ef items, ef <= kHnswLdsEf).  SPLIT = true (larger ef:
// 3 * ef * 8 bytes per query would leave one or two waves per CU): the first kHnswLdsEf items of the results heap
// and 2 * kHnswLdsEf of the candidates heap — the top levels, where the sifts spend their steps — stay in LDS,
// the rest of each heap is HBM scratch.
// PQM = 0: nodes scored from their fp32 rows (hnsw.KNNSearch).  PQM != 0: from their PQ codes, with the
// query's distance table `luts` (ComputeAsymmetricDistance order), the candidate stage of the
// graph -> PQ -> exact-rerank pipeline.
constexpr int kHnswLdsEf = 512;  // LDS items of the results heap (twice that for the candidates): 12 KiB per query
// Split heaps on fp32 rows (ef > 512): neither the vector ALU (41 % busy) nor HBM (4.6 TB/s) is saturated, the waves
// wait on their own chains.  A fourth wave per SIMD (128 registers instead of 161) bought 1.26x (ef 1024: 58.8 -> 46.5
// ms per 8192 queries, ef 2048: 120 -> 95 on one box); with the query in LDS (F32ScorerT<true>: its pieces are read
// where they are used instead of being held beside the row's for a whole chunk: 20 registers spilled instead of 29, no
// query traffic through L1) and 128 heap items in LDS (7 KiB per query in all, so that the 16 waves fit) another 9 %
// (48.5 / 95.7 / 208.8 -> 44.2 / 88.8 / 191.7 ms at ef 1024 / 2048 / 4096), and with 6 row blocks in flight instead of 12
// (91 registers, nothing spilled) 42.7 / 84.6 / 182.4 = 5.4 / 5.2 / 4.9 TB/s of gathered rows.  A fifth wave then fits
// without spills and changes nothing (43.3 / 83.6 / 183.4): the walk is no longer short of waves.  The other
// instantiations keep the compiler's choice.
#ifndef VG_SPLIT_F32_WAVES
#define VG_SPLIT_F32_WAVES 4
#endif
#ifndef VG_SPLIT_PQ_WAVES
#define VG_SPLIT_PQ_WAVES 4
#endif
// PQ walk with its heaps in LDS (direct form: 109 registers, 4 waves per SIMD): a fifth and a sixth wave (74 registers
// with one group of centroid loads in flight, nothing spilled) change nothing — 2.34 / 2.37 / 2.40 ms per 8192 queries at
// ef 128, 8.55 / 8.12 / 8.21 per 32768 — the waves' issue slots are what is used up (each wave issues 39 % of its
// cycles, profiles/r03_pmc_walk_pq_128_issue.csv), so the compiler's choice stays
// r04: the terms of four sub-quantizers are computed as two interleaved packed chains (pq_term8_quad: the s_nop after
// every link of a lone packed chain is gone, 2049 -> 136 in this kernel); left alone the compiler then takes 183
// registers (2 waves per SIMD: 3.06 ms per 8192 queries at ef 128); held to 4 waves 2.28-2.34 ms (2.36 before), 9.68-9.70
// (9.83) at ef 512, Vamana-PQ 1.58-1.60 (1.64): 1-3 %, i.e. the padding was not what bounds the walk - its instruction
// count is (SALU pads co-issue with other waves)
#ifndef VG_LDS_PQ_WAVES
#define VG_LDS_PQ_WAVES 4
#endif
#define VG_HNSW_ATTR __attribute__((amdgpu_waves_per_eu(SPLIT ? (PQM ? VG_SPLIT_PQ_WAVES : VG_SPLIT_F32_WAVES) : (PQM == 2 ? VG_LDS_PQ_WAVES : 1), SPLIT ? (PQM ? VG_SPLIT_PQ_WAVES : VG_SPLIT_F32_WAVES) : 8)))
// UK: the metric is not Dot, so every distance is >= +0 and the heaps compare bit patterns (heap_sift_down_uk).
// PQM: 0 = fp32 rows, 1 = PQ codes scored from the query's table, 2 = PQ codes scored from the codebook (direct form)
// STRICT (UK = false): the second pass over the queries whose first pass met a NaN distance (search_layer)
template <int PQM, bool SPLIT, bool UK, bool STRICT = false>
__global__ __launch_bounds__(64) VG_HNSW_ATTR void hnsw_search_kernel(
    const float *__restrict__ base, int64_t n, int dim, int metric, const uint32_t *__restrict__ l0,
    int m0, int max_level, int m, const uint32_t *__restrict__ slots, const uint32_t *__restrict__ adj,
    const int64_t *__restrict__ level_off, uint32_t entry, const float *__restrict__ queries,
    const uint8_t *__restrict__ pq_rows, int pq_m, const float *__restrict__ luts,
    const int8_t *__restrict__ pq_cb, const float *__restrict__ pq_scales, const float *__restrict__ pq_offsets, int k, int ef,
    int lds_cand, int lds_res /* SPLIT: items of the candidates / results heap kept in LDS */,
    uint32_t *__restrict__ visited_ws, int64_t vis_words, HItem *__restrict__ heap_ws,
    uint32_t *__restrict__ ids, float *__restrict__ scores, vg_search_stats *__restrict__ stats,
    uint8_t *__restrict__ redo /* first pass: redo[q] = 1 when the walk met a NaN distance and stopped; STRICT: only the
                                  queries so marked, every comparison as the reference writes it */,
    const uint8_t *__restrict__ mask, int64_t mask_stride, int ef_keep /* searchLayerWithPostFilter (hnsw.go:1159-1218):
                                  `ef` is the EXPANDED ef the walk runs with; afterwards every result is popped, the rows
                                  whose mask bit is set are kept in that order and pushed back capped at ef_keep */,
    const uint8_t *__restrict__ dead /* g.tombstones (vg_index_set_hnsw_tombstones) or nullptr; read by the STRICT instances,
                                        which answer every query of a tombstoned graph */)
{
    extern __shared__ __attribute__((aligned(8))) unsigned char smem[];
    constexpr bool PQ = PQM != 0;
    const int64_t q = blockIdx.x;
    const int lane = threadIdx.x;
    static_assert(!(STRICT && UK), "the second pass compares floats");
    if constexpr (STRICT) {
        if (!redo[q]) return;
        // the first pass left its marks in this query's visited bitmap
        for (int64_t w = lane; w < vis_words; w += 64) visited_ws[q * vis_words + w] = 0;
        __threadfence();
        __syncthreads();
    }
    float *nb_pair = reinterpret_cast<float *>(smem);
    float *nb_bnd = nb_pair + 64;
    float *qprep = nb_bnd + 64;  // PQ direct form: (pq_m / 2) * 20 floats (a multiple of 8 bytes)
    // fp32 rows with split heaps: the query itself (dim floats, padded to 16 bytes) — F32ScorerT<true>
#ifndef VG_SPLIT_F32_QLDS
#define VG_SPLIT_F32_QLDS 1
#endif
    constexpr bool kQLds = !PQ && SPLIT && VG_SPLIT_F32_QLDS;
    const int qwords = PQ ? (PQM == 2 ? (pq_m >> 1) * kPqPairFloats : 0) : (kQLds ? ((dim + 3) & ~3) : 0);
    HItem *heaps = reinterpret_cast<HItem *>(qprep + qwords);
    typename std::conditional<SPLIT, SplitHeap, HItem *>::type cand, res;
    if constexpr (SPLIT) {
        HItem *lo = heaps;
        cand = SplitHeap{lo, heap_ws + q * 3 * ef, lds_cand};
        res = SplitHeap{lo + lds_cand, heap_ws + q * 3 * ef + 2 * ef, lds_res};
    } else {
        cand = heaps;
        res = cand + 2 * ef;
    }
    uint32_t *vis = visited_ws + q * vis_words;
    typename std::conditional<PQ, PqScorerT<PQM == 2>, F32ScorerT<kQLds>>::type sc;
    if constexpr (PQ) {
        sc.rows = pq_rows;
        sc.lut = luts ? luts + q * static_cast<int64_t>(pq_m) * 256 : nullptr;
        sc.cb = pq_cb;  // direct form (sub-dimension 8): no table
        sc.scales = pq_scales;
        sc.offsets = pq_offsets;
        sc.qv = queries + q * dim;
        sc.qprep = qprep;
        sc.m = pq_m;
        if constexpr (PQM == 2) {
            pq_direct_prepare(qprep, sc.qv, pq_scales, pq_offsets, pq_m, lane);
            __syncthreads();
        }
    } else {
        sc.base = base;
        sc.qv = queries + q * dim;
        if constexpr (kQLds) {
            for (int i = lane; i < dim; i += 64) qprep[i] = sc.qv[i];
            sc.qv = qprep;
            __syncthreads();
        }
        sc.dim = dim;
        sc.metric = metric;
        sc.sub = Sub16::make(lane);
    }

    if constexpr (PQ && !STRICT) {
        // a PQ distance is a NaN only if the query holds a non-finite value (the codes and the codebook cannot): one look
        // at the query instead of a watch on every neighbour list
        bool bad = false;
        for (int i = lane; i < dim; i += 64) bad |= !(fabsf(queries[q * dim + i]) <= 3.40282346638528859811704183484516925440e+38f);
        if (__ballot(bad)) {
            if (lane == 0) redo[q] = 1;
            return;
        }
    }

    // ---- greedySearch through the upper layers --------------------------------------------------
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

    // ---- searchLayerUnfiltered on layer 0 ---------------------------------------------------------
    int res_len = 0;
    LayerStats st;
    auto row0 = [&](uint32_t node) -> const uint32_t * { return l0 + static_cast<int64_t>(node) * m0; };
    bool odd = false;
    search_layer<UK, STRICT>(sc, metric == kMetricL2, lane, row0, m0, cur, cur_d, ef, cand, res, nb_pair, nb_bnd, vis, res_len,
#ifdef VG_NO_ODD_WATCH  // stage probe (tools/build_variant.sh): the first pass without its NaN watch
                             st, nullptr);
#else
                             // (PQ codes hold no NaN: the query was checked once, above — the per-list watch cost the
                             // split-heap PQ walk 5 %: 9.71 -> 10.24 ms per 8192 queries at ef 512)
                             st, STRICT || PQ ? nullptr : &odd, STRICT ? dead : nullptr);
#endif
    if (odd) {  // a NaN distance: this query is answered by the second pass (search_hnsw_impl)
        if (lane == 0) redo[q] = 1;
        return;
    }

    if (mask) {  // hnsw.go:1187-1217
        const uint8_t *mq = mask + q * mask_stride;
        // (the finished candidates heap's storage holds the survivors: 2 * ef items >= res_len)
        int keep = 0;
        while (res_len > 0) {
            const HItem it = heap_pop<true, UK>(res, res_len);
            if (((mq[it.node >> 3] >> (it.node & 7)) & 1) && !(dead && mask_bit(dead, it.node))) heap_put(cand, keep++, it);  // :1198
        }
        for (int i = 0; i < keep; i++) {
            const HItem it = heap_get(cand, i);
            if (res_len < ef_keep)
                heap_push<true>(res, res_len, it);
            else
                res_push_bounded<UK>(res, res_len, it, ef_keep);
        }
    }

    // knnSearchInternal extraction (hnsw.go:1732-1751): drop the worst until k remain, then pop — the k closest
    // in ascending order, equal distances in whatever order the heap's layout pops them.  Ties among the results
    // that are dropped only change the order they are dropped in: when the k + 1 closest distances are distinct the
    // output is THE ascending order of the k closest, and selecting / sorting the items (8 bytes = the key: distance
    // bits above the node id) writes it without the ef dependent sift-downs — ~5 % of a PQ walk at ef 128, and with
    // split heaps every one of them a chain of HBM round trips (fp32 walk, ef 2048: ~15 %).  A tie among those
    // k + 1, or a NaN, takes the pops.
    bool sorted = false;
    if constexpr (UK) {
        if (k < 64) {  // selection: lane i of a register list ends up with the i-th closest (WaveTopK)
            WaveTopK tk;
            tk.init(k + 1);
            bool bad = false;
            for (int i0 = 0; i0 < res_len; i0 += 64) {
                uint64_t key = kKeyMax;
                if (i0 + lane < res_len) {
                    key = heap_load_u64(res, i0 + lane);
                    bad |= static_cast<uint32_t>(key >> 32) > 0x7F800000u;
                }
                tk.offer(key, lane);
            }
            const int have = res_len < k + 1 ? res_len : k + 1;
            const uint32_t next_d = __shfl_down(static_cast<uint32_t>(tk.list >> 32), 1);
            bad |= lane + 1 < have && static_cast<uint32_t>(tk.list >> 32) == next_d;
            if (!__ballot(bad)) {
                sorted = true;
                res_len = res_len < k ? res_len : k;
                if (lane < res_len) {
                    ids[q * k + lane] = static_cast<uint32_t>(tk.list);
                    scores[q * k + lane] = __uint_as_float(static_cast<uint32_t>(tk.list >> 32));
                }
            }
        } else if constexpr (!SPLIT) {  // a sort of a copy in the finished candidates heap's LDS (2 * ef items)
            uint64_t *keys = reinterpret_cast<uint64_t *>(cand);
            if (results_sorted_lds(res, res_len, keys, k + 1, lane)) {
                sorted = true;
                res_len = res_len < k ? res_len : k;
                for (int i = lane; i < res_len; i += 64) {
                    ids[q * k + i] = static_cast<uint32_t>(keys[i]);
                    scores[q * k + i] = __uint_as_float(static_cast<uint32_t>(keys[i] >> 32));
                }
            }
        }
    }
    if (!sorted) {
        while (res_len > k) (void)heap_pop<true, UK>(res, res_len);
    }
    const int nres = res_len;
    for (int i = nres - 1; i >= 0 && !sorted; i--) {
        const HItem it = heap_pop<true, UK>(res, res_len);
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
        stats[q].nodes_visited = st.visited;
        stats[q].distance_computations = st.dc;
        stats[q].distance_short_circuits = st.sc;
        stats[q].pops = st.pops;
        stats[q].descent_distance_computations = st_descent;
#ifdef VG_WALK_TIMING
        stats[q].nodes_visited = st.t_pop + (st.t_cand << 32);     // two 32-bit fields per column (probe only)
        stats[q].distance_computations = st.t_adj + (st.t_res << 32);
        stats[q].distance_short_circuits = st.t_score + (st.n_push << 32);
        stats[q].descent_distance_computations = st.t_push;
#endif
    }
}

// ---- Vamana --------------------------------------------------------------------------------------
enum { kVamanaF32 = 0, kVamanaPQ = 1, kVamanaRaBitQ = 2, kVamanaInt4 = 3, kVamanaPQDirect = 4, kVamanaInt4Direct = 5 /* 4, 5: kernel instances only */ };

__device__ inline float rq_formula_g(float qn, float yn, float dimf, float hamming)
{
    const float t1 = qn - yn;
    const float t1sq = t1 * t1;
    float t2 = 4.0f * qn;
    t2 = t2 * yn;
    t2 = t2 / dimf;
    t2 = t2 * hamming;
    return t1sq + t2;
}

constexpr int kHnswMaxEf = 1 << 20;  // ... in HBM scratch beyond
constexpr int kVamanaMaxK = 512;  // results per query: one per lane up to 64, a sorted LDS list beyond
#ifndef VG_VAMANA_LDS_CAND
#define VG_VAMANA_LDS_CAND 512
#endif
constexpr int kVamanaLdsCand = VG_VAMANA_LDS_CAND;  // items of the exploration heap kept in LDS (4 KiB: 16 waves per CU next to the PQ constants)

template <bool big>
__device__ __forceinline__ uint64_t *vamana_result_list()
{
    if constexpr (big) {
        __shared__ uint64_t list[kVamanaMaxK];
        return list;
    } else {
        return nullptr;
    }
}

// one instance per node scorer and per result-set form: the fp32 scorer's row blocks in flight do not set the
// register budget of the code scorers, and the k <= 64 search carries no LDS result list
// fp32 rows: the gather runs best with 3 waves per SIMD (131 registers: all 12 row blocks of a pass in flight); squeezed
// to a fourth wave it is 6 % slower (3.85 vs 3.63 ms per 8192 queries on one box)
#ifndef VG_VAMANA_F32_MAX_WAVES
#define VG_VAMANA_F32_MAX_WAVES 3
#endif
#ifndef VG_VAMANA_PQ_WAVES
#define VG_VAMANA_PQ_WAVES 4
#endif
// MASKED: vg_search_vamana_filtered (pushToHeap's filter, segment.go:616-627) — a template flag: the unfiltered instances keep
// their register budgets (a runtime mask cost 13 registers: the PQ-direct scorer spilled, the others lost a wave per SIMD)
// STRICT (with big and MASKED): the second pass over the queries whose distances may hold a NaN (a non-finite query value or index
// datum; dot products that can overflow both ways; RaBitQ: 4 |q| |y| overflowing next to a Hamming distance of 0).  The result
// set above is kept by 64-bit keys — a total order — while for the reference's CandidateHeap a NaN is neither better nor worse than
// anything (candidate_queue.go:12-38): which rows it holds, the root the pruning test reads and the order they leave in are then
// decided by the heap's layout.  This pass runs the reference's TryPushBounded / Pop on an LDS array with float comparisons
// (vg_cand_replay.hpp), neighbour by neighbour in list order; a query without risk returns at once.
template <int kind, bool big, bool MASKED = false, bool STRICT = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(kind == kVamanaPQDirect ? VG_VAMANA_PQ_WAVES : 1, kind == kVamanaF32 ? VG_VAMANA_F32_MAX_WAVES : 8))) void vamana_search_kernel(
    int metric, int64_t n, int dim, const uint32_t *__restrict__ graph, int r, uint32_t entry,
    const float *__restrict__ base, const uint8_t *__restrict__ pq_rows, int pq_m,
    const float *__restrict__ luts /* nq * m * 256, or nullptr: terms from the codebook */, const int8_t *__restrict__ pq_cb,
    const float *__restrict__ pq_scales, const float *__restrict__ pq_offsets, const uint8_t *__restrict__ rq_rows,
    const uint8_t *__restrict__ qcodes /* nq * (nb+4) */, int rq_nb, const uint8_t *__restrict__ int4_rows,
    const float *__restrict__ int4_table, const float *__restrict__ int4_min, const float *__restrict__ int4_diff,
    const float *__restrict__ queries, int k, HItem *__restrict__ cand_ws, int64_t cand_cap, uint32_t *__restrict__ visited_ws,
    int64_t vis_words, uint32_t *__restrict__ ids, float *__restrict__ scores,
    vg_search_stats *__restrict__ stats, const uint8_t *__restrict__ mask, int64_t mask_stride,
    const float *__restrict__ risk_absmax /* STRICT: max |x| of the rows (fp32) / max |norm| (RaBitQ), +Inf when one is not finite */)
{
    __shared__ float nb_d[64];
    __shared__ uint64_t res[kVamanaMaxK];  // the result set when k > 64 (STRICT: the CandidateHeap's array)
    // the exploration heap is unbounded in the reference (up to cand_cap items of HBM scratch here); its first
    // kVamanaLdsCand items — the levels every pop and push touches; a k = 10 search scores ~1000 nodes and its
    // heap peaks a little above that — live in LDS: a
    // popped node pushes up to R = 64 neighbours one after the other, each a sift of dependent accesses
    __shared__ HItem cand_lo[kVamanaLdsCand];
    extern __shared__ __attribute__((aligned(16))) float vamana_qprep[];  // PQ direct form: pq_direct_prepare's image
    const int64_t q = blockIdx.x;
    const int lane = threadIdx.x;
    const Sub16 sub = Sub16::make(lane);
    const float *qv = queries + q * dim;
    uint32_t *vis = visited_ws + q * vis_words;
    // (the flat address of cand_lo passes through a register: as a constant expression inside the heap's
    // LDS-or-HBM selects this compiler emits v_cmp with src_shared_base as a VOP2 operand and rejects it)
    HItem *cand_lo_flat = cand_lo;
    asm volatile("" : "+s"(cand_lo_flat));
    const SplitHeap cand{cand_lo_flat, cand_ws + q * cand_cap, kVamanaLdsCand};
    const bool desc = metric != kMetricL2;  // sc.Heap.Reset(s.Metric() != MetricL2), segment.go:597
    const float *lut = luts ? luts + q * static_cast<int64_t>(pq_m) * 256 : nullptr;
    const uint8_t *qc = qcodes ? qcodes + q * static_cast<int64_t>(rq_nb + 4) : nullptr;
    float qn = 0.0f;
    if (kind == kVamanaRaBitQ) {
        const uint32_t b = qc[rq_nb] | (qc[rq_nb + 1] << 8) | (qc[rq_nb + 2] << 16) |
                           (static_cast<uint32_t>(qc[rq_nb + 3]) << 24);
        qn = __uint_as_float(b);
        // the query's sign words in LDS (rq_nb bytes of the dynamic allocation; qcodes rows are 4-byte aligned)
        for (int i = lane; i < (rq_nb >> 2); i += 64)
            reinterpret_cast<uint32_t *>(vamana_qprep)[i] = reinterpret_cast<const uint32_t *>(qc)[i];
        __syncthreads();
    }
    if constexpr (STRICT) {
        bool bad = false;
        for (int j = lane; j < dim; j += 64) bad = bad || !is_finite_f32(qv[j]);
        // (the tests of vg_cand_replay.hpp's scorers: non-finite inputs, or magnitudes whose partial sums could overflow)
        float vm = 0.0f;  // the largest |value| a row can hold
        if (kind == kVamanaF32) {
            vm = risk_absmax[0];
        } else if (kind == kVamanaRaBitQ) {
            const float ma = risk_absmax[0];
            bad = bad || !is_finite_f32(qn) || !is_finite_f32(ma) || !(4.0f * fabsf(qn) * ma < 1e38f) || !(score_bound(fabsf(qn), ma, false) < 1e38f);
        } else if (kind == kVamanaPQ || kind == kVamanaPQDirect) {
            for (int j = lane; j < pq_m; j += 64) {
                bad = bad || !is_finite_f32(pq_scales[j]) || !is_finite_f32(pq_offsets[j]);
                vm = fmaxf(vm, 128.0f * fabsf(pq_scales[j]) + fabsf(pq_offsets[j]));
            }
        } else if (int4_min && int4_diff) {
            for (int j = lane; j < dim; j += 64) {
                bad = bad || !is_finite_f32(int4_min[j]) || !is_finite_f32(int4_diff[j]);
                vm = fmaxf(vm, fabsf(int4_min[j]) + fabsf(int4_diff[j]));
            }
        } else {
            for (int j = lane; j < dim * 16; j += 64) {  // (the lookup table holds the decoded values)
                bad = bad || !is_finite_f32(int4_table[j]);
                vm = fmaxf(vm, fabsf(int4_table[j]));
            }
        }
        if (kind != kVamanaRaBitQ) {
            for (int off = 32; off > 0; off >>= 1) vm = fmaxf(vm, __shfl_xor(vm, off));
            bad = bad || !is_finite_f32(vm);
            const bool as_dot = kind == kVamanaF32 && desc;
            for (int j = lane; j < dim; j += 64) bad = bad || !(score_bound(fabsf(qv[j]), vm, as_dot) * static_cast<float>(dim) < 1e38f);
        }
        if (!__any(bad)) return;
        for (int64_t w = lane; w < vis_words; w += 64) vis[w] = 0u;  // the first pass's marks
        __syncthreads();
    }
    int64_t st_visited = 0, st_dc = 0, st_pops = 0, st_dropped = 0;
    __shared__ vg_f2v int4_pairs[256];  // INT4 direct form: code byte -> (hi / 15, lo / 15)
    if (kind == kVamanaInt4Direct) {
        int4_fill_pairs(int4_pairs, lane, 64);
        __syncthreads();
    }
    if (kind == kVamanaPQDirect) {
        pq_direct_prepare(vamana_qprep, qv, pq_scales, pq_offsets, pq_m, lane);
        __syncthreads();
    }

    // score the nodes held by the lanes in `mask` (one id per lane) into nb_d[lane]
    auto score_mask = [&](uint64_t mask, uint32_t id_lane) {
        if (kind == kVamanaF32) {
            while (mask) {
                const int mine = take4(mask, lane);
                const uint32_t id = __shfl(id_lane, mine < 0 ? 0 : mine);
                if (mine >= 0) {
                    const float *row = base + static_cast<int64_t>(id) * dim;
                    // distFunc(query, vec) = distance.Provider(metric): SquaredL2 or Dot
                    const float d = desc ? exact_pair16<true, kPair>(row, qv, dim, sub)
                                         : exact_pair16<false, kPair>(row, qv, dim, sub);
                    if ((lane & 15) == 0) nb_d[mine] = d;
                }
            }
        } else if ((mask >> lane) & 1) {
            if (kind == kVamanaPQ || kind == kVamanaPQDirect) {
                // ComputeAsymmetricDistance (pq.go:234-260): term(m) = BuildDistanceTable entry, summed
                // sequentially over the sub-quantizers (vg_hnsw_layer.hpp: loads batched, sum order kept)
                const uint8_t *code = pq_rows + static_cast<int64_t>(id_lane) * pq_m;
                nb_d[lane] = kind == kVamanaPQ ? pq_asym_distance(code, lut, pq_m)
                                               : pq_direct_distance(code, pq_cb, pq_scales, pq_offsets, qv, vamana_qprep, pq_m);
            } else if (kind == kVamanaInt4Direct) {
                // the same terms evaluated in place of the table read (vg_device.hpp)
                nb_d[lane] = int4_l2_direct(qv, int4_rows + static_cast<int64_t>(id_lane) * (dim >> 1), dim, int4_min,
                                            int4_diff, int4_pairs);
            } else if (kind == kVamanaInt4) {
                // iq.L2Distance (diskann/segment.go:558-565) = int4L2DistancePrecomputedAvx512 order
                nb_d[lane] = int4_l2_precomputed(qv, int4_rows + static_cast<int64_t>(id_lane) * ((dim + 1) / 2), dim,
                                                 int4_table);
            } else {
                // rows are rq_nb + 4 bytes with rq_nb a multiple of 8: every row is 4-byte aligned, so the
                // bits and the norm are read as dwords, 8 loads in flight at a time; the query's words come from
                // LDS (broadcast reads) — as scalar loads each one was a round trip the pass waited out
                const uint32_t *cw = reinterpret_cast<const uint32_t *>(rq_rows + static_cast<int64_t>(id_lane) * (rq_nb + 4));
                const uint32_t *qw = reinterpret_cast<const uint32_t *>(vamana_qprep);
                const int nw = rq_nb >> 2;
                int h = 0;
                int b0 = 0;
                for (; b0 + 8 <= nw; b0 += 8) {
                    uint32_t x[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) x[u] = cw[b0 + u];
#pragma unroll
                    for (int u = 0; u < 8; u++) h += __popc(x[u] ^ qw[b0 + u]);
                }
                for (; b0 < nw; b0++) h += __popc(cw[b0] ^ qw[b0]);
                const uint32_t yb = cw[nw];
                nb_d[lane] = rq_formula_g(qn, __uint_as_float(yb), static_cast<float>(dim), static_cast<float>(h));
            }
        }
        __syncthreads();
    };

    WaveTopK tk;  // sc.Heap: top-k by (Score, RowID) — candidate_queue.go:12-23
    tk.init(k < 64 ? k : 64);
    // k > 64: the k best keys as a sorted list in LDS instead of one key per lane (wave-uniform bookkeeping)
    int res_n = 0;
    uint64_t res_tau = kKeyMax;
    auto res_insert = [&](uint64_t c) {  // c < res_tau, every lane calls it with the same key
        int lo = 0, hi = res_n;          // first position whose key is >= c (keys are distinct)
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (res[mid] < c)
                lo = mid + 1;
            else
                hi = mid;
        }
        const int newn = res_n < k ? res_n + 1 : k;
        if constexpr (!big) return;
        uint64_t moved[kVamanaMaxK / 64];
#pragma unroll
        for (int u = 0; u < kVamanaMaxK / 64; u++) {
            const int i = lo + lane + u * 64;
            moved[u] = i < newn - 1 ? res[i] : kKeyMax;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < kVamanaMaxK / 64; u++) {
            const int i = lo + lane + u * 64;
            if (i < newn - 1) res[i + 1] = moved[u];
        }
        if (lane == 0) res[lo] = c;
        __syncthreads();
        res_n = newn;
        res_tau = res_n >= k ? res[k - 1] : kKeyMax;
    };
    auto offer = [&](uint64_t key) {
        if (!big) {
            tk.offer(key, lane);
            return;
        }
        uint64_t mask = __ballot(key < res_tau);
        while (mask) {
            const int j = __builtin_ctzll(mask);
            mask &= mask - 1;
            const uint64_t c = readlane_u64(key, j);
            if (c < res_tau) res_insert(c);  // the bound may have dropped since the ballot
        }
    };
    int heap_count = 0;  // min(k, candidates offered): sc.Heap.Len()
    int cand_len = 0;
    CItem *rheap = reinterpret_cast<CItem *>(res);  // STRICT: sc.Heap itself
    auto strict_offer = [&](uint32_t id, float d) {   // TryPushBounded(k), candidate_queue.go:120-132 (uniform over the wave)
        const CItem x{d, id};
        if (heap_count < k) {
            cand_up(rheap, heap_count, x, desc);
            heap_count++;
        } else if (cand_better(x, cand_load(rheap, 0), desc)) {
            cand_down(rheap, 0, heap_count, x, desc);
        }
    };
    (void)strict_offer;

    // start node (segment.go:603-636)
    if (lane == 0) atomicOr(&vis[entry >> 5], 1u << (entry & 31));
    score_mask(1ull, entry);
    const float sd = nb_d[0];
    st_dc++;
    heap_push<false>(cand, cand_len, HItem{entry, sd});
    // pushToHeap (segment.go:616-627): a row whose filter.Matches is false goes to the traversal queue only — sc.Heap, and with it
    // the pruning test below, holds matching rows
    const uint8_t *mq = MASKED && mask ? mask + blockIdx.x * mask_stride : nullptr;
    const bool entry_ok = !MASKED || mask_bit(mq, entry);
    if constexpr (STRICT) {
        if (entry_ok) strict_offer(entry, sd);
    } else {
        offer(lane == 0 && entry_ok ? make_key(sd, entry, desc) : kKeyMax);
        heap_count = entry_ok && 1 < k ? 1 : (entry_ok ? k : 0);
    }
    __syncthreads();

    while (cand_len > 0) {
        const HItem c = heap_pop<false>(cand, cand_len);
        st_pops++;
        if (heap_count >= k) {
            const float worst = STRICT ? cand_load(rheap, 0).score : key_score(big ? res_tau : tk.tau, desc);
            if (c.dist > worst) break;
        }
        const uint32_t id_lane = lane < r ? graph[static_cast<int64_t>(c.node) * r + lane] : VG_INVALID_ID;
        bool fresh = false;
        if (id_lane != VG_INVALID_ID && id_lane < n) {
            const uint32_t bit = 1u << (id_lane & 31);
            fresh = (atomicOr(&vis[id_lane >> 5], bit) & bit) == 0;
        }
        const uint64_t newmask = __ballot(fresh);
        if (!newmask) continue;
        const int nnew = __popcll(newmask);
        st_visited += nnew;
        st_dc += nnew;
        score_mask(newmask, id_lane);
        const float myd = nb_d[lane];
        if constexpr (STRICT) {  // neighbour by neighbour: PushItem, then pushToHeap (segment.go:690-700)
            uint64_t todo = newmask;
            while (todo) {
                const int j = __builtin_ctzll(todo);
                todo &= todo - 1;
                const uint32_t id = static_cast<uint32_t>(__builtin_amdgcn_readlane(id_lane, j));
                const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(myd), j));
                if (cand_len < cand_cap)
                    heap_push<false>(cand, cand_len, HItem{id, d});
                else
                    st_dropped++;
                if (mask_bit(mq, id)) strict_offer(id, d);
            }
            __syncthreads();
            continue;
        }
        // exploration heap: PushItem in the node's neighbour order (segment.go:695)
        if (cand_len + nnew <= cand_cap) {
            heap_push_run_min(cand, cand_len, newmask, id_lane, myd);  // the run of pushes, parents tracked in registers
        } else {
            uint64_t todo = newmask;
            while (todo) {
                const int j = __builtin_ctzll(todo);
                todo &= todo - 1;
                if (cand_len < cand_cap)
                    heap_push<false>(cand, cand_len,
                                     HItem{static_cast<uint32_t>(__builtin_amdgcn_readlane(id_lane, j)),
                                           __int_as_float(__builtin_amdgcn_readlane(__float_as_int(myd), j))});
                else
                    st_dropped++;  // the reference's heap is unbounded: reported, see vg_search_stats
            }
        }
        // result heap: TryPushBounded(k) — a set maintained by (score, id): order-free
        const bool pass = fresh && (!MASKED || mask_bit(mq, id_lane));
        const int npass = MASKED ? __popcll(__ballot(pass)) : nnew;
        offer(pass ? make_key(myd, id_lane, desc) : kKeyMax);
        heap_count = heap_count + npass < k ? heap_count + npass : k;
        __syncthreads();
    }
    if constexpr (STRICT) {  // the engine empties the heap with Pop() (engine/search.go:859-862): reported best first
        const int nres = heap_count;
        for (int i = nres - 1; i >= 0; i--) {
            const CItem it = cand_pop(rheap, heap_count, desc);
            if (lane == 0) {
                ids[q * k + i] = it.row;
                scores[q * k + i] = it.score;
            }
        }
        for (int i = nres + lane; i < k; i += 64) {
            ids[q * k + i] = VG_INVALID_ID;
            scores[q * k + i] = desc ? -INFINITY : INFINITY;
        }
    } else if (big) {
        for (int i = lane; i < k; i += 64) {
            const uint64_t e = i < res_n ? res[i] : kKeyMax;
            ids[q * k + i] = e == kKeyMax ? VG_INVALID_ID : key_row(e);
            scores[q * k + i] = e == kKeyMax ? (desc ? -INFINITY : INFINITY) : key_score(e, desc);
        }
    } else if (lane < k) {
        const uint64_t e = tk.list;
        ids[q * k + lane] = e == kKeyMax ? VG_INVALID_ID : key_row(e);
        scores[q * k + lane] = e == kKeyMax ? (desc ? -INFINITY : INFINITY) : key_score(e, desc);
    }
    if (stats && lane == 0) {
        stats[q].nodes_visited = st_visited;
        stats[q].distance_computations = st_dc;
        stats[q].distance_short_circuits = st_dropped;
        stats[q].pops = st_pops;
        stats[q].descent_distance_computations = 0;
    }
}

// sign bits + norm of each query (RaBitQ Encode, k_rabitq.hip)
int32_t launch_rabitq_encode(const float *d_vectors, int64_t n, int dim, uint8_t *d_codes, hipStream_t st);

}  // namespace vg

namespace vg {
// adjacency ids must be rows of the index or VG_INVALID_ID: the searches index the visited bitmap and the
// row arrays with them unchecked
__global__ void count_bad_ids_kernel(const uint32_t *__restrict__ ids, int64_t count, uint32_t n, unsigned int *__restrict__ bad)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint32_t id = ids[i];
    if (id != VG_INVALID_ID && id >= n) atomicAdd(bad, 1u);
}
}  // namespace vg

static int32_t check_adjacency(const uint32_t *d_ids, int64_t count, int64_t n, hipStream_t st, const char *what)
{
    if (count == 0) return VG_OK;
    vg::DevTmp<unsigned int> bad;
    VG_TRY(bad.init(1, st));
    VG_HIP(hipMemsetAsync(bad.ptr, 0, sizeof(unsigned int), st));
    VG_LAUNCH(vg::count_bad_ids_kernel, dim3(static_cast<unsigned>((count + 255) / 256)), dim3(256), 0, st, d_ids, count,
              static_cast<uint32_t>(n), bad.ptr);
    unsigned int h = 0;
    VG_HIP(hipMemcpyAsync(&h, bad.ptr, sizeof(h), hipMemcpyDeviceToHost, st));
    VG_HIP(hipStreamSynchronize(st));
    VG_CHECK(h == 0, VG_ERR_INVALID_ARG, "%s: %u neighbour ids are neither rows of the index nor VG_INVALID_ID", what, h);
    return VG_OK;
}

template <typename T>
static int32_t replace_device_array(T **slot, const T *src, size_t count, hipStream_t st)
{
    if (*slot) {
        VG_HIP(hipStreamSynchronize(st));
        VG_HIP(hipFree(*slot));
        *slot = nullptr;
    }
    if (count == 0) return VG_OK;
    VG_HIP(hipMalloc(reinterpret_cast<void **>(slot), count * sizeof(T)));
    VG_HIP(hipMemcpyAsync(*slot, src, count * sizeof(T), hipMemcpyDefault, st));
    return VG_OK;
}

VG_API int32_t vg_index_set_hnsw_graph(vg_index *idx, int32_t m0, const uint32_t *l0, int32_t max_level,
                                       int32_t m, const uint32_t *upper_slot, const uint32_t *upper_adj,
                                       const int64_t *level_rows, uint32_t entry_point, void *stream)
{
    VG_CHECK(idx, VG_ERR_INVALID_ARG, "vg_index_set_hnsw_graph: NULL index");
    VG_CHECK(m0 > 0 && m0 <= 64 && m > 0 && m <= 64, VG_ERR_UNSUPPORTED,
             "vg_index_set_hnsw_graph: degrees m0=%d m=%d must be in 1..64", m0, m);
    VG_CHECK(max_level >= 0 && max_level < 64, VG_ERR_INVALID_ARG, "vg_index_set_hnsw_graph: bad max_level");
    VG_CHECK(idx->n == 0 || l0, VG_ERR_INVALID_ARG, "vg_index_set_hnsw_graph: l0 is NULL");
    VG_CHECK(max_level == 0 || (upper_slot && upper_adj && level_rows), VG_ERR_INVALID_ARG,
             "vg_index_set_hnsw_graph: upper-layer arrays are NULL");
    VG_CHECK(idx->n == 0 || entry_point < idx->n, VG_ERR_INVALID_ARG, "vg_index_set_hnsw_graph: entry point out of range");
    VG_HIP(hipSetDevice(idx->ctx->device));
    hipStream_t st = vg::pick_stream(idx->ctx, stream);
    VG_TRY(replace_device_array(&idx->d_hnsw_l0, l0, static_cast<size_t>(idx->n) * m0, st));
    if (idx->d_hnsw_l0_dist) {  // the old graph's edge distances (vg_index_set_hnsw_edge_distances)
        VG_HIP(hipStreamSynchronize(st));
        (void)hipFree(idx->d_hnsw_l0_dist);
        idx->d_hnsw_l0_dist = nullptr;
    }
    std::vector<int64_t> off(max_level + 1, 0);
    for (int l = 0; l < max_level; l++) {
        VG_CHECK(level_rows[l] >= 0, VG_ERR_INVALID_ARG, "vg_index_set_hnsw_graph: negative level_rows");
        off[l + 1] = off[l] + level_rows[l];
    }
    VG_TRY(replace_device_array(&idx->d_hnsw_slot, upper_slot, static_cast<size_t>(max_level) * idx->n, st));
    VG_TRY(replace_device_array(&idx->d_hnsw_adj, upper_adj, static_cast<size_t>(off[max_level]) * m, st));
    VG_TRY(replace_device_array(&idx->d_hnsw_level_off, off.data(), off.size(), st));
    VG_HIP(hipStreamSynchronize(st));
    int32_t bad = check_adjacency(idx->d_hnsw_l0, idx->n * m0, idx->n, st, "vg_index_set_hnsw_graph (layer 0)");
    if (bad == VG_OK) bad = check_adjacency(idx->d_hnsw_adj, off[max_level] * m, idx->n, st, "vg_index_set_hnsw_graph (upper layers)");
    if (bad != VG_OK) {  // a graph that failed the check is not searchable
        (void)hipFree(idx->d_hnsw_l0);
        idx->d_hnsw_l0 = nullptr;
        return bad;
    }
    idx->hnsw_m0 = m0;
    idx->hnsw_m = m;
    idx->hnsw_max_level = max_level;
    idx->hnsw_entry = entry_point;
    return VG_OK;
}

VG_API int32_t vg_index_set_vamana_graph(vg_index *idx, int32_t r, const uint32_t *graph, uint32_t entry_point,
                                         void *stream)
{
    VG_CHECK(idx, VG_ERR_INVALID_ARG, "vg_index_set_vamana_graph: NULL index");
    VG_CHECK(r > 0 && r <= 64, VG_ERR_UNSUPPORTED, "vg_index_set_vamana_graph: degree %d must be in 1..64", r);
    VG_CHECK(idx->n == 0 || graph, VG_ERR_INVALID_ARG, "vg_index_set_vamana_graph: graph is NULL");
    VG_CHECK(idx->n == 0 || entry_point < idx->n, VG_ERR_INVALID_ARG, "vg_index_set_vamana_graph: entry point out of range");
    VG_HIP(hipSetDevice(idx->ctx->device));
    hipStream_t st = vg::pick_stream(idx->ctx, stream);
    VG_TRY(replace_device_array(&idx->d_vamana, graph, static_cast<size_t>(idx->n) * r, st));
    VG_HIP(hipStreamSynchronize(st));
    const int32_t bad = check_adjacency(idx->d_vamana, idx->n * r, idx->n, st, "vg_index_set_vamana_graph");
    if (bad != VG_OK) {
        (void)hipFree(idx->d_vamana);
        idx->d_vamana = nullptr;
        return bad;
    }
    idx->vamana_r = r;
    idx->vamana_entry = entry_point;
    return VG_OK;
}

// Per-query scratch of the graph searches (visited bitmap, HBM part of the heaps, exploration heap) is carved from
// the arena for as many queries as fit under this cap; the rest of the batch goes into further launches.  1 GiB (r02)
// cut 8192 Vamana queries over 1M nodes (650 KB each) into 5 launches of 1650 wavefronts — fewer than the 3072 the
// chip holds.  1/16 of the device's memory, at most 16 GiB.
static int64_t graph_scratch_cap(const vg_ctx *ctx)
{
    const int64_t gib = int64_t(1) << 30;
    return std::min<int64_t>(16 * gib, std::max<int64_t>(gib, ctx->hbm_bytes / 16));
}

static int32_t search_hnsw_impl(vg_index *idx, bool pq, const float *queries, int64_t nq, int32_t k, int32_t ef,
                                uint32_t *ids, float *scores, vg_search_stats *stats, void *stream,
                                const uint8_t *mask = nullptr, int64_t mask_stride = 0, double selectivity = 0.0)
{
    const char *fn = mask ? "vg_search_hnsw_filtered" : pq ? "vg_search_hnsw_pq" : "vg_search_hnsw";
    VG_CHECK(idx, VG_ERR_INVALID_ARG, "%s: NULL index", fn);
    VG_CHECK(nq >= 0 && k >= 0, VG_ERR_INVALID_ARG, "%s: negative nq or k", fn);
    if (nq == 0 || k == 0) return VG_OK;
    VG_CHECK(idx->d_hnsw_l0, VG_ERR_NOT_READY, "%s: index has no HNSW graph", fn);
    VG_CHECK(pq || idx->d_vectors, VG_ERR_NOT_READY, "%s: index has no fp32 vectors", fn);
    VG_CHECK(!pq || (idx->d_pq_rows && idx->pq), VG_ERR_NOT_READY, "%s: index has no PQ codes", fn);
    VG_CHECK(!pq || idx->pq->k == 256, VG_ERR_UNSUPPORTED, "%s: PQ needs numCentroids == 256", fn);
    VG_CHECK(!pq || idx->metric == VG_METRIC_L2, VG_ERR_UNSUPPORTED,
             "%s: ComputeAsymmetricDistance is an L2 distance; metric must be L2", fn);
    VG_CHECK(idx->metric != VG_METRIC_HAMMING, VG_ERR_UNSUPPORTED, "unsupported metric for float32: Hamming");
    VG_CHECK(queries && ids && scores, VG_ERR_INVALID_ARG, "%s: NULL buffer", fn);
    if (ef < k) ef = k;  // determineEF hnsw.go:1891-1894
    VG_CHECK(ef <= vg::kHnswMaxEf, VG_ERR_UNSUPPORTED, "%s: ef=%d exceeds %d", fn, ef, vg::kHnswMaxEf);
    const int ef_keep = ef;
    if (mask) {  // searchLayerWithPostFilter's expanded ef (hnsw.go:1166-1183); ef_keep caps the rebuilt results heap
        int64_t expanded = static_cast<int64_t>(static_cast<double>(ef) * (1.0 + (1.0 - selectivity) * 0.5));
        if (expanded > int64_t(ef) * 2) expanded = int64_t(ef) * 2;
        if (expanded > 500) expanded = 500;
        if (expanded < 1) expanded = 1;
        ef = static_cast<int32_t>(expanded);
    }
    VG_HIP(hipSetDevice(idx->ctx->device));
    hipStream_t st = vg::pick_stream(idx->ctx, stream);
    vg::DevIn<float> q;
    vg::DevOut<uint32_t> oid;
    vg::DevOut<float> osc;
    vg::DevOut<vg_search_stats> ost;
    VG_TRY(q.init(queries, static_cast<size_t>(nq) * idx->dim, st));
    VG_TRY(oid.init(ids, static_cast<size_t>(nq) * k, st));
    VG_TRY(osc.init(scores, static_cast<size_t>(nq) * k, st));
    VG_TRY(ost.init(stats, stats ? static_cast<size_t>(nq) : 0, st));
    vg::DevIn<uint8_t> mk;
    const int64_t mask_bytes = (idx->n + 7) / 8;
    VG_TRY(mk.init(mask, mask ? static_cast<size_t>(mask_stride ? (nq - 1) * mask_stride + mask_bytes : mask_bytes) : 0, st));
    const int64_t vis_words = (idx->n + 31) / 32;
    const int pq_m = pq ? idx->pq->m : 0;
    // heaps in LDS up to kHnswLdsEf (12 KiB per query: the waves of a CU are then bounded by registers, not
    // LDS), beyond it split between those 12 KiB (the top levels) and HBM scratch
    // split heaps: waves in flight beat heap levels in LDS.  The PQ walk (3.75 KiB of query constants in LDS): 128 items
    // = 7.4 KiB per query, 16 waves per CU: ef 1024 / 2048 40.7 / 81.9 -> 35.6 / 68.5 ms per 8192 queries against 256 items
    // (15 waves); the fp32 walk (the query itself in LDS): 128 items as well
#ifndef VG_PQ_LDS_HEAPS_MAX
#define VG_PQ_LDS_HEAPS_MAX 448  // PQ walk: all-LDS heaps at ef 512 are 16.7 KiB per query = 9 waves per CU (12.95 ms per 8192
#endif                            // queries); split with 128 items in LDS: 16 waves, 12.47 ms; at ef 384 the other way round (7.95 vs 9.19)
    // (a post-filter walk capped at 500 may keep more results than it walked with: the heaps are sized for the larger)
    const int ef_walk = ef;
    if (ef_keep > ef) ef = ef_keep;
    const bool lds_heaps = ef <= (pq ? VG_PQ_LDS_HEAPS_MAX : vg::kHnswLdsEf);
    // (swept on one box, ef 1024 / 2048: candidates : results = 768 : 384 46 / 94 ms, 128 : 768 47 / 95, 256 : 1024
    // 50 / 107, 256 : 2048 51 / 124, 1024 : 2048 58 / 152 — waves in flight beat heap levels in LDS)
#ifndef VG_SPLIT_F32_LDS_EF
#define VG_SPLIT_F32_LDS_EF 128
#endif
#ifndef VG_SPLIT_PQ_LDS_EF
#define VG_SPLIT_PQ_LDS_EF 128
#endif
    const int lds_ef = pq ? VG_SPLIT_PQ_LDS_EF : VG_SPLIT_F32_LDS_EF;
    // (after heap_push_run_min the split hardly matters: 128 : 1024, 64 : 1088, 256 : 896, 384 : 768 all within 2 % of this)
    const int lds_cand = 2 * lds_ef, lds_res = lds_ef;
    const int64_t heap_bytes = lds_heaps ? 0 : static_cast<int64_t>(3) * ef * sizeof(vg::HItem);
    // sub-dimension 8: node terms straight from the codebook (vg_hnsw_layer.hpp PqScorer), no per-query table
    const bool pq_direct = pq && idx->pq->subdim == 8 && (reinterpret_cast<uintptr_t>(idx->pq->d_codebooks) & 7) == 0;
    const int64_t lut_bytes = pq_direct ? 0 : static_cast<int64_t>(pq_m) * 256 * sizeof(float);
    int64_t chunk = std::max<int64_t>(1, graph_scratch_cap(idx->ctx) / std::max<int64_t>(vis_words * 4 + heap_bytes + lut_bytes, 1));
    chunk = std::min(chunk, nq);
    vg::ArenaCall ar(idx->ctx, st);
    const int i_vis = ar.add(sizeof(uint32_t) * static_cast<size_t>(chunk) * vis_words);
    const int i_heap = ar.add(static_cast<size_t>(chunk) * heap_bytes);
    const int i_luts = ar.add(static_cast<size_t>(chunk) * lut_bytes);
    const int i_redo = ar.add(static_cast<size_t>(chunk));
    VG_TRY(ar.commit());
    struct { uint32_t *ptr; } vis{ar.get<uint32_t>(i_vis)};
    vg::HItem *heap_ws = lds_heaps ? nullptr : ar.get<vg::HItem>(i_heap);
    float *luts = pq && !pq_direct ? ar.get<float>(i_luts) : nullptr;
    // + 4 items: heap_sift_down_uk reads slots fc .. fc+3 whatever the heap's length
    const size_t lds = static_cast<size_t>((lds_heaps ? 3 * ef : lds_cand + lds_res) + 4) * sizeof(vg::HItem) + 128 * sizeof(float) +
                       (pq_direct ? static_cast<size_t>(pq_m >> 1) * vg::kPqPairFloats * sizeof(float) : 0) +
                       (!pq && !lds_heaps && VG_SPLIT_F32_QLDS ? static_cast<size_t>((idx->dim + 3) & ~3) * sizeof(float) : 0);
    const bool uk = pq || idx->metric != VG_METRIC_DOT;
    uint8_t *redo = ar.get<uint8_t>(i_redo);
    auto kern = pq_direct ? (lds_heaps ? vg::hnsw_search_kernel<2, false, true> : vg::hnsw_search_kernel<2, true, true>)
                : pq      ? (lds_heaps ? vg::hnsw_search_kernel<1, false, true> : vg::hnsw_search_kernel<1, true, true>)
                : uk      ? (lds_heaps ? vg::hnsw_search_kernel<0, false, true> : vg::hnsw_search_kernel<0, true, true>)
                          : (lds_heaps ? vg::hnsw_search_kernel<0, false, false> : vg::hnsw_search_kernel<0, true, false>);
    // the unsigned-key kernels stop a walk at the first NaN / negative distance and mark the query; this pass — the same
    // walk with the float sift-downs, i.e. the reference's comparisons as written — answers the marked queries
    auto kern_f32 = pq_direct ? (lds_heaps ? vg::hnsw_search_kernel<2, false, false, true> : vg::hnsw_search_kernel<2, true, false, true>)
                    : pq      ? (lds_heaps ? vg::hnsw_search_kernel<1, false, false, true> : vg::hnsw_search_kernel<1, true, false, true>)
                              : (lds_heaps ? vg::hnsw_search_kernel<0, false, false, true> : vg::hnsw_search_kernel<0, true, false, true>);
    VG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                               static_cast<int>(lds)));
    VG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern_f32), hipFuncAttributeMaxDynamicSharedMemorySize,
                               static_cast<int>(lds)));
    for (int64_t q0 = 0; q0 < nq; q0 += chunk) {
        const int64_t cnt = std::min(chunk, nq - q0);
        VG_HIP(hipMemsetAsync(vis.ptr, 0, static_cast<size_t>(cnt) * vis_words * 4, st));
        if (luts) VG_TRY(vg::launch_pq_build_table(idx->pq, q.ptr + q0 * idx->dim, cnt, luts, false, st));
        // a graph with tombstones: every query takes the pass that walks as the reference writes it (float comparisons, each lane
        // at its turn) and reads the bitmap — the fast pass is tuned to its register budget without it
        VG_HIP(hipMemsetAsync(redo, idx->d_hnsw_tomb ? 1 : 0, static_cast<size_t>(cnt), st));
        if (!idx->d_hnsw_tomb) {
            vg::ProfScope prof(idx->ctx, pq ? "hnsw_search_pq" : "hnsw_search", st);
            VG_LAUNCH(kern, dim3(static_cast<unsigned>(cnt)), dim3(64), lds, st, idx->d_vectors, idx->n, idx->dim,
                      idx->metric, idx->d_hnsw_l0, idx->hnsw_m0, idx->hnsw_max_level, idx->hnsw_m, idx->d_hnsw_slot,
                      idx->d_hnsw_adj, idx->d_hnsw_level_off, idx->hnsw_entry, q.ptr + q0 * idx->dim,
                      pq ? idx->d_pq_rows : nullptr, pq_m, luts, pq_direct ? idx->pq->d_codebooks : nullptr,
                      pq ? idx->pq->d_scales : nullptr, pq ? idx->pq->d_offsets : nullptr, k, ef_walk, lds_cand, lds_res, vis.ptr,
                      vis_words, heap_ws, oid.ptr + q0 * k, osc.ptr + q0 * k, ost.ptr ? ost.ptr + q0 : nullptr, redo,
                      mk.ptr ? mk.ptr + q0 * mask_stride : nullptr, mask_stride, ef_keep, idx->d_hnsw_tomb);
        }
        // (every workgroup of an ordinary batch leaves at its first instruction)
        vg::ProfScope prof_strict(idx->ctx, idx->d_hnsw_tomb ? (pq ? "hnsw_search_pq" : "hnsw_search") : "hnsw_search_redo", st);
        VG_LAUNCH(kern_f32, dim3(static_cast<unsigned>(cnt)), dim3(64), lds, st, idx->d_vectors, idx->n, idx->dim,
                      idx->metric, idx->d_hnsw_l0, idx->hnsw_m0, idx->hnsw_max_level, idx->hnsw_m, idx->d_hnsw_slot,
                      idx->d_hnsw_adj, idx->d_hnsw_level_off, idx->hnsw_entry, q.ptr + q0 * idx->dim,
                      pq ? idx->d_pq_rows : nullptr, pq_m, luts, pq_direct ? idx->pq->d_codebooks : nullptr,
                      pq ? idx->pq->d_scales : nullptr, pq ? idx->pq->d_offsets : nullptr, k, ef_walk, lds_cand, lds_res, vis.ptr,
                      vis_words, heap_ws, oid.ptr + q0 * k, osc.ptr + q0 * k, ost.ptr ? ost.ptr + q0 : nullptr, redo,
                      mk.ptr ? mk.ptr + q0 * mask_stride : nullptr, mask_stride, ef_keep, idx->d_hnsw_tomb);
    }
    VG_TRY(oid.finish());
    VG_TRY(osc.finish());
    VG_TRY(ost.finish());
    if (oid.on_host() || osc.on_host() || ost.on_host()) VG_HIP(hipStreamSynchronize(st));
    return VG_OK;
}

VG_API int32_t vg_search_hnsw(vg_index *idx, const float *queries, int64_t nq, int32_t k, int32_t ef,
                              uint32_t *ids, float *scores, vg_search_stats *stats, void *stream)
{
    return search_hnsw_impl(idx, false, queries, nq, k, ef, ids, scores, stats, stream);
}

VG_API int32_t vg_search_hnsw_pq(vg_index *idx, const float *queries, int64_t nq, int32_t k, int32_t ef,
                                 uint32_t *ids, float *scores, vg_search_stats *stats, void *stream)
{
    return search_hnsw_impl(idx, true, queries, nq, k, ef, ids, scores, stats, stream);
}

VG_API int32_t vg_search_hnsw_filtered(vg_index *idx, const float *queries, int64_t nq, int32_t k, int32_t ef,
                                       const uint8_t *mask, int64_t mask_stride, double selectivity, uint32_t *ids,
                                       float *scores, vg_search_stats *stats, void *stream)
{
    VG_CHECK(idx, VG_ERR_INVALID_ARG, "vg_search_hnsw_filtered: NULL index");
    VG_CHECK(mask, VG_ERR_INVALID_ARG, "vg_search_hnsw_filtered: NULL mask (vg_search_hnsw is the unfiltered walk)");
    // searchLayer's strategy choice (hnsw.go:1120-1145): above highSelectivityThreshold the unfiltered walk + post-filter;
    // at or below it (0 = unknown) the predicate-aware walk (searchLayerPredicateAware, :1406; k_hnsw_predicate.hip), no tombstones
    if (!(selectivity > 0.3))
        return vg_search_hnsw_predicate(idx, queries, nq, k, ef, mask, mask_stride, nullptr, ids, scores, stats, stream);
    const int64_t mask_bytes = (idx->n + 7) / 8;
    VG_CHECK(mask_stride == 0 || mask_stride >= mask_bytes, VG_ERR_INVALID_ARG,
             "vg_search_hnsw_filtered: mask_stride %lld is shorter than a mask (%lld bytes)", static_cast<long long>(mask_stride),
             static_cast<long long>(mask_bytes));
    return search_hnsw_impl(idx, false, queries, nq, k, ef, ids, scores, stats, stream, mask, mask_stride, selectivity);
}

static int32_t vamana_impl(vg_index *idx, const float *queries, int64_t nq, int32_t k, int32_t kind, const uint8_t *mask,
                           int64_t mask_stride, uint32_t *ids, float *scores, vg_search_stats *stats, void *stream);

VG_API int32_t vg_search_vamana(vg_index *idx, const float *queries, int64_t nq, int32_t k, int32_t kind,
                                uint32_t *ids, float *scores, vg_search_stats *stats, void *stream)
{
    return vamana_impl(idx, queries, nq, k, kind, nullptr, 0, ids, scores, stats, stream);
}

VG_API int32_t vg_search_vamana_filtered(vg_index *idx, const float *queries, int64_t nq, int32_t k, int32_t kind,
                                         const uint8_t *mask, int64_t mask_stride, uint32_t *ids, float *scores,
                                         vg_search_stats *stats, void *stream)
{
    VG_CHECK(idx, VG_ERR_INVALID_ARG, "vg_search_vamana_filtered: NULL index");
    const int64_t mask_bytes = (idx->n + 7) / 8;
    VG_CHECK(mask == nullptr || mask_stride == 0 || mask_stride >= mask_bytes, VG_ERR_INVALID_ARG,
             "vg_search_vamana_filtered: mask_stride %lld is shorter than a mask (%lld bytes)", static_cast<long long>(mask_stride),
             static_cast<long long>(mask_bytes));
    return vamana_impl(idx, queries, nq, k, kind, mask, mask_stride, ids, scores, stats, stream);
}

static int32_t vamana_impl(vg_index *idx, const float *queries, int64_t nq, int32_t k, int32_t kind, const uint8_t *mask,
                           int64_t mask_stride, uint32_t *ids, float *scores, vg_search_stats *stats, void *stream)
{
    VG_CHECK(idx, VG_ERR_INVALID_ARG, "vg_search_vamana: NULL index");
    VG_CHECK(nq >= 0 && k >= 0, VG_ERR_INVALID_ARG, "vg_search_vamana: negative nq or k");
    if (nq == 0 || k == 0) return VG_OK;
    VG_CHECK(idx->d_vamana, VG_ERR_NOT_READY, "vg_search_vamana: index has no Vamana graph");
    VG_CHECK(kind >= 0 && kind <= 3, VG_ERR_INVALID_ARG, "vg_search_vamana: unknown kind %d", kind);
    VG_CHECK(kind != 3 || (idx->d_int4_rows && idx->int4_table), VG_ERR_NOT_READY,
             "vg_search_vamana: index has no INT4 codes");
    VG_CHECK(kind != 0 || idx->d_vectors, VG_ERR_NOT_READY, "vg_search_vamana: index has no fp32 vectors");
    VG_CHECK(kind != 1 || (idx->d_pq_rows && idx->pq), VG_ERR_NOT_READY, "vg_search_vamana: index has no PQ codes");
    VG_CHECK(kind != 2 || idx->d_rq_rows, VG_ERR_NOT_READY, "vg_search_vamana: index has no RaBitQ codes");
    VG_CHECK(kind != 1 || idx->pq->k == 256, VG_ERR_UNSUPPORTED, "vg_search_vamana: PQ needs numCentroids == 256");
    VG_CHECK(idx->metric != VG_METRIC_HAMMING, VG_ERR_UNSUPPORTED, "unsupported metric for float32: Hamming");
    VG_CHECK(queries && ids && scores, VG_ERR_INVALID_ARG, "vg_search_vamana: NULL buffer");
    VG_CHECK(k <= vg::kVamanaMaxK, VG_ERR_UNSUPPORTED, "vg_search_vamana: k=%d exceeds %d", k, vg::kVamanaMaxK);
    VG_HIP(hipSetDevice(idx->ctx->device));
    hipStream_t st = vg::pick_stream(idx->ctx, stream);
    vg::DevIn<float> q;
    vg::DevOut<uint32_t> oid;
    vg::DevOut<float> osc;
    vg::DevOut<vg_search_stats> ost;
    VG_TRY(q.init(queries, static_cast<size_t>(nq) * idx->dim, st));
    VG_TRY(oid.init(ids, static_cast<size_t>(nq) * k, st));
    VG_TRY(osc.init(scores, static_cast<size_t>(nq) * k, st));
    VG_TRY(ost.init(stats, stats ? static_cast<size_t>(nq) : 0, st));
    vg::DevIn<uint8_t> mk;
    VG_TRY(mk.init(mask, mask ? static_cast<size_t>(mask_stride ? (nq - 1) * mask_stride + (idx->n + 7) / 8 : (idx->n + 7) / 8) : 0, st));
    const int64_t vis_words = (idx->n + 31) / 32;
    // The reference's exploration heap is unbounded (ScratchCandidates, diskann/segment.go:641-703).  A node enters it at most once
    // (the visited test), so n items are "unbounded" here.  The unfiltered walk stops at k results long before 65 536; a walk with a
    // selective row filter goes on until k MATCHING rows are found and can fill any smaller heap (pushes were then
    // dropped silently): with a filter the cap is n — nothing is ever dropped — at the price of fewer queries per launch.
    const int64_t cand_cap = mask ? idx->n : std::min<int64_t>(idx->n, 65536);
    const int64_t per_query = vis_words * 4 + cand_cap * 8;
    int64_t chunk = std::max<int64_t>(1, graph_scratch_cap(idx->ctx) / per_query);
    chunk = std::min(chunk, nq);
    const int rq_nb = ((idx->dim + 63) / 64) * 8;
    const int pq_m = idx->pq ? idx->pq->m : 0;
    vg::ArenaCall ar(idx->ctx, st);
    const int i_vis = ar.add(sizeof(uint32_t) * static_cast<size_t>(chunk) * vis_words);
    const int i_cand = ar.add(sizeof(vg::HItem) * static_cast<size_t>(chunk) * cand_cap);
    const bool pq_direct = kind == 1 && idx->pq->subdim == 8 && (reinterpret_cast<uintptr_t>(idx->pq->d_codebooks) & 7) == 0;
    // INT4: terms evaluated in place when the rows are whole 32-element blocks (16-byte aligned codes)
    const bool int4_direct = kind == 3 && idx->dim % 32 == 0 && idx->int4_min && idx->int4_diff &&
                             (reinterpret_cast<uintptr_t>(idx->d_int4_rows) & 15) == 0;
    const int i_luts = ar.add(kind == 1 && !pq_direct ? sizeof(float) * static_cast<size_t>(nq) * pq_m * 256 : 0);
    const int i_qcodes = ar.add(kind == 2 ? static_cast<size_t>(nq) * (rq_nb + 4) : 0);
    VG_TRY(ar.commit());
    struct { uint32_t *ptr; } vis{ar.get<uint32_t>(i_vis)};
    struct { vg::HItem *ptr; } cand{ar.get<vg::HItem>(i_cand)};
    struct { float *ptr; } luts{ar.get<float>(i_luts)};
    struct { uint8_t *ptr; } qcodes{ar.get<uint8_t>(i_qcodes)};
    if (kind == 1 && !pq_direct) {
        VG_TRY(vg::launch_pq_build_table(idx->pq, q.ptr, nq, luts.ptr, false, st));
    }
    if (kind == 2) {
        VG_TRY(vg::launch_rabitq_encode(q.ptr, nq, idx->dim, qcodes.ptr, st));
    }
    for (int64_t q0 = 0; q0 < nq; q0 += chunk) {
        const int64_t cnt = std::min(chunk, nq - q0);
        VG_HIP(hipMemsetAsync(vis.ptr, 0, static_cast<size_t>(cnt) * vis_words * 4, st));
        vg::ProfScope prof(idx->ctx, "vamana_search", st);
        auto launch = [&](auto kernel) -> int32_t {
            VG_LAUNCH(kernel, dim3(static_cast<unsigned>(cnt)), dim3(64),
                      pq_direct ? static_cast<size_t>(pq_m >> 1) * vg::kPqPairFloats * sizeof(float) : (kind == 2 ? static_cast<size_t>(rq_nb) : 0), st,
                      idx->metric, idx->n, idx->dim, idx->d_vamana, idx->vamana_r, idx->vamana_entry,
                      idx->d_vectors, idx->d_pq_rows, pq_m, kind == 1 && !pq_direct ? luts.ptr + q0 * pq_m * 256 : nullptr,
                      pq_direct ? idx->pq->d_codebooks : nullptr, idx->pq ? idx->pq->d_scales : nullptr,
                      idx->pq ? idx->pq->d_offsets : nullptr,
                      idx->d_rq_rows, kind == 2 ? qcodes.ptr + q0 * (rq_nb + 4) : nullptr, rq_nb,
                      idx->d_int4_rows, idx->int4_table, idx->int4_min, idx->int4_diff, q.ptr + q0 * idx->dim, k, cand.ptr, cand_cap, vis.ptr, vis_words, oid.ptr + q0 * k,
                      osc.ptr + q0 * k, ost.ptr ? ost.ptr + q0 : nullptr, mk.ptr ? mk.ptr + q0 * mask_stride : nullptr, mask_stride,
                      kind == 0 ? idx->d_norm_max + 1 : kind == 2 ? idx->d_rq_norms + idx->n : nullptr);
            return VG_OK;
        };
        const bool big = k > 64;
        int32_t rc;
        const int inst = pq_direct ? vg::kVamanaPQDirect : int4_direct ? vg::kVamanaInt4Direct : kind;
#define VG_VAMANA_CASE(K)                                                                                                   \
    case K:                                                                                                                 \
        rc = mk.ptr ? (big ? launch(vg::vamana_search_kernel<K, true, true>) : launch(vg::vamana_search_kernel<K, false, true>)) \
                    : (big ? launch(vg::vamana_search_kernel<K, true>) : launch(vg::vamana_search_kernel<K, false>));       \
        break;
        switch (inst) {
            VG_VAMANA_CASE(0)
            VG_VAMANA_CASE(1)
            VG_VAMANA_CASE(4)
            VG_VAMANA_CASE(5)
            VG_VAMANA_CASE(2)
        default:
            rc = mk.ptr ? (big ? launch(vg::vamana_search_kernel<3, true, true>) : launch(vg::vamana_search_kernel<3, false, true>))
                        : (big ? launch(vg::vamana_search_kernel<3, true>) : launch(vg::vamana_search_kernel<3, false>));
            break;
        }
#undef VG_VAMANA_CASE
        VG_TRY(rc);
        // the queries whose distances may hold a NaN, again with the reference's CandidateHeap (every other query returns at once)
        if (!vg::hook(vg::kHookNoCandReplay)) {
            switch (inst) {
            case 0: rc = launch(vg::vamana_search_kernel<0, true, true, true>); break;
            case 1: rc = launch(vg::vamana_search_kernel<1, true, true, true>); break;
            case 4: rc = launch(vg::vamana_search_kernel<4, true, true, true>); break;
            case 5: rc = launch(vg::vamana_search_kernel<5, true, true, true>); break;
            case 2: rc = launch(vg::vamana_search_kernel<2, true, true, true>); break;
            default: rc = launch(vg::vamana_search_kernel<3, true, true, true>); break;
            }
            VG_TRY(rc);
        }
    }
    VG_TRY(oid.finish());
    VG_TRY(osc.finish());
    VG_TRY(ost.finish());
    if (oid.on_host() || osc.on_host() || ost.on_host()) VG_HIP(hipStreamSynchronize(st));
    return VG_OK;
}
