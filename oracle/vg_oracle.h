/*
 * vg_oracle.h — CPU restatement of vecgo's distance + quantization hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (vecgo_amd/, the
 * C-ABI library) may include, link or call this.  Only tests/, the smoke()
 * check in __graft_entry__.py and bench.py's cpu_baseline leg use it, and
 * only as the checker / the reported CPU baseline.
 *
 * Every function cites the reference file:line (relative to the vecgo repo
 * root) whose arithmetic it follows.  The float kernels reproduce the
 * accumulation ORDER of the reference's AVX-512 C kernels (4x16 lane
 * accumulators, FMA, _mm512_reduce_add_ps tree) in scalar C, so results are
 * bit-identical to the clang-built reference objects (pinned by
 * tests/golden/ fixtures generated from those objects, see
 * tests/golden/make_golden*.py, and cross-checked live against oracle/_ref
 * when that library is present): L2 / Dot / batch / bounded / ADC / Hamming /
 * SQ8 / INT4 kernels.  The parts of the reference written in Go (PQ, RaBitQ,
 * ScalarQuantizer, Int4Quantizer, k-means, the heaps and search loops, the
 * neighbour selections of the builders) cannot be compiled here; they are
 * pinned by the reference tests' known answers (tests/golden/reference_kats.json,
 * tests/test_oracle_*.py).  PARITY UNPINNED for whatever the reference leaves to
 * an unseeded RNG or an unspecified order: trained codebooks, built graphs, the
 * order of equal-distance candidates in robustPrune.
 *
 * Files: vg_oracle.c (kernels, quantizers, heaps, scans, searches), vg_oracle_hnsw_build.c (hnsw.Insert),
 * vg_oracle_opq.c (opq.go, svd.go), vg_cpu_bench.c (the C-threaded harness behind bench.py's CPU twins and the
 * full-size replays; with vgo_set_kernel_hooks the timed loops call the REFERENCE's compiled kernels from
 * oracle/_ref — bit-identical to the restatements, see tests/test_oracle_golden.py).
 *
 * Build: gcc/clang, -ffp-contract=off (every FMA below is an explicit
 * __builtin_fmaf; everything else is a separately rounded IEEE fp32 op).
 */
#ifndef VG_ORACLE_H
#define VG_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- L0 kernels (internal/simd) ------------------------------------- */
float vgo_dot_avx512(const float *a, const float *b, int64_t n);
float vgo_l2_avx512(const float *a, const float *b, int64_t n);
void vgo_l2_batch_avx512(const float *query, const float *targets, int64_t dim,
                         int64_t n, float *out);
void vgo_dot_batch_avx512(const float *query, const float *targets, int64_t dim,
                          int64_t n, float *out);
void vgo_l2_bounded_avx512(const float *a, const float *b, int64_t n, float bound,
                           float *result, int32_t *exceeded);
float vgo_adc_avx512(const float *table, const uint8_t *codes, int64_t m);
float vgo_adc_generic(const float *table, const uint8_t *codes, int64_t m);
int64_t vgo_hamming(const uint8_t *a, const uint8_t *b, int64_t n);
void vgo_scale(float *a, int64_t n, float s);
float vgo_sqrt(float x);

/* int8-codebook PQ helpers: the pure-Go generics that amd64 actually runs */
float vgo_l2_int8_deq(const float *query, const int8_t *code, int64_t n, float scale,
                      float offset);
void vgo_build_table_int8(const float *qsub, const int8_t *codebook, int64_t subdim,
                          int64_t k, float scale, float offset, float *out);
int64_t vgo_nearest_centroid_int8(const float *qsub, const int8_t *codebook,
                                  int64_t subdim, int64_t k, float scale, float offset);

/* ---- L1 codecs (internal/quantization) ------------------------------- */
typedef struct {
    int32_t dim, m, k, subdim;
    const int8_t *codebooks; /* m*k*subdim */
    const float *scales;     /* m */
    const float *offsets;    /* m */
} vgo_pq;

void vgo_pq_build_table(const vgo_pq *pq, const float *query, float *table /* m*k */);
void vgo_pq_encode(const vgo_pq *pq, const float *vec, uint8_t *codes /* m */);
void vgo_pq_decode(const vgo_pq *pq, const uint8_t *codes, float *out /* dim */);
float vgo_pq_asym_distance(const vgo_pq *pq, const float *query, const uint8_t *codes);
/* int8-quantise fp32 centroids of one subspace (pq.go:97-136) */
void vgo_pq_quantize_centroids(const float *centroids, int64_t count, int8_t *out,
                               float *scale, float *offset);
/* Lloyd iterations + k-means++ init for one subspace, deterministic RNG */
void vgo_pq_train_subspace(const float *vectors, int64_t n, int32_t dim, int32_t m, int32_t sub, int32_t k,
                           int32_t iters, uint64_t seed, float *cent);
int vgo_pq_train(const float *vectors, int64_t n, int32_t dim, int32_t m, int32_t k,
                 int32_t iters, uint64_t seed, int8_t *codebooks, float *scales,
                 float *offsets, float *centroids_f32 /* optional m*k*subdim */);

int64_t vgo_rabitq_code_bytes(int32_t dim);
void vgo_rabitq_encode(const float *v, int32_t dim, uint8_t *out);
float vgo_rabitq_distance(const float *query, int32_t dim, const uint8_t *code);
void vgo_binary_encode_u64(const float *v, int32_t dim, float threshold, uint64_t *dst);
float vgo_binary_train(const float *vectors, int64_t n, int32_t dim);
void vgo_binary_decode(const uint8_t *code, int32_t code_bytes, int32_t dim, float threshold, float *out);
int32_t vgo_normalize_l2(float *v, int32_t dim);

/* ---- internal/kmeans -------------------------------------------------- */
enum { VGO_METRIC_L2 = 0, VGO_METRIC_COSINE = 1, VGO_METRIC_DOT = 2, VGO_METRIC_HAMMING = 3 };
int vgo_kmeans_train(const float *vectors, int64_t n, int32_t dim, int32_t k, int32_t metric,
                     int32_t max_iter, uint64_t seed, float *centroids);
int32_t vgo_assign_partition(const float *vec, const float *centroids, int32_t dim, int32_t k,
                             int32_t metric);
int vgo_find_closest_centroids(const float *query, const float *centroids, int32_t dim,
                               int32_t k, int32_t n, int32_t metric, int32_t *out);

/* deterministic counter-based RNG shared with the device code */
uint64_t vgo_rng_u64(uint64_t seed, uint64_t a, uint64_t b, uint64_t c);

/* ---- internal/searcher heaps ------------------------------------------ */
typedef struct {
    uint32_t node;
    float dist;
} vgo_pq_item;
typedef struct {
    vgo_pq_item *items;
    int32_t len, cap;
    int32_t is_max;
} vgo_prioq;
void vgo_prioq_init(vgo_prioq *q, int is_max, int32_t cap);
void vgo_prioq_free(vgo_prioq *q);
void vgo_prioq_push(vgo_prioq *q, vgo_pq_item it);
void vgo_prioq_push_bounded(vgo_prioq *q, vgo_pq_item it, int32_t capacity);
int vgo_prioq_try_push_bounded(vgo_prioq *q, vgo_pq_item it, int32_t max_size);
int vgo_prioq_pop(vgo_prioq *q, vgo_pq_item *out);
/* script of queue operations (op codes = VG_HEAP_* of include/vecgo_hip.h); returns the final length */
int32_t vgo_prioq_replay(int is_max, const int32_t *ops, int32_t n_ops, int32_t *out, uint64_t *final_items, int32_t cap);

/* searcher.VisitedSet (visited.go) literally, driven by a script (see vg_oracle.c) */
int32_t vgo_visited_replay(int32_t capacity, const int64_t *ops, int32_t n_ops, int64_t *out);

typedef struct {
    uint32_t segment_id, row_id;
    float score;
} vgo_cand;
typedef struct {
    vgo_cand *c;
    int32_t len, cap;
    int32_t descending;
} vgo_candheap;
void vgo_candheap_init(vgo_candheap *h, int32_t cap, int descending);
void vgo_candheap_free(vgo_candheap *h);
int vgo_cand_better(vgo_cand a, vgo_cand b, int descending);
int vgo_candheap_try_push_bounded(vgo_candheap *h, vgo_cand x, int32_t k);
int vgo_candheap_pop(vgo_candheap *h, vgo_cand *out);
int vgo_candheap_replace_top(vgo_candheap *h, vgo_cand x);
/* best-first copy (SortedResults) */
int32_t vgo_candheap_sorted(const vgo_candheap *h, vgo_cand *dst);

/* ---- scan / search loops ---------------------------------------------- */
/* flat/segment.go:606-723 — every row scored, CandidateHeap top-k; outputs
 * best-first. Returns number of results (min(k, rows)). */
int32_t vgo_flat_search_f32(const float *base, int64_t n, int32_t dim, int32_t metric,
                            const float *query, int32_t k, uint32_t *ids, float *scores);
int32_t vgo_flat_search_pq(const vgo_pq *pq, const uint8_t *codes, int64_t n,
                           const float *query, int32_t k, uint32_t *ids, float *scores);
int32_t vgo_flat_search_rabitq(const uint8_t *codes, int64_t n, int32_t dim,
                               const float *query, int32_t k, uint32_t *ids, float *scores);
/* segment Rerank: exact distance for given ids (flat/segment.go:754-780) */
void vgo_rerank_f32(const float *base, int32_t dim, int32_t metric, const float *query,
                    const uint32_t *ids, int32_t n, float *scores);

typedef struct {
    int64_t n;
    int32_t dim;
    int32_t metric;      /* VGO_METRIC_* ; distances as hnsw wraps them */
    const float *base;   /* n*dim */
    int32_t m0;          /* layer-0 max degree */
    const uint32_t *l0;  /* n*m0, 0xFFFFFFFF = end of list */
    int32_t max_level;   /* highest level with nodes (0 = only layer 0) */
    int32_t m;           /* upper-layer max degree */
    /* upper layers: for level L in 1..max_level, slot[L-1][node] = row into
     * adj[L-1] or 0xFFFFFFFF; adj[L-1] is count*m ids, 0xFFFFFFFF-padded */
    const uint32_t *const *slot;
    const uint32_t *const *adj;
    uint32_t entry_point;
    /* appended: when pq != NULL the nodes are scored from their PQ codes (n*m bytes) with
     * ComputeAsymmetricDistance (pq.go:234-260) instead of their fp32 rows — distFunc the way
     * diskann/segment.go:536-557 builds it; no SquaredL2Bounded short-circuit then */
    const vgo_pq *pq;
    const uint8_t *codes;
    /* appended: g.tombstones (hnsw.go:95) as a bitmap, bit i of byte i/8 = node i deleted; NULL = none.  A deleted node is
     * walked through but never enters the results (hnsw.go:1381-1390, :1562, :1198, :1485, :1580). */
    const uint8_t *tombstones;
} vgo_hnsw_graph;

typedef struct {
    int64_t nodes_visited, distance_computations, distance_short_circuits, pops;
} vgo_search_stats;

/* hnsw.go:1791 searchExecute = greedySearch + searchLayerUnfiltered(level 0),
 * then knnSearchInternal's extraction: best-first k results. */
int32_t vgo_hnsw_search(const vgo_hnsw_graph *g, const float *query, int32_t k, int32_t ef,
                        uint32_t *ids, float *scores, vgo_search_stats *stats);

/* searchLayerWithPostFilter (hnsw.go:1159-1218) behind searchExecute: selectivity > 0.3 only (-1 otherwise) */
int32_t vgo_hnsw_search_filtered(const vgo_hnsw_graph *g, const float *query, int32_t k, int32_t ef, const uint8_t *mask,
                                 double selectivity, uint32_t *ids, float *scores, vgo_search_stats *stats);

/* searchLayerPredicateAware (hnsw.go:1406-1558) behind searchExecute — selectivity <= 0.3 or unknown.  mask = filter.Matches,
 * deleted = tombstones (NULL = none), l0_dist = the layer-0 lists' cached Neighbor.Dist (n*m0; NULL = recomputed from the rows).
 * stats->distance_short_circuits = ExpansionsSkipped. */
int32_t vgo_hnsw_search_predicate(const vgo_hnsw_graph *g, const float *query, int32_t k, int32_t ef, const uint8_t *mask,
                                  const uint8_t *deleted, const float *l0_dist, uint32_t *ids, float *scores,
                                  vgo_search_stats *stats);

/* hnsw.BruteSearch + scanSegment (hnsw.go:2021-2101) and searchBitmap (:2240-2263) + extraction (:1732-1751):
 * exhaustive scans over the rows whose mask bit is set (NULL = all), each with its own heap discipline */
enum { VGO_BRUTE_SCAN = 0, VGO_BRUTE_BITMAP = 1 };
int32_t vgo_hnsw_brute_search(const vgo_hnsw_graph *g, const float *query, int32_t k, int32_t mode,
                              const uint8_t *mask, uint32_t *ids, float *scores);

/* see vg_oracle.c "Optional kernel hooks": the reference's compiled kernels (oracle/_ref) for the timed CPU
 * baseline; signatures are the reference's C ABI (the _avx512.c files of internal/simd/src) */
typedef struct {
    void (*l2)(float *, float *, int64_t, float *);
    void (*dot)(float *, float *, int64_t, float *);
    void (*l2_bounded)(float *, float *, int64_t, float, float *, int32_t *);
    void (*l2_batch)(float *, float *, int64_t, int64_t, float *);
    void (*adc)(float *, uint8_t *, int64_t, float *, const void *);
    long long (*hamming)(const uint8_t *, const uint8_t *, int64_t);
} vgo_kernel_hooks;
extern vgo_kernel_hooks vgo_hooks;
void vgo_set_kernel_hooks(const vgo_kernel_hooks *h);

/* one layer of the search, shared by vgo_hnsw_search and the builder (hnsw.go:1220-1396) */
void vgo_hnsw_search_layer(const vgo_hnsw_graph *g, const float *query, uint32_t ep, float ep_d,
                           int32_t level, int32_t ef, uint32_t *visited, uint32_t epoch,
                           vgo_prioq *cand, vgo_prioq *res, vgo_search_stats *st);
float vgo_hnsw_node_distance(const vgo_hnsw_graph *g, const float *query, uint32_t id);

/* ---- OptimizedProductQuantizer (vg_oracle_opq.c; opq.go, svd.go) -------------------------------------- */
int32_t vgo_opq_block_size(int32_t dim, int32_t m);
/* rot = [dim/block][block][block] row-major */
void vgo_opq_rotate(const float *rot, int32_t dim, int32_t block, const float *src, float *dst);
void vgo_opq_unrotate(const float *rot, int32_t dim, int32_t block, const float *src, float *dst);
void vgo_procrustes(float *m_matrix /* n*n, destroyed */, int32_t n, float *r_out);
void vgo_opq_accumulate_m(const float *x, const float *y, int64_t n, int32_t dim, int32_t block, float *m_out);
int vgo_opq_train(const float *vectors, int64_t n, int32_t dim, int32_t m, int32_t k, int32_t opq_iters,
                  int32_t pq_iters, uint64_t seed, float *rot, int8_t *codebooks, float *scales, float *offsets);

/* ---- HNSW construction (vg_oracle_hnsw_build.c) ------------------------------------------------
 * hnsw.go:713-984 insert / insertNode with the ids and levels of ApplyInsert (hnsw.go:629-684:
 * ids are the row numbers, level = layerForApplyInsert(id) hnsw.go:2103-2116), selectNeighbors
 * (:986-1106) and addConnection / addConnectionPrune (:455-555).  Nodes are inserted in id order in
 * batches: every node of a batch searches the graph as it stood when the batch began (the view a
 * goroutine of ApplyBatchInsert has of the nodes its siblings are inserting), then the batch's links
 * are applied in id order.  batch size = clamp(inserted / growth_div, 1, max_batch); max_batch = 1 is
 * the reference's sequential Insert loop. */
int32_t vgo_hnsw_level_for_id(uint64_t id, int32_t m);
/* levels[n] (may be NULL), level_rows[l] = nodes with level >= l+1 for l < 63; returns the highest level */
int32_t vgo_hnsw_build_layout(int64_t n, int32_t m, int32_t *levels, int64_t *level_rows);
/* next batch size of the schedule above */
int64_t vgo_hnsw_build_batch(int64_t inserted, int64_t n, int32_t max_batch, int32_t growth_div);
/* l0: n*2m ids; slots: max_level*n; adj: (sum level_rows)*m ids, level tables concatenated (the C-ABI's
 * vg_index_set_hnsw_graph layout); every list is 0xFFFFFFFF-terminated.  Returns 0, or -1 on bad input. */
int32_t vgo_hnsw_build(const float *base, int64_t n, int32_t dim, int32_t metric, int32_t m, int32_t ef,
                       int32_t max_batch, int32_t growth_div, uint32_t *l0, uint32_t *slots, uint32_t *adj,
                       uint32_t *entry_point, int32_t *max_level);

int32_t vgo_hnsw_search_ws(const vgo_hnsw_graph *g, const float *query, int32_t k, int32_t ef,
                           uint32_t *ids, float *scores, vgo_search_stats *stats, uint32_t *visited_ws,
                           uint32_t epoch);

enum { VGO_VAMANA_F32 = 0, VGO_VAMANA_PQ = 1, VGO_VAMANA_RABITQ = 2, VGO_VAMANA_INT4 = 3 };
typedef struct {
    int64_t n;
    int32_t dim;
    int32_t r;             /* max degree */
    const uint32_t *graph; /* n*r, 0xFFFFFFFF = none */
    uint32_t entry_point;
    int32_t kind;
    int32_t metric;        /* for VGO_VAMANA_F32: L2 / Dot */
    const float *base;     /* F32 */
    const vgo_pq *pq;      /* PQ */
    const uint8_t *codes;  /* PQ: n*m ; RABITQ: n*code_bytes ; INT4: n*ceil(dim/2) */
    const float *int4_table; /* INT4: BuildInt4LookupTable, dim*16 (appended field) */
} vgo_vamana;
/* diskann/segment.go:503-706 searchInternal, no filters */
int32_t vgo_vamana_search(const vgo_vamana *v, const float *query, int32_t k,
                          uint32_t *ids, float *scores, vgo_search_stats *stats);
/* with `filter` (segment.go:616-627: pushToHeap skips a row whose filter.Matches is false); mask bit i = Matches(i) */
int32_t vgo_vamana_search_filtered(const vgo_vamana *v, const float *query, int32_t k, const uint8_t *mask, uint32_t *ids,
                                   float *scores, vgo_search_stats *stats);

/* SQ8 (sq8_avx512.c:59-103, quantizer.go:27-250, flat/segment.go:517-604) */
void vgo_sq8u_l2_batch(const float *query, const uint8_t *codes, const float *mins, const float *inv_scales,
                       int64_t dim, int64_t n, float *out);
void vgo_sq8_train(const float *vectors, int64_t n, int32_t dim, float *mins, float *maxs, float *scales,
                   float *inv_scales);
void vgo_sq8_encode(const float *v, int32_t dim, const float *mins, const float *maxs, const float *scales,
                    uint8_t *out);
void vgo_sq8_decode(const uint8_t *code, int32_t dim, const float *mins, const float *inv_scales, float *out);
int32_t vgo_flat_search_sq8(const uint8_t *codes, int64_t n, int32_t dim, const float *mins,
                            const float *inv_scales, const float *query, int32_t k, uint32_t *ids,
                            float *scores);

float vgo_sq8_dot(const float *q, const uint8_t *code, int32_t dim, const float *mins, const float *inv_scales);
/* flat.Segment.Search (flat/segment.go:447-751) over a whole flat segment: scan type by quantization
 * (sq_mins != NULL: SQ8 L2Distance / DotProduct by metric; else pq != NULL: ADC; else fp32), IVF partitions probed when
 * num_partitions > 1 (:727-744). */
typedef struct {
    int64_t n;
    int32_t dim, metric;
    const float *base;          /* n*dim fp32 rows */
    const vgo_pq *pq;           /* PQ segment: codes = n*m */
    const uint8_t *codes;       /* PQ or SQ8 codes */
    const float *sq_mins, *sq_inv_scales; /* SQ8 segment */
    int32_t num_partitions;
    const float *centroids;     /* num_partitions*dim */
    const uint32_t *part_offsets; /* num_partitions+1 */
} vgo_flat_segment;
int32_t vgo_flat_segment_search(const vgo_flat_segment *s, const float *query, int32_t k, int32_t nprobes,
                                uint32_t *ids, float *scores);
/* with a row filter (segment.go:631-635, :559-561): mask bit i of byte i/8 = filter.Matches(i); NULL = none */
int32_t vgo_flat_segment_search_filtered(const vgo_flat_segment *s, const float *query, int32_t k, int32_t nprobes,
                                         const uint8_t *mask, uint32_t *ids, float *scores);

/* INT4 (int4_avx512.c, int4.go, kernels.go:94-103) */
float vgo_int4_l2(const float *query, const uint8_t *code, int64_t dim, const float *min_val, const float *diff);
void vgo_int4_l2_batch(const float *query, const uint8_t *codes, int64_t dim, int64_t n, const float *min_val,
                       const float *diff, float *out);
void vgo_int4_build_lut(const float *min_val, const float *diff, int32_t dim, float *table);
float vgo_int4_l2_precomputed(const float *query, const uint8_t *code, int64_t dim, const float *table);
void vgo_int4_train(const float *vectors, int64_t n, int32_t dim, float *min_val, float *diff);
void vgo_int4_encode(const float *v, int32_t dim, const float *min_val, const float *diff, uint8_t *out);
void vgo_int4_decode(const uint8_t *code, int32_t dim, const float *min_val, const float *diff, float *out);

/* construction-time neighbour selection (diskann/writer.go:571-625, hnsw.go:1009-1106) */
int32_t vgo_robust_prune(const float *base, int64_t n, int32_t dim, int32_t metric, uint32_t node,
                         const uint32_t *cands, int32_t nc, int32_t r, float alpha, uint32_t *out);
int32_t vgo_hnsw_select_neighbors(const float *base, int64_t n, int32_t dim, int32_t metric,
                                  const uint32_t *cand_ids, const float *cand_dists, int32_t nc, int32_t m,
                                  uint32_t *out);

#ifdef __cplusplus
}
#endif
#endif
