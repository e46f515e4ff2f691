/*
 * vecgo_hip.h — C ABI of libvecgo_hip.so: the MI355X (gfx950) implementation of
 * vecgo's distance + quantization hot path.
 *
 * This is the drop-in boundary.  Every entry point names the reference interface
 * (file:line under the vecgo repository root) it stands in for.  Plain C types,
 * caller-owned buffers, no callbacks; INTEGRATION.md shows the cgo binding a
 * vecgo maintainer would add (internal/simd/kernels_hip.go + quantization shims).
 *
 * Conventions
 *   - Every function returns a vg_status (0 = ok, negative = error) and never
 *     aborts; vg_last_error() gives the message for the calling thread.
 *   - Data pointers may be HOST or DEVICE (HBM) pointers; the library detects
 *     which (hipPointerGetAttributes).  Host buffers are staged through HBM and
 *     the call returns after the results are back in the caller's buffer (the Go
 *     `//go:noescape` contract: no pointer is retained).  When every buffer of a
 *     call is a device pointer the call only enqueues work on `stream` and
 *     returns; results are ready when the stream reaches that point.
 *   - `stream` is a hipStream_t passed as void*.  NULL = the context's own (non-blocking)
 *     stream — right for host-buffer callers such as cgo.  A caller that produces device
 *     buffers on HIP's legacy default stream (handle 0, e.g. PyTorch's default stream) must
 *     pass VG_STREAM_LEGACY so that its work and the library's are ordered.
 *   - Index / quantizer handles may be used from many threads at once for the
 *     read-only calls (search, encode, build table); create / set / train /
 *     destroy must be externally serialised (internal/quantization/doc.go:120-123).
 *   - Zero-length inputs succeed and produce empty / zero outputs
 *     (internal/simd/kernels_amd64.go:291-297).
 *   - Result ids are uint32 row ids (model.RowID); unused result slots hold
 *     id 0xFFFFFFFF and score +Inf (L2-like) or -Inf (Dot).
 */
#ifndef VECGO_HIP_H
#define VECGO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever a struct the caller allocates, or an entry point's argument list, changes; vg_abi_version()
 * returns the value the library was built with — a host compares the two before its first call.
 *   2: vg_search_stats has FIVE int64 (descent_distance_computations was appended): vg_search_hnsw / _hnsw_pq /
 *      _vamana write nq * 5 values. */
#define VG_ABI_VERSION 2
/* Bumped whenever entry points are ADDED (nothing existing changes): a binding that wants a newer entry point compares
 * vg_abi_minor() with the value below before it looks the symbol up, instead of failing on first use.
 *   1: (r04) vg_search_hnsw_brute, vg_debug_heap_replay
 *   2: (r05) vg_abi_minor itself; no other symbol added — vg_search_hnsw / _hnsw_pq answer NaN distances as the reference
 *      does, vg_kmeans_* decide assignments on the matrix cores, the k-means++ running sum of vg_pq_train is blocked
 *   3: (r05) vg_search_hnsw_filtered
 *   4: (r05) vg_search_flat_filtered
 *   5: (r05) vg_search_vamana_filtered
 *   6: (r05) vg_search_hnsw_predicate, vg_index_set_hnsw_edge_distances; vg_search_hnsw_filtered serves every selectivity
 *   7: (r05) vg_index_set_hnsw_tombstones
 *   8: (r05) vg_segment_search_filtered
 *   9: (r05) vg_index_enable_sq8_nomination
 *  10: (r06) vg_index_enable_pq_nomination; NaN scores answered as the reference's heaps answer them (see "NaN scores") */
#define VG_ABI_MINOR 10
#define VG_INVALID_ID 0xFFFFFFFFu
#define VG_STREAM_LEGACY ((void *)1) /* == hipStreamLegacy */

/* error conventions: Go (value, error) strings in parentheses are what the Go
 * shim maps each code to (internal/quantization/pq.go:148-153,190,496;
 * rabitq.go:53,124) */
typedef enum vg_status {
    VG_OK = 0,
    VG_ERR_INVALID_ARG = -1,        /* nil/negative/inconsistent arguments */
    VG_ERR_DIM_MISMATCH = -2,       /* "vector dimension mismatch" */
    VG_ERR_NOT_TRAINED = -3,        /* "ProductQuantizer not trained" */
    VG_ERR_CODE_LENGTH = -4,        /* "codes length mismatch" / "invalid code length" */
    VG_ERR_UNSUPPORTED = -5,        /* "unsupported metric for float32: ..." / shape not supported */
    VG_ERR_OUT_OF_MEMORY = -6,
    VG_ERR_HIP = -7,                /* a HIP runtime call failed */
    VG_ERR_NO_DEVICE = -8,          /* no gfx950 device / extension not usable */
    VG_ERR_NOT_READY = -9,          /* index lacks the data this search needs */
    VG_ERR_FORMAT = -10,            /* segment image: "invalid magic number", "file too short for ..." */
    VG_ERR_CHECKSUM = -11           /* segment image: "checksum mismatch: expected %x, got %x" */
} vg_status;

/* distance.Metric — distance/distance.go:66-73 */
typedef enum vg_metric {
    VG_METRIC_L2 = 0,
    VG_METRIC_COSINE = 1,
    VG_METRIC_DOT = 2,
    VG_METRIC_HAMMING = 3
} vg_metric;

typedef struct vg_ctx vg_ctx;     /* one per (process, GPU) */
typedef struct vg_pq vg_pq;       /* quantization.ProductQuantizer — pq.go:20-29 */
typedef struct vg_index vg_index; /* device-resident rows / codes / graph of one segment */
typedef struct vg_sq8 vg_sq8;     /* quantization.ScalarQuantizer — quantizer.go:27-39 */
typedef struct vg_int4 vg_int4;   /* quantization.Int4Quantizer — int4.go:12-20 */
typedef struct vg_segment vg_segment; /* an opened flat / DiskANN segment image: index + quantizers */

/* ---- context ------------------------------------------------------------ */
int32_t vg_abi_version(void);
int32_t vg_abi_minor(void);
/* device = HIP ordinal (LOCAL_RANK in a one-process-per-GPU job) */
int32_t vg_ctx_create(int32_t device, vg_ctx **out);
int32_t vg_ctx_destroy(vg_ctx *ctx);
int32_t vg_ctx_synchronize(vg_ctx *ctx, void *stream);
const char *vg_last_error(void);
const char *vg_status_string(int32_t status);
/* name (e.g. "gfx950"), CU count and HBM bytes of the context's device */
int32_t vg_ctx_device_info(vg_ctx *ctx, char *arch, int32_t arch_len, int32_t *compute_units,
                           int64_t *hbm_bytes);

/* per-kernel timing with HIP events on the launching stream (the analogue of the reference's
 * per-query FilterGateStats.SearchTimeNanos, searcher/searcher.go:114-137).  While enabled,
 * every launch of the named hot kernels is bracketed by an event pair.  vg_profile_read
 * synchronises, then returns the number of launches and their summed duration since the last
 * read for `kernel` ("flat_gemm", "pq_adc_scan", "rabitq_scan", "flat_select", "topk_merge",
 * "hnsw_search", "vamana_search"), and clears those records. */
int32_t vg_profile_enable(vg_ctx *ctx, int32_t on);
int32_t vg_profile_read(vg_ctx *ctx, const char *kernel, int64_t *launches, double *total_ms);

/* ---- ProductQuantizer (internal/quantization/pq.go) ------------------------ */
/* NewProductQuantizer pq.go:36-64: dim % m == 0, 0 < k <= 256 */
int32_t vg_pq_create(vg_ctx *ctx, int32_t dim, int32_t m, int32_t k, vg_pq **out);
int32_t vg_pq_destroy(vg_pq *pq);
/* SetCodebooks pq.go:458-464: int8 codebooks m*k*(dim/m), scales[m], offsets[m] */
int32_t vg_pq_set_codebooks(vg_pq *pq, const int8_t *codebooks, const float *scales,
                            const float *offsets);
/* Codebooks pq.go:452-455 (copies out) */
int32_t vg_pq_get_codebooks(vg_pq *pq, int8_t *codebooks, float *scales, float *offsets);
int32_t vg_pq_is_trained(vg_pq *pq);
/* BuildDistanceTable pq.go:468-491, batched: tables[nq][m*k] (entry = a8 term,
 * internal/simd/kernels.go:354-374 generic order, no FMA) */
int32_t vg_pq_build_distance_table(vg_pq *pq, const float *queries, int64_t nq, float *tables,
                                   void *stream);

/* Train pq.go:68-143: per sub-quantizer k-means++ init + <= iters Lloyd iterations on fp32
 * sub-vectors, then int8 quantisation of the centroids (pq.go:97-136).  vectors[n*dim].
 * The reference draws from the unseeded global math/rand (pq.go:294,308,314,409), so trained
 * codebooks are not reproducible there; here every draw comes from a counter-based stream
 * keyed by `seed` (the CPU oracle uses the same stream, so GPU == oracle bit for bit).
 * The reference uses iters = 20.
 * STATED DEVIATION from the reference's arithmetic (not a restatement): the k-means++ seeding adds
 * minDistSq up — for the total and again for the pick — in blocks of 64 with a pairwise tree inside a
 * block and a sequential sum over the block totals, where pq.go:296-336 runs ONE fp32 accumulator over
 * all n values in index order.  For the same draw the picked row can therefore differ from the
 * reference's where the target lies within rounding distance of a prefix sum; the reference's own draws
 * are unseeded, so no trained codebook of it can be reproduced either way (trained
 * codebooks are quality-parity only).  The oracle implements the same blocked sum (oracle/vg_oracle.c). */
int32_t vg_pq_train(vg_pq *pq, const float *vectors, int64_t n, int32_t iters, uint64_t seed,
                    void *stream);
/* The same for the sub-quantizers [sub_begin, sub_begin + sub_count) only.  The reference trains
 * the sub-quantizers as independent goroutines (pq.go:83-138); on several GPUs each rank trains a
 * range, the ranges are exchanged (vg_pq_get_codebooks_range -> all-gather -> vg_pq_set_codebooks)
 * and, the random stream being keyed by (seed, sub-quantizer), the result equals vg_pq_train's bit
 * for bit.  The quantizer counts as trained only after the full range [0, m) was trained in one
 * call or the codebooks were set. */
int32_t vg_pq_train_subset(vg_pq *pq, const float *vectors, int64_t n, int32_t iters, uint64_t seed,
                           int32_t sub_begin, int32_t sub_count, void *stream);
/* codebooks[sub_count*k*subdim], scales[sub_count], offsets[sub_count] of that range (trained or not) */
int32_t vg_pq_get_codebooks_range(vg_pq *pq, int32_t sub_begin, int32_t sub_count, int8_t *codebooks,
                                  float *scales, float *offsets);
/* Encode pq.go:147-176, batched: codes[n*m]; code = FindNearestCentroidInt8
 * (internal/simd/kernels.go:376-396: strict '<', lowest index wins ties) */
int32_t vg_pq_encode(vg_pq *pq, const float *vectors, int64_t n, uint8_t *codes, void *stream);
/* Decode pq.go:185-229, batched: out[n*dim] = float32(code)*scale + offset */
int32_t vg_pq_decode(vg_pq *pq, const uint8_t *codes, int64_t n, float *out, void *stream);
/* ComputeAsymmetricDistance pq.go:234-260 (no LUT, terms summed sequentially over the
 * sub-quantizers), one query against n codes: out[n] */
int32_t vg_pq_asymmetric_distance_batch(vg_pq *pq, const float *query, const uint8_t *codes,
                                        int64_t n, float *out, void *stream);

/* ---- OptimizedProductQuantizer (internal/quantization/opq.go, svd.go) ------------------------------------
 * A block-diagonal rotation in front of a ProductQuantizer; blocks = vg_opq_block_size (opq.go:38-58: the
 * multiple of dim/m dividing dim nearest 32; the whole vector up to 64 dims; blocks above 64 dims are refused).
 *   vg_opq_rotate     rotateVector (:196-215): out[b*bs+i] = simd.Dot(R_b[i], v_b), dotProductAvx512 order
 *   vg_opq_encode     Encode (:218-231) = rotate + ProductQuantizer.Encode        -> codes[n*m]
 *   vg_opq_decode     Decode (:234-269) = ProductQuantizer.Decode + R^T, summed left to right
 *   vg_opq_asymmetric_distance_batch   ComputeAsymmetricDistance (:272-286): rotate the query once, then PQ's
 *   vg_opq_train      Train (:89-193): num_iterations x { rotate all, ProductQuantizer.Train(pq_iters; the
 *                     reference's 20) from scratch with the stream seed + iteration, M_b = sum_i x_b^T yhat_b in
 *                     vector order, R_b = Procrustes(M_b) by one-sided Jacobi SVD (svd.go) }.  As in the reference
 *                     the last rotations are solved after the last PQ training.  Same seeded stream as
 *                     vg_pq_train: equals the CPU restatement bit for bit; the reference itself is unseeded.
 *   vg_opq_pq         the inner ProductQuantizer (codebooks: vg_pq_get_codebooks / vg_pq_set_codebooks)
 *   vg_opq_get/set_rotations   [nblocks][block][block] fp32 row-major, host memory. */
typedef struct vg_opq vg_opq;
int32_t vg_opq_block_size(int32_t dim, int32_t m);
int32_t vg_opq_create(vg_ctx *ctx, int32_t dim, int32_t m, int32_t k, int32_t num_iterations, vg_opq **out);
int32_t vg_opq_destroy(vg_opq *opq);
vg_pq *vg_opq_pq(vg_opq *opq);
int32_t vg_opq_is_trained(vg_opq *opq);
int32_t vg_opq_get_rotations(vg_opq *opq, int32_t *block, int32_t *nblocks, float *rotations);
int32_t vg_opq_set_rotations(vg_opq *opq, const float *rotations);
int32_t vg_opq_train(vg_opq *opq, const float *vectors, int64_t n, int32_t pq_iters, uint64_t seed, void *stream);
int32_t vg_opq_rotate(vg_opq *opq, const float *vectors, int64_t n, float *out, void *stream);
int32_t vg_opq_encode(vg_opq *opq, const float *vectors, int64_t n, uint8_t *codes, void *stream);
int32_t vg_opq_decode(vg_opq *opq, const uint8_t *codes, int64_t n, float *out, void *stream);
int32_t vg_opq_asymmetric_distance_batch(vg_opq *opq, const float *query, const uint8_t *codes, int64_t n,
                                         float *out, void *stream);

/* ---- RaBitQ / binary (internal/quantization/rabitq.go, binary.go) -------------------- */
/* BytesTotal rabitq.go:187-190: ((dim+63)/64)*8 + 4 */
int64_t vg_rabitq_code_bytes(int32_t dim);
/* Encode rabitq.go:51-78, batched: codes[n*code_bytes] = sign bits (bit i -> byte i/8, bit
 * i%8; v >= 0 sets the bit) padded to 8-byte words, then the little-endian fp32 norm
 * sqrt(Dot(v,v)) (dotProductAvx512 order, float64 sqrt rounded to fp32: simd/doc.go:58-60) */
int32_t vg_rabitq_encode(vg_ctx *ctx, int32_t dim, const float *vectors, int64_t n, uint8_t *codes,
                         void *stream);
/* Distance rabitq.go:119-176 of one query against n codes (reference layout): out[n] =
 * (|q|-|y|)^2 + (4*|q|*|y|/dim)*hamming, evaluated left to right in fp32 */
int32_t vg_rabitq_distance_batch(vg_ctx *ctx, int32_t dim, const float *query, const uint8_t *codes,
                                 int64_t n, float *out, void *stream);
/* simd.Hamming kernels.go:71 (hammingAvx512, popcount_avx512.c:25-46), one nbytes-long code
 * against n contiguous codes: out[n] (exact integers) */
int32_t vg_hamming_batch(vg_ctx *ctx, const uint8_t *a, const uint8_t *codes, int64_t nbytes, int64_t n,
                         int32_t *out, void *stream);

/* Test hooks (process-wide): force an alternative path so that tests can compare the paths bit for bit.
 * Names: VG_FLAT_NO_SMALL_TILE, VG_FLAT_UNFUSED, VG_FLAT_NO_SCAN, VG_FLAT_FORCE_EXACT, VG_FLAT_NO_DMA,
 * VG_FLAT_DEBUG, VG_PROBE_NO_GROUP, VG_ADC_BIGK_EXHAUSTIVE, VG_BUILD_DEBUG.  The environment variable of the same
 * name ("1") gives the initial value, read once; the search entry points never call getenv. */
int32_t vg_debug_set_hook(const char *name, int32_t on);
/* Test entry point: a script of searcher.PriorityQueue operations (queue.go) replayed by ONE wave on the device
 * heap the graph searches use (csrc/vg_heap.hpp) — how the reference's own queue tests run against it.
 * ops[n_ops*4] = {op, node, float32 bits of distance, arg (capacity / maxSize)}; out[n_ops*3] = {flag, node, bits}:
 * flag = 1 pushed / accepted / item returned, 0 rejected / empty (VG_HEAP_LEN: the length).  is_max: NewPriorityQueue's
 * isMaxHeap; unsigned_keys != 0 selects the scalar-ALU sift-downs the walks use for distances >= +0.  final_items =
 * the heap array afterwards, item i = node | bits << 32; the script may hold at most cap (<= 8192) items. */
enum { VG_HEAP_PUSH = 0, VG_HEAP_POP = 1, VG_HEAP_PUSH_BOUNDED = 2, VG_HEAP_TRY_PUSH_BOUNDED = 3, VG_HEAP_TOP = 4,
       VG_HEAP_MIN_ITEM = 5, VG_HEAP_RESET = 6, VG_HEAP_LEN = 7 };
int32_t vg_debug_heap_replay(vg_ctx *ctx, int32_t is_max, int32_t unsigned_keys, const int32_t *ops, int32_t n_ops,
                             int32_t *out, int32_t *final_len, uint64_t *final_items, int32_t cap, void *stream);

/* ---- resident index ---------------------------------------------------------- */
int32_t vg_index_create(vg_ctx *ctx, int64_t n, int32_t dim, int32_t metric, vg_index **out);
int32_t vg_index_destroy(vg_index *idx);
/* PQ codes of a flat / DiskANN segment: n*m bytes row-major
 * (flat/segment.go:678-680 `codes[i*m:(i+1)*m]`, diskann/segment.go:316).  The
 * library keeps its own HBM copy (re-tiled for coalesced 16-byte loads); the
 * quantizer handle must outlive the index. */
int32_t vg_index_set_pq_codes(vg_index *idx, vg_pq *pq, const uint8_t *codes, void *stream);

/* RaBitQ codes of a DiskANN segment: n*code_bytes row-major (diskann/segment.go:1393-1409).
 * Re-tiled in HBM like the PQ codes (bits in 16-byte groups per 64-row tile, norms apart). */
int32_t vg_index_set_rabitq_codes(vg_index *idx, const uint8_t *codes, void *stream);

/* HNSW graph of a memtable shard (internal/hnsw): only the adjacency the search reads.
 *   l0[n*m0]            layer-0 neighbour ids in the node's stored order (node.go:62-80
 *                       GetConnectionsRaw order), VG_INVALID_ID terminates a shorter list;
 *                       m0 = 2*M <= 64 (hnsw.go:34-37 default M=32)
 *   max_level           highest level that has nodes (0 = layer 0 only)
 *   upper_slot[max_level*n]   for level L>=1: row of node in that level's table or VG_INVALID_ID
 *   upper_adj, level_rows[max_level]  tables of level_rows[L-1]*m ids each, concatenated
 *   entry_point         g.entryPointAtomic (hnsw.go:1800)
 * Needs vg_index_set_vectors (the scoring reads the fp32 rows). */
int32_t vg_index_set_hnsw_graph(vg_index *idx, int32_t m0, const uint32_t *l0, int32_t max_level,
                                int32_t m, const uint32_t *upper_slot, const uint32_t *upper_adj,
                                const int64_t *level_rows, uint32_t entry_point, void *stream);
/* HNSW construction over the index's fp32 rows: hnsw.Insert for rows 0..n-1 (hnsw.go:713-984 insert /
 * insertNode, :986-1106 selectNeighborsHeuristic, :455-555 addConnection / addConnectionPrune, :885-900
 * updateEntryPoint) with ApplyInsert's ids and levels (:629-684: id = row number, level =
 * layerForApplyInsert(id) :2103-2116, M0 = 2M :28).  Rows are inserted in id order in batches of
 * clamp(inserted / growth_div, 1, max_batch): every node of a batch searches the graph as it stood when
 * the batch began (what ApplyBatchInsert's goroutines see of one another), then the batch's links are
 * applied in id order; max_batch = 1 is the sequential Insert loop.  ef_construction = Options.EF
 * (hnsw.go:37 default 300).  The graph replaces the index's HNSW graph (vg_search_hnsw reads it).
 * 2 <= m <= 32, ef_construction <= 1024. */
int32_t vg_hnsw_build(vg_index *idx, int32_t m, int32_t ef_construction, int32_t max_batch,
                      int32_t growth_div, void *stream);
/* layerForApplyInsert (hnsw.go:2103-2116) with layerMultiplier = 1/ln(m) (hnsw.go:218) */
int32_t vg_hnsw_level_for_id(uint64_t id, int32_t m);
/* The index's HNSW graph in vg_index_set_hnsw_graph's layout.  Every output may be NULL; call once for
 * the sizes (m0, m, max_level, level_rows[max_level]), then with buffers: l0[n*m0],
 * upper_slot[max_level*n], upper_adj[sum(level_rows)*m] (host or device). */
int32_t vg_index_get_hnsw_graph(const vg_index *idx, int32_t *m0, int32_t *m, int32_t *max_level,
                                uint32_t *entry_point, int64_t *level_rows, uint32_t *l0,
                                uint32_t *upper_slot, uint32_t *upper_adj, void *stream);
/* Vamana graph of a DiskANN segment: graph[n*r] (diskann/segment.go:671-681; VG_INVALID_ID =
 * empty slot), entry point = header.Entrypoint.  Scoring uses whichever of the index's data the
 * search call names. */
int32_t vg_index_set_vamana_graph(vg_index *idx, int32_t r, const uint32_t *graph, uint32_t entry_point,
                                  void *stream);

/* fp32 rows of the segment, n*dim row-major — the layout of
 * vectorstore.ColumnarStore (internal/vectorstore/columnar.go:21-24) and of
 * flat.Segment.vectors (flat/segment.go:692).  Copied to HBM. */
int32_t vg_index_set_vectors(vg_index *idx, const float *base, void *stream);

/* ---- L0 batch kernels: the dispatch seam (internal/simd/kernels.go:11-30) ------- */
/* simd.SquaredL2Batch / simd.DotBatch (kernels.go:61-68 → batch_avx512.c:19-143):
 * one query against n contiguous targets; out[n].  Bit-identical to the AVX-512
 * kernels (4x16 FMA accumulators, 16-wide tail, reduce_add tree, FMA scalar tail). */
int32_t vg_squared_l2_batch(vg_ctx *ctx, const float *query, const float *targets, int64_t dim,
                            int64_t n, float *out, void *stream);
int32_t vg_dot_batch(vg_ctx *ctx, const float *query, const float *targets, int64_t dim,
                     int64_t n, float *out, void *stream);

/* simd.SquaredL2Bounded (kernels.go:173 -> bounded_l2_avx512.c:19-108), one query against n
 * contiguous targets with one bound each (bounds[n]) or a shared one (bounds[1], n_bounds=1).
 * (dist[i], exceeded[i]) = the pair the reference returns: the bounded kernel's reduction order run to completion
 * when the bound holds, and when it does not the PARTIAL total of the 64-float block at which the reference exits
 * (bounded_l2_avx512.c:60-75; an excess reached only in the 8-wide / scalar remainder returns the full sum). */
int32_t vg_squared_l2_bounded_batch(vg_ctx *ctx, const float *query, const float *targets, int64_t dim,
                                    int64_t n, const float *bounds, int64_t n_bounds, float *dist,
                                    int32_t *exceeded, void *stream);
/* simd.PqAdcLookup (kernels.go:56 -> pqAdcLookupAvx512, floats_avx512.c:135-167), batched:
 * one table (m*256 fp32, stride 256) against n codes of m bytes: out[n] */
int32_t vg_pq_adc_lookup_batch(vg_ctx *ctx, const float *table, const uint8_t *codes, int64_t m, int64_t n,
                               float *out, void *stream);

/* ---- internal/kmeans ------------------------------------------------------------------------- */
/* TrainKMeans kmeans.go:16-138: Lloyd on flat vectors[n*dim]; init = the first k entries of a
 * random permutation (seeded counter stream instead of the unseeded rand.Perm, kmeans.go:25),
 * assignment by SquaredL2Batch argmin (strict <) or DotBatch argmax (strict >) for Dot/Cosine,
 * update = index-ordered fp32 sums * (1/count), empty cluster -> random point, stop when no
 * assignment changed.  *produced = 0 and nothing written when n < k (the reference returns
 * (nil, nil), kmeans.go:17-20); metric Hamming -> VG_ERR_UNSUPPORTED. */
int32_t vg_kmeans_train(vg_ctx *ctx, const float *vectors, int64_t n, int32_t dim, int32_t k, int32_t metric,
                        int32_t max_iter, uint64_t seed, float *centroids, int32_t *produced, void *stream);
/* AssignPartition kmeans.go:142-196, batched: out[n] = index of the closest centroid */
int32_t vg_kmeans_assign(vg_ctx *ctx, const float *vectors, int64_t n, int32_t dim, const float *centroids,
                         int32_t k, int32_t metric, int32_t *out, void *stream);
/* FindClosestCentroids kmeans.go:217-280: the nprobe closest centroid indices for one query
 * (selection when nprobe <= k/4 && nprobe < 16, otherwise a sort by distance; ties by index).
 * Returns min(nprobe, k) indices in out; *n_out receives that count. */
int32_t vg_find_closest_centroids(vg_ctx *ctx, const float *query, const float *centroids, int32_t dim,
                                  int32_t k, int32_t nprobe, int32_t metric, int32_t *out, int32_t *n_out,
                                  void *stream);

/* ---- BinaryQuantizer (internal/quantization/binary.go) and NormalizeL2InPlace ------------------------
 * vg_binary_train   Train (:59-79): *threshold = float32(mean of every component, accumulated in float64).
 *                   The reference walks one float64 accumulator over the values in order; the device adds
 *                   float64 chunk sums, so the float32 result can differ by one ulp when the mean falls on
 *                   a rounding boundary (never observed) — the only entry point here that is not bit-exact.
 * vg_binary_encode  Encode / EncodeUint64Into (:86-154): bit i = (v[i] >= threshold), ceil(dim/64) little-
 *                   endian uint64 words per vector (threshold 0 = the sign bits RaBitQ stores).
 * vg_binary_decode  Decode (:173-188): threshold +- 0.5; code_bytes = bytes per code the caller holds
 *                   (bits beyond it read as 0).
 * vg_binary_hamming_batch   ComputeHammingDistance (:158-171) of one float query against n codes:
 *                   out[i] = popcount(encode(query) xor code_i) (HammingDistance :221-239 = simd.Hamming). */
int64_t vg_binary_code_bytes(int32_t dim);
int32_t vg_binary_train(vg_ctx *ctx, int32_t dim, const float *vectors, int64_t n, float *threshold, void *stream);
int32_t vg_binary_encode(vg_ctx *ctx, int32_t dim, float threshold, const float *vectors, int64_t n,
                         uint8_t *codes, void *stream);
int32_t vg_binary_decode(vg_ctx *ctx, int32_t dim, float threshold, const uint8_t *codes, int64_t n,
                         int32_t code_bytes, float *out, void *stream);
int32_t vg_binary_hamming_batch(vg_ctx *ctx, int32_t dim, float threshold, const float *query,
                                const uint8_t *codes, int64_t n, int32_t *out, void *stream);
/* distance.NormalizeL2InPlace (distance/distance.go:40-53) of n rows in place: norm2 = simd.Dot(v, v)
 * (dotProductAvx512 order), inv = 1 / simd.Sqrt(norm2) (float64 sqrt rounded to float32, float32 divide),
 * v[i] *= inv (scaleAvx512).  ok[row] (may be NULL) = 0 where the norm is zero (the row is left as it is —
 * the reference returns false) or dim == 0, else 1.  What cosine callers run before every insert / query
 * (engine/search.go:171-184, hnsw.go:799-818). */
int32_t vg_normalize_l2(vg_ctx *ctx, float *vectors, int64_t n, int32_t dim, uint8_t *ok, void *stream);

/* ---- searches ---------------------------------------------------------------- */
/* Segment.Rerank (flat/segment.go:754-780, diskann/segment.go:1093-1116,
 * engine/search.go:914-965): exact distance.SquaredL2 / distance.Dot
 * (squaredL2Avx512 / dotProductAvx512 order) of each query against its own nc
 * candidate rows, then the best k by (Score, RowID).  cand_ids[nq*nc] may hold
 * VG_INVALID_ID (skipped); a row listed twice is scored and reported twice, as in the reference's
 * loop.  k <= 512. */
int32_t vg_rerank(vg_index *idx, const float *queries, int64_t nq, const uint32_t *cand_ids,
                  int32_t nc, int32_t k, uint32_t *ids, float *scores, void *stream);
/* exact scores only, scores[nq*nc] in candidate order (invalid ids → +Inf / -Inf) */
int32_t vg_score_candidates(vg_index *idx, const float *queries, int64_t nq,
                            const uint32_t *cand_ids, int32_t nc, float *scores, void *stream);

/* flat.Segment.Search, fp32 branch (flat/segment.go:691-701) == exact brute force:
 * distance.SquaredL2 / distance.Dot of every row (squaredL2Avx512 / dotProductAvx512
 * order), best k by (Score, RowID) (segment.go:714-721).  (hnsw.BruteSearch, which keeps its results in a
 * PriorityQueue and reports HNSW distances, is vg_search_hnsw_brute.)  Candidates come from a batched query x base fp32 MFMA GEMM
 * (||x||^2 - 2 q.x); the survivors are re-scored in the reference's summation order and the
 * result is verified against the GEMM error bound (a query that fails the check is
 * recomputed by the exhaustive exact kernel), so ids and scores equal the reference's.
 * Metric L2 → ascending squared L2; Dot / Cosine → descending dot product
 * (distance.Provider, distance/distance.go:91-106).  k <= 512: up to k = 48 the 64 best GEMM
 * scores per query are re-scored and proved; above that every row under the query's threshold is
 * (the threshold is taken deeper in the sample for k > 64, ~3k rows pass it, sorted in LDS); batches
 * of up to 4 queries with k <= 64 take the exhaustive exact scan instead. */
int32_t vg_search_flat(vg_index *idx, const float *queries, int64_t nq, int32_t k, uint32_t *ids,
                       float *scores, void *stream);
/* Optional bfloat16 FILTER for vg_search_flat (no reference counterpart; the results stay the reference's).  on != 0:
 * the index keeps a bfloat16 copy of its fp32 rows (n*dim*2 bytes more HBM, dim rounded up to a multiple of 64 with
 * zeros; made from the rows attached by vg_index_set_vectors, to be enabled again after the rows are replaced).  Searches
 * of more than 4 queries (smaller batches take the exact scan, which has no nomination step) then NOMINATE with
 * v_mfma_f32_32x32x16_bf16 over the copies instead of the fp32 MFMA GEMM; the nominated rows are re-scored from the fp32 rows in the reference's summation order exactly as without the filter,
 * and the proof that no other row can enter the top k widens its margin by what the rounding of the copies can change
 * (<= 2^-8 * 1.02 * (|q|^2 + max|x|^2)); a query whose proof fails goes to the exhaustive exact kernel as before.
 * Ids and scores are therefore bit-identical with and without the filter.  on == 0 drops the copy. */
int32_t vg_index_enable_bf16_filter(vg_index *idx, int32_t on, void *stream);

/* diagnostics of vg_search_flat since vg_index_set_vectors: how many queries were searched and
 * how many of them were answered by the exhaustive kernel instead of GEMM candidates + proof
 * (either pointer may be NULL).  Synchronises the stream. */
int32_t vg_index_flat_stats(vg_index *idx, int64_t *queries, int64_t *exhaustive, void *stream);

/* exhaustive scan of the RaBitQ codes: RaBitQuantizer.Distance (rabitq.go:119-176) for every
 * row, best k by (Score, RowID).  The reference only scores RaBitQ codes node by node inside the
 * Vamana search (diskann/segment.go:512-535); the scan is the sharded config-5 workload
 * (SURVEY.md §8d).  k <= 512 (pages of 64 results, one scan per page). */
int32_t vg_search_rabitq(vg_index *idx, const float *queries, int64_t nq, int32_t k, uint32_t *ids,
                         float *scores, void *stream);

/* ---- SQ8 (internal/quantization/quantizer.go, internal/simd/src/sq8_avx512.c) ---------- */
/* NewScalarQuantizer quantizer.go:119-125 */
int32_t vg_sq8_create(vg_ctx *ctx, int32_t dim, vg_sq8 **out);
int32_t vg_sq8_destroy(vg_sq8 *sq);
int32_t vg_sq8_is_trained(vg_sq8 *sq);
/* Train quantizer.go:127-180: per-dimension min / max over vectors[n*dim]; a constant dimension
 * gets max = min + 1e-6; scale = 255 / range, invScale = range / 255 */
int32_t vg_sq8_train(vg_sq8 *sq, const float *vectors, int64_t n, void *stream);
/* SetBounds quantizer.go:52-78 (max - min < 1e-9 zeroes both scales) */
int32_t vg_sq8_set_bounds(vg_sq8 *sq, const float *mins, const float *maxs);
/* Mins / Maxs and the derived scales, dim floats each (any pointer may be NULL) */
int32_t vg_sq8_get_params(vg_sq8 *sq, float *mins, float *maxs, float *scales, float *inv_scales);
/* EncodeInto quantizer.go:198-222, batched: codes[n*dim] = uint8((clamp(v) - min) * scale + 0.5) */
int32_t vg_sq8_encode(vg_sq8 *sq, const float *vectors, int64_t n, uint8_t *codes, void *stream);
/* DecodeInto quantizer.go:240-250, batched: out[n*dim] = float32(code) * invScale + min */
int32_t vg_sq8_decode(vg_sq8 *sq, const uint8_t *codes, int64_t n, float *out, void *stream);
/* L2DistanceBatch quantizer.go:93-106 = simd.Sq8uL2BatchPerDimension in
 * sq8uL2BatchPerDimensionAvx512 order (sq8_avx512.c:59-103): out[n] */
int32_t vg_sq8_l2_distance_batch(vg_sq8 *sq, const float *query, const uint8_t *codes, int64_t n,
                                 float *out, void *stream);
/* codes[n*dim] in the reference's row-major layout (flat/segment.go s.codes); re-tiled on the
 * device for the scan.  The quantizer must outlive the index. */
int32_t vg_index_set_sq8_codes(vg_index *idx, vg_sq8 *sq, const uint8_t *codes, void *stream);
/* flat.Segment.Search, SQ8 branch: L2 segments score every row with L2DistanceBatch
 * (flat/segment.go:517-604), Dot / Cosine segments with ScalarQuantizer.DotProduct (:659-667,
 * quantizer.go:109-119: a sequential fp32 loop, largest first); best k by (Score, RowID).  k <= 512
 * (beyond 64 results the scan runs once per page of 64, each page after the previous one's last key). */
/* Optional bfloat16 NOMINATION for batches of vg_search_sq8 (no reference counterpart; the results stay the reference's).
 * on != 0: keeps the dequantised rows rounded to bfloat16 (rows * dim * 2 bytes, dim rounded up to a multiple of 64: twice
 * the codes) and their norms.  A batch of 5 or more queries (k <= 256) is then nominated by the bf16 MFMA GEMM of the flat
 * search, its 64 best rows per query (k > 48: every row below a sampled threshold) re-scored with L2Distance / DotProduct
 * (by the metric) from the CODES, and a bound on the nomination's error proves no other row can enter the k best; a query
 * whose proof fails is scanned as before.  Filtered batches over an unpartitioned segment (vg_search_flat_filtered,
 * VG_SCAN_SQ8) and partitions probed by 12 or more queries each (k <= 160, dim % 4 == 0) take it too.  Other shapes keep
 * the scan.  Dropped by vg_index_set_sq8_codes.  (VG_ABI_MINOR 9.) */
int32_t vg_index_enable_sq8_nomination(vg_index *idx, int32_t on, void *stream);
int32_t vg_search_sq8(vg_index *idx, const float *queries, int64_t nq, int32_t k, uint32_t *ids,
                      float *scores, void *stream);

/* ---- INT4 (internal/quantization/int4.go, internal/simd/src/int4_avx512.c) -------------- */
int32_t vg_int4_create(vg_ctx *ctx, int32_t dim, vg_int4 **out);
int32_t vg_int4_destroy(vg_int4 *iq);
int32_t vg_int4_is_trained(vg_int4 *iq);
/* Train int4.go:29-62: per-dimension min and diff = max - min (0 -> 1), then
 * simd.BuildInt4LookupTable (kernels.go:94-103) */
int32_t vg_int4_train(vg_int4 *iq, const float *vectors, int64_t n, void *stream);
/* UnmarshalBinary int4.go:190-219: min[dim], diff[dim] as stored, lookup table rebuilt */
int32_t vg_int4_set_params(vg_int4 *iq, const float *min_val, const float *diff);
/* min[dim], diff[dim], table[dim*16] (any pointer may be NULL) */
int32_t vg_int4_get_params(vg_int4 *iq, float *min_val, float *diff, float *table);
int64_t vg_int4_code_bytes(int32_t dim); /* (dim + 1) / 2 */
/* Encode int4.go:65-105, batched: two dimensions per byte, the even one in the high nibble,
 * quant = byte(math.Round(float64(clamp((v - min) / diff)) * 15)) */
int32_t vg_int4_encode(vg_int4 *iq, const float *vectors, int64_t n, uint8_t *codes, void *stream);
/* Decode int4.go:108-130, batched: float32(q)/15.0*diff + min */
int32_t vg_int4_decode(vg_int4 *iq, const uint8_t *codes, int64_t n, float *out, void *stream);
/* precomputed = 0: L2DistanceBatch (int4.go:150-164) = int4L2DistanceBatchAvx512 order
 * (int4_avx512.c:191-299); precomputed = 1: L2Distance (int4.go:133-147) =
 * int4L2DistancePrecomputedAvx512 order (:127-189) for each of the n codes.  out[n]. */
int32_t vg_int4_l2_distance_batch(vg_int4 *iq, const float *query, const uint8_t *codes, int64_t n,
                                  int32_t precomputed, float *out, void *stream);
/* INT4 codes of a DiskANN segment (diskann/segment.go:378-416), n * ceil(dim/2) bytes; scored by
 * vg_search_vamana kind 3.  The quantizer must outlive the index. */
int32_t vg_index_set_int4_codes(vg_index *idx, vg_int4 *iq, const uint8_t *codes, void *stream);

/* ---- graph construction: neighbour selection ---------------------- */
/* Vamana robustPrune (diskann/writer.go:571-625) for n_nodes nodes at once.  cands[n_nodes*nc]
 * holds, per node, the search results and the node's current neighbours (what the reference's
 * `unique` map holds); VG_INVALID_ID entries, duplicates and the node itself are dropped.
 * Candidates are visited by ascending (distance to the node, id) — the reference's order among
 * equal distances is unspecified (map iteration + unstable sort) — and kept unless
 * alpha * dist(candidate, kept) < dist(candidate, node) for some already kept one.
 * out[n_nodes*r] (VG_INVALID_ID padded), counts[n_nodes].  nc <= 1024, r <= 256. */
int32_t vg_robust_prune(vg_index *idx, const uint32_t *nodes, int64_t n_nodes, const uint32_t *cands,
                        int32_t nc, int32_t r, float alpha, uint32_t *out, int32_t *counts, void *stream);
/* HNSW selectNeighborsHeuristic (hnsw.go:1009-1106: applyHeuristic + fillUpNeighbors; all the
 * candidates when there are at most m).  cand_ids / cand_dists[n_nodes*nc]: nearest first, with the
 * distance to the source as the caller's queue held it; lists may end in VG_INVALID_ID.
 * out[n_nodes*m], counts[n_nodes].  nc <= 1024, m <= 256. */
int32_t vg_hnsw_select_neighbors(vg_index *idx, int64_t n_nodes, const uint32_t *cand_ids,
                                 const float *cand_dists, int32_t nc, int32_t m, uint32_t *out,
                                 int32_t *counts, void *stream);

/* ---- IVF partitions of a flat segment (flat/segment.go:187-207, :727-749) ------------------- */
/* centroids[num_partitions*dim] and part_offsets[num_partitions+1] (first row of every partition,
 * non-decreasing, last <= rows) as the flat writer lays them out (rows grouped by partition).
 * num_partitions == 0 removes them. */
int32_t vg_index_set_partitions(vg_index *idx, const float *centroids, const uint32_t *part_offsets,
                                int32_t num_partitions, void *stream);
enum { VG_SCAN_F32 = 0, VG_SCAN_PQ = 1, VG_SCAN_SQ8 = 2 };
/* The partition-probed scan of flat.Segment.Search: per query kmeans.FindClosestCentroids(nprobes)
 * (kmeans.go:217-280; nprobes <= 0 means 1), then the chosen scan over those partitions' row ranges
 * only, one top-k by (score, row id).  With at most one partition it is vg_search_flat /
 * vg_search_pq_adc / vg_search_sq8.  k <= 512 (pages of 64 results), nprobes <= 64.  Equal centroid distances: the
 * reference's selection loop (kmeans.go:255-269, taken for nprobes <= partitions / 4 && nprobes < 16) is replayed, so the
 * probed partitions are the reference's also among duplicated centroids; its full sort leaves ties unpinned (by id here). */
int32_t vg_search_flat_probed(vg_index *idx, const float *queries, int64_t nq, int32_t k, int32_t nprobes,
                              int32_t scan, uint32_t *ids, float *scores, void *stream);
/* flat.Segment.Search with `filter segment.Filter` set (flat/segment.go:447): a row whose filter.Matches(rowID) is false
 * is skipped (fp32 / PQ rows before they are scored, :631-635; SQ8 L2 rows after their batch was scored, :559-561) — the
 * k best (score, row id) keys of the matching rows of the probed partitions, or of the whole segment when it has at most
 * one partition.  mask: bit i of byte i/8 set = filter.Matches(i) (tombstones are the caller's: clear their bits); query q
 * reads mask + q * mask_stride (0 = one mask for the batch, else >= ceil(rows/8)); NULL = vg_search_flat_probed.  Block
 * skipping by field statistics (MatchesBlock, :614-628) only ever skips rows no filter bit is set for.  k <= 512,
 * nprobes <= 64.  (VG_ABI_MINOR 4.) */
int32_t vg_search_flat_filtered(vg_index *idx, const float *queries, int64_t nq, int32_t k, int32_t nprobes,
                                int32_t scan, const uint8_t *mask, int64_t mask_stride, uint32_t *ids, float *scores,
                                void *stream);

/* ---- on-disk segment images --------------------------------------- */
enum { VG_QUANT_NONE = 0, VG_QUANT_PQ = 1, VG_QUANT_SQ8 = 3, VG_QUANT_RABITQ = 5, VG_QUANT_INT4 = 6 }; /* quantization.Type, types.go:6-14 */
typedef struct vg_segment_info {
    uint64_t segment_id;
    int64_t rows;
    int32_t dim, metric;
    int32_t kind;          /* 0 = flat segment, 1 = DiskANN segment */
    int32_t quantization;  /* VG_QUANT_* */
    int32_t pq_m, pq_k;
    int32_t max_degree, search_list_size; /* DiskANN R and L (diskann/format.go:26-27) */
    uint32_t entrypoint;
    int32_t num_partitions; /* flat segment: IVF partitions (flat/format.go:37) */
} vg_segment_info;
/* flat.Open (flat/segment.go:105-300) over a whole segment file held in host memory (mmap or
 * read): header (flat/format.go:11-165), optional CRC32C of the body, SQ8 bounds / PQ codebooks,
 * codes and fp32 rows are uploaded as they lie in the file.  The result owns a vg_index (and the
 * quantizer the file describes); search it with vg_search_flat / vg_search_sq8 / vg_search_pq_adc
 * / vg_rerank.  Errors carry the reference's messages ("invalid magic number", "unsupported
 * version", "file too short for vectors", "checksum mismatch: ..."). */
int32_t vg_segment_open_flat(vg_ctx *ctx, const void *image, int64_t size, int32_t verify_checksum,
                             vg_segment **out, void *stream);
/* flat.Segment.Search (flat/segment.go:447-751, filter == nil) for a batch of queries: the scan type
 * follows the segment (SQ8 codes: L2Distance / DotProduct by metric; PQ: table lookups, L2 segments
 * only — see vg_segment.hip; else fp32 rows) and, when the segment has more than one IVF partition,
 * only the nprobes closest partitions are scanned (:727-744; nprobes <= 0 means 1).  k <= 512 and
 * nprobes <= 64 on the partitioned path.
 * On a DiskANN segment: diskann.Segment.Search (diskann/segment.go:487-706) = vg_search_vamana with
 * the distFn of the segment's quantization (RaBitQ, else PQ, else INT4, else fp32); nprobes is unused,
 * like the search-list size the reference computes and never reads. */
int32_t vg_segment_search(vg_segment *seg, const float *queries, int64_t nq, int32_t k, int32_t nprobes,
                          uint32_t *ids, float *scores, void *stream);
/* Segment.Search with `filter` set, by what the file holds: vg_search_flat_filtered (flat/segment.go:631-635) or
 * vg_search_vamana_filtered (diskann/segment.go:616-627).  mask: bit i of byte i/8 = filter.Matches(i), query q's at
 * mask + q * mask_stride (0 = one for the batch); NULL = vg_segment_search.  (VG_ABI_MINOR 8.) */
int32_t vg_segment_search_filtered(vg_segment *seg, const float *queries, int64_t nq, int32_t k, int32_t nprobes,
                                   const uint8_t *mask, int64_t mask_stride, uint32_t *ids, float *scores, void *stream);
/* diskann segment (diskann/format.go:8-119, segment.go:165-440,1393-1408): fp32 rows, the
 * N x R uint32 graph and entry point, PQ codebooks + codes or RaBitQ codes; search with
 * or INT4 parameters + codes; search with vg_search_vamana (kind 0 / 1 / 2 / 3).  The header's
 * CompressionType byte is ignored exactly as segment.go ignores it: the reference writer records
 * LZ4 there by default (writer.go:92) while writing every section raw (writer.go:697-740). */
int32_t vg_segment_open_diskann(vg_ctx *ctx, const void *image, int64_t size, int32_t verify_checksum,
                                vg_segment **out, void *stream);
int32_t vg_segment_get_info(vg_segment *seg, vg_segment_info *info);
vg_index *vg_segment_index(vg_segment *seg); /* borrowed: valid until vg_segment_close */
vg_pq *vg_segment_pq(vg_segment *seg);       /* NULL unless the segment is PQ-quantized */
vg_sq8 *vg_segment_sq8(vg_segment *seg);     /* NULL unless the segment is SQ8-quantized */
vg_int4 *vg_segment_int4(vg_segment *seg);   /* NULL unless the segment is INT4-quantized */
int32_t vg_segment_close(vg_segment *seg);
/* hash.CRC32C (internal/hash/crc32c.go:15-17) */
uint32_t vg_crc32c(const void *data, int64_t size);

/* per-query counters, the reference's FilterGateStats (searcher/searcher.go:114-137).  vg_search_vamana has
 * no short-circuit path; it reports in distance_short_circuits the candidates it could NOT push because the
 * per-query exploration heap (min(rows, 65536) entries; the reference's is unbounded) was full — 0 in every
 * search that followed the reference exactly. */
typedef struct vg_search_stats {
    int64_t nodes_visited, distance_computations, distance_short_circuits, pops;
    /* not a FilterGateStats field: rows scored by greedySearch on the way down through the upper layers
     * (hnsw.go:1897-1934 does not count them) — part of a query's gathered bytes all the same */
    int64_t descent_distance_computations;
} vg_search_stats;

/* hnsw.KNNSearch (hnsw.go:1650-1755): greedySearch through the upper layers (:1897-1934), then
 * searchLayerUnfiltered on layer 0 (:1220-1396) with the reference's exact heap semantics
 * (searcher/queue.go 4-ary heaps: bounded result heap of ef, exploration heap capped at 2*ef with
 * the adaptive shrink, strict comparisons), distance.SquaredL2Bounded short-circuit
 * (bounded_l2_avx512.c order) once ef results exist, distFunc otherwise (L2 / -Dot / 0.5*L2:
 * vectorstore/columnar.go:29-50).  One wavefront per query, many queries in flight; per query
 * the result equals the sequential reference's.  ids/scores[nq*k] best first; stats[nq] may be
 * NULL.  k <= ef (a smaller ef is raised to k, determineEF hnsw.go:1891-1894); the two heaps of a query live
 * in LDS up to ef = 512 and in HBM scratch beyond.  NaN distances (a NaN or an Inf in a row or in a query): every
 * comparison with a NaN is false in the reference (queue.go:75-82,199-203, hnsw.go:1376), and the result is the
 * reference's here too — a walk that scores a NaN stops and the query is answered by a second pass that compares
 * floats exactly as the reference's loop is written (r04 and before: unspecified, and an Inf in a query could fault). */
int32_t vg_search_hnsw(vg_index *idx, const float *queries, int64_t nq, int32_t k, int32_t ef,
                       uint32_t *ids, float *scores, vg_search_stats *stats, void *stream);
/* searchExecute with a filter and a selectivity hint above highSelectivityThreshold = 0.3 (hnsw.go:1107-1146, :1791-1835):
 * searchLayerWithPostFilter (hnsw.go:1159-1218) — the walk above with ef expanded to ef * (1 + (1 - selectivity) / 2)
 * (at most 2 ef, at most 500), every result then popped worst first, the rows whose mask bit is set kept in that order and
 * pushed back capped at ef — and knnSearchInternal's extraction.  mask: bit i of byte i/8 set = row i passes
 * (filter.Matches and not tombstoned); query q reads mask + q * mask_stride (0 = one mask for the batch, else >=
 * ceil(n/8)).  `ef` is what determineEF returned (its bitmap-cardinality expansion, hnsw.go:1863-1889, is the caller's).
 * selectivity <= 0.3 (0 = no hint): the reference walks predicate-aware there — vg_search_hnsw_predicate below with no
 * tombstones (VG_ABI_MINOR 6; before it this returned VG_ERR_UNSUPPORTED).  (VG_ABI_MINOR 3.) */
int32_t vg_search_hnsw_filtered(vg_index *idx, const float *queries, int64_t nq, int32_t k, int32_t ef,
                                const uint8_t *mask, int64_t mask_stride, double selectivity, uint32_t *ids,
                                float *scores, vg_search_stats *stats, void *stream);
/* searchExecute with a filter and a selectivity hint at or below 0.3, or none: searchLayerPredicateAware (hnsw.go:1406-1558) on
 * layer 0 — per neighbour, in list order: filter.Matches and the tombstone bit BEFORE any distance; a passing live node is
 * scored; a rejected one is navigated by its cached edge distance (Neighbor.Dist > 0, node.go:62-80) while the results hold
 * fewer than ef/2 items, scored while they hold fewer than ef unless more than 10 rejections came in a row or its edge
 * distance exceeds 1.5 x the worst result, and skipped once they hold ef; the navigation queue is unbounded, the results are
 * bounded by ef and take passing live nodes only; then knnSearchInternal's extraction.  mask as above (filter.Matches only);
 * deleted: the tombstone bitmap, one for the batch, NULL = none.  stats: nodes_visited, distance_computations,
 * distance_short_circuits = ExpansionsSkipped, pops.  ef <= 4096.  fp32 rows.  Needs the layer-0 edge distances:
 * vg_index_set_hnsw_edge_distances, else they are recomputed from the rows at the first call.  (VG_ABI_MINOR 6.) */
int32_t vg_search_hnsw_predicate(vg_index *idx, const float *queries, int64_t nq, int32_t k, int32_t ef,
                                 const uint8_t *mask, int64_t mask_stride, const uint8_t *deleted, uint32_t *ids,
                                 float *scores, vg_search_stats *stats, void *stream);
/* The cached Neighbor.Dist of the layer-0 lists, l0_dist[n*m0] slot for slot with vg_index_set_hnsw_graph's l0 (the low 32
 * bits of the reference's neighbour words, node.go:67-80); NULL = recompute them from the fp32 rows as the distance between
 * the two nodes, which is what the insert stored (hnsw.go:516, :550, :964).  Dropped when the graph is replaced. */
int32_t vg_index_set_hnsw_edge_distances(vg_index *idx, const float *l0_dist, void *stream);
/* g.tombstones (hnsw.go:95, Delete :1601-1617) as a bitmap: bit i of byte i/8 set = node i deleted, ceil(n/8) bytes; NULL = no
 * deleted node.  A deleted node is walked through but never enters the results, so it never moves the bound either
 * (searchLayerUnfiltered hnsw.go:1381-1390, processEntryPointUnfiltered :1559-1565, the post-filter's re-filter :1198,
 * searchLayerPredicateAware :1485, :1580).  Read by vg_search_hnsw, _hnsw_pq, _hnsw_filtered and — when its own `deleted`
 * argument is NULL — vg_search_hnsw_predicate; while it is set the unfiltered / post-filter walks run as the pass that
 * compares as the reference writes it (the one that answers NaN distances), somewhat slower than the tuned walk.
 * vg_search_hnsw_brute takes the tombstones inside its mask, as before.  (VG_ABI_MINOR 7.) */
int32_t vg_index_set_hnsw_tombstones(vg_index *idx, const uint8_t *deleted, void *stream);
/* The same walk scored from the nodes' PQ codes instead of their fp32 rows: distFunc =
 * pq.ComputeAsymmetricDistance (pq.go:234-260), the way the reference scores graph nodes from PQ codes
 * (diskann/segment.go:536-557); no SquaredL2Bounded short-circuit (that kernel reads fp32 rows).  scores =
 * the PQ distances.  Candidate stage of graph -> PQ -> exact rerank (engine/search.go:914-965): ask for
 * k = ef candidates, then vg_rerank.  Needs vg_index_set_pq_codes (numCentroids 256) and metric L2. */
int32_t vg_search_hnsw_pq(vg_index *idx, const float *queries, int64_t nq, int32_t k, int32_t ef,
                          uint32_t *ids, float *scores, vg_search_stats *stats, void *stream);

/* diskann.Segment.searchInternal (diskann/segment.go:503-706), filters nil.  kind selects the
 * distFn: 0 = fp32 rows (distance.Provider(metric), :582-588), 1 = PQ
 * ComputeAsymmetricDistance (:536-541, terms summed sequentially over the sub-quantizers),
 * 2 = RaBitQ Distance (:512-519), 3 = INT4 L2Distance (:558-565).  Unbounded exploration
 * min-heap, top-k CandidateHeap,
 * stop when the popped candidate is worse than the k-th result.  k <= 512. */
int32_t vg_search_vamana(vg_index *idx, const float *queries, int64_t nq, int32_t k, int32_t kind,
                         uint32_t *ids, float *scores, vg_search_stats *stats, void *stream);
/* The same with `filter segment.Filter` set: pushToHeap (diskann/segment.go:616-627) returns before TryPushBounded for a row
 * whose filter.Matches is false — the traversal queue still takes the node, and the stop test reads the heap of MATCHING rows,
 * so a selective filter walks further before it stops.  mask: bit i of byte i/8 set = filter.Matches(i) (and the metadata
 * filter, :620-622, if any — the host ANDs them); query q reads mask + q * mask_stride (0 = one mask for the batch, else >=
 * ceil(rows/8)); NULL = vg_search_vamana.  (VG_ABI_MINOR 5.) */
int32_t vg_search_vamana_filtered(vg_index *idx, const float *queries, int64_t nq, int32_t k, int32_t kind,
                                  const uint8_t *mask, int64_t mask_stride, uint32_t *ids, float *scores,
                                  vg_search_stats *stats, void *stream);

/* The HNSW index's two EXHAUSTIVE paths, each with the heap it is written with (searcher/queue.go) — which ids
 * survive a tie at the k-th distance and the order of equal distances in the result follow from the heap's layout,
 * so the heap is replayed operation by operation (vg_search_flat / vg_rerank order by (Score, RowID) instead, the
 * CandidateHeap order of flat/segment.go:714-721, and report dot products; this entry reports HNSW distances):
 *   VG_BRUTE_SCAN    hnsw.BruteSearch + scanSegment (hnsw.go:2021-2101): PriorityQueue(max); len < k -> PushItem,
 *                    else `d < top.Distance` -> PopItem + PushItem; res[len-1 .. 0] = PopItem().
 *   VG_BRUTE_BITMAP  searchBitmap (hnsw.go:2240-2263) + knnSearchInternal's extraction (:1732-1751):
 *                    s.Candidates.TryPushBounded(item, k) (at capacity `d >= top` is rejected, else the item replaces
 *                    the root and sifts down); popped, reversed.
 * Rows are visited in ascending id (node segments in order; bm.ForEach ascending, segment/segment.go:154).  mask: bit i
 * of byte i/8 set = row i takes part — for SCAN the rows that are live nodes and pass `filter`, for BITMAP the
 * bitmap minus the tombstones; NULL = every row.  Query q reads mask + q*mask_stride (mask_stride 0 = one mask for
 * the batch, else >= ceil(n/8)).  Distances as the index wraps them (hnsw.go:2218-2238, columnar.go:37-44): L2 ->
 * squared L2, Dot -> -dot, Cosine -> 0.5 * squared L2 (rows and queries normalised by the caller, as for
 * vg_search_hnsw).  ids/scores[nq*k] best first; unused slots VG_INVALID_ID / +Inf.  k <= 1024.  NaN distances (a NaN
 * or Inf in a row or query; dot products overflowing both ways) are answered as the reference's loops answer them — `d <
 * top.Distance` never admits a NaN once the heap is full, `d >= top.Distance` never rejects one — by a pass that decides every
 * row against the live top (r05 and before: unspecified; see "NaN scores" below). */
enum { VG_BRUTE_SCAN = 0, VG_BRUTE_BITMAP = 1 };
int32_t vg_search_hnsw_brute(vg_index *idx, const float *queries, int64_t nq, int32_t k, int32_t mode,
                             const uint8_t *mask, int64_t mask_stride, uint32_t *ids, float *scores, void *stream);

/* engine fan-in (engine/search.go:904-908: per-segment candidate lists merged into one
 * bounded heap, ordered by searcher/candidate_queue.go:12-23).  Here the "segments" are row
 * shards, one per GPU: lists[l] holds nq*k (id, score) results of shard l, ids local to the
 * shard; id_offsets[l] (may be NULL = all 0) is added to make them global.  Output: the k best
 * of the union per query, best first.  metric picks the direction (L2 ascending, Dot/Cosine
 * descending).  ids_in/scores_in are [lists][nq][k] contiguous. */
int32_t vg_merge_topk(vg_ctx *ctx, const uint32_t *ids_in, const float *scores_in, int32_t lists,
                      int64_t nq, int32_t k, int32_t metric, const uint32_t *id_offsets,
                      uint32_t *ids, float *scores, void *stream);
/* The same merge over the all-gathered image of vg_comm_all_gather_topk: packed[lists][2][nq][k] uint32 in
 * device memory, [l][0] = ids of list l, [l][1] = the bit patterns of its fp32 scores. */
int32_t vg_merge_topk_packed(vg_ctx *ctx, const uint32_t *packed, int32_t lists, int64_t nq, int32_t k,
                             int32_t metric, const uint32_t *id_offsets, uint32_t *ids, float *scores,
                             void *stream);

/* ---- multi-GPU: the one exchange step of a row-sharded search --------------------------------------
 * One process (or OS thread) per GPU, each with its own vg_ctx and its shard of the rows resident
 * (rows are independent, ids are global row numbers).  The reference merges per-segment
 * candidate lists into one bounded heap in-process (engine/search.go:835-908); here the lists cross GPUs:
 * ONE ncclAllGather (RCCL over xGMI) of nq*k*8 bytes per rank on the caller's stream, then vg_merge_topk
 * with the reference's tie-break on every rank.  RCCL is dlopen'ed on first use.
 *   vg_comm_unique_id   rank 0 fills id[VG_COMM_ID_BYTES] (ncclGetUniqueId); the host hands the bytes to
 *                       the other ranks by whatever channel it has (Go: a pipe, a file, its RPC layer)
 *   vg_comm_create      collective: every rank calls it with the same id (ncclCommInitRank)
 *   vg_comm_all_gather_topk   local_ids / local_scores [nq*k] = this rank's vg_search_* output (local row
 *                       numbers), id_offsets[world] = first global row of every shard; ids / scores [nq*k]
 *                       = the merged global top-k, identical on every rank
 *   vg_comm_all_gather  raw bytes (device buffers): e.g. the codebooks of a PQ trained by sub-quantizer
 *                       ranges (vg_pq_train_subset + vg_pq_get_codebooks_range) */
#define VG_COMM_ID_BYTES 128
typedef struct vg_comm vg_comm;
int32_t vg_comm_unique_id(uint8_t *id);
int32_t vg_comm_create(vg_ctx *ctx, int32_t world, int32_t rank, const uint8_t *id, vg_comm **out);
int32_t vg_comm_destroy(vg_comm *comm);
int32_t vg_comm_info(const vg_comm *comm, int32_t *world, int32_t *rank);
/* vg_comm_probe: load RCCL without creating anything (NOT a collective): every rank calls it and the host agrees on
 * the outcome BEFORE any rank enters vg_comm_create, which is a collective (ncclCommInitRank) — a rank that cannot
 * load RCCL would otherwise leave its peers waiting inside it.  An RCCL that is already mapped into the process (e.g.
 * PyTorch's torch/lib/librccl.so) is reused, never a second copy; rccl_path receives the file the symbols came from.
 * vg_comm_describe: what RCCL itself says about the communicator (ncclCommCount / ncclCommUserRank /
 * ncclCommCuDevice), for the run's record: a line that claims n_gpus = N should carry rccl_ranks = N. */
int32_t vg_comm_probe(char *rccl_path, int32_t len);
int32_t vg_comm_describe(const vg_comm *comm, int32_t *rccl_ranks, int32_t *rccl_rank, int32_t *rccl_device,
                         int32_t *reused_mapped_rccl, char *rccl_path, int32_t len);
int32_t vg_comm_all_gather(vg_comm *comm, const void *send, void *recv, int64_t bytes_per_rank, void *stream);
int32_t vg_comm_all_gather_topk(vg_comm *comm, const uint32_t *local_ids, const float *local_scores,
                                int64_t nq, int32_t k, int32_t metric, const uint32_t *id_offsets,
                                uint32_t *ids, float *scores, void *stream);

/* NaN scores in the exhaustive searches and the beam search (vg_search_flat, vg_search_pq_adc, vg_search_sq8, vg_search_rabitq,
 * vg_search_flat_probed / _filtered, vg_search_hnsw_brute, vg_search_vamana / _filtered; VG_ABI_MINOR 10).  The scans keep their best k by a 64-bit key (score
 * bits, row id) — a total order, which is what the reference's heaps implement while no score is a NaN.  For a NaN every
 * comparison of candidate_queue.go:12-38 / queue.go:75-82,199-203 is false: a NaN that enters while the heap fills stays, at
 * the root it is never replaced (rows better than everything kept are turned away), as a first child it stops a sift.  That
 * outcome is DEFINED — the loops are sequential — and these entry points return it: a query whose INPUTS could produce a NaN
 * or an infinite score (a non-finite query value; non-finite rows, quantizer parameters or stored norms; magnitudes whose
 * partial sums can overflow: dim * max|q| * max|x| >= 1e38 for dot products, dim * (max|q| + max|x|)^2 >= 1e38 for squared
 * distances; RaBitQ: 4 |q| |y| >= 1e38 — +Inf scores are ties the heap breaks by row id inside its history, which the scans'
 * `score < bound` pre-tests do not reproduce while the bound is +Inf) is answered a second time by the reference's
 * heap replayed operation by operation with float comparisons, rows in the reference's order, and overwrites the first answer.
 * ids / scores then hold what the engine takes out of the heap — Pop() until empty (engine/search.go:859-862) — best first; a
 * NaN score's sign and payload are the instruction set's, not the algorithm's.  Rare by construction (such inputs are garbage)
 * and slow by design: one workgroup walks all rows per query — measured at 1M x 768 (tools/nan_replay_time.py): a call of 1024
 * queries takes 11.6 ms with none at risk, 90 ms with one, 101 ms with 64, 698 ms with all of them; an index holding a non-finite
 * row sends EVERY query there.  Every other query pays one extra kernel launch per call that returns at once (the probed searches: two —
 * the probe lists are selected again for the replay).  vg_search_flat_probed / vg_search_flat_filtered (and vg_segment_search
 * through them) take part: the rows a query's filter lets through, the probed partitions' ranges in FindClosestCentroids'
 * order — its selection loop and, up to 12 partitions, its full sort (Go's insertion sort, where a NaN distance compares equal
 * to everything) are replayed; with more than 12 partitions AND NaN centroid distances the full sort is pdqsort proper, whose
 * order is not restated (NaN distances sort last).  vg_merge_topk / _packed (and vg_comm_all_gather_topk through them) replay
 * the engine's fan-in (engine/search.go:904-918: every list from its last valid entry to its first into one heap with
 * TryPushBounded, then popped) for a query whose lists hold a NaN.  Not covered: vg_rerank — the order in which the engine hands
 * it the candidates comes out of an unstable sort (search.go:921), so there is no defined outcome to reproduce; NaN scores
 * order as the largest keys there. */

/* flat.Segment.Search, PQ branch (flat/segment.go:476-483 LUT, :678-689 ADC
 * = simd.PqAdcLookup in pqAdcLookupAvx512 order, :714-721 top-k with the
 * (Score, RowID) tie-break of searcher/candidate_queue.go:12-23).
 * queries[nq*dim] → ids[nq*k], scores[nq*k], best first.  k <= 1024. */
int32_t vg_search_pq_adc(vg_index *idx, const float *queries, int64_t nq, int32_t k,
                         uint32_t *ids, float *scores, void *stream);
/* Optional bfloat16 NOMINATION for batches of vg_search_pq_adc (no reference counterpart; the results stay the reference's).
 * A row's table sum is the squared distance to its DECODED vector (ProductQuantizer.Decode, pq.go:185-229: the table's
 * entries, pq.go:468-491, are computed from the same fp32 centroid values).  on != 0 keeps the decoded rows rounded to
 * bfloat16 (rows * dim * 2 bytes, dim rounded up to a multiple of 64 — 2 * dim / m times the codes: 16x at 8 dimensions per
 * sub-quantizer) and their norms.  A batch of queries x rows >= 24M (k <= 256, numCentroids == 256) is then nominated by the
 * bf16 MFMA GEMM of the flat search, its 64 best rows per query (k > 48: every row below a sampled threshold) re-scored
 * from the CODES against the query's BuildDistanceTable in pqAdcLookupAvx512 order, and a bound on the nomination's error
 * proves no other row can enter the k best; a query whose proof fails is scanned as before.  Smaller batches and the
 * filtered / probed searches keep the table scan.  Dropped by vg_index_set_pq_codes.  (VG_ABI_MINOR 10.) */
int32_t vg_index_enable_pq_nomination(vg_index *idx, int32_t on, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* VECGO_HIP_H */
