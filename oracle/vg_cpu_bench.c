/*
 * vg_cpu_bench.c — the timed CPU baseline of bench.py (TEST / MEASUREMENT INFRASTRUCTURE ONLY, see
 * vg_oracle.h): the reference's CPU algorithm on the host's cores with the reference's concurrency model,
 * one query per thread (one goroutine per query, benchmark_test FastConcurrentSearch), nothing but C
 * inside the timed region.  The loops are the oracle's restatements (flat/segment.go scan loops,
 * hnsw.go searchExecute, diskann/segment.go searchInternal); when oracle/_ref is present the caller installs
 * the reference's own compiled AVX-512 kernels through vgo_set_kernel_hooks, so the arithmetic runs at the
 * reference's speed ("kind": "reference"), otherwise the scalar restatement runs ("kind": "port").
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

#include "vg_oracle.h"

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ---- NUMA-interleaved copy of a corpus ---------------------------------------------------------------
 * A corpus that one thread first-touched sits on one socket's DRAM; the scan baseline would then measure
 * that socket's memory controller.  mbind(MPOL_INTERLEAVE) over the online nodes before the pages are
 * touched (raw syscall: no libnuma in the image); failure is not an error, the copy is then ordinary. */
typedef struct {
    char *dst;
    const char *src;
    size_t bytes;
} copy_job;
static void *copy_worker(void *p)
{
    copy_job *j = (copy_job *)p;
    memcpy(j->dst, j->src, j->bytes);
    return NULL;
}

static int online_nodes(void)
{
    FILE *f = fopen("/sys/devices/system/node/online", "r");
    if (!f) return 1;
    int lo = 0, hi = 0, n = fscanf(f, "%d-%d", &lo, &hi);
    fclose(f);
    if (n == 2) return hi + 1;
    return 1;
}

void *vgo_bench_interleaved_copy(const void *src, size_t bytes, int32_t *interleaved)
{
    size_t len = (bytes + 4095) & ~(size_t)4095;
    void *p = mmap(NULL, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (p == MAP_FAILED) return NULL;
    int nodes = online_nodes();
    int ok = 0;
    if (nodes > 1 && nodes <= 64) {
        unsigned long mask = nodes >= 64 ? ~0ul : ((1ul << nodes) - 1);
#ifdef SYS_mbind
        ok = syscall(SYS_mbind, p, len, 3 /* MPOL_INTERLEAVE */, &mask, (unsigned long)(nodes + 1), 0u) == 0;
#endif
    }
    if (interleaved) *interleaved = ok ? nodes : 0;
    enum { T = 16 };
    pthread_t th[T];
    copy_job jobs[T];
    size_t per = ((bytes / T) + 4095) & ~(size_t)4095;
    int used = 0;
    for (int t = 0; t < T; t++) {
        size_t off = (size_t)t * per;
        if (off >= bytes) break;
        jobs[t].dst = (char *)p + off;
        jobs[t].src = (const char *)src + off;
        jobs[t].bytes = off + per <= bytes ? per : bytes - off;
        pthread_create(&th[t], NULL, copy_worker, &jobs[t]);
        used++;
    }
    for (int t = 0; t < used; t++) pthread_join(th[t], NULL);
    return p;
}

void vgo_bench_free(void *p, size_t bytes)
{
    if (p) munmap(p, (bytes + 4095) & ~(size_t)4095);
}

/* ---- one query per thread until the deadline ------------------------------------------------------- */
enum { VGO_BENCH_FLAT = 0, VGO_BENCH_HNSW = 1, VGO_BENCH_ADC = 2, VGO_BENCH_RABITQ = 3, VGO_BENCH_VAMANA = 4, VGO_BENCH_SQ8 = 5,
       VGO_BENCH_HNSW_PQ_RERANK = 6 };

typedef struct {
    int32_t kind;
    /* corpus */
    const float *base;
    int64_t n;
    int32_t dim, metric;
    const vgo_hnsw_graph *hnsw;
    const vgo_vamana *vamana;
    const vgo_pq *pq;
    const uint8_t *codes;
    /* queries */
    const float *queries;
    int64_t nq;
    int32_t k, ef;
    /* outputs of the first pass over the queries (may be NULL): ids[nq*k] */
    uint32_t *ids;
    int64_t *dist_comps; /* [nq] HNSW / Vamana: DistanceComputations */
    float *scores;       /* [nq*k] next to ids (may be NULL) */
    const float *sq_mins, *sq_inv_scales; /* SQ8 scan */
} vgo_bench_job;

typedef struct {
    const vgo_bench_job *job;
    int tid, nthreads;
    double deadline;
    int64_t done;
} worker;

/* per-thread scratch of the rerank pipeline: allocated once per worker, not per query */
typedef struct {
    uint32_t *cand;
    float *cs;
    vgo_cand *sorted;
    int32_t cap;
} rerank_scratch;

static void run_one(const vgo_bench_job *j, int64_t qi, uint32_t *ids, float *scores, uint32_t *visited, uint32_t epoch,
                    rerank_scratch *rs)
{
    const float *q = j->queries + qi * j->dim;
    vgo_search_stats st = {0, 0, 0, 0};
    int32_t r = 0;
    switch (j->kind) {
    case VGO_BENCH_FLAT:
        r = vgo_flat_search_f32(j->base, j->n, j->dim, j->metric, q, j->k, ids, scores);
        break;
    case VGO_BENCH_HNSW:
        r = vgo_hnsw_search_ws(j->hnsw, q, j->k, j->ef, ids, scores, &st, visited, epoch);
        break;
    case VGO_BENCH_HNSW_PQ_RERANK: {
        /* the metric's named pipeline: the graph walked on PQ codes (hnsw graph with ->pq set: distFunc =
         * ComputeAsymmetricDistance, pq.go:234-260) for ef candidates, then Segment.Rerank — the exact distance of
         * every candidate (flat/segment.go:766-772) — and the best k by (Score, RowID) (engine/search.go:914-965) */
        const int32_t ef = j->ef > j->k ? j->ef : j->k;
        uint32_t *cand = rs->cand;
        float *cs = rs->cs;
        const int32_t nc = vgo_hnsw_search_ws(j->hnsw, q, ef, ef, cand, cs, &st, visited, epoch);
        vgo_rerank_f32(j->hnsw->base, j->hnsw->dim, j->hnsw->metric, q, cand, nc, cs);
        vgo_candheap h;
        vgo_candheap_init(&h, j->k > 0 ? j->k : 1, j->hnsw->metric != VGO_METRIC_L2);
        for (int32_t c = 0; c < nc; c++) vgo_candheap_try_push_bounded(&h, (vgo_cand){0, cand[c], cs[c]}, j->k);
        vgo_cand *sorted = rs->sorted;
        r = vgo_candheap_sorted(&h, sorted);
        for (int32_t c = 0; c < r; c++) {
            ids[c] = sorted[c].row_id;
            scores[c] = sorted[c].score;
        }
        vgo_candheap_free(&h);
        break;
    }
    case VGO_BENCH_ADC:
        r = vgo_flat_search_pq(j->pq, j->codes, j->n, q, j->k, ids, scores);
        break;
    case VGO_BENCH_RABITQ:
        r = vgo_flat_search_rabitq(j->codes, j->n, j->dim, q, j->k, ids, scores);
        break;
    case VGO_BENCH_VAMANA:
        r = vgo_vamana_search(j->vamana, q, j->k, ids, scores, &st);
        break;
    case VGO_BENCH_SQ8:
        r = vgo_flat_search_sq8(j->codes, j->n, j->dim, j->sq_mins, j->sq_inv_scales, q, j->k, ids, scores);
        break;
    }
    for (int i = r; i < j->k; i++) ids[i] = 0xFFFFFFFFu;
    if (j->dist_comps) j->dist_comps[qi] = st.distance_computations;
}

static void *bench_worker(void *p)
{
    worker *w = (worker *)p;
    const vgo_bench_job *j = w->job;
    uint32_t *ids = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(j->k > 0 ? j->k : 1));
    float *scores = (float *)malloc(sizeof(float) * (size_t)(j->k > 0 ? j->k : 1));
    const int walks = j->kind == VGO_BENCH_HNSW || j->kind == VGO_BENCH_HNSW_PQ_RERANK;
    uint32_t *visited = walks ? (uint32_t *)calloc((size_t)j->hnsw->n, sizeof(uint32_t)) : NULL;
    uint32_t epoch = 0;
    rerank_scratch rs = {NULL, NULL, NULL, 0};
    if (j->kind == VGO_BENCH_HNSW_PQ_RERANK) {
        rs.cap = j->ef > j->k ? j->ef : j->k;
        if (rs.cap < 1) rs.cap = 1;
        rs.cand = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)rs.cap);
        rs.cs = (float *)malloc(sizeof(float) * (size_t)rs.cap);
        rs.sorted = (vgo_cand *)malloc(sizeof(vgo_cand) * (size_t)rs.cap);
    }
    int64_t i = w->tid;
    for (;;) {
        const int64_t qi = i % j->nq;
        const int first_pass = i < j->nq;
        if (++epoch == 0 && visited) { /* visited.go:60-75: clear on wrap-around */
            memset(visited, 0, sizeof(uint32_t) * (size_t)j->hnsw->n);
            epoch = 1;
        }
        const int keep = first_pass && j->ids;
        run_one(j, qi, keep ? j->ids + qi * j->k : ids, (keep && j->scores) ? j->scores + qi * j->k : scores, visited, epoch,
                &rs);
        w->done++;
        i += w->nthreads;
        if (now_s() >= w->deadline) break;
    }
    free(ids);
    free(scores);
    free(visited);
    free(rs.cand);
    free(rs.cs);
    free(rs.sorted);
    return NULL;
}

/* Runs for about budget_s seconds of wall time (every thread finishes the query it is on).  Returns
 * queries completed; *seconds = wall time from the first thread's start to the last one's end. */
int64_t vgo_bench_run(const vgo_bench_job *job, int32_t nthreads, double budget_s, double *seconds)
{
    if (nthreads < 1) nthreads = 1;
    worker *ws = (worker *)calloc((size_t)nthreads, sizeof(worker));
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    const double t0 = now_s();
    for (int t = 0; t < nthreads; t++) {
        ws[t].job = job;
        ws[t].tid = t;
        ws[t].nthreads = nthreads;
        ws[t].deadline = t0 + budget_s;
        pthread_create(&th[t], NULL, bench_worker, &ws[t]);
    }
    int64_t total = 0;
    for (int t = 0; t < nthreads; t++) {
        pthread_join(th[t], NULL);
        total += ws[t].done;
    }
    *seconds = now_s() - t0;
    free(ws);
    free(th);
    return total;
}

/* ---- build-side twins: one UNIT per thread until the deadline ------------------------------------------
 * The functions north_star names that are not searches — kmeans assignment, PQ Train / Encode /
 * BuildDistanceTable, Segment.Rerank, hnsw.BruteSearch — timed the same way as the searches above: the oracle's
 * restatement of the reference's loop, one unit (a row, a query, a sub-quantizer) per C thread, the reference's
 * compiled AVX-512 kernels through the hooks where the reference calls internal/simd (the int8-dequantised PQ
 * kernels are generic Go on amd64, kernels.go:354-396: the scalar restatement IS the reference's arithmetic). */
enum { VGO_BUILD_KM_ASSIGN = 0, VGO_BUILD_PQ_ENCODE = 1, VGO_BUILD_PQ_LUT = 2, VGO_BUILD_RERANK = 3, VGO_BUILD_BRUTE = 4,
       VGO_BUILD_PQ_TRAIN_SUB = 5 };

typedef struct {
    int32_t kind;
    /* units: rows (KM_ASSIGN, PQ_ENCODE), queries (PQ_LUT, RERANK, BRUTE), sub-quantizers (PQ_TRAIN_SUB) */
    const float *units;
    int64_t n_units;
    int32_t dim, metric;
    const float *centroids; /* KM_ASSIGN: k x dim */
    int32_t k;
    const vgo_pq *pq;       /* PQ_ENCODE, PQ_LUT */
    const float *base;      /* RERANK: corpus rows */
    const uint32_t *cand;   /* RERANK: [n_units][nc] */
    int32_t nc, topk;
    const vgo_hnsw_graph *hnsw; /* BRUTE */
    int32_t mode;
    int64_t train_n;        /* PQ_TRAIN_SUB: `units` = training rows [train_n][dim] */
    int32_t pq_m, pq_k, iters;
    uint64_t seed;
    /* outputs of the first pass over the units (may be NULL) */
    int32_t *out_assign;    /* [n_units] */
    uint8_t *out_codes;     /* [n_units][m] */
    uint32_t *out_ids;      /* [n_units][topk] */
    float *out_scores;      /* [n_units][topk] */
    float *out_cent;        /* [pq_m][pq_k][dim/pq_m] */
} vgo_build_job;

typedef struct {
    const vgo_build_job *job;
    int tid, nthreads, one_pass;
    double deadline;
    int64_t done;
} build_worker;

static void *build_thread(void *p)
{
    build_worker *w = (build_worker *)p;
    const vgo_build_job *j = w->job;
    const int32_t topk = j->topk > 0 ? j->topk : 1;
    float *dists = (float *)malloc(sizeof(float) * (size_t)(j->k > 0 ? j->k : 1));
    uint8_t *codes = j->pq ? (uint8_t *)malloc((size_t)j->pq->m) : NULL;
    float *table = (j->kind == VGO_BUILD_PQ_LUT) ? (float *)malloc(sizeof(float) * (size_t)j->pq->m * j->pq->k) : NULL;
    float *cs = (float *)malloc(sizeof(float) * (size_t)(j->nc > 0 ? j->nc : 1));
    vgo_cand *sorted = (vgo_cand *)malloc(sizeof(vgo_cand) * (size_t)topk);
    uint32_t *ids = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)topk);
    float *scores = (float *)malloc(sizeof(float) * (size_t)topk);
    float *cent = (j->kind == VGO_BUILD_PQ_TRAIN_SUB)
                      ? (float *)malloc(sizeof(float) * (size_t)j->pq_k * (size_t)(j->dim / j->pq_m)) : NULL;
    volatile float sink = 0.0f;
    int64_t i = w->tid;
    for (;;) {
        const int64_t u = i % j->n_units;
        const int first = i < j->n_units;
        switch (j->kind) {
        case VGO_BUILD_KM_ASSIGN: { /* kmeans.go:54-99 / :142-196: SquaredL2Batch (DotBatch) over the k centroids, strict compare */
            const float *v = j->units + u * j->dim;
            int best = 0;
            if (j->metric == VGO_METRIC_L2) {
                if (vgo_hooks.l2_batch)
                    vgo_hooks.l2_batch((float *)v, (float *)j->centroids, j->dim, j->k, dists);
                else
                    vgo_l2_batch_avx512(v, j->centroids, j->dim, j->k, dists);
                float md = dists[0];
                for (int c = 1; c < j->k; c++)
                    if (dists[c] < md) {
                        md = dists[c];
                        best = c;
                    }
            } else {
                vgo_dot_batch_avx512(v, j->centroids, j->dim, j->k, dists);
                float mx = dists[0];
                for (int c = 1; c < j->k; c++)
                    if (dists[c] > mx) {
                        mx = dists[c];
                        best = c;
                    }
            }
            if (first && j->out_assign) j->out_assign[u] = best;
            break;
        }
        case VGO_BUILD_PQ_ENCODE: { /* pq.go:147-176 */
            uint8_t *dst = (first && j->out_codes) ? j->out_codes + u * j->pq->m : codes;
            vgo_pq_encode(j->pq, j->units + u * j->dim, dst);
            break;
        }
        case VGO_BUILD_PQ_LUT: /* pq.go:468-491 */
            vgo_pq_build_table(j->pq, j->units + u * j->dim, table);
            sink += table[0];
            break;
        case VGO_BUILD_RERANK: { /* flat/segment.go:754-780 + engine/search.go:914-965 */
            const uint32_t *cd = j->cand + u * j->nc;
            vgo_rerank_f32(j->base, j->dim, j->metric, j->units + u * j->dim, cd, j->nc, cs);
            vgo_candheap h;
            vgo_candheap_init(&h, topk, j->metric != VGO_METRIC_L2);
            for (int32_t c = 0; c < j->nc; c++) vgo_candheap_try_push_bounded(&h, (vgo_cand){0, cd[c], cs[c]}, topk);
            const int32_t r = vgo_candheap_sorted(&h, sorted);
            if (first && j->out_ids)
                for (int32_t c = 0; c < topk; c++) {
                    j->out_ids[u * topk + c] = c < r ? sorted[c].row_id : 0xFFFFFFFFu;
                    if (j->out_scores) j->out_scores[u * topk + c] = c < r ? sorted[c].score : 0.0f;
                }
            vgo_candheap_free(&h);
            break;
        }
        case VGO_BUILD_BRUTE: { /* hnsw.go:2021-2101 */
            uint32_t *oi = (first && j->out_ids) ? j->out_ids + u * topk : ids;
            float *os = (first && j->out_scores) ? j->out_scores + u * topk : scores;
            const int32_t r = vgo_hnsw_brute_search(j->hnsw, j->units + u * j->dim, topk, j->mode, NULL, oi, os);
            for (int32_t c = r; c < topk; c++) oi[c] = 0xFFFFFFFFu;
            break;
        }
        case VGO_BUILD_PQ_TRAIN_SUB: { /* pq.go:78-95: one goroutine per sub-quantizer */
            float *dst = (first && j->out_cent) ? j->out_cent + u * (int64_t)j->pq_k * (j->dim / j->pq_m) : cent;
            vgo_pq_train_subspace(j->units, j->train_n, j->dim, j->pq_m, (int32_t)u, j->pq_k, j->iters, j->seed, dst);
            break;
        }
        }
        w->done++;
        i += w->nthreads;
        if (w->one_pass ? i >= j->n_units : now_s() >= w->deadline) break;
    }
    (void)sink;
    free(dists);
    free(codes);
    free(table);
    free(cs);
    free(sorted);
    free(ids);
    free(scores);
    free(cent);
    return NULL;
}

/* as vgo_bench_run: ~budget_s seconds of wall time (0: every thread runs exactly one unit; < 0: exactly one pass over
 * all the units — the multi-threaded batch form the parity tests use as their checker); returns units completed */
int64_t vgo_bench_build_run(const vgo_build_job *job, int32_t nthreads, double budget_s, double *seconds)
{
    if (nthreads < 1) nthreads = 1;
    build_worker *ws = (build_worker *)calloc((size_t)nthreads, sizeof(build_worker));
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    const double t0 = now_s();
    for (int t = 0; t < nthreads; t++) {
        ws[t].job = job;
        ws[t].tid = t;
        ws[t].nthreads = nthreads;
        ws[t].deadline = t0 + budget_s;
        ws[t].one_pass = budget_s < 0 && (int64_t)t < job->n_units;
        if (budget_s < 0 && !ws[t].one_pass) continue; /* more threads than units */
        pthread_create(&th[t], NULL, build_thread, &ws[t]);
    }
    int64_t total = 0;
    for (int t = 0; t < nthreads; t++) {
        if (budget_s < 0 && !ws[t].one_pass) continue;
        pthread_join(th[t], NULL);
        total += ws[t].done;
    }
    *seconds = now_s() - t0;
    free(ws);
    free(th);
    return total;
}
