/*
 * vg_oracle_hnsw_build.c — CPU restatement of HNSW construction (TEST INFRASTRUCTURE ONLY, see
 * vg_oracle.h).  Follows internal/hnsw/hnsw.go of the reference:
 *   insert / performInsertion / insertNode   :713-984
 *   selectNeighbors{,Simple,Heuristic}, extractSortedCandidates, applyHeuristic, fillUpNeighbors :986-1106
 *   addConnection / addConnectionSimple / addConnectionPrune :455-555
 *   updateEntryPoint :885-900, layerForApplyInsert :2103-2116
 * Nothing is cached or approximated here: every heuristic decision recomputes its pair distance with the
 * reference's kernel order (vgo_l2_avx512 / vgo_dot_avx512) and every candidate order comes out of the
 * reference's 4-ary heap (vgo_prioq), so ties fall the way the reference's heap lets them fall.
 *
 * PARITY UNPINNED against the reference binary (no Go toolchain): pinned by restatement only.  The
 * reference's own ApplyBatchInsert is not reproducible run to run (goroutine interleaving); the batch
 * rule used here is the deterministic member of that family described in vg_oracle.h.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "vg_oracle.h"

#define INVALID 0xFFFFFFFFu

/* hnsw.go:2103-2116 with layerMultiplier = 1 / ln(M) (hnsw.go:218) */
int32_t vgo_hnsw_level_for_id(uint64_t id, int32_t m)
{
    uint64_t x = id + 0x9e3779b97f4a7c15ull;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
    x ^= x >> 31;
    const double inv = 1.0 / 9007199254740992.0; /* 1 / (1 << 53) */
    uint64_t u = x >> 11;
    double r = (double)u * inv;
    if (r == 0) r = inv;
    const double mult = 1.0 / log((double)m);
    return (int32_t)floor(-log(r) * mult);
}

int32_t vgo_hnsw_build_layout(int64_t n, int32_t m, int32_t *levels, int64_t *level_rows)
{
    int32_t top = 0;
    for (int l = 0; l < 63; l++) level_rows[l] = 0;
    for (int64_t i = 0; i < n; i++) {
        int32_t lv = vgo_hnsw_level_for_id((uint64_t)i, m);
        if (lv > 62) lv = 62;
        if (levels) levels[i] = lv;
        if (lv > top) top = lv;
        for (int l = 0; l < lv; l++) level_rows[l]++;
    }
    return top;
}

int64_t vgo_hnsw_build_batch(int64_t inserted, int64_t n, int32_t max_batch, int32_t growth_div)
{
    int64_t b = inserted / (growth_div > 0 ? growth_div : 1);
    if (b > max_batch) b = max_batch;
    if (b < 1) b = 1;
    if (b > n - inserted) b = n - inserted;
    return b;
}

typedef struct {
    const float *base;
    int64_t n;
    int32_t dim, metric, m, m0;
    uint32_t *l0;     /* n*m0 */
    float *l0_d;      /* cached distances, node.go Neighbor{ID, Dist} */
    int32_t *l0_cnt;
    uint32_t *slots;  /* max_level*n */
    uint32_t *adj;    /* rows*m */
    float *adj_d;
    int32_t *adj_cnt;
    int64_t level_off[64];
    int32_t top;
} builder;

static void row_of(builder *b, uint32_t node, int level, uint32_t **ids, float **d, int32_t **cnt, int *deg)
{
    if (level == 0) {
        *ids = b->l0 + (int64_t)node * b->m0;
        *d = b->l0_d + (int64_t)node * b->m0;
        *cnt = b->l0_cnt + node;
        *deg = b->m0;
        return;
    }
    int64_t r = b->level_off[level - 1] + b->slots[(int64_t)(level - 1) * b->n + node];
    *ids = b->adj + r * b->m;
    *d = b->adj_d + r * b->m;
    *cnt = b->adj_cnt + r;
    *deg = b->m;
}

/* h.distanceFunc (newDistanceFunc hnsw.go:2218-2238) */
static float pair_dist(const builder *b, uint32_t x, uint32_t y)
{
    const float *a = b->base + (int64_t)x * b->dim, *c = b->base + (int64_t)y * b->dim;
    if (b->metric == VGO_METRIC_DOT) return -vgo_dot_avx512(a, c, b->dim);
    float d = vgo_l2_avx512(a, c, b->dim);
    return b->metric == VGO_METRIC_COSINE ? 0.5f * d : d;
}

/* selectNeighbors hnsw.go:986-1024 (Heuristic = true, the default); consumes the max-heap `q`.
 * out: best-first items, returns their number (<= m). */
static int select_neighbors(const builder *b, vgo_prioq *q, int m, vgo_pq_item *out, vgo_pq_item *temp)
{
    vgo_pq_item it;
    if (q->len <= m) { /* selectNeighborsSimple :993-1009 */
        int n = 0;
        while (q->len > 0) {
            vgo_prioq_pop(q, &it);
            out[n++] = it;
        }
        for (int i = 0, j = n - 1; i < j; i++, j--) {
            vgo_pq_item t = out[i];
            out[i] = out[j];
            out[j] = t;
        }
        return n;
    }
    int nt = 0; /* extractSortedCandidates :1026-1046 */
    while (q->len > 0) {
        vgo_prioq_pop(q, &it);
        temp[nt++] = it;
    }
    for (int i = 0, j = nt - 1; i < j; i++, j--) {
        vgo_pq_item t = temp[i];
        temp[i] = temp[j];
        temp[j] = t;
    }
    int sel = 0; /* applyHeuristic :1048-1085 */
    for (int i = 0; i < nt; i++) {
        if (sel >= m) break;
        int good = 1;
        for (int s = 0; s < sel; s++) {
            float d = pair_dist(b, temp[i].node, out[s].node);
            if (d < temp[i].dist) {
                good = 0;
                break;
            }
        }
        if (good) out[sel++] = temp[i];
    }
    for (int i = 0; i < nt && sel < m; i++) { /* fillUpNeighbors :1087-1106 */
        int found = 0;
        for (int s = 0; s < sel; s++)
            if (out[s].node == temp[i].node) {
                found = 1;
                break;
            }
        if (!found) out[sel++] = temp[i];
    }
    return sel;
}

/* addConnection hnsw.go:455-499 (+ Simple :501-518, Prune :520-555) */
static void add_connection(builder *b, vgo_prioq *q, vgo_pq_item *out, vgo_pq_item *temp, uint32_t source,
                           uint32_t target, int level, float dist)
{
    uint32_t *ids;
    float *d;
    int32_t *cnt;
    int deg;
    row_of(b, source, level, &ids, &d, &cnt, &deg);
    for (int i = 0; i < *cnt; i++)
        if (ids[i] == target) return;
    if (*cnt < deg) {
        ids[*cnt] = target;
        d[*cnt] = dist;
        (*cnt)++;
        return;
    }
    q->len = 0;
    for (int i = 0; i < *cnt; i++) vgo_prioq_push(q, (vgo_pq_item){ids[i], d[i]});
    vgo_prioq_push(q, (vgo_pq_item){target, dist});
    int k = select_neighbors(b, q, deg, out, temp);
    for (int i = 0; i < deg; i++) {
        ids[i] = i < k ? out[i].node : INVALID;
        d[i] = i < k ? out[i].dist : 0.0f;
    }
    *cnt = k;
}

int32_t vgo_hnsw_build(const float *base, int64_t n, int32_t dim, int32_t metric, int32_t m, int32_t ef,
                       int32_t max_batch, int32_t growth_div, uint32_t *l0, uint32_t *slots, uint32_t *adj,
                       uint32_t *entry_point, int32_t *max_level)
{
    if (n <= 0 || dim <= 0 || m < 2 || ef < 1 || max_batch < 1 || growth_div < 1) return -1;
    builder b;
    memset(&b, 0, sizeof b);
    b.base = base;
    b.n = n;
    b.dim = dim;
    b.metric = metric;
    b.m = m;
    b.m0 = 2 * m; /* mmax0Multiplier hnsw.go:28 */
    int32_t *levels = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    int64_t level_rows[63];
    b.top = vgo_hnsw_build_layout(n, m, levels, level_rows);
    int64_t rows = 0;
    for (int l = 0; l < b.top; l++) {
        b.level_off[l] = rows;
        rows += level_rows[l];
    }
    b.l0 = l0;
    b.slots = slots;
    b.adj = adj;
    b.l0_d = (float *)calloc((size_t)n * b.m0, sizeof(float));
    b.l0_cnt = (int32_t *)calloc((size_t)n, sizeof(int32_t));
    b.adj_d = (float *)calloc((size_t)(rows > 0 ? rows : 1) * m, sizeof(float));
    b.adj_cnt = (int32_t *)calloc((size_t)(rows > 0 ? rows : 1), sizeof(int32_t));
    for (int64_t i = 0; i < n * b.m0; i++) l0[i] = INVALID;
    for (int64_t i = 0; i < rows * m; i++) adj[i] = INVALID;
    for (int l = 0; l < b.top; l++) { /* slot = rank of the node among the nodes of that level, id order */
        uint32_t next = 0;
        for (int64_t i = 0; i < n; i++) slots[(int64_t)l * n + i] = levels[i] >= l + 1 ? next++ : INVALID;
    }

    const uint32_t *slot_ptrs[63];
    const uint32_t *adj_ptrs[63];
    for (int l = 0; l < b.top; l++) {
        slot_ptrs[l] = slots + (int64_t)l * n;
        adj_ptrs[l] = adj + b.level_off[l] * m;
    }
    vgo_hnsw_graph g;
    memset(&g, 0, sizeof g);
    g.n = n;
    g.dim = dim;
    g.metric = metric;
    g.base = base;
    g.m0 = b.m0;
    g.l0 = l0;
    g.m = m;
    g.slot = slot_ptrs;
    g.adj = adj_ptrs;

    uint32_t *visited = (uint32_t *)calloc((size_t)n, sizeof(uint32_t));
    uint32_t epoch = 0;
    vgo_prioq cand, res, pq;
    vgo_prioq_init(&cand, 0, ef * 2);
    vgo_prioq_init(&res, 1, ef);
    vgo_prioq_init(&pq, 1, b.m0 + 1);
    const int tmax = (ef > b.m0 + 1 ? ef : b.m0 + 1) + 1;
    vgo_pq_item *out = (vgo_pq_item *)malloc(sizeof(vgo_pq_item) * (size_t)tmax);
    vgo_pq_item *temp = (vgo_pq_item *)malloc(sizeof(vgo_pq_item) * (size_t)tmax);

    /* first node: performInsertion :857-864 */
    uint32_t entry = 0;
    int32_t cur_top = levels[0];
    int64_t done = 1;
    vgo_search_stats st = {0, 0, 0, 0};
    while (done < n) {
        const int64_t bsz = vgo_hnsw_build_batch(done, n, max_batch, growth_div);
        /* every node of the batch: insertNode :902-984 against the graph of the batch start */
        for (int64_t t = done; t < done + bsz; t++) {
            const float *vec = base + t * dim;
            const int32_t lt = levels[t];
            uint32_t cur = entry;
            float cur_d = vgo_hnsw_node_distance(&g, vec, cur);
            for (int level = cur_top; level > lt; level--) { /* :918-934 */
                int changed = 1;
                while (changed) {
                    changed = 0;
                    uint32_t *ids;
                    float *dd;
                    int32_t *cnt;
                    int deg;
                    row_of(&b, cur, level, &ids, &dd, &cnt, &deg);
                    for (int i = 0; i < *cnt; i++) {
                        float d = vgo_hnsw_node_distance(&g, vec, ids[i]);
                        if (d < cur_d) {
                            cur = ids[i];
                            cur_d = d;
                            changed = 1;
                        }
                    }
                }
            }
            for (int level = lt < cur_top ? lt : cur_top; level >= 0; level--) { /* :940-981 */
                epoch++;
                vgo_hnsw_search_layer(&g, vec, cur, cur_d, level, ef, visited, epoch, &cand, &res, &st);
                vgo_pq_item best = res.items[0]; /* MinItem queue.go:46-57 */
                for (int i = 1; i < res.len; i++)
                    if (res.items[i].dist < best.dist) best = res.items[i];
                cur = best.node;
                cur_d = best.dist;
                const int maxc = level == 0 ? b.m0 : m;
                int k = select_neighbors(&b, &res, maxc, out, temp);
                uint32_t *ids;
                float *dd;
                int32_t *cnt;
                int deg;
                row_of(&b, (uint32_t)t, level, &ids, &dd, &cnt, &deg);
                for (int i = 0; i < k; i++) { /* setConnections :445-453 */
                    ids[i] = out[i].node;
                    dd[i] = out[i].dist;
                }
                *cnt = k;
            }
        }
        /* the batch's back links, id order (insertNode :975-979) */
        for (int64_t t = done; t < done + bsz; t++) {
            const int32_t lt = levels[t];
            for (int level = lt < cur_top ? lt : cur_top; level >= 0; level--) {
                uint32_t *ids;
                float *dd;
                int32_t *cnt;
                int deg;
                row_of(&b, (uint32_t)t, level, &ids, &dd, &cnt, &deg);
                for (int i = 0; i < *cnt; i++) add_connection(&b, &pq, out, temp, ids[i], (uint32_t)t, level, dd[i]);
            }
        }
        for (int64_t t = done; t < done + bsz; t++) /* updateEntryPoint :885-900 */
            if (levels[t] > cur_top) {
                cur_top = levels[t];
                entry = (uint32_t)t;
            }
        done += bsz;
    }
    *entry_point = entry;
    *max_level = cur_top;
    free(out);
    free(temp);
    vgo_prioq_free(&cand);
    vgo_prioq_free(&res);
    vgo_prioq_free(&pq);
    free(visited);
    free(b.l0_d);
    free(b.l0_cnt);
    free(b.adj_d);
    free(b.adj_cnt);
    free(levels);
    return 0;
}
