/* The C ABI from plain C99 — the language cgo compiles include/vecgo_hip.h as.  Every call here is one the Go shim
 * (go/) makes through `C.vg_*`; the expected values are the reference's own known answers (internal/simd/floats_test.go,
 * internal/hnsw/hnsw_test.go:104-159).  Exit code 0 = all checks passed, 77 = no gfx950 device (there is no CPU
 * fallback), 1 = a check failed. */
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "vecgo_hip.h"

static int g_fail = 0;
#define EXPECT(c)                                                        \
    do {                                                                 \
        if (!(c)) {                                                      \
            fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, #c);      \
            g_fail++;                                                    \
        }                                                                \
    } while (0)

int main(void)
{
    vg_ctx *ctx = NULL;
    int32_t st = vg_ctx_create(0, &ctx);
    if (st == VG_ERR_NO_DEVICE || ctx == NULL) {
        printf("no gfx950 device: %s (no CPU fallback)\n", vg_last_error());
        return 77;
    }
    EXPECT(vg_abi_version() == VG_ABI_VERSION);
    EXPECT(strcmp(vg_status_string(VG_OK), vg_status_string(VG_ERR_INVALID_ARG)) != 0);

    /* simd.SquaredL2Batch / DotBatch: floats_test.go:17-25, 61-65 */
    {
        const float q[3] = {1, 2, 3}, t[6] = {4, 5, 6, 1, 2, 3};
        float out[2] = {-1, -1};
        EXPECT(vg_squared_l2_batch(ctx, q, t, 3, 2, out, NULL) == VG_OK);
        EXPECT(out[0] == 27.0f && out[1] == 0.0f);
        EXPECT(vg_dot_batch(ctx, q, t, 3, 2, out, NULL) == VG_OK);
        EXPECT(out[0] == 32.0f && out[1] == 14.0f);
        /* SquaredL2Bounded floats_test.go:76-163: bound 10 is exceeded, 27 and 100 are not */
        float dist[1];
        int32_t exc[1];
        const float b10 = 10.0f, b27 = 27.0f;
        EXPECT(vg_squared_l2_bounded_batch(ctx, q, t, 3, 1, &b10, 1, dist, exc, NULL) == VG_OK && exc[0] == 1 && dist[0] >= 10.0f);
        EXPECT(vg_squared_l2_bounded_batch(ctx, q, t, 3, 1, &b27, 1, dist, exc, NULL) == VG_OK && exc[0] == 0 && dist[0] == 27.0f);
        /* zero-length inputs succeed (kernels_amd64.go:291-297) */
        EXPECT(vg_squared_l2_batch(ctx, q, t, 3, 0, out, NULL) == VG_OK);
    }
    /* simd.Hamming: distance_test.go:68-71 */
    {
        const uint8_t a[2] = {255, 0}, c[4] = {0, 255, 255, 0};
        int32_t out[2];
        EXPECT(vg_hamming_batch(ctx, a, c, 2, 2, out, NULL) == VG_OK && out[0] == 16 && out[1] == 0);
    }
    /* errors come back as a status and a message, never an abort */
    {
        vg_pq *pq = NULL;
        EXPECT(vg_pq_create(ctx, 10, 3, 256, &pq) != VG_OK && pq == NULL && strlen(vg_last_error()) > 0);
        EXPECT(vg_pq_create(ctx, 16, 2, 256, &pq) == VG_OK && pq != NULL);
        EXPECT(vg_pq_is_trained(pq) == 0);
        uint8_t code[2];
        const float v[16] = {0};
        EXPECT(vg_pq_encode(pq, v, 1, code, NULL) == VG_ERR_NOT_TRAINED);
        EXPECT(vg_pq_destroy(pq) == VG_OK);
    }
    /* hnsw_test.go:104-159: three rows under the Dot metric; build, KNNSearch and BruteSearch return [1, 0, 2] with
     * distances -dot; flat.Segment.Search (vg_search_flat) reports the dot products, largest first */
    {
        const float rows[9] = {1, 0, 0, 2, 0, 0, -1, 0, 0}, q[3] = {1, 0, 0};
        vg_index *idx = NULL;
        EXPECT(vg_index_create(ctx, 3, 3, VG_METRIC_DOT, &idx) == VG_OK);
        EXPECT(vg_index_set_vectors(idx, rows, NULL) == VG_OK);
        EXPECT(vg_hnsw_build(idx, 8, 50, 1, 1, NULL) == VG_OK);
        uint32_t ids[3];
        float sc[3];
        EXPECT(vg_search_hnsw(idx, q, 1, 3, 100, ids, sc, NULL, NULL) == VG_OK);
        EXPECT(ids[0] == 1 && ids[1] == 0 && ids[2] == 2 && sc[0] == -2.0f && sc[1] == -1.0f && sc[2] == 1.0f);
        EXPECT(vg_search_hnsw_brute(idx, q, 1, 3, VG_BRUTE_SCAN, NULL, 0, ids, sc, NULL) == VG_OK);
        EXPECT(ids[0] == 1 && ids[1] == 0 && ids[2] == 2 && sc[0] == -2.0f && sc[1] == -1.0f && sc[2] == 1.0f);
        const uint8_t mask = 0x05; /* rows 0 and 2 */
        EXPECT(vg_search_hnsw_brute(idx, q, 1, 3, VG_BRUTE_BITMAP, &mask, 0, ids, sc, NULL) == VG_OK);
        EXPECT(ids[0] == 0 && ids[1] == 2 && ids[2] == VG_INVALID_ID && sc[0] == -1.0f && sc[1] == 1.0f && isinf(sc[2]));
        EXPECT(vg_search_flat(idx, q, 1, 3, ids, sc, NULL) == VG_OK);
        EXPECT(ids[0] == 1 && ids[1] == 0 && ids[2] == 2 && sc[0] == 2.0f && sc[1] == 1.0f && sc[2] == -1.0f);
        EXPECT(vg_search_hnsw_brute(idx, q, 1, 3, 9, NULL, 0, ids, sc, NULL) == VG_ERR_INVALID_ARG);
        EXPECT(vg_index_destroy(idx) == VG_OK);
    }
    EXPECT(vg_ctx_destroy(ctx) == VG_OK);
    if (g_fail) {
        fprintf(stderr, "%d check(s) failed\n", g_fail);
        return 1;
    }
    printf("C ABI from C99: all checks passed\n");
    return 0;
}
