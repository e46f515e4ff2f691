/*
 * vg_oracle.c — CPU restatement of vecgo's distance + quantization hot path.
 * TEST INFRASTRUCTURE ONLY (see vg_oracle.h).  Own code; follows the
 * reference's arithmetic, cites file:line (relative to the vecgo root).
 */
#include "vg_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* Optional kernel hooks (bench.py's cpu_baseline only): when oracle/_ref is   */
/* present the timed CPU loops call the REFERENCE's compiled AVX-512 kernels   */
/* instead of the scalar restatements below (bit-identical results, see        */
/* tests/test_oracle_golden.py), so that the reported baseline is the          */
/* reference's arithmetic at the reference's speed.  Set once, before threads. */
/* ------------------------------------------------------------------ */
vgo_kernel_hooks vgo_hooks = {0, 0, 0, 0, 0, 0};
void vgo_set_kernel_hooks(const vgo_kernel_hooks *h)
{
    if (h)
        vgo_hooks = *h;
    else
        memset(&vgo_hooks, 0, sizeof vgo_hooks);
}
static inline float hk_l2(const float *a, const float *b, int64_t n)
{
    if (vgo_hooks.l2 && n > 0) {
        float r;
        vgo_hooks.l2((float *)a, (float *)b, n, &r);
        return r;
    }
    return vgo_l2_avx512(a, b, n);
}
static inline float hk_dot(const float *a, const float *b, int64_t n)
{
    if (vgo_hooks.dot && n > 0) {
        float r;
        vgo_hooks.dot((float *)a, (float *)b, n, &r);
        return r;
    }
    return vgo_dot_avx512(a, b, n);
}
static inline float hk_adc(const float *table, const uint8_t *codes, int64_t m)
{
    if (vgo_hooks.adc && m > 0) {
        /* kernels_amd64.go:38-44 pqAdcOffsets[i] = i*256, 64-byte aligned */
        static const int32_t offs[16] __attribute__((aligned(64))) = {0, 256, 512, 768, 1024, 1280, 1536, 1792,
                                                                      2048, 2304, 2560, 2816, 3072, 3328, 3584, 3840};
        float r;
        vgo_hooks.adc((float *)table, (uint8_t *)codes, m, &r, offs);
        return r;
    }
    return vgo_adc_avx512(table, codes, m);
}
static inline int64_t hk_hamming(const uint8_t *a, const uint8_t *b, int64_t n)
{
    if (vgo_hooks.hamming) return (int64_t)vgo_hooks.hamming(a, b, n);
    return vgo_hamming(a, b, n);
}
static inline void hk_l2_bounded(const float *a, const float *b, int64_t n, float bound, float *out, int32_t *ex)
{
    if (vgo_hooks.l2_bounded && n > 0)
        vgo_hooks.l2_bounded((float *)a, (float *)b, n, bound, out, ex);
    else
        vgo_l2_bounded_avx512(a, b, n, bound, out, ex);
}


#if defined(__clang__)
#pragma clang fp contract(off)
#elif defined(__GNUC__)
#pragma GCC optimize("fp-contract=off")
#endif

#define FMA(a, b, c) __builtin_fmaf((a), (b), (c))

/* ------------------------------------------------------------------ */
/* reductions                                                          */
/* ------------------------------------------------------------------ */

/* _mm512_reduce_add_ps as clang lowers it (internal/simd/floats_avx512.s:
 * extractf64x4 + add, extractf128 + add, permilpd/shufpd + add, movshdup + add):
 * (i,i+8) -> (i,i+4) -> (i,i+2) -> (0,1). */
static inline float reduce16(const float *s)
{
    float a[8], b[4], c[2];
    for (int i = 0; i < 8; i++) a[i] = s[i] + s[i + 8];
    for (int i = 0; i < 4; i++) b[i] = a[i] + a[i + 4];
    for (int i = 0; i < 2; i++) c[i] = b[i] + b[i + 2];
    return c[0] + c[1];
}

/* hsum512 of internal/simd/src/bounded_l2_avx512.c:6-16: 256-halves, 128-halves,
 * then two _mm_hadd_ps: ((b0+b1)+(b2+b3)). */
static inline float hsum512(const float *s)
{
    float a[8], b[4];
    for (int i = 0; i < 8; i++) a[i] = s[i] + s[i + 8];
    for (int i = 0; i < 4; i++) b[i] = a[i] + a[i + 4];
    return (b[0] + b[1]) + (b[2] + b[3]);
}

/* (sum1+sum2) , (sum3+sum4), then their sum — floats_avx512.c:50-53 */
static inline void combine4(float acc[4][16], float *s)
{
    for (int l = 0; l < 16; l++) s[l] = (acc[0][l] + acc[1][l]) + (acc[2][l] + acc[3][l]);
}

/* ------------------------------------------------------------------ */
/* L0 kernels                                                          */
/* ------------------------------------------------------------------ */

/* internal/simd/src/floats_avx512.c:12-65 dotProductAvx512 */
float vgo_dot_avx512(const float *a, const float *b, int64_t n)
{
    float acc[4][16];
    memset(acc, 0, sizeof acc);
    int64_t epoch = n / 64;
    for (int64_t e = 0; e < epoch; e++) {
        const float *pa = a + e * 64, *pb = b + e * 64;
        for (int k = 0; k < 4; k++)
            for (int l = 0; l < 16; l++)
                acc[k][l] = FMA(pa[k * 16 + l], pb[k * 16 + l], acc[k][l]);
    }
    float s[16];
    combine4(acc, s);
    float total = reduce16(s);
    /* scalar tail: clang contracts `total += a*b` into vfmadd231ss (floats_avx512.s) */
    for (int64_t i = epoch * 64; i < n; i++) total = FMA(a[i], b[i], total);
    return total;
}

/* internal/simd/src/floats_avx512.c:69-129 squaredL2Avx512 */
float vgo_l2_avx512(const float *a, const float *b, int64_t n)
{
    float acc[4][16];
    memset(acc, 0, sizeof acc);
    int64_t epoch = n / 64;
    for (int64_t e = 0; e < epoch; e++) {
        const float *pa = a + e * 64, *pb = b + e * 64;
        for (int k = 0; k < 4; k++)
            for (int l = 0; l < 16; l++) {
                float d = pa[k * 16 + l] - pb[k * 16 + l];
                acc[k][l] = FMA(d, d, acc[k][l]);
            }
    }
    float s[16];
    combine4(acc, s);
    float total = reduce16(s);
    for (int64_t i = epoch * 64; i < n; i++) {
        float d = a[i] - b[i];
        total = FMA(d, d, total);
    }
    return total;
}

/* internal/simd/src/batch_avx512.c:19-83 squaredL2BatchAvx512 (one target) */
static float l2_batch_one(const float *q, const float *t, int64_t dim)
{
    float acc[4][16];
    memset(acc, 0, sizeof acc);
    int64_t j = 0;
    for (; j <= dim - 64; j += 64)
        for (int k = 0; k < 4; k++)
            for (int l = 0; l < 16; l++) {
                float d = q[j + k * 16 + l] - t[j + k * 16 + l];
                acc[k][l] = FMA(d, d, acc[k][l]);
            }
    float s[16];
    combine4(acc, s);
    for (; j <= dim - 16; j += 16) /* 16-wide tail goes into the combined register */
        for (int l = 0; l < 16; l++) {
            float d = q[j + l] - t[j + l];
            s[l] = FMA(d, d, s[l]);
        }
    float total = reduce16(s);
    for (; j < dim; j++) {
        float d = q[j] - t[j];
        total = FMA(d, d, total);
    }
    return total;
}

/* internal/simd/src/batch_avx512.c:86-143 dotBatchAvx512 (one target) */
static float dot_batch_one(const float *q, const float *t, int64_t dim)
{
    float acc[4][16];
    memset(acc, 0, sizeof acc);
    int64_t j = 0;
    for (; j <= dim - 64; j += 64)
        for (int k = 0; k < 4; k++)
            for (int l = 0; l < 16; l++)
                acc[k][l] = FMA(q[j + k * 16 + l], t[j + k * 16 + l], acc[k][l]);
    float s[16];
    combine4(acc, s);
    for (; j <= dim - 16; j += 16)
        for (int l = 0; l < 16; l++) s[l] = FMA(q[j + l], t[j + l], s[l]);
    float total = reduce16(s);
    for (; j < dim; j++) total = FMA(q[j], t[j], total);
    return total;
}

void vgo_l2_batch_avx512(const float *query, const float *targets, int64_t dim, int64_t n,
                         float *out)
{
    for (int64_t i = 0; i < n; i++) out[i] = l2_batch_one(query, targets + i * dim, dim);
}

void vgo_dot_batch_avx512(const float *query, const float *targets, int64_t dim, int64_t n,
                          float *out)
{
    for (int64_t i = 0; i < n; i++) out[i] = dot_batch_one(query, targets + i * dim, dim);
}

/* internal/simd/src/bounded_l2_avx512.c:19-108 squaredL2BoundedAvx512 */
void vgo_l2_bounded_avx512(const float *a, const float *b, int64_t n, float bound,
                           float *result, int32_t *exceeded)
{
    float acc[4][16];
    memset(acc, 0, sizeof acc);
    float total = 0.0f;
    float s[16];
    int64_t i = 0;
    while (i + 64 <= n) {
        for (int k = 0; k < 4; k++)
            for (int l = 0; l < 16; l++) {
                float d = a[i + k * 16 + l] - b[i + k * 16 + l];
                acc[k][l] = FMA(d, d, acc[k][l]);
            }
        i += 64;
        combine4(acc, s);
        total = hsum512(s);
        if (total > bound) {
            *result = total;
            *exceeded = 1;
            return;
        }
    }
    combine4(acc, s);
    total = hsum512(s);
    for (; i + 8 <= n; i += 8) { /* AVX2 8-wide remainder: mul (no FMA), hadd tree */
        float q[8];
        for (int l = 0; l < 8; l++) {
            float d = a[i + l] - b[i + l];
            q[l] = d * d;
        }
        float p0 = q[0] + q[4], p1 = q[1] + q[5], p2 = q[2] + q[6], p3 = q[3] + q[7];
        total += (p0 + p1) + (p2 + p3);
    }
    for (; i < n; i++) {
        float d = a[i] - b[i];
        total = FMA(d, d, total);
    }
    *result = total;
    *exceeded = (total > bound) ? 1 : 0;
}

/* internal/simd/src/floats_avx512.c:135-167 pqAdcLookupAvx512 (table stride is 256
 * regardless of K: kernels_amd64.go:38-44) */
float vgo_adc_avx512(const float *table, const uint8_t *codes, int64_t m)
{
    float s[16];
    memset(s, 0, sizeof s);
    int64_t i = 0;
    for (; i <= m - 16; i += 16)
        for (int l = 0; l < 16; l++) s[l] = s[l] + table[(i + l) * 256 + codes[i + l]];
    float total = reduce16(s);
    for (; i < m; i++) total += table[i * 256 + codes[i]];
    return total;
}

/* internal/simd/kernels.go:247-253 pqAdcLookupGeneric */
float vgo_adc_generic(const float *table, const uint8_t *codes, int64_t m)
{
    float sum = 0.0f;
    for (int64_t i = 0; i < m; i++) sum += table[i * 256 + codes[i]];
    return sum;
}

/* internal/simd/src/popcount_avx512.c:25-46 hammingAvx512 (integer: order-free) */
int64_t vgo_hamming(const uint8_t *a, const uint8_t *b, int64_t n)
{
    int64_t r = 0;
    int64_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t x, y;
        memcpy(&x, a + i, 8);
        memcpy(&y, b + i, 8);
        r += __builtin_popcountll(x ^ y);
    }
    for (; i < n; i++) r += __builtin_popcount((unsigned)(a[i] ^ b[i]));
    return r;
}

/* internal/simd/src/floats_avx512.c:174-218 scaleAvx512 */
void vgo_scale(float *a, int64_t n, float s)
{
    for (int64_t i = 0; i < n; i++) a[i] *= s;
}

/* internal/simd/doc.go:58-60 Sqrt = float32(math.Sqrt(float64(x))) */
float vgo_sqrt(float x) { return (float)sqrt((double)x); }

/* internal/simd/kernels.go:354-362 squaredL2Int8DequantizedGeneric — Go on
 * amd64 (GOAMD64=v1) does not fuse: mul, add, sub, mul, add each rounded. */
float vgo_l2_int8_deq(const float *query, const int8_t *code, int64_t n, float scale,
                      float offset)
{
    float sum = 0.0f;
    for (int64_t i = 0; i < n; i++) {
        float v = (float)code[i] * scale;
        v = v + offset;
        float d = query[i] - v;
        float dd = d * d;
        sum = sum + dd;
    }
    return sum;
}

/* internal/simd/kernels.go:364-374 buildDistanceTableInt8Generic */
void vgo_build_table_int8(const float *qsub, const int8_t *codebook, int64_t subdim, int64_t k,
                          float scale, float offset, float *out)
{
    if (subdim <= 0) return;
    for (int64_t c = 0; c < k; c++)
        out[c] = vgo_l2_int8_deq(qsub, codebook + c * subdim, subdim, scale, offset);
}

/* internal/simd/kernels.go:376-396 findNearestCentroidInt8Generic (strict <) */
int64_t vgo_nearest_centroid_int8(const float *qsub, const int8_t *codebook, int64_t subdim,
                                  int64_t k, float scale, float offset)
{
    if (subdim <= 0 || k <= 0) return 0;
    int64_t best = 0;
    float bd = vgo_l2_int8_deq(qsub, codebook, subdim, scale, offset);
    for (int64_t c = 1; c < k; c++) {
        float d = vgo_l2_int8_deq(qsub, codebook + c * subdim, subdim, scale, offset);
        if (d < bd) {
            bd = d;
            best = c;
        }
    }
    return best;
}

/* ------------------------------------------------------------------ */
/* deterministic RNG (the reference uses unseeded math/rand: pq.go:294,   */
/* kmeans.go:25 — construction is "parity unpinned"; this stream makes the */
/* oracle and the device code agree with each other)                       */
/* ------------------------------------------------------------------ */
static inline uint64_t splitmix64(uint64_t x)
{
    x += 0x9e3779b97f4a7c15ULL;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL;
    return x ^ (x >> 31);
}

uint64_t vgo_rng_u64(uint64_t seed, uint64_t a, uint64_t b, uint64_t c)
{
    uint64_t h = splitmix64(seed);
    h = splitmix64(h ^ a);
    h = splitmix64(h ^ b);
    h = splitmix64(h ^ c);
    return h;
}

/* uniform float in [0,1) with 24 random bits */
static inline float rng_f32(uint64_t r) { return (float)(r >> 40) * (1.0f / 16777216.0f); }

/* ------------------------------------------------------------------ */
/* Product quantizer                                                    */
/* ------------------------------------------------------------------ */

/* internal/quantization/pq.go:468-491 BuildDistanceTable */
void vgo_pq_build_table(const vgo_pq *pq, const float *query, float *table)
{
    for (int m = 0; m < pq->m; m++)
        vgo_build_table_int8(query + (int64_t)m * pq->subdim,
                             pq->codebooks + (int64_t)m * pq->k * pq->subdim, pq->subdim,
                             pq->k, pq->scales[m], pq->offsets[m], table + (int64_t)m * pq->k);
}

/* internal/quantization/pq.go:147-176 Encode */
void vgo_pq_encode(const vgo_pq *pq, const float *vec, uint8_t *codes)
{
    for (int m = 0; m < pq->m; m++)
        codes[m] = (uint8_t)vgo_nearest_centroid_int8(
            vec + (int64_t)m * pq->subdim, pq->codebooks + (int64_t)m * pq->k * pq->subdim,
            pq->subdim, pq->k, pq->scales[m], pq->offsets[m]);
}

/* internal/quantization/pq.go:185-229 Decode */
void vgo_pq_decode(const vgo_pq *pq, const uint8_t *codes, float *out)
{
    for (int m = 0; m < pq->m; m++) {
        const int8_t *src =
            pq->codebooks + ((int64_t)m * pq->k + codes[m]) * pq->subdim;
        for (int i = 0; i < pq->subdim; i++) {
            float v = (float)src[i] * pq->scales[m];
            out[m * pq->subdim + i] = v + pq->offsets[m];
        }
    }
}

/* internal/quantization/pq.go:234-260 ComputeAsymmetricDistance (sequential over m) */
float vgo_pq_asym_distance(const vgo_pq *pq, const float *query, const uint8_t *codes)
{
    float distance = 0.0f;
    for (int m = 0; m < pq->m; m++) {
        const int8_t *c = pq->codebooks + ((int64_t)m * pq->k + codes[m]) * pq->subdim;
        distance += vgo_l2_int8_deq(query + (int64_t)m * pq->subdim, c, pq->subdim,
                                    pq->scales[m], pq->offsets[m]);
    }
    return distance;
}

/* internal/quantization/pq.go:97-136: scale=(max-min)/255, offset=min+128*scale,
 * q = clamp(round((v-min)/scale),0,255)-128 ; math.Round = half away from zero */
void vgo_pq_quantize_centroids(const float *centroids, int64_t count, int8_t *out, float *scale,
                               float *offset)
{
    float mn = 3.40282346638528859811704183484516925440e+38f, mx = -mn;
    for (int64_t i = 0; i < count; i++) {
        if (centroids[i] < mn) mn = centroids[i];
        if (centroids[i] > mx) mx = centroids[i];
    }
    if (mx == mn) mx = mn + 1e-6f;
    float sc = (mx - mn) / 255.0f;
    float of = mn + 128.0f * sc;
    *scale = sc;
    *offset = of;
    for (int64_t i = 0; i < count; i++) {
        float t = (centroids[i] - mn) / sc;
        double r = round((double)t);
        int val = (int)r;
        if (val < 0) val = 0;
        if (val > 255) val = 255;
        out[i] = (int8_t)(val - 128);
    }
}

/* pq.go:416-433 findNearestCentroid over fp32 centroids (strict <, from MaxFloat32) */
static int nearest_f32(const float *v, const float *centroids, int k, int sd)
{
    float best = 3.40282346638528859811704183484516925440e+38f;
    int bi = 0;
    for (int c = 0; c < k; c++) {
        float d = vgo_l2_avx512(v, centroids + (int64_t)c * sd, sd);
        if (d < best) {
            best = d;
            bi = c;
        }
    }
    return bi;
}

/* ---- the running sum of k-means++ seeding: ONE definition, shared with the device (k_pq_train.hip) ----------------
 * DEVIATION from pq.go:296-336, stated once here.  The reference adds minDistSq[] up with one fp32 accumulator in
 * index order (`sum += minDistSq[i]`, :300,:332) and walks the same running sum again to pick the next centroid
 * (`cumsum += d; if cumsum >= target`, :317-323): per new centroid a chain of n dependent additions, 255 times per
 * sub-quantizer.  The reference draws its random numbers from the unseeded global math/rand (:294,:307,:314), so no
 * run of it can be reproduced bit for bit anyway (quality parity); what IS pinned is this oracle against
 * the device.  Both therefore use the same BLOCKED running sum, still in index order:
 *   T_b   = the sum of block b = elements 64 b .. 64 b + 63 (missing ones count as +0) by the balanced pairwise tree
 *           ((m0 + m1) + (m2 + m3)) + ... : stride 1, 2, 4, 8, 16, 32
 *   P_b   = P_(b-1) + T_b, one fp32 accumulator over the blocks in order; sum = P_last
 *   pick  = the first block b* whose P_b is not below the target (as the reference's walk, a NaN prefix stops it too),
 *           then cum = P_(b*-1); cum += m[i] over the block in order; the first i with cum >= target; if the block's
 *           sequential walk ends below the target (the tree and the walk round differently) its last element.
 * The sampling law is the reference's (index i with probability ~ minDistSq[i] / sum); the roundings of the sum are not. */
#define VGO_PP_BLOCK 64
static float pp_block_total(const float *m, int64_t i0, int64_t n)
{
    float a[VGO_PP_BLOCK];
    for (int t = 0; t < VGO_PP_BLOCK; t++) a[t] = i0 + t < n ? m[i0 + t] : 0.0f;
    for (int s = 1; s < VGO_PP_BLOCK; s <<= 1)
        for (int t = 0; t + s < VGO_PP_BLOCK; t += 2 * s) a[t] = a[t] + a[t + s];
    return a[0];
}
/* prefixes P_b of the block totals; returns sum */
static float pp_prefixes(const float *mind, int64_t n, float *pref)
{
    const int64_t nblk = (n + VGO_PP_BLOCK - 1) / VGO_PP_BLOCK;
    float run = 0.0f;
    for (int64_t b = 0; b < nblk; b++) {
        run = run + pp_block_total(mind, b * VGO_PP_BLOCK, n);
        pref[b] = run;
    }
    return run;
}
static int64_t pp_pick(const float *mind, int64_t n, const float *pref, float target)
{
    const int64_t nblk = (n + VGO_PP_BLOCK - 1) / VGO_PP_BLOCK;
    int64_t b = 0;
    while (b < nblk && pref[b] < target) b++;
    if (b == nblk) return 0; /* cannot happen for a finite sum (P_last = sum >= target); pq.go:317 `chosen := 0` */
    const int64_t i0 = b * VGO_PP_BLOCK, i1 = i0 + VGO_PP_BLOCK < n ? i0 + VGO_PP_BLOCK : n;
    float cum = b ? pref[b - 1] : 0.0f;
    for (int64_t i = i0; i < i1; i++) {
        cum = cum + mind[i];
        if (cum >= target) return i;
    }
    return i1 - 1;
}

/* pq.go:275-414 kmeans for ONE subspace.  RNG draws come from
 * vgo_rng_u64(seed, subspace, purpose, counter) in place of math/rand. */
static void pq_kmeans_subspace(const float *vectors, int64_t n, int dim, int sub, int sd, int k,
                               int iters, uint64_t seed, float *cent)
{
    const float *base = vectors + (int64_t)sub * sd;
    /* initializeCentroids pq.go:281-338 */
    if (n < k) {
        for (int i = 0; i < k; i++)
            memcpy(cent + (int64_t)i * sd, base + (i % n) * (int64_t)dim, sizeof(float) * sd);
    } else {
        uint64_t ctr = 0;
        int64_t first = (int64_t)(vgo_rng_u64(seed, (uint64_t)sub, 1, ctr++) % (uint64_t)n);
        memcpy(cent, base + first * dim, sizeof(float) * sd);
        float *mind = (float *)malloc(sizeof(float) * n);
        float *pref = (float *)malloc(sizeof(float) * (size_t)((n + VGO_PP_BLOCK - 1) / VGO_PP_BLOCK));
        for (int64_t i = 0; i < n; i++) mind[i] = vgo_l2_avx512(base + i * dim, cent, sd);
        float sum = pp_prefixes(mind, n, pref);
        for (int c = 1; c < k; c++) {
            if (sum == 0.0f) {
                int64_t idx = (int64_t)(vgo_rng_u64(seed, (uint64_t)sub, 1, ctr++) % (uint64_t)n);
                memcpy(cent + (int64_t)c * sd, base + idx * dim, sizeof(float) * sd);
                continue;
            }
            float target = rng_f32(vgo_rng_u64(seed, (uint64_t)sub, 1, ctr++)) * sum;
            int64_t chosen = pp_pick(mind, n, pref, target);
            memcpy(cent + (int64_t)c * sd, base + chosen * dim, sizeof(float) * sd);
            if (c == k - 1) break; /* the reference updates minDistSq once more; nothing reads it */
            for (int64_t i = 0; i < n; i++) {
                float d = vgo_l2_avx512(base + i * dim, cent + (int64_t)c * sd, sd);
                if (d < mind[i]) mind[i] = d;
            }
            sum = pp_prefixes(mind, n, pref);
        }
        free(mind);
        free(pref);
    }
    /* runKMeansIterations pq.go:340-351 */
    int32_t *assign = (int32_t *)calloc((size_t)n, sizeof(int32_t));
    int64_t *counts = (int64_t *)malloc(sizeof(int64_t) * k);
    float *nc = (float *)malloc(sizeof(float) * (size_t)k * sd);
    for (int it = 0; it < iters; it++) {
        int changed = 0;
        for (int64_t i = 0; i < n; i++) { /* assignClusters pq.go:353-386 */
            int a = nearest_f32(base + i * dim, cent, k, sd);
            if (assign[i] != a) {
                assign[i] = a;
                changed = 1;
            }
        }
        if (!changed) break;
        /* updateCentroids pq.go:388-414: sequential fp32 sums, then / float32(count) */
        memset(counts, 0, sizeof(int64_t) * k);
        memset(nc, 0, sizeof(float) * (size_t)k * sd);
        for (int64_t i = 0; i < n; i++) {
            int c = assign[i];
            counts[c]++;
            for (int j = 0; j < sd; j++) nc[(int64_t)c * sd + j] += base[i * dim + j];
        }
        for (int c = 0; c < k; c++) {
            if (counts[c] > 0) {
                for (int j = 0; j < sd; j++)
                    cent[(int64_t)c * sd + j] = nc[(int64_t)c * sd + j] / (float)counts[c];
            } else {
                int64_t idx = (int64_t)(vgo_rng_u64(seed, (uint64_t)sub, 2 + (uint64_t)it,
                                                    (uint64_t)c) %
                                        (uint64_t)n);
                memcpy(cent + (int64_t)c * sd, base + idx * dim, sizeof(float) * sd);
            }
        }
    }
    free(assign);
    free(counts);
    free(nc);
}

/* one sub-quantizer of Train (the unit the reference hands to a goroutine, pq.go:78-95): for the timed CPU twin */
void vgo_pq_train_subspace(const float *vectors, int64_t n, int32_t dim, int32_t m, int32_t sub, int32_t k,
                           int32_t iters, uint64_t seed, float *cent /* k * dim/m */)
{
    pq_kmeans_subspace(vectors, n, dim, sub, dim / m, k, iters, seed, cent);
}

/* internal/quantization/pq.go:68-143 Train */
int vgo_pq_train(const float *vectors, int64_t n, int32_t dim, int32_t m, int32_t k,
                 int32_t iters, uint64_t seed, int8_t *codebooks, float *scales,
                 float *offsets, float *centroids_f32)
{
    if (n <= 0 || m <= 0 || dim % m != 0 || k <= 0 || k > 256) return -1;
    int sd = dim / m;
    float *cent = (float *)malloc(sizeof(float) * (size_t)k * sd);
    for (int sub = 0; sub < m; sub++) {
        pq_kmeans_subspace(vectors, n, dim, sub, sd, k, iters, seed, cent);
        if (centroids_f32)
            memcpy(centroids_f32 + (int64_t)sub * k * sd, cent, sizeof(float) * (size_t)k * sd);
        vgo_pq_quantize_centroids(cent, (int64_t)k * sd, codebooks + (int64_t)sub * k * sd,
                                  &scales[sub], &offsets[sub]);
    }
    free(cent);
    return 0;
}

/* ------------------------------------------------------------------ */
/* RaBitQ / binary                                                      */
/* ------------------------------------------------------------------ */
int64_t vgo_rabitq_code_bytes(int32_t dim) { return (int64_t)((dim + 63) / 64) * 8 + 4; }

/* internal/quantization/rabitq.go:51-78 Encode */
void vgo_rabitq_encode(const float *v, int32_t dim, uint8_t *out)
{
    int64_t nb = (int64_t)((dim + 63) / 64) * 8;
    memset(out, 0, (size_t)nb + 4);
    float norm = vgo_sqrt(vgo_dot_avx512(v, v, dim));
    for (int i = 0; i < dim; i++)
        if (v[i] >= 0.0f) out[(i / 64) * 8 + (i % 64) / 8] |= (uint8_t)(1u << (i % 8));
    memcpy(out + nb, &norm, 4); /* little endian f32 */
}

/* internal/quantization/binary.go:59-79 Train: one float64 accumulator over every component in order */
float vgo_binary_train(const float *vectors, int64_t n, int32_t dim)
{
    double sum = 0.0;
    int64_t count = 0;
    for (int64_t i = 0; i < n; i++)
        for (int j = 0; j < dim; j++) {
            sum += (double)vectors[i * dim + j];
            count++;
        }
    return count > 0 ? (float)(sum / (double)count) : 0.0f;
}

/* internal/quantization/binary.go:173-188 Decode */
void vgo_binary_decode(const uint8_t *code, int32_t code_bytes, int32_t dim, float threshold, float *out)
{
    for (int i = 0; i < dim; i++) {
        int byte = i / 8, bit = i % 8;
        out[i] = (byte < code_bytes && (code[byte] & (1u << bit))) ? threshold + 0.5f : threshold - 0.5f;
    }
}

/* distance/distance.go:40-53 NormalizeL2InPlace; returns 0 for an empty or zero-norm vector */
int32_t vgo_normalize_l2(float *v, int32_t dim)
{
    if (dim == 0) return 0;
    float norm2 = vgo_dot_avx512(v, v, dim);
    if (norm2 == 0.0f) return 0;
    float inv = 1.0f / vgo_sqrt(norm2);
    vgo_scale(v, dim, inv);
    return 1;
}

/* internal/quantization/binary.go:130-154 EncodeUint64Into */
void vgo_binary_encode_u64(const float *v, int32_t dim, float threshold, uint64_t *dst)
{
    int nw = (dim + 63) / 64;
    memset(dst, 0, sizeof(uint64_t) * (size_t)nw);
    for (int i = 0; i < dim; i++)
        if (v[i] >= threshold) dst[i / 64] |= (uint64_t)1 << (i % 64);
}

/* internal/quantization/rabitq.go:119-176 Distance */
float vgo_rabitq_distance(const float *query, int32_t dim, const uint8_t *code)
{
    int nw = (dim + 63) / 64;
    int64_t nb = (int64_t)nw * 8;
    float ynorm;
    memcpy(&ynorm, code + nb, 4);
    float qnorm = vgo_sqrt(hk_dot(query, query, dim));
    uint64_t qc[64];
    uint64_t *q = nw <= 64 ? qc : (uint64_t *)malloc(sizeof(uint64_t) * (size_t)nw);
    vgo_binary_encode_u64(query, dim, 0.0f, q);
    float hamming = (float)hk_hamming((const uint8_t *)q, code, nb);
    if (q != qc) free(q);
    float t1 = qnorm - ynorm;
    float t1sq = t1 * t1;
    float t2 = 4.0f * qnorm;
    t2 = t2 * ynorm;
    t2 = t2 / (float)dim;
    t2 = t2 * hamming;
    return t1sq + t2;
}

/* ------------------------------------------------------------------ */
/* internal/kmeans                                                      */
/* ------------------------------------------------------------------ */
static int argbest_batch(const float *vec, const float *centroids, int dim, int k, int metric,
                         float *dists)
{
    if (metric == VGO_METRIC_L2) { /* kmeans.go:60-70 */
        vgo_l2_batch_avx512(vec, centroids, dim, k, dists);
        int best = 0;
        float md = dists[0];
        for (int j = 1; j < k; j++)
            if (dists[j] < md) {
                md = dists[j];
                best = j;
            }
        return best;
    }
    /* kmeans.go:71-81 Dot/Cosine: higher is better */
    vgo_dot_batch_avx512(vec, centroids, dim, k, dists);
    int best = 0;
    float mx = dists[0];
    for (int j = 1; j < k; j++)
        if (dists[j] > mx) {
            mx = dists[j];
            best = j;
        }
    return best;
}

/* internal/kmeans/kmeans.go:16-138 TrainKMeans.  rand.Perm / rand.Intn replaced by
 * the deterministic stream (Fisher-Yates over vgo_rng_u64). Returns 1 when n<k
 * (reference returns (nil,nil)), -1 on bad metric, 0 ok. */
int vgo_kmeans_train(const float *vectors, int64_t n, int32_t dim, int32_t k, int32_t metric,
                     int32_t max_iter, uint64_t seed, float *centroids)
{
    if (metric != VGO_METRIC_L2 && metric != VGO_METRIC_DOT && metric != VGO_METRIC_COSINE)
        return -1;
    if (n < k) return 1;
    int64_t *perm = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    for (int64_t i = 0; i < n; i++) perm[i] = i;
    for (int64_t i = 0; i < k && i < n - 1; i++) { /* partial Fisher-Yates: first k entries */
        int64_t j = i + (int64_t)(vgo_rng_u64(seed, 0, 1, (uint64_t)i) % (uint64_t)(n - i));
        int64_t t = perm[i];
        perm[i] = perm[j];
        perm[j] = t;
    }
    for (int i = 0; i < k; i++)
        memcpy(centroids + (int64_t)i * dim, vectors + perm[i] * dim, sizeof(float) * dim);
    free(perm);

    int32_t *assign = (int32_t *)calloc((size_t)n, sizeof(int32_t));
    int64_t *counts = (int64_t *)malloc(sizeof(int64_t) * k);
    float *sums = (float *)malloc(sizeof(float) * (size_t)k * dim);
    float *dists = (float *)malloc(sizeof(float) * k);
    for (int it = 0; it < max_iter; it++) {
        int changed = 0;
        for (int64_t i = 0; i < n; i++) {
            int b = argbest_batch(vectors + i * dim, centroids, dim, k, metric, dists);
            if (assign[i] != b) {
                assign[i] = b;
                changed = 1;
            }
        }
        if (!changed) break;
        memset(sums, 0, sizeof(float) * (size_t)k * dim);
        memset(counts, 0, sizeof(int64_t) * k);
        for (int64_t i = 0; i < n; i++) { /* kmeans.go:107-119 */
            int c = assign[i];
            for (int d = 0; d < dim; d++) sums[(int64_t)c * dim + d] += vectors[i * dim + d];
            counts[c]++;
        }
        for (int j = 0; j < k; j++) { /* kmeans.go:121-135: sums * (1/count) */
            if (counts[j] > 0) {
                float scale = 1.0f / (float)counts[j];
                for (int d = 0; d < dim; d++)
                    centroids[(int64_t)j * dim + d] = sums[(int64_t)j * dim + d] * scale;
            } else {
                int64_t idx = (int64_t)(vgo_rng_u64(seed, 0, 2 + (uint64_t)it, (uint64_t)j) %
                                        (uint64_t)n);
                memcpy(centroids + (int64_t)j * dim, vectors + idx * dim, sizeof(float) * dim);
            }
        }
    }
    free(assign);
    free(counts);
    free(sums);
    free(dists);
    return 0;
}

/* internal/kmeans/kmeans.go:142-196 AssignPartition */
int32_t vgo_assign_partition(const float *vec, const float *centroids, int32_t dim, int32_t k,
                             int32_t metric)
{
    if (metric != VGO_METRIC_L2 && metric != VGO_METRIC_DOT && metric != VGO_METRIC_COSINE)
        return -1;
    float *dists = (float *)malloc(sizeof(float) * k);
    int b = argbest_batch(vec, centroids, dim, k, metric, dists);
    free(dists);
    return b;
}

typedef struct {
    int32_t id;
    float dist;
} cdist;

/* internal/kmeans/kmeans.go:217-280 FindClosestCentroids.  The selection path
 * (n <= k/4 && n < 16) is restated exactly; the full-sort path uses
 * slices.SortFunc (pdqsort, unstable) in the reference — ties there are
 * "unpinned"; this restatement breaks ties by id. */
int vgo_find_closest_centroids(const float *query, const float *centroids, int32_t dim,
                               int32_t k, int32_t n, int32_t metric, int32_t *out)
{
    if (metric != VGO_METRIC_L2 && metric != VGO_METRIC_DOT && metric != VGO_METRIC_COSINE)
        return -1;
    if (n > k) n = k;
    cdist *d = (cdist *)malloc(sizeof(cdist) * k);
    float *vals = (float *)malloc(sizeof(float) * k);
    if (metric == VGO_METRIC_L2) {
        vgo_l2_batch_avx512(query, centroids, dim, k, vals);
        for (int i = 0; i < k; i++) d[i] = (cdist){i, vals[i]};
    } else {
        vgo_dot_batch_avx512(query, centroids, dim, k, vals);
        for (int i = 0; i < k; i++) d[i] = (cdist){i, -vals[i]};
    }
    if (n <= k / 4 && n < 16) {
        for (int i = 0; i < n; i++) {
            int mi = i;
            for (int j = i + 1; j < k; j++)
                if (d[j].dist < d[mi].dist) mi = j;
            cdist t = d[i];
            d[i] = d[mi];
            d[mi] = t;
            out[i] = d[i].id;
        }
    } else {
        /* slices.SortFunc(dists, cmpCentroidDistByDist) (kmeans.go:205-213,272): for up to 12 entries Go's pdqsort IS this
         * insertion sort (sort/zsortanyfunc.go insertionSortCmpFunc: an entry moves left while cmp < 0, i.e. dist <) — stable, ids
         * ascending on entry, so the id comparison below never fires there, and a NaN distance (cmp == 0 against everything) stops
         * every move, exactly as here.  Beyond 12 entries pdqsort proper runs: without ties and NaN any sort gives its order; WITH
         * them its order is not restated (ties go by id here; NaN distances stay where the insertion sort leaves them). */
        for (int i = 1; i < k; i++) {
            cdist x = d[i];
            int j = i - 1;
            while (j >= 0 && (d[j].dist > x.dist || (d[j].dist == x.dist && d[j].id > x.id))) {
                d[j + 1] = d[j];
                j--;
            }
            d[j + 1] = x;
        }
        for (int i = 0; i < n; i++) out[i] = d[i].id;
    }
    free(d);
    free(vals);
    return n;
}

/* ------------------------------------------------------------------ */
/* searcher.PriorityQueue — internal/searcher/queue.go:25-290 (4-ary)   */
/* ------------------------------------------------------------------ */
void vgo_prioq_init(vgo_prioq *q, int is_max, int32_t cap)
{
    q->is_max = is_max;
    q->len = 0;
    q->cap = cap > 16 ? cap : 16;
    q->items = (vgo_pq_item *)malloc(sizeof(vgo_pq_item) * (size_t)q->cap);
}
void vgo_prioq_free(vgo_prioq *q)
{
    free(q->items);
    q->items = NULL;
}

static void prioq_sift_up(vgo_prioq *q, int i) /* queue.go:161-183 */
{
    vgo_pq_item it = q->items[i];
    if (q->is_max) {
        while (i > 0) {
            int p = (i - 1) / 4;
            if (it.dist <= q->items[p].dist) break;
            q->items[i] = q->items[p];
            i = p;
        }
    } else {
        while (i > 0) {
            int p = (i - 1) / 4;
            if (it.dist >= q->items[p].dist) break;
            q->items[i] = q->items[p];
            i = p;
        }
    }
    q->items[i] = it;
}

static void prioq_sift_down(vgo_prioq *q, int i) /* queue.go:221-290 */
{
    int n = q->len;
    vgo_pq_item it = q->items[i];
    for (;;) {
        int fc = 4 * i + 1;
        if (fc >= n) break;
        int best = fc;
        float bd = q->items[fc].dist;
        int lc = fc + 4;
        if (lc > n) lc = n;
        if (q->is_max) {
            for (int c = fc + 1; c < lc; c++)
                if (q->items[c].dist > bd) {
                    best = c;
                    bd = q->items[c].dist;
                }
            if (it.dist >= bd) break;
        } else {
            for (int c = fc + 1; c < lc; c++)
                if (q->items[c].dist < bd) {
                    best = c;
                    bd = q->items[c].dist;
                }
            if (it.dist <= bd) break;
        }
        q->items[i] = q->items[best];
        i = best;
    }
    q->items[i] = it;
}

void vgo_prioq_push(vgo_prioq *q, vgo_pq_item it) /* queue.go:59-62 */
{
    if (q->len == q->cap) {
        q->cap *= 2;
        q->items = (vgo_pq_item *)realloc(q->items, sizeof(vgo_pq_item) * (size_t)q->cap);
    }
    q->items[q->len++] = it;
    prioq_sift_up(q, q->len - 1);
}

void vgo_prioq_push_bounded(vgo_prioq *q, vgo_pq_item it, int32_t capacity) /* queue.go:67-92 */
{
    if (q->len < capacity) {
        vgo_prioq_push(q, it);
        return;
    }
    vgo_pq_item top = q->items[0];
    if (q->is_max) {
        if (it.dist < top.dist) {
            q->items[0] = it;
            prioq_sift_down(q, 0);
        }
    } else {
        if (it.dist > top.dist) {
            q->items[0] = it;
            prioq_sift_down(q, 0);
        }
    }
}

int vgo_prioq_try_push_bounded(vgo_prioq *q, vgo_pq_item it, int32_t max_size) /* :190-215 */
{
    if (q->len < max_size) {
        vgo_prioq_push(q, it);
        return 1;
    }
    vgo_pq_item top = q->items[0];
    if (q->is_max) {
        if (it.dist >= top.dist) return 0;
    } else {
        if (it.dist <= top.dist) return 0;
    }
    q->items[0] = it;
    prioq_sift_down(q, 0);
    return 1;
}

int vgo_prioq_pop(vgo_prioq *q, vgo_pq_item *out) /* queue.go:114-129 */
{
    if (q->len == 0) return 0;
    *out = q->items[0];
    q->items[0] = q->items[q->len - 1];
    q->len--;
    if (q->len > 0) prioq_sift_down(q, 0);
    return 1;
}

/* ------------------------------------------------------------------ */
/* searcher.CandidateHeap — internal/searcher/candidate_queue.go        */
/* ------------------------------------------------------------------ */
int vgo_cand_better(vgo_cand a, vgo_cand b, int desc) /* :12-23 */
{
    if (a.score != b.score) return desc ? a.score > b.score : a.score < b.score;
    if (a.segment_id != b.segment_id) return a.segment_id < b.segment_id;
    return a.row_id < b.row_id;
}
static int cand_worse(vgo_cand a, vgo_cand b, int desc) /* :27-38 */
{
    if (a.score != b.score) return desc ? a.score < b.score : a.score > b.score;
    if (a.segment_id != b.segment_id) return a.segment_id > b.segment_id;
    return a.row_id > b.row_id;
}
void vgo_candheap_init(vgo_candheap *h, int32_t cap, int descending)
{
    h->cap = cap > 4 ? cap : 4;
    h->len = 0;
    h->descending = descending;
    h->c = (vgo_cand *)malloc(sizeof(vgo_cand) * (size_t)h->cap);
}
void vgo_candheap_free(vgo_candheap *h)
{
    free(h->c);
    h->c = NULL;
}
static void cand_up(vgo_candheap *h, int j) /* :141-152 */
{
    vgo_cand it = h->c[j];
    while (j > 0) {
        int i = (j - 1) / 4;
        if (!cand_worse(it, h->c[i], h->descending)) break;
        h->c[j] = h->c[i];
        j = i;
    }
    h->c[j] = it;
}
static void cand_down(vgo_candheap *h, int i, int n) /* :156-183 */
{
    vgo_cand it = h->c[i];
    for (;;) {
        int fc = 4 * i + 1;
        if (fc >= n) break;
        int best = fc, lc = fc + 4;
        if (lc > n) lc = n;
        for (int c = fc + 1; c < lc; c++)
            if (cand_worse(h->c[c], h->c[best], h->descending)) best = c;
        if (!cand_worse(h->c[best], it, h->descending)) break;
        h->c[i] = h->c[best];
        i = best;
    }
    h->c[i] = it;
}
int vgo_candheap_try_push_bounded(vgo_candheap *h, vgo_cand x, int32_t k) /* :120-132 */
{
    if (h->len < k) {
        if (h->len == h->cap) {
            h->cap *= 2;
            h->c = (vgo_cand *)realloc(h->c, sizeof(vgo_cand) * (size_t)h->cap);
        }
        h->c[h->len++] = x;
        cand_up(h, h->len - 1);
        return 1;
    }
    if (h->len > 0 && vgo_cand_better(x, h->c[0], h->descending)) {
        h->c[0] = x;
        cand_down(h, 0, h->len);
        return 1;
    }
    return 0;
}
/* SortedResults :191-211 — (score, segment, row) is a total order, so any correct
 * sort gives the reference's sequence; insertion sort keeps this file dependency-free */
int32_t vgo_candheap_sorted(const vgo_candheap *h, vgo_cand *dst)
{
    for (int i = 0; i < h->len; i++) {
        vgo_cand x = h->c[i];
        int j = i - 1;
        while (j >= 0 && vgo_cand_better(x, dst[j], h->descending)) {
            dst[j + 1] = dst[j];
            j--;
        }
        dst[j + 1] = x;
    }
    return h->len;
}

/* CandidateHeap.Pop :75-82 (Swap(0, n), down(0, n), truncate) — how the searches' results leave the heap (emit_sorted) — and
 * ReplaceTop :100-103 */
int vgo_candheap_pop(vgo_candheap *h, vgo_cand *out)
{
    if (h->len == 0) return 0;
    int n = h->len - 1;
    vgo_cand t = h->c[0];
    h->c[0] = h->c[n];
    h->c[n] = t;
    cand_down(h, 0, n);
    *out = h->c[n];
    h->len = n;
    return 1;
}
int vgo_candheap_replace_top(vgo_candheap *h, vgo_cand x) /* TryReplaceTop :104-111 */
{
    if (h->len == 0) return 0;
    h->c[0] = x;
    cand_down(h, 0, h->len);
    return 1;
}

/* A script of PriorityQueue operations (the op codes of include/vecgo_hip.h VG_HEAP_*: 0 PushItem, 1 PopItem,
 * 2 PushItemBounded, 3 TryPushBounded, 4 TopItem, 5 MinItem, 6 Reset, 7 Len): ops[i] = {op, node, float bits, arg},
 * out[i] = {flag, node, float bits}.  How the reference's queue tests (queue_test.go) and random scripts are run
 * against this heap and, by the same script, against the device heap (vg_debug_heap_replay). */
int32_t vgo_prioq_replay(int is_max, const int32_t *ops, int32_t n_ops, int32_t *out, uint64_t *final_items, int32_t cap)
{
    vgo_prioq q;
    vgo_prioq_init(&q, is_max, cap > 0 ? cap : 1);
    for (int i = 0; i < n_ops; i++) {
        int op = ops[4 * i], arg = ops[4 * i + 3], flag = 0;
        vgo_pq_item it, res = {0, 0.0f};
        it.node = (uint32_t)ops[4 * i + 1];
        memcpy(&it.dist, &ops[4 * i + 2], 4);
        switch (op) {
        case 0:
            vgo_prioq_push(&q, it);
            flag = 1;
            break;
        case 1:
            flag = vgo_prioq_pop(&q, &res);
            break;
        case 2: /* PushItemBounded reports nothing; the flag says whether the item went in (capacity 0 on an
                 * empty queue would index an empty slice in the reference: skipped) */
            if (q.len < arg) {
                flag = 1;
                vgo_prioq_push_bounded(&q, it, arg);
            } else if (q.len > 0) {
                flag = is_max ? it.dist < q.items[0].dist : it.dist > q.items[0].dist;
                vgo_prioq_push_bounded(&q, it, arg);
            }
            break;
        case 3:
            flag = vgo_prioq_try_push_bounded(&q, it, arg);
            break;
        case 4:
            if (q.len) {
                res = q.items[0];
                flag = 1;
            }
            break;
        case 5: /* MinItem queue.go:46-57 */
            if (q.len) {
                res = q.items[0];
                for (int j = 1; j < q.len; j++)
                    if (q.items[j].dist < res.dist) res = q.items[j];
                flag = 1;
            }
            break;
        case 6:
            q.len = 0;
            flag = 1;
            break;
        case 7:
            flag = q.len;
            break;
        default:
            flag = -1;
        }
        out[3 * i] = flag;
        out[3 * i + 1] = (int32_t)res.node;
        memcpy(&out[3 * i + 2], &res.dist, 4);
    }
    int32_t len = q.len;
    for (int i = 0; i < len && i < cap; i++) {
        uint32_t b;
        memcpy(&b, &q.items[i].dist, 4);
        final_items[i] = (uint64_t)q.items[i].node | ((uint64_t)b << 32);
    }
    vgo_prioq_free(&q);
    return len;
}

/* searcher.VisitedSet (visited.go:12-129) as written: uint8 epochs, 0 = never visited, Reset = epoch++ and a full clear
 * only on wrap-around, growth to max(2 * len, id + 1).  The searches above use a uint32 epoch array owned by the caller
 * (the same observable set; vg_cpu_bench.c keeps one per thread) — this literal restatement exists so that the
 * reference's own tests (visited_test.go) run against SOMETHING that follows the file line by line.  A script:
 * ops[i] = {op, id-or-capacity}: 0 Visit, 1 Visited, 2 CheckAndVisit, 3 Reset, 4 EnsureCapacity, 5 Capacity;
 * out[i] = the call's result (Visited / CheckAndVisit: 0/1, Capacity: len, others 0). */
int32_t vgo_visited_replay(int32_t capacity, const int64_t *ops, int32_t n_ops, int64_t *out)
{
    int64_t len = capacity > 0 ? capacity : 0;
    uint8_t *v = (uint8_t *)calloc((size_t)(len > 0 ? len : 1), 1);
    uint8_t epoch = 1; /* NewVisitedSet :21-26 */
    for (int i = 0; i < n_ops; i++) {
        const int64_t op = ops[2 * i], id = ops[2 * i + 1];
        int64_t r = 0;
        if ((op == 0 || op == 2 || op == 4) && (op == 4 ? id > len : id >= len)) { /* grow :121-128 */
            const int64_t want = op == 4 ? id : id + 1;
            const int64_t cap = len * 2 > want ? len * 2 : want;
            uint8_t *nv = (uint8_t *)calloc((size_t)cap, 1);
            memcpy(nv, v, (size_t)len);
            free(v);
            v = nv;
            len = cap;
        }
        switch (op) {
        case 0: v[id] = epoch; break;                                   /* Visit :29-35 */
        case 1: r = id < len && v[id] == epoch; break;                  /* Visited :38-44 */
        case 2: r = v[id] == epoch; v[id] = epoch; break;               /* CheckAndVisit :50-65 */
        case 3:                                                         /* Reset :102-109 */
            epoch++;
            if (epoch == 0) {
                epoch = 1;
                memset(v, 0, (size_t)len);
            }
            break;
        case 5: r = len; break;                                         /* Capacity :118-120 */
        default: break;
        }
        out[i] = r;
    }
    free(v);
    return (int32_t)len;
}

/* ------------------------------------------------------------------ */
/* flat scans — internal/segment/flat/segment.go:606-723                */
/* ------------------------------------------------------------------ */
/* What a caller of Segment.Search sees of the heap: the engine empties it with Pop() — worst first — into its buffer
 * (engine/search.go:859-862 per segment, :915-918 after the merge); this file reports that sequence reversed, best first.
 * (score, segment, row) is a total order on candidates WITHOUT NaN scores, and then any correct sort gives this sequence; a NaN
 * score is neither better nor worse than anything (candidate_queue.go:12-38), the heap's layout — the history of every accepted
 * candidate — then decides both which candidates it holds and the order they leave in, so it is popped, not sorted. */
static int32_t emit_sorted(vgo_candheap *h, uint32_t *ids, float *scores)
{
    int32_t n = h->len;
    vgo_cand c;
    for (int i = n - 1; i >= 0; i--) {
        vgo_candheap_pop(h, &c);
        ids[i] = c.row_id;
        scores[i] = c.score;
    }
    return n;
}

int32_t vgo_flat_search_f32(const float *base, int64_t n, int32_t dim, int32_t metric,
                            const float *query, int32_t k, uint32_t *ids, float *scores)
{
    int desc = metric != VGO_METRIC_L2; /* segment.go:449 */
    vgo_candheap h;
    vgo_candheap_init(&h, k, desc);
    for (int64_t i = 0; i < n; i++) { /* segment.go:691-701 */
        float d = desc ? hk_dot(query, base + i * dim, dim) : hk_l2(query, base + i * dim, dim);
        vgo_candheap_try_push_bounded(&h, (vgo_cand){0, (uint32_t)i, d}, k);
    }
    int32_t r = emit_sorted(&h, ids, scores);
    vgo_candheap_free(&h);
    return r;
}

int32_t vgo_flat_search_pq(const vgo_pq *pq, const uint8_t *codes, int64_t n,
                           const float *query, int32_t k, uint32_t *ids, float *scores)
{
    /* the reference builds the table with stride K but looks it up with stride 256
     * (pq.go:474 vs kernels.go:249): only K == 256 is self-consistent */
    float *table = (float *)calloc((size_t)pq->m * 256, sizeof(float));
    float *t = (float *)malloc(sizeof(float) * (size_t)pq->m * pq->k);
    vgo_pq_build_table(pq, query, t);
    for (int m = 0; m < pq->m; m++)
        memcpy(table + (int64_t)m * 256, t + (int64_t)m * pq->k, sizeof(float) * pq->k);
    free(t);
    vgo_candheap h;
    vgo_candheap_init(&h, k, 0);
    for (int64_t i = 0; i < n; i++) { /* segment.go:678-689 */
        float d = hk_adc(table, codes + i * pq->m, pq->m);
        vgo_candheap_try_push_bounded(&h, (vgo_cand){0, (uint32_t)i, d}, k);
    }
    int32_t r = emit_sorted(&h, ids, scores);
    vgo_candheap_free(&h);
    free(table);
    return r;
}

int32_t vgo_flat_search_rabitq(const uint8_t *codes, int64_t n, int32_t dim,
                               const float *query, int32_t k, uint32_t *ids, float *scores)
{
    int64_t cb = vgo_rabitq_code_bytes(dim);
    vgo_candheap h;
    vgo_candheap_init(&h, k, 0);
    for (int64_t i = 0; i < n; i++) {
        float d = vgo_rabitq_distance(query, dim, codes + i * cb);
        vgo_candheap_try_push_bounded(&h, (vgo_cand){0, (uint32_t)i, d}, k);
    }
    int32_t r = emit_sorted(&h, ids, scores);
    vgo_candheap_free(&h);
    return r;
}

void vgo_rerank_f32(const float *base, int32_t dim, int32_t metric, const float *query,
                    const uint32_t *ids, int32_t n, float *scores)
{
    for (int i = 0; i < n; i++) { /* flat/segment.go:766-772 */
        const float *v = base + (int64_t)ids[i] * dim;
        scores[i] = metric == VGO_METRIC_L2 ? hk_l2(query, v, dim) : hk_dot(query, v, dim);
    }
}

/* ------------------------------------------------------------------ */
/* HNSW search — internal/hnsw/hnsw.go                                  */
/* ------------------------------------------------------------------ */
/* vectorstore/columnar.go:29-50 snapshot distance: L2, -Dot, 0.5*L2 */
static float hnsw_dist(const vgo_hnsw_graph *g, const float *q, uint32_t id)
{
    if (g->pq) return vgo_pq_asym_distance(g->pq, q, g->codes + (int64_t)id * g->pq->m);
    const float *v = g->base + (int64_t)id * g->dim;
    switch (g->metric) {
    case VGO_METRIC_L2:
        return hk_l2(v, q, g->dim);
    case VGO_METRIC_DOT:
        return -hk_dot(v, q, g->dim);
    default:
        return 0.5f * hk_l2(v, q, g->dim);
    }
}

/* neighbour list of `node` on `level`: layer 0 rows hold m0 ids, upper rows m ids, 0xFFFFFFFF ends a list */
static const uint32_t *hnsw_neighbours(const vgo_hnsw_graph *g, uint32_t node, int level, int *deg)
{
    if (level == 0) {
        *deg = g->m0;
        return g->l0 + (int64_t)node * g->m0;
    }
    *deg = g->m;
    uint32_t slot = g->slot[level - 1][node];
    if (slot == 0xFFFFFFFFu) {
        *deg = 0;
        return NULL;
    }
    return g->adj[level - 1] + (int64_t)slot * g->m;
}

/* searchLayerUnfiltered hnsw.go:1220-1396 (initializeSearch :1567, processEntryPointUnfiltered :1560).
 * `visited` is an epoch array of n words: a node counts as visited when visited[id] == epoch.
 * `cand` (min-heap) and `res` (max-heap) are reset here; `res` is left exactly as the search left it
 * (insertNode reads it through MinItem and selectNeighbors, hnsw.go:942-957). */
void vgo_hnsw_search_layer(const vgo_hnsw_graph *g, const float *query, uint32_t ep, float ep_d,
                           int32_t level, int32_t ef, uint32_t *visited, uint32_t epoch,
                           vgo_prioq *candp, vgo_prioq *resp, vgo_search_stats *stp)
{
    vgo_search_stats st = *stp;
    vgo_prioq cand = *candp, res = *resp;
    cand.len = 0;
    res.len = 0;
#define TOMB(id) (g->tombstones && ((g->tombstones[(id) >> 3] >> ((id) & 7)) & 1))
    visited[ep] = epoch;
    vgo_prioq_push(&cand, (vgo_pq_item){ep, ep_d});
    if (!TOMB(ep)) vgo_prioq_push(&res, (vgo_pq_item){ep, ep_d}); /* processEntryPointUnfiltered hnsw.go:1559-1565 */

    int use_sc = g->metric == VGO_METRIC_L2 && !g->pq;
    int cap = ef * 2;
    int stagnant = 0;
    float last_best = 3.40282346638528859811704183484516925440e+38f;
    const int min_cap = ef + ef * 3 / 4;

    vgo_pq_item c;
    while (cand.len > 0) {
        vgo_prioq_pop(&cand, &c);
        st.pops++;
        if (res.len > 0) {
            vgo_pq_item worst = res.items[0];
            if (c.dist > worst.dist && res.len >= ef) break;
            if (worst.dist < last_best * 0.999f) { /* float32 * untyped const -> float32 */
                last_best = worst.dist;
                stagnant = 0;
            } else if (res.len >= ef) {
                stagnant++;
                if (stagnant >= 8 && cap > min_cap) {
                    cap -= ef / 8;
                    if (cap < min_cap) cap = min_cap;
                    stagnant = 0;
                }
            }
        }
        int deg;
        const uint32_t *nb = hnsw_neighbours(g, c.node, level, &deg);
        int has_bound = res.len >= ef;
        float bound = has_bound ? res.items[0].dist : 0.0f;
        for (int i = 0; i < deg && nb[i] != 0xFFFFFFFFu; i++) {
            uint32_t id = nb[i];
            if (visited[id] == epoch) continue;
            visited[id] = epoch;
            st.nodes_visited++;
            float nd;
            if (use_sc && has_bound) {
                int32_t ex;
                hk_l2_bounded(query, g->base + (int64_t)id * g->dim, g->dim, bound, &nd, &ex);
                st.distance_computations++;
                if (ex) {
                    st.distance_short_circuits++;
                    continue;
                }
            } else {
                nd = hnsw_dist(g, query, id);
                st.distance_computations++;
            }
            if (has_bound && nd > bound) continue;
            vgo_prioq_try_push_bounded(&cand, (vgo_pq_item){id, nd}, cap);
            if (TOMB(id)) continue; /* hnsw.go:1381-1390: only live nodes enter the results (and move the bound) */
            vgo_prioq_push_bounded(&res, (vgo_pq_item){id, nd}, ef);
            if (res.len >= ef) {
                bound = res.items[0].dist;
                has_bound = 1;
            }
        }
    }
#undef TOMB
    *candp = cand;
    *resp = res;
    *stp = st;
}

float vgo_hnsw_node_distance(const vgo_hnsw_graph *g, const float *query, uint32_t id)
{
    return hnsw_dist(g, query, id);
}

int32_t vgo_hnsw_search(const vgo_hnsw_graph *g, const float *query, int32_t k, int32_t ef,
                        uint32_t *ids, float *scores, vgo_search_stats *stats)
{
    return vgo_hnsw_search_ws(g, query, k, ef, ids, scores, stats, NULL, 1);
}

/* the same with a caller-owned visited array (n words, all != epoch): searcher.VisitedSet is an epoch array
 * reused across queries (visited.go:12-129), which is what a timed loop should pay for */
int32_t vgo_hnsw_search_ws(const vgo_hnsw_graph *g, const float *query, int32_t k, int32_t ef,
                           uint32_t *ids, float *scores, vgo_search_stats *stats, uint32_t *visited_ws,
                           uint32_t epoch)
{
    vgo_search_stats st = {0, 0, 0, 0};
    if (ef < k) ef = k; /* determineEF hnsw.go:1891-1894 */

    /* greedySearch hnsw.go:1897-1934 */
    uint32_t cur = g->entry_point;
    float cur_d = hnsw_dist(g, query, cur);
    for (int level = g->max_level; level > 0; level--) {
        int changed = 1;
        while (changed) {
            changed = 0;
            uint32_t slot = g->slot[level - 1][cur];
            if (slot == 0xFFFFFFFFu) continue;
            const uint32_t *nb = g->adj[level - 1] + (int64_t)slot * g->m;
            for (int i = 0; i < g->m && nb[i] != 0xFFFFFFFFu; i++) {
                float d = hnsw_dist(g, query, nb[i]);
                if (d < cur_d) {
                    cur = nb[i];
                    cur_d = d;
                    changed = 1;
                }
            }
        }
    }

    /* searchLayerUnfiltered hnsw.go:1220-1396 on layer 0 */
    uint32_t *visited = visited_ws ? visited_ws : (uint32_t *)calloc((size_t)g->n, sizeof(uint32_t));
    vgo_prioq cand, res;
    vgo_prioq_init(&cand, 0, ef * 2);
    vgo_prioq_init(&res, 1, ef);
    vgo_hnsw_search_layer(g, query, cur, cur_d, 0, ef, visited, epoch, &cand, &res, &st);
    /* knnSearchInternal extraction hnsw.go:1732-1751 */
    vgo_pq_item it;
    while (res.len > k) vgo_prioq_pop(&res, &it);
    int32_t nres = res.len;
    for (int i = nres - 1; i >= 0; i--) {
        vgo_prioq_pop(&res, &it);
        ids[i] = it.node;
        scores[i] = it.dist;
    }
    vgo_prioq_free(&cand);
    vgo_prioq_free(&res);
    if (!visited_ws) free(visited);
    if (stats) *stats = st;
    return nres;
}

/* searchExecute with a filter whose selectivity hint is above highSelectivityThreshold (hnsw.go:1107-1146):
 * searchLayerWithPostFilter (hnsw.go:1159-1218) — the unfiltered walk with an expanded ef, then every result popped
 * (worst first), the ones that pass the filter kept in that order and pushed back (PushItem below ef, PushItemBounded
 * at it) — then knnSearchInternal's extraction.  `mask`: bit i of byte i/8 = row i passes (filter.Matches and not
 * tombstoned); `ef` is what determineEF returned (its bitmap-cardinality expansion, hnsw.go:1863-1889, is the
 * caller's).  Returns -1 for a selectivity at or below the threshold (searchLayerPredicateAware is not restated). */
int32_t vgo_hnsw_search_filtered(const vgo_hnsw_graph *g, const float *query, int32_t k, int32_t ef, const uint8_t *mask,
                                 double selectivity, uint32_t *ids, float *scores, vgo_search_stats *stats)
{
    if (!(selectivity > 0.3)) return -1;
    vgo_search_stats st = {0, 0, 0, 0};
    if (ef < k) ef = k;
    uint32_t cur = g->entry_point;
    float cur_d = hnsw_dist(g, query, cur);
    for (int level = g->max_level; level > 0; level--) { /* greedySearch hnsw.go:1897-1934 */
        int changed = 1;
        while (changed) {
            changed = 0;
            uint32_t slot = g->slot[level - 1][cur];
            if (slot == 0xFFFFFFFFu) continue;
            const uint32_t *nb = g->adj[level - 1] + (int64_t)slot * g->m;
            for (int i = 0; i < g->m && nb[i] != 0xFFFFFFFFu; i++) {
                float d = hnsw_dist(g, query, nb[i]);
                if (d < cur_d) {
                    cur = nb[i];
                    cur_d = d;
                    changed = 1;
                }
            }
        }
    }
    /* hnsw.go:1166-1183 */
    int32_t expanded = (int32_t)((double)ef * (1.0 + (1.0 - selectivity) * 0.5));
    if (expanded > ef * 2) expanded = ef * 2;
    if (expanded > 500) expanded = 500;
    if (expanded < 1) expanded = 1;
    uint32_t *visited = (uint32_t *)calloc((size_t)g->n, sizeof(uint32_t));
    vgo_prioq cand, res;
    vgo_prioq_init(&cand, 0, expanded * 2);
    vgo_prioq_init(&res, 1, expanded > ef ? expanded : ef);
    vgo_hnsw_search_layer(g, query, cur, cur_d, 0, expanded, visited, 1, &cand, &res, &st);
    /* hnsw.go:1187-1217: pop everything, keep what passes, push back capped at ef */
    vgo_pq_item *keep = (vgo_pq_item *)malloc(sizeof(vgo_pq_item) * (size_t)(res.len > 0 ? res.len : 1));
    int32_t nkeep = 0;
    vgo_pq_item it;
    while (res.len > 0) {
        vgo_prioq_pop(&res, &it);
        if ((!mask || ((mask[it.node >> 3] >> (it.node & 7)) & 1)) &&
            !(g->tombstones && ((g->tombstones[it.node >> 3] >> (it.node & 7)) & 1)))
            keep[nkeep++] = it; /* hnsw.go:1198 */
    }
    for (int32_t i = 0; i < nkeep; i++) {
        if (res.len < ef)
            vgo_prioq_push(&res, keep[i]);
        else
            vgo_prioq_push_bounded(&res, keep[i], ef);
    }
    free(keep);
    while (res.len > k) vgo_prioq_pop(&res, &it); /* extraction hnsw.go:1732-1751 */
    int32_t nres = res.len;
    for (int i = nres - 1; i >= 0; i--) {
        vgo_prioq_pop(&res, &it);
        ids[i] = it.node;
        scores[i] = it.dist;
    }
    vgo_prioq_free(&cand);
    vgo_prioq_free(&res);
    free(visited);
    if (stats) *stats = st;
    return nres;
}

/* searchExecute with a filter whose selectivity hint is at or below highSelectivityThreshold, or unknown (hnsw.go:1107-1146):
 * searchLayerPredicateAware (hnsw.go:1406-1558) on layer 0 after the usual greedy descent — restated statement by statement:
 *   processEntryPoint (:1574-1583): the entry point always goes to the navigation queue (PushItem), to the results only when
 *     it passes the filter and is not tombstoned (PushItem);
 *   loop: pop the closest navigation candidate; stop when the results hold ef items and it is farther than the worst;
 *   per neighbour, in list order: CheckAndVisit; passesFilter / isDeleted; consecutiveFilterMisses (reset by a pass);
 *     a passing live node's distance is computed; a rejected one's comes from the cached edge distance (next.Dist > 0, else
 *     computed) while results < ef/2; while results < ef it is skipped after more than 10 consecutive misses or when its
 *     edge distance exceeds 1.5 x the worst result, else computed; with ef results it is skipped;
 *     then, unless the results are full and the node is farther than the worst: PushItem to the navigation queue and — passing,
 *     live nodes only — PushItemBounded(ef) to the results.
 * mask: bit i = filter.Matches(i); deleted: bit i = tombstoned (NULL = none); l0_dist: the layer-0 lists' cached Neighbor.Dist
 * values, n*m0 (node.go:62-80), NULL = recomputed as the distance between the two rows (what the insert stored, hnsw.go:516,550).
 * stats: nodes_visited, distance_computations, distance_short_circuits = ExpansionsSkipped, pops. */
int32_t vgo_hnsw_search_predicate(const vgo_hnsw_graph *g, const float *query, int32_t k, int32_t ef, const uint8_t *mask,
                                  const uint8_t *deleted, const float *l0_dist, uint32_t *ids, float *scores,
                                  vgo_search_stats *stats)
{
#define BIT(m, i) (((m)[(i) >> 3] >> ((i) & 7)) & 1)
    vgo_search_stats st = {0, 0, 0, 0};
    if (ef < k) ef = k;
    if (!deleted) deleted = g->tombstones;
    uint32_t cur = g->entry_point;
    float cur_d = hnsw_dist(g, query, cur);
    for (int level = g->max_level; level > 0; level--) { /* greedySearch hnsw.go:1897-1934 */
        int changed = 1;
        while (changed) {
            changed = 0;
            uint32_t slot = g->slot[level - 1][cur];
            if (slot == 0xFFFFFFFFu) continue;
            const uint32_t *nb = g->adj[level - 1] + (int64_t)slot * g->m;
            for (int i = 0; i < g->m && nb[i] != 0xFFFFFFFFu; i++) {
                float d = hnsw_dist(g, query, nb[i]);
                if (d < cur_d) {
                    cur = nb[i];
                    cur_d = d;
                    changed = 1;
                }
            }
        }
    }
    uint8_t *visited = (uint8_t *)calloc((size_t)g->n, 1);
    vgo_prioq cand, res;
    vgo_prioq_init(&cand, 0, ef * 2);
    vgo_prioq_init(&res, 1, ef + 1);
    visited[cur] = 1; /* initializeSearch */
    vgo_prioq_push(&cand, (vgo_pq_item){cur, cur_d});
    if ((!mask || BIT(mask, cur)) && !(deleted && BIT(deleted, cur))) vgo_prioq_push(&res, (vgo_pq_item){cur, cur_d});
    int misses = 0;
    vgo_pq_item c;
    while (cand.len > 0) {
        vgo_prioq_pop(&cand, &c);
        st.pops++;
        if (res.len >= ef && c.dist > res.items[0].dist) break;
        const uint32_t *nb = g->l0 + (int64_t)c.node * g->m0;
        for (int i = 0; i < g->m0 && nb[i] != 0xFFFFFFFFu; i++) {
            const uint32_t id = nb[i];
            if (visited[id]) continue;
            visited[id] = 1;
            st.nodes_visited++;
            const int passes = !mask || BIT(mask, id);
            const int dead = deleted && BIT(deleted, id);
            if (passes) misses = 0;
            else misses++;
            float nd;
            if (passes && !dead) {
                nd = hnsw_dist(g, query, id);
                st.distance_computations++;
            } else {
                const float edge = l0_dist ? l0_dist[(int64_t)c.node * g->m0 + i]
                                           : hnsw_dist(g, g->base + (int64_t)c.node * g->dim, id);
                if (res.len < ef / 2) {
                    if (edge > 0) {
                        nd = edge;
                    } else {
                        nd = hnsw_dist(g, query, id);
                        st.distance_computations++;
                    }
                } else if (res.len < ef) {
                    if (misses > 10) {
                        st.distance_short_circuits++;
                        continue;
                    }
                    if (edge > 0 && res.len > 0 && edge > res.items[0].dist * 1.5f) {
                        st.distance_short_circuits++;
                        continue;
                    }
                    nd = hnsw_dist(g, query, id);
                    st.distance_computations++;
                } else {
                    st.distance_short_circuits++;
                    continue;
                }
            }
            if (res.len >= ef && nd > res.items[0].dist) continue;
            vgo_prioq_push(&cand, (vgo_pq_item){id, nd});
            if (passes && !dead) vgo_prioq_push_bounded(&res, (vgo_pq_item){id, nd}, ef);
        }
    }
#undef BIT
    vgo_pq_item it;
    while (res.len > k) vgo_prioq_pop(&res, &it); /* extraction hnsw.go:1732-1751 */
    int32_t nres = res.len;
    for (int i = nres - 1; i >= 0; i--) {
        vgo_prioq_pop(&res, &it);
        ids[i] = it.node;
        scores[i] = it.dist;
    }
    vgo_prioq_free(&cand);
    vgo_prioq_free(&res);
    free(visited);
    if (stats) *stats = st;
    return nres;
}

/* hnsw.BruteSearch hnsw.go:2021-2073 + scanSegment :2075-2101, and searchBitmap :2240-2263 + knnSearchInternal's
 * extraction :1732-1751 — the two exhaustive paths of the HNSW index, each with the heap discipline it is WRITTEN
 * with (they differ, and with them the order in which equal distances leave the heap):
 *   VGO_BRUTE_SCAN    max-heap PriorityQueue; len < k -> PushItem, else `d < top.Distance` -> PopItem + PushItem
 *                     (the last leaf goes to the root and sifts down, then the new item sifts up from the end);
 *   VGO_BRUTE_BITMAP  s.Candidates (max-heap, searcher.go) with TryPushBounded(k): at capacity `d >= top` is
 *                     rejected, otherwise the new item REPLACES the root and sifts down.
 * Rows are visited in ascending id (node segments in order / bm.ForEach ascending, segment.go:154); `mask`
 * (bit i of byte i/8, NULL = every row) holds what survives node.IsZero()/filter (scan) or bitmap minus tombstones
 * (bitmap).  Results: popped into res[len-1 .. 0] (BruteSearch) resp. popped then reversed (extraction) — the
 * same best-first array.  Distances as the index wraps them (hnsw_dist: L2, -dot, 0.5*L2). */
int32_t vgo_hnsw_brute_search(const vgo_hnsw_graph *g, const float *query, int32_t k, int32_t mode,
                              const uint8_t *mask, uint32_t *ids, float *scores)
{
    if (k <= 0) return 0;
    vgo_prioq q;
    vgo_prioq_init(&q, 1, k + 1);
    for (int64_t i = 0; i < g->n; i++) {
        if (mask && !((mask[i >> 3] >> (i & 7)) & 1)) continue;
        vgo_pq_item it = {(uint32_t)i, hnsw_dist(g, query, (uint32_t)i)};
        if (mode == VGO_BRUTE_BITMAP) {
            vgo_prioq_try_push_bounded(&q, it, k);
        } else if (q.len < k) {
            vgo_prioq_push(&q, it);
        } else if (it.dist < q.items[0].dist) {
            vgo_pq_item dropped;
            vgo_prioq_pop(&q, &dropped);
            vgo_prioq_push(&q, it);
        }
    }
    int32_t nres = q.len;
    for (int i = nres - 1; i >= 0; i--) {
        vgo_pq_item it;
        vgo_prioq_pop(&q, &it);
        ids[i] = it.node;
        scores[i] = it.dist;
    }
    vgo_prioq_free(&q);
    return nres;
}

/* ------------------------------------------------------------------ */
/* Vamana beam search — internal/segment/diskann/segment.go:503-706     */
/* ------------------------------------------------------------------ */
static float vamana_dist(const vgo_vamana *v, const float *q, uint32_t id)
{
    switch (v->kind) {
    case VGO_VAMANA_PQ: /* segment.go:536-541 */
        return vgo_pq_asym_distance(v->pq, q, v->codes + (int64_t)id * v->pq->m);
    case VGO_VAMANA_RABITQ: /* segment.go:512-519 */
        return vgo_rabitq_distance(q, v->dim, v->codes + (int64_t)id * vgo_rabitq_code_bytes(v->dim));
    case VGO_VAMANA_INT4: /* segment.go:558-565: iq.L2Distance -> lookup-table kernel (int4.go:140-147) */
        return vgo_int4_l2_precomputed(q, v->codes + (int64_t)id * ((v->dim + 1) / 2), v->dim, v->int4_table);
    default: /* segment.go:582-588, distFunc = Provider(metric) */
        return v->metric == VGO_METRIC_L2
                   ? vgo_l2_avx512(q, v->base + (int64_t)id * v->dim, v->dim)
                   : vgo_dot_avx512(q, v->base + (int64_t)id * v->dim, v->dim);
    }
}

int32_t vgo_vamana_search(const vgo_vamana *v, const float *query, int32_t k, uint32_t *ids,
                          float *scores, vgo_search_stats *stats)
{
    return vgo_vamana_search_filtered(v, query, k, NULL, ids, scores, stats);
}

/* searchInternal with `filter` set (diskann/segment.go:616-627): pushToHeap returns before TryPushBounded for a row whose
 * filter.Matches is false — the traversal queue still takes it, the pruning test reads the heap of MATCHING rows only.
 * mask: bit i of byte i/8 = filter.Matches(i); NULL = no filter. */
int32_t vgo_vamana_search_filtered(const vgo_vamana *v, const float *query, int32_t k, const uint8_t *mask, uint32_t *ids,
                                   float *scores, vgo_search_stats *stats)
{
#define VAMANA_PASS(id) (!mask || ((mask[(id) >> 3] >> ((id) & 7)) & 1))
    vgo_search_stats st = {0, 0, 0, 0};
    int desc = v->metric != VGO_METRIC_L2; /* segment.go:597 */
    uint8_t *visited = (uint8_t *)calloc((size_t)v->n, 1);
    vgo_prioq cand;
    vgo_prioq_init(&cand, 0, 256);
    vgo_candheap heap;
    vgo_candheap_init(&heap, k, desc);

    uint32_t start = v->entry_point;
    visited[start] = 1;
    float sd = vamana_dist(v, query, start);
    st.distance_computations++;
    vgo_prioq_push(&cand, (vgo_pq_item){start, sd});
    if (VAMANA_PASS(start)) vgo_candheap_try_push_bounded(&heap, (vgo_cand){0, start, sd}, k);

    vgo_pq_item c;
    while (cand.len > 0) {
        vgo_prioq_pop(&cand, &c);
        st.pops++;
        if (heap.len >= k) {
            if (c.dist > heap.c[0].score) break;
        }
        const uint32_t *nb = v->graph + (int64_t)c.node * v->r;
        for (int i = 0; i < v->r; i++) {
            uint32_t id = nb[i];
            if (id == 0xFFFFFFFFu) continue;
            if (visited[id]) continue;
            visited[id] = 1;
            st.nodes_visited++;
            float d = vamana_dist(v, query, id);
            st.distance_computations++;
            vgo_prioq_push(&cand, (vgo_pq_item){id, d});
            if (VAMANA_PASS(id)) vgo_candheap_try_push_bounded(&heap, (vgo_cand){0, id, d}, k);
        }
    }
#undef VAMANA_PASS
    int32_t r = emit_sorted(&heap, ids, scores);
    vgo_prioq_free(&cand);
    vgo_candheap_free(&heap);
    free(visited);
    if (stats) *stats = st;
    return r;
}

/* ------------------------------------------------------------------ */
/* SQ8                                           */
/* ------------------------------------------------------------------ */
/* internal/simd/src/sq8_avx512.c:59-103 sq8uL2BatchPerDimensionAvx512, one row: 16 lane
 * accumulators over 16-element blocks (rec = fma(code, invScale, min); diff = q - rec;
 * sum[l] = fma(diff, diff, sum[l])), _mm512_reduce_add_ps, then a scalar tail that clang
 * contracts (rec = fma, total = fma(diff, diff, total)). */
static float sq8u_l2_one(const float *q, const uint8_t *code, const float *mins, const float *inv, int64_t dim)
{
    float sum[16];
    memset(sum, 0, sizeof sum);
    int64_t j = 0;
    for (; j <= dim - 16; j += 16)
        for (int l = 0; l < 16; l++) {
            float rec = FMA((float)code[j + l], inv[j + l], mins[j + l]);
            float diff = q[j + l] - rec;
            sum[l] = FMA(diff, diff, sum[l]);
        }
    float total = reduce16(sum);
    for (; j < dim; j++) {
        float rec = FMA((float)code[j], inv[j], mins[j]);
        float diff = q[j] - rec;
        total = FMA(diff, diff, total);
    }
    return total;
}

void vgo_sq8u_l2_batch(const float *query, const uint8_t *codes, const float *mins, const float *inv_scales,
                       int64_t dim, int64_t n, float *out)
{
    for (int64_t i = 0; i < n; i++) out[i] = sq8u_l2_one(query, codes + i * dim, mins, inv_scales, dim);
}

/* internal/quantization/quantizer.go:127-180 ScalarQuantizer.Train: per-dimension min / max
 * (a constant dimension gets max = min + 1e-6, :168-170), scale = 255 / range,
 * invScale = range / 255 (fp32 divisions) */
void vgo_sq8_train(const float *vectors, int64_t n, int32_t dim, float *mins, float *maxs, float *scales,
                   float *inv_scales)
{
    for (int d = 0; d < dim; d++) {
        mins[d] = 3.40282346638528859811704183484516925440e+38f;
        maxs[d] = -3.40282346638528859811704183484516925440e+38f;
    }
    for (int64_t i = 0; i < n; i++)
        for (int d = 0; d < dim; d++) {
            float v = vectors[i * dim + d];
            if (v < mins[d]) mins[d] = v;
            if (v > maxs[d]) maxs[d] = v;
        }
    for (int d = 0; d < dim; d++) {
        if (mins[d] == maxs[d]) maxs[d] = mins[d] + 1e-6f;
        float range = maxs[d] - mins[d];
        scales[d] = 255.0f / range;
        inv_scales[d] = range / 255.0f;
    }
}

/* quantizer.go:198-222 EncodeInto: clamp to [min, max], (val - min) * scale, uint8(x + 0.5)
 * (Go float->uint8 conversion truncates; the value is within [0.5, 255.5]) */
void vgo_sq8_encode(const float *v, int32_t dim, const float *mins, const float *maxs, const float *scales,
                    uint8_t *out)
{
    for (int d = 0; d < dim; d++) {
        float val = v[d];
        if (val < mins[d]) val = mins[d];
        else if (val > maxs[d]) val = maxs[d];
        float normalized = (val - mins[d]) * scales[d];
        float r = normalized + 0.5f;
        out[d] = (uint8_t)(int32_t)r;
    }
}

/* quantizer.go:240-250 DecodeInto: float32(code) * invScale + min (two roundings, Go does not fuse) */
void vgo_sq8_decode(const uint8_t *code, int32_t dim, const float *mins, const float *inv_scales, float *out)
{
    for (int d = 0; d < dim; d++) {
        float t = (float)code[d] * inv_scales[d];
        out[d] = t + mins[d];
    }
}

/* flat/segment.go:517-604: the SQ8 branch of Segment.Search (L2 only): L2DistanceBatch over
 * 256-row batches, candidates pushed with the (Score, RowID) order */
int32_t vgo_flat_search_sq8(const uint8_t *codes, int64_t n, int32_t dim, const float *mins,
                            const float *inv_scales, const float *query, int32_t k, uint32_t *ids,
                            float *scores)
{
    vgo_candheap h;
    vgo_candheap_init(&h, k, 0);
    for (int64_t i = 0; i < n; i++) {
        float d = sq8u_l2_one(query, codes + i * dim, mins, inv_scales, dim);
        vgo_candheap_try_push_bounded(&h, (vgo_cand){0, (uint32_t)i, d}, k);
    }
    int32_t r = emit_sorted(&h, ids, scores);
    vgo_candheap_free(&h);
    return r;
}

/* ScalarQuantizer.DotProduct — internal/quantization/quantizer.go:109-119: a scalar Go loop, no FMA
 * on amd64: val = mins[i] + float32(code[i])*invScales[i]; dot += q[i]*val */
float vgo_sq8_dot(const float *q, const uint8_t *code, int32_t dim, const float *mins, const float *inv_scales)
{
    float dot = 0.0f;
    for (int32_t i = 0; i < dim; i++) {
        float t = (float)code[i] * inv_scales[i];
        float val = mins[i] + t;
        float p = q[i] * val;
        dot = dot + p;
    }
    return dot;
}

/* ------------------------------------------------------------------ */
/* flat.Segment.Search with IVF partitions — flat/segment.go:447-751     */
/* ------------------------------------------------------------------ */
/* The scan type follows the segment's quantization (PQ table lookups :678-689, SQ8 batches for L2
 * :517-604, fp32 otherwise :691-701); with more than one partition only the nprobes closest
 * centroids' row ranges are scanned (:727-744, nprobes <= 0 -> 1), all into the same bounded heap.
 * The heap's order is total (score, then row id), so the order of the probed partitions does not
 * matter; only the probed SET does, and FindClosestCentroids' ties are measure-zero. */
int32_t vgo_flat_segment_search(const vgo_flat_segment *s, const float *query, int32_t k, int32_t nprobes,
                                uint32_t *ids, float *scores)
{
    return vgo_flat_segment_search_filtered(s, query, k, nprobes, NULL, ids, scores);
}

/* The same with `filter segment.Filter` set: a row whose filter.Matches(rowID) is false is skipped before it is scored
 * (segment.go:631-635; the SQ8 batch branch scores the batch first and skips the row at :559-561 — the candidates are the
 * same).  mask: bit i of byte i/8 = filter.Matches(i); NULL = no filter. */
int32_t vgo_flat_segment_search_filtered(const vgo_flat_segment *s, const float *query, int32_t k, int32_t nprobes,
                                         const uint8_t *mask, uint32_t *ids, float *scores)
{
    /* segment.go:657-701: SQ8 codes first (L2Distance / DotProduct by metric), else PQ, else fp32.
     * The reference's heap direction follows the metric for every branch (:449); for a Dot- or
     * Cosine-metric PQ segment that keeps the k LARGEST squared-L2 ADC distances — restated as written. */
    const int use_sq = s->sq_mins != NULL;
    const int use_pq = !use_sq && s->pq != NULL;
    const int desc = s->metric != VGO_METRIC_L2;
    float *table = NULL;
    if (use_pq) {
        table = (float *)calloc((size_t)s->pq->m * 256, sizeof(float));
        float *t = (float *)malloc(sizeof(float) * (size_t)s->pq->m * s->pq->k);
        vgo_pq_build_table(s->pq, query, t);
        for (int m = 0; m < s->pq->m; m++)
            memcpy(table + (int64_t)m * 256, t + (int64_t)m * s->pq->k, sizeof(float) * s->pq->k);
        free(t);
    }
    vgo_candheap h;
    vgo_candheap_init(&h, k, desc);
    int32_t np = 1;
    int32_t *parts = NULL;
    if (s->num_partitions > 1) {
        if (nprobes <= 0) nprobes = 1;
        parts = (int32_t *)malloc(sizeof(int32_t) * (size_t)s->num_partitions);
        np = vgo_find_closest_centroids(query, s->centroids, s->dim, s->num_partitions, nprobes, s->metric, parts);
    }
    for (int32_t j = 0; j < np; j++) {
        int64_t start = 0, end = s->n;
        if (parts) {
            start = s->part_offsets[parts[j]];
            end = s->part_offsets[parts[j] + 1];
        }
        for (int64_t i = start; i < end; i++) {
            float d;
            if (mask && !((mask[i >> 3] >> (i & 7)) & 1)) continue;
            if (use_pq)
                d = vgo_adc_avx512(table, s->codes + i * s->pq->m, s->pq->m);
            else if (use_sq)
                d = s->metric == VGO_METRIC_L2
                        ? sq8u_l2_one(query, s->codes + i * s->dim, s->sq_mins, s->sq_inv_scales, s->dim)
                        : vgo_sq8_dot(query, s->codes + i * s->dim, s->dim, s->sq_mins, s->sq_inv_scales);
            else
                d = desc ? vgo_dot_avx512(query, s->base + i * s->dim, s->dim)
                         : vgo_l2_avx512(query, s->base + i * s->dim, s->dim);
            vgo_candheap_try_push_bounded(&h, (vgo_cand){0, (uint32_t)i, d}, k);
        }
    }
    int32_t r = emit_sorted(&h, ids, scores);
    vgo_candheap_free(&h);
    free(parts);
    free(table);
    return r;
}

/* ------------------------------------------------------------------ */
/* INT4                                          */
/* ------------------------------------------------------------------ */
/* 1/15 as the reference materialises it: bits 0x3d888889 (int4_avx512.c:13-19,35-38) */
static float inv15(void)
{
    uint32_t b = 0x3d888889u;
    float f;
    memcpy(&f, &b, 4);
    return f;
}

/* element j of a packed code: even j = high nibble of byte j/2, odd j = low nibble (int4.go:93) */
static inline float nib(const uint8_t *code, int64_t j)
{
    uint8_t b = code[j / 2];
    return (float)((j & 1) ? (b & 0x0F) : ((b >> 4) & 0x0F));
}

/* internal/simd/src/int4_avx512.c:21-125 int4L2DistanceAvx512 == :191-299 int4L2DistanceBatchAvx512
 * per row: two 16-lane accumulators; 64-element blocks put sub-blocks 0,1 into sum1 and 2,3 into
 * sum2, 32-element blocks both into sum1; dq = fma(f * inv15, diff, min) (the product rounded first);
 * sum1 + sum2, _mm512_reduce_add_ps, then a scalar tail that clang contracts the same way. */
float vgo_int4_l2(const float *query, const uint8_t *code, int64_t dim, const float *min_val, const float *diff)
{
    float s1[16], s2[16];
    memset(s1, 0, sizeof s1);
    memset(s2, 0, sizeof s2);
    const float sc = inv15();
    int64_t i = 0;
#define INT4_BLOCK(acc, base)                                                  \
    for (int l = 0; l < 16; l++) {                                             \
        int64_t j = (base) + l;                                                \
        float f = nib(code, j) * sc;                                           \
        float dq = FMA(f, diff[j], min_val[j]);                                \
        float d = query[j] - dq;                                               \
        acc[l] = FMA(d, d, acc[l]);                                            \
    }
    for (; i <= dim - 64; i += 64) {
        INT4_BLOCK(s1, i)
        INT4_BLOCK(s1, i + 16)
        INT4_BLOCK(s2, i + 32)
        INT4_BLOCK(s2, i + 48)
    }
    for (; i <= dim - 32; i += 32) {
        INT4_BLOCK(s1, i)
        INT4_BLOCK(s1, i + 16)
    }
#undef INT4_BLOCK
    for (int l = 0; l < 16; l++) s1[l] = s1[l] + s2[l];
    float total = reduce16(s1);
    for (; i < dim; i++) { /* `for (; i < dim; i += 2)` with the i+1 < dim guard, element by element */
        float f = nib(code, i) * sc;
        float v = FMA(f, diff[i], min_val[i]);
        float d = query[i] - v;
        total = FMA(d, d, total);
    }
    return total;
}

void vgo_int4_l2_batch(const float *query, const uint8_t *codes, int64_t dim, int64_t n, const float *min_val,
                       const float *diff, float *out)
{
    int64_t cs = (dim + 1) / 2;
    for (int64_t j = 0; j < n; j++) out[j] = vgo_int4_l2(query, codes + j * cs, dim, min_val, diff);
}

/* internal/simd/kernels.go:94-103 BuildInt4LookupTable: (float32(q)/15.0)*diff + min, three
 * separately rounded fp32 operations (Go on amd64 does not fuse) */
void vgo_int4_build_lut(const float *min_val, const float *diff, int32_t dim, float *table)
{
    for (int d = 0; d < dim; d++)
        for (int q = 0; q < 16; q++) {
            float a = (float)q / 15.0f;
            float b = a * diff[d];
            table[d * 16 + q] = b + min_val[d];
        }
}

/* internal/simd/src/int4_avx512.c:127-189 int4L2DistancePrecomputedAvx512: one 16-lane accumulator
 * over 16-element blocks of table values, reduce tree, contracted scalar tail.  This is what
 * Int4Quantizer.L2Distance runs once the table exists (int4.go:140-147), i.e. the DiskANN node scorer. */
float vgo_int4_l2_precomputed(const float *query, const uint8_t *code, int64_t dim, const float *table)
{
    float sum[16];
    memset(sum, 0, sizeof sum);
    int64_t i = 0;
    for (; i <= dim - 16; i += 16)
        for (int l = 0; l < 16; l++) {
            int64_t j = i + l;
            float v = table[j * 16 + (int)nib(code, j)];
            float d = query[j] - v;
            sum[l] = FMA(d, d, sum[l]);
        }
    float total = reduce16(sum);
    for (; i < dim; i++) {
        float v = table[i * 16 + (int)nib(code, i)];
        float d = query[i] - v;
        total = FMA(d, d, total);
    }
    return total;
}

/* internal/quantization/int4.go:29-62 Train: min / max per dimension starting from vectors[0],
 * diff = max - min, 0 -> 1 */
void vgo_int4_train(const float *vectors, int64_t n, int32_t dim, float *min_val, float *diff)
{
    for (int d = 0; d < dim; d++) {
        float mn = vectors[d], mx = vectors[d];
        for (int64_t i = 1; i < n; i++) {
            float v = vectors[i * dim + d];
            if (v < mn) mn = v;
            if (v > mx) mx = v;
        }
        min_val[d] = mn;
        float df = mx - mn;
        diff[d] = df == 0.0f ? 1.0f : df;
    }
}

static uint8_t int4_quant(float v, float mn, float df)
{
    float norm = (v - mn) / df; /* int4.go:75-81 */
    if (norm < 0.0f) norm = 0.0f;
    else if (norm > 1.0f) norm = 1.0f;
    return (uint8_t)round((double)norm * 15.0); /* byte(math.Round(float64(norm) * 15)) */
}

/* int4.go:65-105 Encode: two dimensions per byte, the even one in the high nibble */
void vgo_int4_encode(const float *v, int32_t dim, const float *min_val, const float *diff, uint8_t *out)
{
    for (int i = 0; i < dim; i += 2) {
        uint8_t q1 = int4_quant(v[i], min_val[i], diff[i]);
        uint8_t q2 = i + 1 < dim ? int4_quant(v[i + 1], min_val[i + 1], diff[i + 1]) : 0;
        out[i / 2] = (uint8_t)((q1 << 4) | (q2 & 0x0F));
    }
}

/* int4.go:108-130 Decode: float32(q)/15.0*diff + min, left to right, no fusion */
void vgo_int4_decode(const uint8_t *code, int32_t dim, const float *min_val, const float *diff, float *out)
{
    for (int i = 0; i < dim; i++) {
        float a = nib(code, i) / 15.0f;
        float b = a * diff[i];
        out[i] = b + min_val[i];
    }
}

/* ------------------------------------------------------------------ */
/* construction-time neighbour selection         */
/* ------------------------------------------------------------------ */
static float provider_dist(const float *base, int32_t dim, int32_t metric, uint32_t a, uint32_t b)
{
    /* distance.Provider(metric) (distance.go:91-106): SquaredL2 or Dot */
    const float *va = base + (int64_t)a * dim, *vb = base + (int64_t)b * dim;
    return metric == VGO_METRIC_L2 ? vgo_l2_avx512(va, vb, dim) : vgo_dot_avx512(va, vb, dim);
}

/* diskann/writer.go:571-625 robustPrune.  `cands` is what the reference's `unique` map holds
 * (search results + current neighbours); duplicates, the node itself, 0xFFFFFFFF and ids >= n are
 * dropped here.  The reference sorts with slices.SortFunc on dist only, over a map iteration order
 * (non-deterministic among equal distances): ties are ordered by id here.  Returns the number of
 * selected ids written to out[0..r). */
int32_t vgo_robust_prune(const float *base, int64_t n, int32_t dim, int32_t metric, uint32_t node,
                         const uint32_t *cands, int32_t nc, int32_t r, float alpha, uint32_t *out)
{
    typedef struct { float d; uint32_t id; } dn;
    dn *c = (dn *)malloc(sizeof(dn) * (size_t)(nc > 0 ? nc : 1));
    int m = 0;
    for (int i = 0; i < nc; i++) {
        uint32_t id = cands[i];
        if (id == 0xFFFFFFFFu || (int64_t)id >= n || id == node) continue;
        int dup = 0;
        for (int j = 0; j < m; j++)
            if (c[j].id == id) { dup = 1; break; }
        if (dup) continue;
        c[m].id = id;
        c[m].d = provider_dist(base, dim, metric, id, node); /* w.dist(w.vectors[id], nodeVec) */
        m++;
    }
    for (int i = 1; i < m; i++) { /* ascending (dist, id) */
        dn x = c[i];
        int j = i - 1;
        while (j >= 0 && (c[j].d > x.d || (c[j].d == x.d && c[j].id > x.id))) { c[j + 1] = c[j]; j--; }
        c[j + 1] = x;
    }
    int sel = 0;
    for (int i = 0; i < m && sel < r; i++) {
        int diverse = 1;
        for (int s = 0; s < sel; s++) {
            float dcs = provider_dist(base, dim, metric, c[i].id, out[s]);
            float lhs = alpha * dcs;
            if (lhs < c[i].d) { diverse = 0; break; }
        }
        if (diverse) out[sel++] = c[i].id;
    }
    free(c);
    return sel;
}

/* hnsw.go:1009-1106 selectNeighborsHeuristic: candidates (nearest first, with their distance to the
 * source as the caller's queue held it) -> at most m neighbours: all of them if nc <= m; else the
 * relative-neighbourhood heuristic (applyHeuristic: keep a candidate unless it is closer to an
 * already kept one than to the source), then fillUpNeighbors in candidate order.  distanceFunc as
 * hnsw wraps it: L2 -> SquaredL2, Cosine -> 0.5 * SquaredL2, Dot -> -Dot. */
int32_t vgo_hnsw_select_neighbors(const float *base, int64_t n, int32_t dim, int32_t metric,
                                  const uint32_t *cand_ids, const float *cand_dists, int32_t nc, int32_t m,
                                  uint32_t *out)
{
    if (nc <= m) {
        for (int i = 0; i < nc; i++) out[i] = cand_ids[i];
        return nc;
    }
    int sel = 0;
    for (int i = 0; i < nc && sel < m; i++) {
        int good = 1;
        const float *cv = base + (int64_t)cand_ids[i] * dim;
        for (int s = 0; s < sel; s++) {
            const float *rv = base + (int64_t)out[s] * dim;
            float d = metric == VGO_METRIC_DOT ? -vgo_dot_avx512(cv, rv, dim) : vgo_l2_avx512(cv, rv, dim);
            if (metric == VGO_METRIC_COSINE) d = 0.5f * d;
            if (d < cand_dists[i]) { good = 0; break; }
        }
        if (good) out[sel++] = cand_ids[i];
    }
    for (int i = 0; i < nc && sel < m; i++) { /* fillUpNeighbors */
        int found = 0;
        for (int s = 0; s < sel; s++)
            if (out[s] == cand_ids[i]) { found = 1; break; }
        if (!found) out[sel++] = cand_ids[i];
    }
    (void)n;
    return sel;
}
