/*
 * vg_oracle_opq.c — CPU restatement of OptimizedProductQuantizer (TEST INFRASTRUCTURE ONLY, see vg_oracle.h):
 *   internal/quantization/opq.go   block size :38-58, Train :89-193, rotateVector :196-215, Encode :218-231,
 *                                  Decode :234-269, ComputeAsymmetricDistance :272-286
 *   internal/quantization/svd.go   one-sided Jacobi svd :13-58, performJacobiIterations :60-91, applyRotation
 *                                  :93-119, computeProcrustesRotation :126-178, determinant :180-224
 * Go on amd64 does not fuse a*b+c: every multiply and add below is separately rounded (-ffp-contract=off);
 * simd.Dot inside rotateVector is dotProductAvx512's order (vgo_dot_avx512).
 * PARITY UNPINNED against the reference binary: Train calls ProductQuantizer.Train, which draws from the
 * unseeded global math/rand — the codebooks (and with them the rotations) are not reproducible in the
 * reference itself.  Pinned here: everything given the rotations and codebooks, and the Procrustes solver given M.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "vg_oracle.h"

/* opq.go:38-58: blocks are multiples of the sub-vector size that divide the dimension, nearest to 32 */
int32_t vgo_opq_block_size(int32_t dim, int32_t m)
{
    const int32_t sub = dim / m;
    int32_t block = dim;
    if (dim > 64) {
        int32_t best = 1000;
        for (int32_t b = sub; b <= dim; b += sub)
            if (dim % b == 0) {
                int32_t diff = b > 32 ? b - 32 : 32 - b;
                if (diff < best) {
                    best = diff;
                    block = b;
                }
            }
    }
    return block;
}

/* rot: [nblocks][block][block] row-major.  rotateVector opq.go:196-215 */
void vgo_opq_rotate(const float *rot, int32_t dim, int32_t block, const float *src, float *dst)
{
    for (int32_t b = 0; b * block < dim; b++) {
        const float *r = rot + (int64_t)b * block * block;
        const float *s = src + (int64_t)b * block;
        for (int32_t i = 0; i < block; i++) dst[b * block + i] = vgo_dot_avx512(r + (int64_t)i * block, s, block);
    }
}

/* the inverse rotation of Decode opq.go:246-266: dst[i] = sum_j rot[j][i] * src[j], left to right */
void vgo_opq_unrotate(const float *rot, int32_t dim, int32_t block, const float *src, float *dst)
{
    for (int32_t b = 0; b * block < dim; b++) {
        const float *r = rot + (int64_t)b * block * block;
        const float *s = src + (int64_t)b * block;
        for (int32_t i = 0; i < block; i++) {
            float sum = 0.0f;
            for (int32_t j = 0; j < block; j++) {
                float p = r[(int64_t)j * block + i] * s[j];
                sum = sum + p;
            }
            dst[b * block + i] = sum;
        }
    }
}

/* svd.go:60-119 */
static int jacobi_sweep(float *u, float *v, int m, int n, double tol)
{
    int changed = 0;
    for (int i = 0; i < n - 1; i++)
        for (int j = i + 1; j < n; j++) {
            float alpha = 0.0f, beta = 0.0f, gamma = 0.0f;
            for (int k = 0; k < m; k++) {
                float a = u[k * n + i], b = u[k * n + j];
                float aa = a * a, bb = b * b, ab = a * b;
                alpha = alpha + aa;
                beta = beta + bb;
                gamma = gamma + ab;
            }
            if (alpha < 1e-12f || beta < 1e-12f) continue;
            float ab = alpha * beta;
            if (fabs((double)gamma) < tol * sqrt((double)ab)) continue;
            changed = 1;
            float num = beta - alpha, den = 2.0f * gamma;
            float zeta = num / den;
            float zz = zeta * zeta;
            float one_zz = 1.0f + zz;
            float root = (float)sqrt((double)one_zz);
            float t;
            if (zeta > 0.0f)
                t = 1.0f / (zeta + root);
            else
                t = -1.0f / (-zeta + root);
            float tt = t * t;
            float c = 1.0f / (float)sqrt((double)(1.0f + tt));
            float s = c * t;
            for (int k = 0; k < m; k++) {
                float t1 = u[k * n + i], t2 = u[k * n + j];
                float c1 = c * t1, s2 = s * t2, s1 = s * t1, c2 = c * t2;
                u[k * n + i] = c1 - s2;
                u[k * n + j] = s1 + c2;
            }
            for (int k = 0; k < n; k++) {
                float t1 = v[k * n + i], t2 = v[k * n + j];
                float c1 = c * t1, s2 = s * t2, s1 = s * t1, c2 = c * t2;
                v[k * n + i] = c1 - s2;
                v[k * n + j] = s1 + c2;
            }
        }
    return changed;
}

/* svd.go:180-224 */
static float determinant(const float *matrix, int n)
{
    if (n == 0) return 0.0f;
    float *t = (float *)malloc(sizeof(float) * (size_t)n * n);
    memcpy(t, matrix, sizeof(float) * (size_t)n * n);
    int *rowp = (int *)malloc(sizeof(int) * (size_t)n);
    for (int i = 0; i < n; i++) rowp[i] = i;
    float det = 1.0f;
    for (int i = 0; i < n; i++) {
        int pivot = i;
        for (int j = i + 1; j < n; j++)
            if (fabs((double)t[rowp[j] * n + i]) > fabs((double)t[rowp[pivot] * n + i])) pivot = j;
        if (pivot != i) {
            int tmp = rowp[i];
            rowp[i] = rowp[pivot];
            rowp[pivot] = tmp;
            det = det * -1.0f;
        }
        float *ri = t + (int64_t)rowp[i] * n;
        if (ri[i] == 0.0f) {
            det = 0.0f;
            break;
        }
        det = det * ri[i];
        for (int j = i + 1; j < n; j++) {
            float *rj = t + (int64_t)rowp[j] * n;
            float factor = rj[i] / ri[i];
            for (int k = i + 1; k < n; k++) {
                float p = factor * ri[k];
                rj[k] = rj[k] - p;
            }
        }
    }
    free(rowp);
    free(t);
    return det;
}

/* computeProcrustesRotation svd.go:126-178: M (n x n, row-major, DESTROYED: it becomes U) -> R = U V^T, with the
 * column of U of the smallest singular value flipped when det(R) < 0 */
void vgo_procrustes(float *mm, int32_t n, float *r)
{
    float *v = (float *)calloc((size_t)n * n, sizeof(float));
    float *sigma = (float *)malloc(sizeof(float) * (size_t)n);
    for (int i = 0; i < n; i++) v[i * n + i] = 1.0f;
    float *u = mm;
    for (int it = 0; it < 100; it++)
        if (!jacobi_sweep(u, v, n, n, 1e-5)) break;
    for (int j = 0; j < n; j++) { /* svd.go:41-55 */
        float sum = 0.0f;
        for (int i = 0; i < n; i++) {
            float p = u[i * n + j] * u[i * n + j];
            sum = sum + p;
        }
        sigma[j] = (float)sqrt((double)sum);
        if (sigma[j] > 1e-10f) {
            float inv = 1.0f / sigma[j];
            for (int i = 0; i < n; i++) u[i * n + j] = u[i * n + j] * inv;
        }
    }
    int min_idx = 0;
    float min_sigma = sigma[0];
    for (int i = 1; i < n; i++)
        if (sigma[i] < min_sigma) {
            min_sigma = sigma[i];
            min_idx = i;
        }
    for (int pass = 0; pass < 2; pass++) {
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) {
                float sum = 0.0f;
                for (int k = 0; k < n; k++) {
                    float p = u[i * n + k] * v[j * n + k];
                    sum = sum + p;
                }
                r[i * n + j] = sum;
            }
        if (pass == 1 || !(determinant(r, n) < 0.0f)) break;
        for (int i = 0; i < n; i++) u[i * n + min_idx] = u[i * n + min_idx] * -1.0f;
    }
    free(sigma);
    free(v);
}

/* M_b[r][c] = sum_i x_i[b*block + r] * y_i[b*block + c], i in order, mul and add separately rounded (opq.go:150-176) */
void vgo_opq_accumulate_m(const float *x, const float *y, int64_t n, int32_t dim, int32_t block, float *m_out)
{
    const int nb = dim / block;
    memset(m_out, 0, sizeof(float) * (size_t)nb * block * block);
    for (int64_t i = 0; i < n; i++)
        for (int b = 0; b < nb; b++) {
            const float *xb = x + i * dim + (int64_t)b * block, *yb = y + i * dim + (int64_t)b * block;
            float *mb = m_out + (int64_t)b * block * block;
            for (int r = 0; r < block; r++) {
                float xr = xb[r];
                for (int c = 0; c < block; c++) {
                    float p = xr * yb[c];
                    mb[r * block + c] = mb[r * block + c] + p;
                }
            }
        }
}

/* Train opq.go:89-193.  Every outer iteration re-trains the PQ from scratch on the rotated vectors (seeded
 * stream: seed + iteration), then re-solves every block's rotation; the final state is the last rotations with
 * the codebooks trained under the previous ones, exactly as the reference leaves it. */
int vgo_opq_train(const float *vectors, int64_t n, int32_t dim, int32_t m, int32_t k, int32_t opq_iters,
                  int32_t pq_iters, uint64_t seed, float *rot, int8_t *codebooks, float *scales, float *offsets)
{
    if (n <= 0 || dim <= 0 || m <= 0 || dim % m) return -1;
    const int32_t block = vgo_opq_block_size(dim, m);
    const int nb = dim / block;
    memset(rot, 0, sizeof(float) * (size_t)nb * block * block);
    for (int b = 0; b < nb; b++)
        for (int i = 0; i < block; i++) rot[((int64_t)b * block + i) * block + i] = 1.0f;
    float *rotated = (float *)malloc(sizeof(float) * (size_t)n * dim);
    float *recon = (float *)malloc(sizeof(float) * (size_t)n * dim);
    float *mm = (float *)malloc(sizeof(float) * (size_t)nb * block * block);
    uint8_t *code = (uint8_t *)malloc((size_t)m);
    vgo_pq pq = {dim, m, k, dim / m, codebooks, scales, offsets};
    int rc = 0;
    for (int it = 0; it < opq_iters && rc == 0; it++) {
        for (int64_t i = 0; i < n; i++) vgo_opq_rotate(rot, dim, block, vectors + i * dim, rotated + i * dim);
        rc = vgo_pq_train(rotated, n, dim, m, k, pq_iters, seed + (uint64_t)it, codebooks, scales, offsets, NULL);
        if (rc) break;
        for (int64_t i = 0; i < n; i++) {
            vgo_pq_encode(&pq, rotated + i * dim, code);
            vgo_pq_decode(&pq, code, recon + i * dim);
        }
        vgo_opq_accumulate_m(vectors, recon, n, dim, block, mm);
        for (int b = 0; b < nb; b++)
            vgo_procrustes(mm + (int64_t)b * block * block, block, rot + (int64_t)b * block * block);
    }
    free(code);
    free(mm);
    free(recon);
    free(rotated);
    return rc;
}
