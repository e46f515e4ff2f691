// k_kmeans.hip — internal/kmeans (TrainKMeans, AssignPartition, FindClosestCentroids) plus the
// remaining batched L0 seams (SquaredL2Bounded, PqAdcLookup).
#include <algorithm>
#include <numeric>

#include "vg_device.hpp"
#include "vg_exact.hpp"
#include "vg_internal.hpp"

namespace vg {

__host__ __device__ inline uint64_t km_splitmix64(uint64_t x)
{
    x += 0x9e3779b97f4a7c15ULL;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL;
    return x ^ (x >> 31);
}
__host__ __device__ inline uint64_t km_rng(uint64_t seed, uint64_t a, uint64_t b, uint64_t c)
{
    uint64_t h = km_splitmix64(seed);
    h = km_splitmix64(h ^ a);
    h = km_splitmix64(h ^ b);
    h = km_splitmix64(h ^ c);
    return h;
}

// assignment (kmeans.go:54-99): 16 lanes per point, centroids visited in index order;
// SquaredL2Batch / DotBatch order (batch_avx512.c), strict comparison keeps the lowest index
template <bool DOT>
__global__ __launch_bounds__(256) void km_assign_kernel(const float *__restrict__ vectors, int64_t n, int dim,
                                                        const float *__restrict__ centroids, int k,
                                                        int32_t *__restrict__ assign, int *__restrict__ changed)
{
    const Sub16 sub = Sub16::make(threadIdx.x);
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 16 + (threadIdx.x >> 4);
    if (i >= n) return;
    const float *v = vectors + i * dim;
    int best = 0;
    float bd = exact_pair16<DOT, kBatch>(centroids, v, dim, sub);
    for (int c = 1; c < k; c++) {
        const float d = exact_pair16<DOT, kBatch>(centroids + static_cast<int64_t>(c) * dim, v, dim, sub);
        if (DOT ? (d > bd) : (d < bd)) {
            bd = d;
            best = c;
        }
    }
    if ((threadIdx.x & 15) == 0) {
        if (changed && assign[i] != best) *changed = 1;
        assign[i] = best;
    }
}

// member lists in index order: one thread per cluster walks the assignment array
__global__ void km_members_kernel(const int32_t *__restrict__ assign, int64_t n, int k,
                                  const int64_t *__restrict__ offsets, int64_t *__restrict__ members)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= k) return;
    int64_t at = offsets[c];
    for (int64_t i = 0; i < n; i++)
        if (assign[i] == c) members[at++] = i;
}

__global__ void km_count_kernel(const int32_t *__restrict__ assign, int64_t n, int k, int64_t *__restrict__ counts)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= k) return;
    int64_t cnt = 0;
    for (int64_t i = 0; i < n; i++) cnt += assign[i] == c;
    counts[c] = cnt;
}

// update (kmeans.go:107-135): per (cluster, coordinate) the sum runs over the members in index
// order (= the reference's single loop over i), then sums * (1/count)
__global__ __launch_bounds__(256) void km_update_kernel(const float *__restrict__ vectors, int64_t n, int dim,
                                                        int k, int iter, uint64_t seed,
                                                        const int64_t *__restrict__ counts,
                                                        const int64_t *__restrict__ offsets,
                                                        const int64_t *__restrict__ members,
                                                        float *__restrict__ centroids)
{
    const int c = blockIdx.y;
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= dim) return;
    const int64_t cnt = counts[c];
    float *dst = centroids + static_cast<int64_t>(c) * dim + d;
    if (cnt > 0) {
        const int64_t *mem = members + offsets[c];
        float sum = 0.0f;
        for (int64_t j = 0; j < cnt; j++) sum += vectors[mem[j] * dim + d];
        const float scale = 1.0f / static_cast<float>(cnt);
        *dst = sum * scale;
    } else {
        const int64_t idx = static_cast<int64_t>(km_rng(seed, 0, 2 + static_cast<uint64_t>(iter), c) %
                                                 static_cast<uint64_t>(n));
        *dst = vectors[idx * dim + d];
    }
}

__global__ void km_gather_rows_kernel(const float *__restrict__ vectors, int dim, const int64_t *__restrict__ rows,
                                      int k, float *__restrict__ out)
{
    const int64_t gid = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (gid >= static_cast<int64_t>(k) * dim) return;
    out[gid] = vectors[rows[gid / dim] * dim + gid % dim];
}

__global__ __launch_bounds__(256) void bounded_batch_kernel(const float *__restrict__ query,
                                                            const float *__restrict__ targets, int dim, int64_t n,
                                                            const float *__restrict__ bounds, int64_t n_bounds,
                                                            float *__restrict__ dist, int32_t *__restrict__ exceeded)
{
    const Sub16 sub = Sub16::make(threadIdx.x);
    const int64_t groups = static_cast<int64_t>(gridDim.x) * 16;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * 16 + (threadIdx.x >> 4); i < n; i += groups) {
        const float v = exact_pair16<false, kBounded>(targets + i * dim, query, dim, sub);
        if ((threadIdx.x & 15) == 0) {
            const float b = bounds[n_bounds == 1 ? 0 : i];
            dist[i] = v;
            exceeded[i] = v > b ? 1 : 0;
        }
    }
}

// pqAdcLookupAvx512 (floats_avx512.c:135-167): thread per code row
__global__ void adc_lookup_batch_kernel(const float *__restrict__ table, const uint8_t *__restrict__ codes, int m,
                                        int64_t n, float *__restrict__ out)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t *c = codes + i * m;
    float acc[16];
#pragma unroll
    for (int l = 0; l < 16; l++) acc[l] = 0.0f;
    int j = 0;
    for (; j + 16 <= m; j += 16) {
#pragma unroll
        for (int l = 0; l < 16; l++) acc[l] = acc[l] + table[(j + l) * 256 + c[j + l]];
    }
    float total = reduce16_regs(acc);
    for (; j < m; j++) total = total + table[j * 256 + c[j]];
    out[i] = total;
}

}  // namespace vg

static bool km_metric_ok(int32_t metric) { return metric == VG_METRIC_L2 || metric == VG_METRIC_DOT || metric == VG_METRIC_COSINE; }

VG_API int32_t vg_kmeans_assign(vg_ctx *ctx, const float *vectors, int64_t n, int32_t dim, const float *centroids,
                                int32_t k, int32_t metric, int32_t *out, void *stream)
{
    VG_CHECK(ctx, VG_ERR_INVALID_ARG, "vg_kmeans_assign: ctx is NULL");
    VG_CHECK(km_metric_ok(metric), VG_ERR_UNSUPPORTED, "unsupported metric for float32: %d", metric);
    VG_CHECK(n >= 0 && dim > 0 && k > 0, VG_ERR_INVALID_ARG, "vg_kmeans_assign: bad sizes");
    if (n == 0) return VG_OK;
    VG_CHECK(vectors && centroids && out, VG_ERR_INVALID_ARG, "vg_kmeans_assign: NULL buffer");
    VG_HIP(hipSetDevice(ctx->device));
    hipStream_t st = vg::pick_stream(ctx, stream);
    vg::DevIn<float> v, c;
    vg::DevOut<int32_t> o;
    VG_TRY(v.init(vectors, static_cast<size_t>(n) * dim, st));
    VG_TRY(c.init(centroids, static_cast<size_t>(k) * dim, st));
    VG_TRY(o.init(out, static_cast<size_t>(n), st));
    const unsigned gx = static_cast<unsigned>((n + 15) / 16);
    if (metric == VG_METRIC_L2)
        VG_LAUNCH(vg::km_assign_kernel<false>, dim3(gx), dim3(256), 0, st, v.ptr, n, dim, c.ptr, k, o.ptr,
                           static_cast<int *>(nullptr));
    else
        VG_LAUNCH(vg::km_assign_kernel<true>, dim3(gx), dim3(256), 0, st, v.ptr, n, dim, c.ptr, k, o.ptr,
                           static_cast<int *>(nullptr));
    VG_TRY(o.finish());
    if (o.on_host()) VG_HIP(hipStreamSynchronize(st));
    return VG_OK;
}

VG_API int32_t vg_kmeans_train(vg_ctx *ctx, const float *vectors, int64_t n, int32_t dim, int32_t k, int32_t metric,
                               int32_t max_iter, uint64_t seed, float *centroids, int32_t *produced, void *stream)
{
    VG_CHECK(ctx, VG_ERR_INVALID_ARG, "vg_kmeans_train: ctx is NULL");
    if (produced) *produced = 0;
    VG_CHECK(dim > 0 && k > 0 && n >= 0 && max_iter >= 0, VG_ERR_INVALID_ARG, "vg_kmeans_train: bad sizes");
    if (n < k) return VG_OK;  // kmeans.go:17-20: not enough vectors to cluster
    VG_CHECK(km_metric_ok(metric), VG_ERR_UNSUPPORTED, "unsupported metric for float32: %d", metric);
    VG_CHECK(vectors && centroids, VG_ERR_INVALID_ARG, "vg_kmeans_train: NULL buffer");
    VG_HIP(hipSetDevice(ctx->device));
    hipStream_t st = vg::pick_stream(ctx, stream);
    vg::DevIn<float> v;
    vg::DevOut<float> cent;
    VG_TRY(v.init(vectors, static_cast<size_t>(n) * dim, st));
    VG_TRY(cent.init(centroids, static_cast<size_t>(k) * dim, st));
    // first k entries of the permutation (partial Fisher-Yates over the counter stream)
    std::vector<int64_t> perm(static_cast<size_t>(n));
    std::iota(perm.begin(), perm.end(), int64_t(0));
    for (int64_t i = 0; i < k && i < n - 1; i++) {
        const int64_t j = i + static_cast<int64_t>(vg::km_rng(seed, 0, 1, static_cast<uint64_t>(i)) %
                                                   static_cast<uint64_t>(n - i));
        std::swap(perm[static_cast<size_t>(i)], perm[static_cast<size_t>(j)]);
    }
    vg::DevTmp<int64_t> rows, counts, offsets, members;
    vg::DevTmp<int32_t> assign;
    vg::DevTmp<int> changed;
    VG_TRY(rows.init(static_cast<size_t>(k), st));
    VG_TRY(counts.init(static_cast<size_t>(k), st));
    VG_TRY(offsets.init(static_cast<size_t>(k), st));
    VG_TRY(members.init(static_cast<size_t>(n), st));
    VG_TRY(assign.init(static_cast<size_t>(n), st));
    VG_TRY(changed.init(1, st));
    VG_HIP(hipMemcpyAsync(rows.ptr, perm.data(), sizeof(int64_t) * k, hipMemcpyHostToDevice, st));
    VG_HIP(hipStreamSynchronize(st));  // perm is a local
    const int64_t tot = static_cast<int64_t>(k) * dim;
    VG_LAUNCH(vg::km_gather_rows_kernel, dim3(static_cast<unsigned>((tot + 255) / 256)), dim3(256), 0, st,
                       v.ptr, dim, rows.ptr, k, cent.ptr);
    VG_HIP(hipMemsetAsync(assign.ptr, 0, sizeof(int32_t) * static_cast<size_t>(n), st));
    std::vector<int64_t> hcounts(static_cast<size_t>(k)), hoff(static_cast<size_t>(k));
    const unsigned gx = static_cast<unsigned>((n + 15) / 16);
    const unsigned kx = static_cast<unsigned>((k + 63) / 64);
    for (int it = 0; it < max_iter; it++) {
        VG_HIP(hipMemsetAsync(changed.ptr, 0, sizeof(int), st));
        if (metric == VG_METRIC_L2)
            VG_LAUNCH(vg::km_assign_kernel<false>, dim3(gx), dim3(256), 0, st, v.ptr, n, dim, cent.ptr, k,
                               assign.ptr, changed.ptr);
        else
            VG_LAUNCH(vg::km_assign_kernel<true>, dim3(gx), dim3(256), 0, st, v.ptr, n, dim, cent.ptr, k,
                               assign.ptr, changed.ptr);
        int hchanged = 0;
        VG_HIP(hipMemcpyAsync(&hchanged, changed.ptr, sizeof(int), hipMemcpyDeviceToHost, st));
        VG_LAUNCH(vg::km_count_kernel, dim3(kx), dim3(64), 0, st, assign.ptr, n, k, counts.ptr);
        VG_HIP(hipMemcpyAsync(hcounts.data(), counts.ptr, sizeof(int64_t) * k, hipMemcpyDeviceToHost, st));
        VG_HIP(hipStreamSynchronize(st));  // training is not a hot path: one sync per Lloyd iteration
        if (!hchanged) break;               // kmeans.go:101-103
        int64_t run = 0;
        for (int c = 0; c < k; c++) {
            hoff[static_cast<size_t>(c)] = run;
            run += hcounts[static_cast<size_t>(c)];
        }
        VG_HIP(hipMemcpyAsync(offsets.ptr, hoff.data(), sizeof(int64_t) * k, hipMemcpyHostToDevice, st));
        VG_LAUNCH(vg::km_members_kernel, dim3(kx), dim3(64), 0, st, assign.ptr, n, k, offsets.ptr,
                           members.ptr);
        VG_LAUNCH(vg::km_update_kernel, dim3(static_cast<unsigned>((dim + 255) / 256), k), dim3(256), 0, st,
                           v.ptr, n, dim, k, it, seed, counts.ptr, offsets.ptr, members.ptr, cent.ptr);
        VG_HIP(hipStreamSynchronize(st));  // hoff is reused next iteration
    }
    VG_TRY(cent.finish());
    VG_HIP(hipStreamSynchronize(st));
    if (produced) *produced = 1;
    return VG_OK;
}

VG_API int32_t vg_find_closest_centroids(vg_ctx *ctx, const float *query, const float *centroids, int32_t dim,
                                         int32_t k, int32_t nprobe, int32_t metric, int32_t *out, int32_t *n_out,
                                         void *stream)
{
    VG_CHECK(ctx, VG_ERR_INVALID_ARG, "vg_find_closest_centroids: ctx is NULL");
    if (n_out) *n_out = 0;
    VG_CHECK(km_metric_ok(metric), VG_ERR_UNSUPPORTED, "unsupported metric for float32: %d", metric);
    VG_CHECK(dim > 0 && k >= 0 && nprobe >= 0, VG_ERR_INVALID_ARG, "vg_find_closest_centroids: bad sizes");
    if (k == 0 || nprobe == 0) return VG_OK;
    VG_CHECK(query && centroids && out, VG_ERR_INVALID_ARG, "vg_find_closest_centroids: NULL buffer");
    int n = nprobe > k ? k : nprobe;
    std::vector<float> d(static_cast<size_t>(k));
    int32_t s = metric == VG_METRIC_L2 ? vg_squared_l2_batch(ctx, query, centroids, dim, k, d.data(), stream)
                                       : vg_dot_batch(ctx, query, centroids, dim, k, d.data(), stream);
    if (s != VG_OK) return s;
    if (metric != VG_METRIC_L2)
        for (auto &x : d) x = -x;  // kmeans.go:238-241
    std::vector<int32_t> id(static_cast<size_t>(k));
    std::iota(id.begin(), id.end(), 0);
    if (n <= k / 4 && n < 16) {  // kmeans.go:254-268 selection
        for (int i = 0; i < n; i++) {
            int mi = i;
            for (int j = i + 1; j < k; j++)
                if (d[static_cast<size_t>(j)] < d[static_cast<size_t>(mi)]) mi = j;
            std::swap(d[static_cast<size_t>(i)], d[static_cast<size_t>(mi)]);
            std::swap(id[static_cast<size_t>(i)], id[static_cast<size_t>(mi)]);
            out[i] = id[static_cast<size_t>(i)];
        }
    } else {  // kmeans.go:271-278 full sort (pdqsort in the reference: ties unpinned; here by index)
        std::vector<int32_t> order(id);
        std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
            return d[static_cast<size_t>(a)] < d[static_cast<size_t>(b)];
        });
        for (int i = 0; i < n; i++) out[i] = order[static_cast<size_t>(i)];
    }
    if (n_out) *n_out = n;
    return VG_OK;
}

VG_API int32_t vg_squared_l2_bounded_batch(vg_ctx *ctx, const float *query, const float *targets, int64_t dim,
                                           int64_t n, const float *bounds, int64_t n_bounds, float *dist,
                                           int32_t *exceeded, void *stream)
{
    VG_CHECK(ctx, VG_ERR_INVALID_ARG, "vg_squared_l2_bounded_batch: ctx is NULL");
    if (n <= 0) return VG_OK;
    VG_CHECK(dim >= 0 && dim < (1 << 30), VG_ERR_INVALID_ARG, "vg_squared_l2_bounded_batch: bad dim");
    VG_CHECK(bounds && dist && exceeded && (n_bounds == 1 || n_bounds == n), VG_ERR_INVALID_ARG,
             "vg_squared_l2_bounded_batch: bounds must have 1 or n entries");
    VG_CHECK(dim == 0 || (query && targets), VG_ERR_INVALID_ARG, "vg_squared_l2_bounded_batch: NULL buffer");
    VG_HIP(hipSetDevice(ctx->device));
    hipStream_t st = vg::pick_stream(ctx, stream);
    vg::DevIn<float> q, t, b;
    vg::DevOut<float> od;
    vg::DevOut<int32_t> oe;
    VG_TRY(q.init(query, static_cast<size_t>(dim), st));
    VG_TRY(t.init(targets, static_cast<size_t>(n) * dim, st));
    VG_TRY(b.init(bounds, static_cast<size_t>(n_bounds), st));
    VG_TRY(od.init(dist, static_cast<size_t>(n), st));
    VG_TRY(oe.init(exceeded, static_cast<size_t>(n), st));
    int64_t blocks = std::min<int64_t>((n + 15) / 16, 4096);
    VG_LAUNCH(vg::bounded_batch_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, st, q.ptr, t.ptr,
                       static_cast<int>(dim), n, b.ptr, n_bounds, od.ptr, oe.ptr);
    VG_TRY(od.finish());
    VG_TRY(oe.finish());
    if (od.on_host() || oe.on_host()) VG_HIP(hipStreamSynchronize(st));
    return VG_OK;
}

VG_API int32_t vg_pq_adc_lookup_batch(vg_ctx *ctx, const float *table, const uint8_t *codes, int64_t m, int64_t n,
                                      float *out, void *stream)
{
    VG_CHECK(ctx, VG_ERR_INVALID_ARG, "vg_pq_adc_lookup_batch: ctx is NULL");
    VG_CHECK(m >= 0 && n >= 0 && m < (1 << 20), VG_ERR_INVALID_ARG, "vg_pq_adc_lookup_batch: bad sizes");
    if (n == 0) return VG_OK;
    VG_CHECK(out && (m == 0 || (table && codes)), VG_ERR_INVALID_ARG, "vg_pq_adc_lookup_batch: NULL buffer");
    VG_HIP(hipSetDevice(ctx->device));
    hipStream_t st = vg::pick_stream(ctx, stream);
    vg::DevIn<float> t;
    vg::DevIn<uint8_t> c;
    vg::DevOut<float> o;
    VG_TRY(t.init(table, static_cast<size_t>(m) * 256, st));
    VG_TRY(c.init(codes, static_cast<size_t>(n) * m, st));
    VG_TRY(o.init(out, static_cast<size_t>(n), st));
    VG_LAUNCH(vg::adc_lookup_batch_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, st,
                       t.ptr, c.ptr, static_cast<int>(m), n, o.ptr);
    VG_TRY(o.finish());
    if (o.on_host()) VG_HIP(hipStreamSynchronize(st));
    return VG_OK;
}
