// k_probe.hip — flat.Segment.Search over a partitioned (IVF) flat segment
// (internal/segment/flat/segment.go:727-749): with more than one partition the reference scans only
// the row ranges of the nprobes centroids closest to the query (kmeans.FindClosestCentroids,
// kmeans.go:217-280; nprobes <= 0 means 1), every range into the same bounded heap.  The heap order
// is total — (score, row id) — so the result is the k best keys of the union of the probed ranges:
//   1. per query: distances to the P centroids in SquaredL2Batch / DotBatch order, the nprobes best
//   2. per (query, probe[, slice]): scan of that partition's rows with the segment's scan type
//      (fp32 pair order here; PQ table lookups k_adc.hip; SQ8 k_sq8.hip), a k-list each
//   3. one merge of the lists per query
#include <algorithm>

#include "vg_device.hpp"
#include "vg_exact.hpp"
#include "vg_internal.hpp"

namespace vg {

int32_t launch_topk_merge(const uint64_t *partial, int64_t nq, int lists, int k, bool descending,
                          uint32_t *ids, float *scores, hipStream_t st, const int *only_if = nullptr,
                          const int *always = nullptr);
int32_t launch_pq_build_table(const vg_pq *pq, const float *d_queries, int64_t nq, float *d_tables,
                              bool scan_layout, hipStream_t st);
int32_t launch_probe_scan_adc(const vg_index *idx, const float *tables, const uint32_t *probes, int64_t nq, int np,
                              int split, int k, uint64_t *partial, hipStream_t st);
int32_t launch_probe_scan_sq8(const vg_index *idx, const float *queries, const uint32_t *probes, int64_t nq, int np,
                              int sub, int k, uint64_t *partial, hipStream_t st);

// ---- 1. the nprobes closest centroids (kmeans.go:217-280) -----------------------------------------
// 16 lanes per centroid, batch-kernel order; for Dot / Cosine the reference sorts -dot ascending,
// i.e. the largest dot products first, which is the DOT key order.
template <bool DOT>
__global__ __launch_bounds__(256) void probe_select_kernel(const float *__restrict__ queries, int dim,
                                                           const float *__restrict__ centroids, int parts, int np,
                                                           uint32_t *__restrict__ probes)
{
    __shared__ uint64_t lists[4 * 64];
    __shared__ int valid[4];
    __shared__ uint64_t best[64];
    const int64_t q = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const Sub16 sub = Sub16::make(tid);
    const float *qv = queries + q * dim;
    WaveTopK tk;
    tk.init(np);
    for (int c0 = wave * 4; c0 < parts; c0 += 16) {
        const int c = c0 + (lane >> 4);
        uint64_t key = kKeyMax;
        if (c < parts) {
            const float v = exact_pair16<DOT, kBatch>(centroids + static_cast<int64_t>(c) * dim, qv, dim, sub);
            if ((lane & 15) == 0) key = make_key(v, static_cast<uint32_t>(c), DOT);
        }
        tk.offer(key, lane);
    }
    wg_rank_merge<4>(tk, lists, valid, wave, lane, tid, np, best);
    __syncthreads();
    if (tid < np) probes[q * np + tid] = key_row(best[tid]);
}

// ---- 2. fp32 scan of one probed partition (segment.go:691-701) ------------------------------------
template <bool DOT>
__global__ __launch_bounds__(256) void probe_scan_f32_kernel(const float *__restrict__ base, int dim,
                                                             const float *__restrict__ queries,
                                                             const uint32_t *__restrict__ probes,
                                                             const uint32_t *__restrict__ part_off, int np, int sub_n,
                                                             int k, uint64_t *__restrict__ partial)
{
    __shared__ uint64_t lists[4 * 64];
    __shared__ int valid[4];
    const int s = blockIdx.x, j = blockIdx.y;
    const int64_t q = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const Sub16 sub = Sub16::make(tid);
    const uint32_t p = probes[q * np + j];
    const int64_t R0 = part_off[p], R1 = part_off[p + 1];
    const int64_t r0 = R0 + (R1 - R0) * s / sub_n, r1 = R0 + (R1 - R0) * (s + 1) / sub_n;
    const float *qv = queries + q * dim;
    WaveTopK tk;
    tk.init(k);
    for (int64_t i0 = r0 + wave * 4; i0 < r1; i0 += 16) {  // 4 rows per wave step, one per 16-lane group
        const int64_t i = i0 + (lane >> 4);
        uint64_t key = kKeyMax;
        if (i < r1) {
            const float v = exact_pair16<DOT, kPair>(base + i * dim, qv, dim, sub);
            if ((lane & 15) == 0) key = make_key(v, static_cast<uint32_t>(i), DOT);
        }
        tk.offer(key, lane);
    }
    wg_rank_merge<4>(tk, lists, valid, wave, lane, tid, k, partial + ((q * np + j) * sub_n + s) * k);
}

}  // namespace vg

VG_API int32_t vg_index_set_partitions(vg_index *idx, const float *centroids, const uint32_t *part_offsets,
                                       int32_t num_partitions, void *stream)
{
    VG_CHECK(idx, VG_ERR_INVALID_ARG, "vg_index_set_partitions: NULL index");
    VG_CHECK(num_partitions >= 0, VG_ERR_INVALID_ARG, "vg_index_set_partitions: negative partition count");
    VG_HIP(hipSetDevice(idx->ctx->device));
    hipStream_t st = vg::pick_stream(idx->ctx, stream);
    VG_HIP(hipStreamSynchronize(st));  // earlier searches may still read the old tables
    if (idx->d_centroids) VG_HIP(hipFree(idx->d_centroids));
    if (idx->d_part_off) VG_HIP(hipFree(idx->d_part_off));
    idx->d_centroids = nullptr;
    idx->d_part_off = nullptr;
    idx->num_partitions = 0;
    if (num_partitions == 0) return VG_OK;
    VG_CHECK(centroids && part_offsets, VG_ERR_INVALID_ARG, "vg_index_set_partitions: NULL buffer");
    // the offsets come from a file: check them once here instead of in every scan
    std::vector<uint32_t> off(static_cast<size_t>(num_partitions) + 1);
    hipPointerAttribute_t attr;
    const bool on_device = hipPointerGetAttributes(&attr, part_offsets) == hipSuccess && attr.type == hipMemoryTypeDevice;
    if (!on_device) (void)hipGetLastError();
    VG_HIP(hipMemcpy(off.data(), part_offsets, off.size() * sizeof(uint32_t), hipMemcpyDefault));
    for (size_t p = 0; p + 1 < off.size(); p++)
        VG_CHECK(off[p] <= off[p + 1], VG_ERR_INVALID_ARG, "vg_index_set_partitions: partition offsets decrease at %zu", p);
    VG_CHECK(static_cast<int64_t>(off.back()) <= idx->n, VG_ERR_INVALID_ARG,
             "vg_index_set_partitions: partition offsets end at %u, the index has %lld rows", off.back(),
             static_cast<long long>(idx->n));
    const size_t cbytes = sizeof(float) * static_cast<size_t>(num_partitions) * idx->dim;
    VG_HIP(hipMalloc(reinterpret_cast<void **>(&idx->d_centroids), cbytes));
    VG_HIP(hipMalloc(reinterpret_cast<void **>(&idx->d_part_off), off.size() * sizeof(uint32_t)));
    VG_HIP(hipMemcpyAsync(idx->d_centroids, centroids, cbytes, hipMemcpyDefault, st));
    VG_HIP(hipMemcpyAsync(idx->d_part_off, off.data(), off.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    VG_HIP(hipStreamSynchronize(st));  // `off` is a local; the caller's buffers are free again
    idx->num_partitions = num_partitions;
    return VG_OK;
}

VG_API int32_t vg_search_flat_probed(vg_index *idx, const float *queries, int64_t nq, int32_t k, int32_t nprobes,
                                     int32_t scan, uint32_t *ids, float *scores, void *stream)
{
    VG_CHECK(idx, VG_ERR_INVALID_ARG, "vg_search_flat_probed: NULL index");
    VG_CHECK(scan == VG_SCAN_F32 || scan == VG_SCAN_PQ || scan == VG_SCAN_SQ8, VG_ERR_INVALID_ARG,
             "vg_search_flat_probed: unknown scan type %d", scan);
    if (idx->num_partitions <= 1) {  // segment.go:745-749: one range, the whole segment
        if (scan == VG_SCAN_PQ) return vg_search_pq_adc(idx, queries, nq, k, ids, scores, stream);
        if (scan == VG_SCAN_SQ8) return vg_search_sq8(idx, queries, nq, k, ids, scores, stream);
        return vg_search_flat(idx, queries, nq, k, ids, scores, stream);
    }
    VG_CHECK(nq >= 0 && k >= 0, VG_ERR_INVALID_ARG, "vg_search_flat_probed: negative nq or k");
    if (nq == 0 || k == 0) return VG_OK;
    VG_CHECK(queries && ids && scores, VG_ERR_INVALID_ARG, "vg_search_flat_probed: NULL buffer");
    VG_CHECK(k <= 64, VG_ERR_UNSUPPORTED, "vg_search_flat_probed: k=%d exceeds 64", k);
    VG_CHECK(idx->metric != VG_METRIC_HAMMING, VG_ERR_UNSUPPORTED, "unsupported metric for float32: Hamming");
    int np = nprobes <= 0 ? 1 : nprobes;  // segment.go:728-731
    if (np > idx->num_partitions) np = idx->num_partitions;  // kmeans.go:219-221
    VG_CHECK(np <= 64, VG_ERR_UNSUPPORTED, "vg_search_flat_probed: nprobes=%d exceeds 64", np);
    const bool dot = idx->metric != VG_METRIC_L2;
    if (scan == VG_SCAN_F32) {
        VG_CHECK(idx->d_vectors, VG_ERR_NOT_READY, "vg_search_flat_probed: index has no fp32 vectors");
    } else if (scan == VG_SCAN_PQ) {
        VG_CHECK(idx->pq && idx->d_pq_tiles, VG_ERR_NOT_READY, "vg_search_flat_probed: index has no PQ codes");
        VG_CHECK(idx->pq->k == 256, VG_ERR_UNSUPPORTED,
                 "vg_search_flat_probed: LUT scan needs numCentroids == 256 (got %d)", idx->pq->k);
    } else {
        VG_CHECK(idx->sq && idx->d_sq_tiles, VG_ERR_NOT_READY, "vg_search_flat_probed: index has no SQ8 codes");
    }
    VG_HIP(hipSetDevice(idx->ctx->device));
    hipStream_t st = vg::pick_stream(idx->ctx, stream);
    vg::DevIn<float> q;
    vg::DevOut<uint32_t> oid;
    vg::DevOut<float> osc;
    VG_TRY(q.init(queries, static_cast<size_t>(nq) * idx->dim, st));
    VG_TRY(oid.init(ids, static_cast<size_t>(nq) * k, st));
    VG_TRY(osc.init(scores, static_cast<size_t>(nq) * k, st));

    // enough workgroups to fill the device when there are few (query, probe) pairs
    const int64_t pairs = nq * np;
    const int want = 4 * idx->ctx->compute_units;
    int sub = static_cast<int>(std::min<int64_t>(32, std::max<int64_t>(1, (want + pairs - 1) / pairs)));
    int split = static_cast<int>(std::min<int64_t>(np, std::max<int64_t>(1, (idx->ctx->compute_units + nq - 1) / nq)));
    const int lists = scan == VG_SCAN_PQ ? split : np * sub;
    const int lut_words = scan == VG_SCAN_PQ ? (((idx->pq->m >> 4) + 1) >> 1) * 8192 + (idx->pq->m & 15) * 256 : 0;

    vg::ArenaCall ar(idx->ctx, st);
    const int i_probes = ar.add(sizeof(uint32_t) * static_cast<size_t>(nq) * np);
    const int i_partial = ar.add(sizeof(uint64_t) * static_cast<size_t>(nq) * lists * k);
    const int i_tables = ar.add(sizeof(float) * static_cast<size_t>(nq) * lut_words);
    VG_TRY(ar.commit());
    uint32_t *probes = ar.get<uint32_t>(i_probes);
    uint64_t *partial = ar.get<uint64_t>(i_partial);
    float *tables = ar.get<float>(i_tables);

    if (dot)
        VG_LAUNCH(vg::probe_select_kernel<true>, dim3(static_cast<unsigned>(nq)), dim3(256), 0, st, q.ptr, idx->dim,
                  idx->d_centroids, idx->num_partitions, np, probes);
    else
        VG_LAUNCH(vg::probe_select_kernel<false>, dim3(static_cast<unsigned>(nq)), dim3(256), 0, st, q.ptr, idx->dim,
                  idx->d_centroids, idx->num_partitions, np, probes);
    if (scan == VG_SCAN_F32) {
        for (int64_t q0 = 0; q0 < nq; q0 += 65535) {
            const int64_t cnt = std::min<int64_t>(65535, nq - q0);
            const dim3 grid(static_cast<unsigned>(sub), static_cast<unsigned>(np), static_cast<unsigned>(cnt));
            vg::ProfScope prof(idx->ctx, "flat_probe", st);
            if (dot)
                VG_LAUNCH(vg::probe_scan_f32_kernel<true>, grid, dim3(256), 0, st, idx->d_vectors, idx->dim,
                          q.ptr + q0 * idx->dim, probes + q0 * np, idx->d_part_off, np, sub, k, partial + q0 * lists * k);
            else
                VG_LAUNCH(vg::probe_scan_f32_kernel<false>, grid, dim3(256), 0, st, idx->d_vectors, idx->dim,
                          q.ptr + q0 * idx->dim, probes + q0 * np, idx->d_part_off, np, sub, k, partial + q0 * lists * k);
        }
    } else if (scan == VG_SCAN_PQ) {
        VG_TRY(vg::launch_pq_build_table(idx->pq, q.ptr, nq, tables, true, st));
        VG_TRY(vg::launch_probe_scan_adc(idx, tables, probes, nq, np, split, k, partial, st));
    } else {
        VG_TRY(vg::launch_probe_scan_sq8(idx, q.ptr, probes, nq, np, sub, k, partial, st));
    }
    // table-lookup scores are squared L2 (ascending); fp32 and SQ8 (L2Distance / DotProduct) follow the metric
    VG_TRY(vg::launch_topk_merge(partial, nq, lists, k, scan != VG_SCAN_PQ && dot, oid.ptr, osc.ptr, st));
    VG_TRY(oid.finish());
    VG_TRY(osc.finish());
    if (oid.on_host() || osc.on_host()) VG_HIP(hipStreamSynchronize(st));
    return VG_OK;
}
