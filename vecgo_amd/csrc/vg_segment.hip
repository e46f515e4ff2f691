// vg_segment.hip — SURVEY.md §8(f) rank 2: the reference's on-disk segment images straight to a
// resident index ("zero-parse": the file is already laid out as the arrays the kernels want).
//   flat segment     internal/segment/flat/format.go:11-165 (header), segment.go:105-300 (Open)
//   DiskANN segment  internal/segment/diskann/format.go:8-119 (header), segment.go:165-440, 1393-1408
// Only the sections on the hot path are read: fp32 rows, quantizer parameters, codes, the graph.
// Primary keys, metadata, block statistics and the inverted index stay with the host database.
// Host code only (no kernels of its own): everything goes through the C ABI of the other files.
#include <cstring>

#include "vg_internal.hpp"
#include "vg_segment_layout.hpp"

struct vg_segment {
    vg_ctx *ctx = nullptr;
    vg_index *idx = nullptr;
    vg_pq *pq = nullptr;
    vg_sq8 *sq = nullptr;
    vg_int4 *iq = nullptr;
    vg_segment_info info{};
};

namespace {

using namespace vg::seglayout;

void close_partial(vg_segment *s)
{
    if (!s) return;
    if (s->idx) (void)vg_index_destroy(s->idx);
    if (s->pq) (void)vg_pq_destroy(s->pq);
    if (s->sq) (void)vg_sq8_destroy(s->sq);
    if (s->iq) (void)vg_int4_destroy(s->iq);
    delete s;
}

// a failed open must not leak: run `expr`, on error drop the half-built segment and propagate
#define SEG_TRY(expr)               \
    do {                            \
        int32_t _s = (expr);        \
        if (_s != VG_OK) {          \
            close_partial(seg);     \
            return _s;              \
        }                           \
    } while (0)
#define SEG_CHECK(cond, status, ...)      \
    do {                                  \
        if (!(cond)) {                    \
            ::vg::set_error(__VA_ARGS__); \
            close_partial(seg);           \
            return (status);              \
        }                                 \
    } while (0)

}  // namespace

VG_API uint32_t vg_crc32c(const void *data, int64_t size)
{
    return data && size > 0 ? crc32c(static_cast<const uint8_t *>(data), static_cast<size_t>(size)) : 0;
}

VG_API int32_t vg_segment_open_flat(vg_ctx *ctx, const void *image, int64_t size, int32_t verify_checksum,
                                    vg_segment **out, void *stream)
{
    VG_CHECK(out, VG_ERR_INVALID_ARG, "vg_segment_open_flat: out is NULL");
    *out = nullptr;
    VG_CHECK(ctx && image && size >= 0, VG_ERR_INVALID_ARG, "vg_segment_open_flat: NULL context or image");
    const uint8_t *data = static_cast<const uint8_t *>(image);
    // every untrusted byte is read by parse_flat (vg_segment_layout.hpp): what follows touches the image only through the
    // sections it returned, all of them inside [0, size)
    FlatLayout L;
    Error err;
    if (parse_flat(data, static_cast<uint64_t>(size), verify_checksum != 0, L, err) != VG_OK) {
        ::vg::set_error("%s", err.text.c_str());
        return err.status;
    }
    vg_segment *seg = new vg_segment;
    seg->ctx = ctx;
    vg_segment_info &h = seg->info;
    h.kind = 0;
    h.segment_id = L.segment_id;
    h.rows = L.rows;
    h.dim = L.dim;
    h.metric = L.metric;
    const uint64_t n = L.rows, dim = static_cast<uint64_t>(L.dim);
    SEG_TRY(vg_index_create(ctx, h.rows, h.dim, h.metric, &seg->idx));
    if (L.qtype == 1) {  // segment.go:209-233: mins[dim] then maxs[dim], SetBounds, codes n*dim
        SEG_TRY(vg_sq8_create(ctx, h.dim, &seg->sq));
        std::vector<float> mm(2 * dim);  // the image is only byte-aligned
        memcpy(mm.data(), data + L.sq_bounds.off, L.sq_bounds.bytes);
        SEG_TRY(vg_sq8_set_bounds(seg->sq, mm.data(), mm.data() + dim));
        h.quantization = VG_QUANT_SQ8;
        // segment.go:659-667: an SQ8 segment is scored from its codes for every metric (L2Distance / DotProduct)
        SEG_TRY(vg_index_set_sq8_codes(seg->idx, seg->sq, data + L.codes.off, stream));
    } else if (L.qtype == 2) {  // segment.go:234-281: m, k, scales[m], offsets[m], codebooks[m*k*dsub], codes n*m
        const uint64_t m = L.pq_m, k = L.pq_k;
        SEG_TRY(vg_pq_create(ctx, h.dim, static_cast<int32_t>(m), static_cast<int32_t>(k), &seg->pq));
        std::vector<float> so(2 * m);
        memcpy(so.data(), data + L.pq_scales_offsets.off, L.pq_scales_offsets.bytes);
        SEG_TRY(vg_pq_set_codebooks(seg->pq, reinterpret_cast<const int8_t *>(data + L.pq_codebooks.off), so.data(),
                                    so.data() + m));
        h.quantization = VG_QUANT_PQ;
        h.pq_m = static_cast<int32_t>(m);
        h.pq_k = static_cast<int32_t>(k);
        if (k == 256) SEG_TRY(vg_index_set_pq_codes(seg->idx, seg->pq, data + L.codes.off, stream));
    } else {
        h.quantization = VG_QUANT_NONE;
    }
    if (n) {  // segment.go:283-289
        if (reinterpret_cast<uintptr_t>(data + L.vectors.off) % 4 == 0) {
            SEG_TRY(vg_index_set_vectors(seg->idx, reinterpret_cast<const float *>(data + L.vectors.off), stream));
        } else {
            std::vector<float> v(n * dim);
            memcpy(v.data(), data + L.vectors.off, L.vectors.bytes);
            SEG_TRY(vg_index_set_vectors(seg->idx, v.data(), stream));
        }
    }
    h.num_partitions = static_cast<int32_t>(L.partitions);
    if (L.partitions > 0) {  // segment.go:187-207: centroids [P*dim] fp32, partition offsets [P+1] uint32
        std::vector<float> cent(static_cast<size_t>(L.partitions) * dim);  // the image is only byte-aligned
        std::vector<uint32_t> poff(static_cast<size_t>(L.partitions) + 1);
        memcpy(cent.data(), data + L.centroids.off, L.centroids.bytes);
        memcpy(poff.data(), data + L.part_offsets.off, L.part_offsets.bytes);
        SEG_TRY(vg_index_set_partitions(seg->idx, cent.data(), poff.data(), static_cast<int32_t>(L.partitions), stream));
    }
    *out = seg;
    return VG_OK;
}

VG_API int32_t vg_segment_search(vg_segment *seg, const float *queries, int64_t nq, int32_t k, int32_t nprobes,
                                 uint32_t *ids, float *scores, void *stream)
{
    return vg_segment_search_filtered(seg, queries, nq, k, nprobes, nullptr, 0, ids, scores, stream);
}

VG_API int32_t vg_segment_search_filtered(vg_segment *seg, const float *queries, int64_t nq, int32_t k, int32_t nprobes,
                                          const uint8_t *mask, int64_t mask_stride, uint32_t *ids, float *scores, void *stream)
{
    VG_CHECK(seg && seg->idx, VG_ERR_INVALID_ARG, "vg_segment_search: NULL segment");
    if (seg->info.kind == 1 && mask) {  // pushToHeap's filter (diskann/segment.go:616-627)
        const int32_t kind = seg->info.quantization == VG_QUANT_RABITQ ? 2
                             : seg->info.quantization == VG_QUANT_PQ   ? 1
                             : seg->info.quantization == VG_QUANT_INT4 ? 3
                                                                       : 0;
        return vg_search_vamana_filtered(seg->idx, queries, nq, k, kind, mask, mask_stride, ids, scores, nullptr, stream);
    }
    if (seg->info.kind == 1) {
        // diskann.Segment.Search (diskann/segment.go:487-706): the search-list size it derives from k and
        // RefineFactor is never read by searchInternal, so the call is the beam search with the distFn the
        // segment's quantization selects (:512-588: RaBitQ, else PQ, else INT4, else fp32 rows)
        const int32_t kind = seg->info.quantization == VG_QUANT_RABITQ ? 2
                             : seg->info.quantization == VG_QUANT_PQ   ? 1
                             : seg->info.quantization == VG_QUANT_INT4 ? 3
                                                                       : 0;
        return vg_search_vamana(seg->idx, queries, nq, k, kind, ids, scores, nullptr, stream);
    }
    // flat/segment.go:657-701: SQ8 codes if the segment has them, else PQ table lookups, else fp32 rows
    int32_t scan = VG_SCAN_F32;
    if (seg->info.quantization == VG_QUANT_SQ8) {
        scan = VG_SCAN_SQ8;
    } else if (seg->info.quantization == VG_QUANT_PQ) {
        // the reference's heap direction follows the segment metric (segment.go:449) while AdcDistance is always a
        // squared L2: a Dot / Cosine PQ segment keeps its k largest ADC distances there — and here
        scan = VG_SCAN_PQ;
    }
    // (a NULL mask is the unfiltered probed scan; filter.Matches: flat/segment.go:631-635)
    return vg_search_flat_filtered(seg->idx, queries, nq, k, nprobes, scan, mask, mask_stride, ids, scores, stream);
}

VG_API int32_t vg_segment_open_diskann(vg_ctx *ctx, const void *image, int64_t size, int32_t verify_checksum,
                                       vg_segment **out, void *stream)
{
    VG_CHECK(out, VG_ERR_INVALID_ARG, "vg_segment_open_diskann: out is NULL");
    *out = nullptr;
    VG_CHECK(ctx && image && size >= 0, VG_ERR_INVALID_ARG, "vg_segment_open_diskann: NULL context or image");
    const uint8_t *data = static_cast<const uint8_t *>(image);
    // (every untrusted byte is read by parse_diskann: see vg_segment_open_flat)
    DiskLayout L;
    Error err;
    if (parse_diskann(data, static_cast<uint64_t>(size), verify_checksum != 0, L, err) != VG_OK) {
        ::vg::set_error("%s", err.text.c_str());
        return err.status;
    }
    vg_segment *seg = new vg_segment;
    seg->ctx = ctx;
    vg_segment_info &h = seg->info;
    h.kind = 1;
    h.segment_id = L.segment_id;
    h.rows = L.rows;
    h.dim = L.dim;
    h.metric = L.metric;
    h.max_degree = L.max_degree_raw > 0x7FFFFFFFu ? 0x7FFFFFFF : static_cast<int32_t>(L.max_degree_raw);
    h.search_list_size = L.search_list_size;
    h.entrypoint = L.entrypoint;
    h.pq_m = static_cast<int32_t>(L.pq_m);
    h.pq_k = static_cast<int32_t>(L.pq_k);
    const uint64_t n = L.rows, dim = static_cast<uint64_t>(L.dim);
    SEG_TRY(vg_index_create(ctx, h.rows, h.dim, h.metric, &seg->idx));
    auto aligned_or_copy = [&](const Section &sec, std::vector<uint32_t> &tmp) -> const void * {
        if (reinterpret_cast<uintptr_t>(data + sec.off) % 4 == 0) return data + sec.off;
        tmp.resize((sec.bytes + 3) / 4);
        memcpy(tmp.data(), data + sec.off, sec.bytes);
        return tmp.data();
    };
    std::vector<uint32_t> tmp;
    if (n) {
        SEG_TRY(vg_index_set_vectors(seg->idx, static_cast<const float *>(aligned_or_copy(L.vectors, tmp)), stream));
        if (L.graph.present)
            SEG_TRY(vg_index_set_vamana_graph(seg->idx, h.max_degree, static_cast<const uint32_t *>(aligned_or_copy(L.graph, tmp)),
                                              h.entrypoint, stream));
    }
    h.quantization = VG_QUANT_NONE;
    if (L.qtype == 1) {  // segment.go:305-376 loadPQ: codes n*m; scales[m], offsets[m], codebooks[m*k*subDim]
        const uint64_t m = L.pq_m, k = L.pq_k;
        SEG_TRY(vg_pq_create(ctx, h.dim, h.pq_m, h.pq_k, &seg->pq));
        std::vector<float> so(2 * m);
        memcpy(so.data(), data + L.pq_scales_offsets.off, L.pq_scales_offsets.bytes);
        SEG_TRY(vg_pq_set_codebooks(seg->pq, reinterpret_cast<const int8_t *>(data + L.pq_codebooks.off), so.data(), so.data() + m));
        if (n && k == 256) SEG_TRY(vg_index_set_pq_codes(seg->idx, seg->pq, data + L.pq_codes.off, stream));
        h.quantization = VG_QUANT_PQ;
    } else if (L.qtype == 5) {  // segment.go:1393-1408 loadRaBitQ
        if (n) SEG_TRY(vg_index_set_rabitq_codes(seg->idx, data + L.rabitq_codes.off, stream));
        h.quantization = VG_QUANT_RABITQ;
    } else if (L.qtype == 6) {  // segment.go:378-416 loadINT4
        SEG_TRY(vg_int4_create(ctx, h.dim, &seg->iq));
        std::vector<float> md(2 * dim);
        memcpy(md.data(), data + L.int4_params.off, L.int4_params.bytes);
        SEG_TRY(vg_int4_set_params(seg->iq, md.data(), md.data() + dim));
        if (n) SEG_TRY(vg_index_set_int4_codes(seg->idx, seg->iq, data + L.int4_codes.off, stream));
        h.quantization = VG_QUANT_INT4;
    }
    *out = seg;
    return VG_OK;
}

VG_API int32_t vg_segment_get_info(vg_segment *seg, vg_segment_info *info)
{
    VG_CHECK(seg && info, VG_ERR_INVALID_ARG, "vg_segment_get_info: NULL argument");
    *info = seg->info;
    return VG_OK;
}

VG_API vg_index *vg_segment_index(vg_segment *seg) { return seg ? seg->idx : nullptr; }
VG_API vg_pq *vg_segment_pq(vg_segment *seg) { return seg ? seg->pq : nullptr; }
VG_API vg_sq8 *vg_segment_sq8(vg_segment *seg) { return seg ? seg->sq : nullptr; }
VG_API vg_int4 *vg_segment_int4(vg_segment *seg) { return seg ? seg->iq : nullptr; }

VG_API int32_t vg_segment_close(vg_segment *seg)
{
    close_partial(seg);
    return VG_OK;
}
