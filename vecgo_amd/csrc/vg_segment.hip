// vg_segment.hip — SURVEY.md §8(f) rank 2: the reference's on-disk segment images straight to a
// resident index ("zero-parse": the file is already laid out as the arrays the kernels want).
//   flat segment     internal/segment/flat/format.go:11-165 (header), segment.go:105-300 (Open)
//   DiskANN segment  internal/segment/diskann/format.go:8-119 (header), segment.go:165-440, 1393-1408
// Only the sections on the hot path are read: fp32 rows, quantizer parameters, codes, the graph.
// Primary keys, metadata, block statistics and the inverted index stay with the host database.
// Host code only (no kernels of its own): everything goes through the C ABI of the other files.
#include <cstring>

#include "vg_internal.hpp"

struct vg_segment {
    vg_ctx *ctx = nullptr;
    vg_index *idx = nullptr;
    vg_pq *pq = nullptr;
    vg_sq8 *sq = nullptr;
    vg_int4 *iq = nullptr;
    vg_segment_info info{};
};

namespace {

uint32_t rd32(const uint8_t *p)
{
    return static_cast<uint32_t>(p[0]) | (static_cast<uint32_t>(p[1]) << 8) | (static_cast<uint32_t>(p[2]) << 16) |
           (static_cast<uint32_t>(p[3]) << 24);
}
uint16_t rd16(const uint8_t *p) { return static_cast<uint16_t>(p[0] | (p[1] << 8)); }
uint64_t rd64(const uint8_t *p) { return static_cast<uint64_t>(rd32(p)) | (static_cast<uint64_t>(rd32(p + 4)) << 32); }

// CRC-32C (Castagnoli, reflected polynomial 0x82F63B78) — internal/hash/crc32c.go:15-17
struct Crc32cTables {
    uint32_t t[8][256];
    Crc32cTables()
    {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
            t[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; i++)
            for (int s = 1; s < 8; s++) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xFF];
    }
};

uint32_t crc32c(const uint8_t *data, size_t n)
{
    static const Crc32cTables tables;  // built once, thread-safe (C++11 static initialisation)
    const uint32_t(&table)[8][256] = tables.t;
    uint32_t c = 0xFFFFFFFFu;
    while (n >= 8) {  // slicing-by-8
        const uint32_t lo = rd32(data) ^ c, hi = rd32(data + 4);
        c = table[7][lo & 0xFF] ^ table[6][(lo >> 8) & 0xFF] ^ table[5][(lo >> 16) & 0xFF] ^ table[4][lo >> 24] ^
            table[3][hi & 0xFF] ^ table[2][(hi >> 8) & 0xFF] ^ table[1][(hi >> 16) & 0xFF] ^ table[0][hi >> 24];
        data += 8;
        n -= 8;
    }
    while (n--) c = table[0][(c ^ *data++) & 0xFF] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

constexpr uint32_t kFlatMagic = 0x56454331;     // "VEC1" flat/format.go:12
constexpr uint32_t kDiskMagic = 0x4449534B;     // "DISK" diskann/format.go:9
constexpr size_t kFlatHeader = 152;             // flat/format.go:113
constexpr size_t kDiskHeader = 160;             // diskann/format.go:49

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

int32_t verify_body(const uint8_t *data, uint64_t size, size_t header, uint32_t want)
{
    if (want == 0 || size <= header) return VG_OK;  // segment.go: `Checksum != 0`
    const uint32_t got = crc32c(data + header, static_cast<size_t>(size - header));
    if (got != want) {
        ::vg::set_error("checksum mismatch: expected %x, got %x", want, got);
        return VG_ERR_CHECKSUM;
    }
    return VG_OK;
}

}  // namespace

VG_API uint32_t vg_crc32c(const void *data, int64_t size)
{
    return data && size > 0 ? crc32c(static_cast<const uint8_t *>(data), static_cast<size_t>(size)) : 0;
}

// off + a*b*c <= len with every quantity an untrusted 64-bit header field: checked by division, no product is
// ever formed before it is known to fit (a wrapped product passes an `x + y >= x` guard)
static bool fits(uint64_t len, uint64_t off, uint64_t a, uint64_t b = 1, uint64_t c = 1)
{
    if (off > len) return false;
    if (a == 0 || b == 0 || c == 0) return true;
    const uint64_t room = len - off;
    if (a > room) return false;
    const uint64_t per_a = room / a;
    if (b > per_a) return false;
    return c <= per_a / b;
}

VG_API int32_t vg_segment_open_flat(vg_ctx *ctx, const void *image, int64_t size, int32_t verify_checksum,
                                    vg_segment **out, void *stream)
{
    VG_CHECK(out, VG_ERR_INVALID_ARG, "vg_segment_open_flat: out is NULL");
    *out = nullptr;
    VG_CHECK(ctx && image && size >= 0, VG_ERR_INVALID_ARG, "vg_segment_open_flat: NULL context or image");
    const uint8_t *data = static_cast<const uint8_t *>(image);
    const uint64_t len = static_cast<uint64_t>(size);
    VG_CHECK(len >= kFlatHeader, VG_ERR_FORMAT, "buffer too small for header");  // format.go:138-140
    VG_CHECK(rd32(data) == kFlatMagic, VG_ERR_FORMAT, "invalid magic number");
    VG_CHECK(rd32(data + 4) == 1, VG_ERR_FORMAT, "unsupported version");
    vg_segment *seg = new vg_segment;
    seg->ctx = ctx;
    vg_segment_info &h = seg->info;
    h.kind = 0;
    h.segment_id = rd64(data + 8);
    h.rows = rd32(data + 16);
    h.dim = static_cast<int32_t>(rd32(data + 20));
    h.metric = data[24];
    const uint32_t partitions = rd32(data + 28);
    const uint64_t cent_off = rd64(data + 40), poff_off = rd64(data + 48);
    const int qtype = data[32];  // format.go:22-26: 0 none, 1 SQ8, 2 PQ
    const uint64_t q_off = rd64(data + 56), codes_off = rd64(data + 64), vec_off = rd64(data + 72);
    const uint32_t checksum = rd32(data + 104);
    SEG_CHECK(h.dim > 0, VG_ERR_FORMAT, "flat segment: dimension 0");
    SEG_CHECK(h.metric <= VG_METRIC_DOT, VG_ERR_UNSUPPORTED, "flat segment: metric %d has no float32 kernels", h.metric);
    if (verify_checksum) SEG_TRY(verify_body(data, len, kFlatHeader, checksum));
    const uint64_t n = static_cast<uint64_t>(h.rows), dim = static_cast<uint64_t>(h.dim);
    SEG_TRY(vg_index_create(ctx, h.rows, h.dim, h.metric, &seg->idx));
    if (qtype == 1) {  // segment.go:209-233: mins[dim] then maxs[dim], SetBounds, codes n*dim
        SEG_CHECK(fits(len, q_off, dim, 8), VG_ERR_FORMAT, "file too short for quantization metadata");
        SEG_CHECK(fits(len, codes_off, n, dim), VG_ERR_FORMAT, "file too short for codes");
        SEG_TRY(vg_sq8_create(ctx, h.dim, &seg->sq));
        std::vector<float> mm(2 * dim);  // the image is only byte-aligned
        memcpy(mm.data(), data + q_off, dim * 8);
        SEG_TRY(vg_sq8_set_bounds(seg->sq, mm.data(), mm.data() + dim));
        h.quantization = VG_QUANT_SQ8;
        // segment.go:659-667: an SQ8 segment is scored from its codes for every metric (L2Distance / DotProduct)
        SEG_TRY(vg_index_set_sq8_codes(seg->idx, seg->sq, data + codes_off, stream));
    } else if (qtype == 2) {  // segment.go:234-281: m, k, scales[m], offsets[m], codebooks[m*k*dsub], codes n*m
        SEG_CHECK(fits(len, q_off, 8), VG_ERR_FORMAT, "file too short for PQ metadata");
        const uint64_t m = rd32(data + q_off), k = rd32(data + q_off + 4);
        SEG_CHECK(m > 0 && dim % m == 0, VG_ERR_FORMAT, "flat segment: %llu sub-quantizers do not divide dimension %llu",
                  static_cast<unsigned long long>(m), static_cast<unsigned long long>(dim));
        SEG_CHECK(fits(len, q_off, 8 + m * 8) && fits(len, q_off + 8 + m * 8, m, k, dim / m), VG_ERR_FORMAT,
                  "file too short for PQ metadata");
        SEG_CHECK(fits(len, codes_off, n, m), VG_ERR_FORMAT, "file too short for codes");
        SEG_TRY(vg_pq_create(ctx, h.dim, static_cast<int32_t>(m), static_cast<int32_t>(k), &seg->pq));
        std::vector<float> so(2 * m);
        memcpy(so.data(), data + q_off + 8, m * 8);
        SEG_TRY(vg_pq_set_codebooks(seg->pq, reinterpret_cast<const int8_t *>(data + q_off + 8 + m * 8), so.data(),
                                    so.data() + m));
        h.quantization = VG_QUANT_PQ;
        h.pq_m = static_cast<int32_t>(m);
        h.pq_k = static_cast<int32_t>(k);
        if (k == 256) SEG_TRY(vg_index_set_pq_codes(seg->idx, seg->pq, data + codes_off, stream));
    } else {
        SEG_CHECK(qtype == 0, VG_ERR_FORMAT, "flat segment: unknown quantization type %d", qtype);
        h.quantization = VG_QUANT_NONE;
    }
    SEG_CHECK(fits(len, vec_off, n, dim, 4), VG_ERR_FORMAT, "file too short for vectors");  // segment.go:283-289
    if (n) {
        if (reinterpret_cast<uintptr_t>(data + vec_off) % 4 == 0) {
            SEG_TRY(vg_index_set_vectors(seg->idx, reinterpret_cast<const float *>(data + vec_off), stream));
        } else {
            std::vector<float> v(n * dim);
            memcpy(v.data(), data + vec_off, n * dim * 4);
            SEG_TRY(vg_index_set_vectors(seg->idx, v.data(), stream));
        }
    }
    h.num_partitions = static_cast<int32_t>(partitions);
    if (partitions > 0) {  // segment.go:187-207: centroids [P*dim] fp32, partition offsets [P+1] uint32
        const uint64_t cbytes = static_cast<uint64_t>(partitions) * dim * 4, pbytes = (static_cast<uint64_t>(partitions) + 1) * 4;
        SEG_CHECK(fits(len, cent_off, static_cast<uint64_t>(partitions), dim, 4), VG_ERR_FORMAT, "file too short for centroids");
        SEG_CHECK(fits(len, poff_off, static_cast<uint64_t>(partitions) + 1, 4), VG_ERR_FORMAT,
                  "file too short for partition offsets");
        std::vector<float> cent(static_cast<size_t>(partitions) * dim);  // the image is only byte-aligned
        std::vector<uint32_t> poff(static_cast<size_t>(partitions) + 1);
        memcpy(cent.data(), data + cent_off, cbytes);
        memcpy(poff.data(), data + poff_off, pbytes);
        SEG_TRY(vg_index_set_partitions(seg->idx, cent.data(), poff.data(), static_cast<int32_t>(partitions), stream));
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
    const uint64_t len = static_cast<uint64_t>(size);
    VG_CHECK(len >= kDiskHeader, VG_ERR_FORMAT, "buffer too small for header");  // format.go:81-83
    VG_CHECK(rd32(data) == kDiskMagic, VG_ERR_FORMAT, "invalid magic number");
    const uint32_t version = rd32(data + 4);
    VG_CHECK(version == 2 || version == 1, VG_ERR_FORMAT, "unsupported version");
    vg_segment *seg = new vg_segment;
    seg->ctx = ctx;
    vg_segment_info &h = seg->info;
    h.kind = 1;
    h.segment_id = rd64(data + 8);
    h.rows = rd32(data + 16);
    h.dim = static_cast<int32_t>(rd32(data + 20));
    h.metric = data[24];
    const uint32_t max_degree_raw = rd32(data + 25);
    h.max_degree = max_degree_raw > 0x7FFFFFFFu ? 0x7FFFFFFF : static_cast<int32_t>(max_degree_raw);
    h.search_list_size = static_cast<int32_t>(rd32(data + 29));
    h.entrypoint = rd32(data + 33);
    const int qtype = data[37];  // quantization.Type (types.go:6-14): 1 PQ, 5 RaBitQ, 6 INT4
    h.pq_m = rd16(data + 38);
    h.pq_k = rd16(data + 40);
    // data[42] = CompressionType (format.go:32).  The writer records its option there (LZ4 by default,
    // writer.go:92,676) but streams every section raw (writer.go:697-740) and Open never consults the
    // field (segment.go:165-440), so neither does this reader: a default reference segment has 1 here.
    const uint64_t vec_off = rd64(data + 48), graph_off = rd64(data + 56), pq_codes_off = rd64(data + 64),
                   bq_codes_off = rd64(data + 72), cb_off = rd64(data + 80), pk_off = rd64(data + 88);
    const uint32_t checksum = rd32(data + 120);
    SEG_CHECK(h.dim > 0, VG_ERR_FORMAT, "diskann segment: dimension 0");
    SEG_CHECK(h.metric <= VG_METRIC_DOT, VG_ERR_UNSUPPORTED, "diskann segment: metric %d has no float32 kernels", h.metric);
    if (verify_checksum) SEG_TRY(verify_body(data, len, kDiskHeader, checksum));
    const uint64_t n = static_cast<uint64_t>(h.rows), dim = static_cast<uint64_t>(h.dim);
    SEG_CHECK(fits(len, pk_off, n, 8), VG_ERR_FORMAT, "file size too small: expected at least %llu, got %llu",
              static_cast<unsigned long long>(pk_off + n * 8), static_cast<unsigned long long>(len));  // segment.go:177-182
    SEG_CHECK(fits(len, vec_off, n, dim, 4), VG_ERR_FORMAT, "vector section out of bounds");
    // the reference reads a node's neighbour list when a search visits it (segment.go:1376-1391) and never
    // checks the graph section in Open: a segment whose graph section does not fit (or whose degree this
    // library cannot walk) opens without a graph, and searching it fails then
    const uint64_t r = max_degree_raw;
    const bool graph_ok = r >= 1 && r <= 64 && fits(len, graph_off, n, r, 4);
    SEG_TRY(vg_index_create(ctx, h.rows, h.dim, h.metric, &seg->idx));
    auto aligned_or_copy = [&](uint64_t off, uint64_t bytes, std::vector<uint32_t> &tmp) -> const void * {
        if (reinterpret_cast<uintptr_t>(data + off) % 4 == 0) return data + off;
        tmp.resize((bytes + 3) / 4);
        memcpy(tmp.data(), data + off, bytes);
        return tmp.data();
    };
    std::vector<uint32_t> tmp;
    if (n) {
        SEG_TRY(vg_index_set_vectors(seg->idx, static_cast<const float *>(aligned_or_copy(vec_off, n * dim * 4, tmp)),
                                     stream));
        if (graph_ok)
            SEG_TRY(vg_index_set_vamana_graph(seg->idx, h.max_degree,
                                              static_cast<const uint32_t *>(aligned_or_copy(graph_off, n * r * 4, tmp)),
                                              h.entrypoint, stream));
    }
    h.quantization = VG_QUANT_NONE;
    if (qtype == 1) {  // segment.go:305-376 loadPQ: codes n*m; scales[m], offsets[m], codebooks[m*k*subDim]
        const uint64_t m = static_cast<uint64_t>(h.pq_m), k = static_cast<uint64_t>(h.pq_k);
        SEG_CHECK(m > 0 && dim % m == 0, VG_ERR_FORMAT, "diskann segment: %llu sub-quantizers do not divide dimension %llu",
                  static_cast<unsigned long long>(m), static_cast<unsigned long long>(dim));
        SEG_CHECK(fits(len, pq_codes_off, n, m), VG_ERR_FORMAT, "PQ codes section out of bounds");
        SEG_CHECK(fits(len, cb_off, m * 8) && fits(len, cb_off + m * 8, m, k, dim / m), VG_ERR_FORMAT,
                  "failed to read PQ codebooks: out of bounds");
        SEG_TRY(vg_pq_create(ctx, h.dim, h.pq_m, h.pq_k, &seg->pq));
        std::vector<float> so(2 * m);
        memcpy(so.data(), data + cb_off, m * 8);
        SEG_TRY(vg_pq_set_codebooks(seg->pq, reinterpret_cast<const int8_t *>(data + cb_off + m * 8), so.data(),
                                    so.data() + m));
        if (n && k == 256) SEG_TRY(vg_index_set_pq_codes(seg->idx, seg->pq, data + pq_codes_off, stream));
        h.quantization = VG_QUANT_PQ;
    } else if (qtype == 5) {  // segment.go:1393-1408 loadRaBitQ: n * (((dim+63)/64)*8 + 4) bytes
        const uint64_t per = static_cast<uint64_t>(vg_rabitq_code_bytes(h.dim));
        SEG_CHECK(fits(len, bq_codes_off, n, per), VG_ERR_FORMAT, "RaBitQ codes section out of bounds");
        if (n) SEG_TRY(vg_index_set_rabitq_codes(seg->idx, data + bq_codes_off, stream));
        h.quantization = VG_QUANT_RABITQ;
    } else if (qtype == 6) {  // segment.go:378-416 loadINT4: params = [dim u32][min f32 x dim][diff f32 x dim]
        SEG_CHECK(cb_off != 0, VG_ERR_FORMAT, "missing INT4 params");
        SEG_CHECK(pk_off > cb_off, VG_ERR_FORMAT, "invalid INT4 params size");
        const uint64_t psize = pk_off - cb_off;
        SEG_CHECK(fits(len, cb_off, psize), VG_ERR_FORMAT, "INT4 params out of bounds");
        SEG_CHECK(psize >= 4, VG_ERR_FORMAT, "data too short");                       // int4.go:191-193
        SEG_CHECK(rd32(data + cb_off) == dim && psize == 4 + dim * 8, VG_ERR_FORMAT, "data size mismatch");
        SEG_CHECK(pq_codes_off != 0, VG_ERR_FORMAT, "missing INT4 codes");
        const uint64_t cs = (dim + 1) / 2;
        SEG_CHECK(fits(len, pq_codes_off, n, cs), VG_ERR_FORMAT, "INT4 codes out of bounds");
        SEG_TRY(vg_int4_create(ctx, h.dim, &seg->iq));
        std::vector<float> md(2 * dim);
        memcpy(md.data(), data + cb_off + 4, dim * 8);
        SEG_TRY(vg_int4_set_params(seg->iq, md.data(), md.data() + dim));
        if (n) SEG_TRY(vg_index_set_int4_codes(seg->idx, seg->iq, data + pq_codes_off, stream));
        h.quantization = VG_QUANT_INT4;
    } else {
        SEG_CHECK(qtype == 0, VG_ERR_UNSUPPORTED,
                  "diskann segment: quantization type %d (OPQ / SQ8 / BQ) has no device scorer yet", qtype);
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
