// vg_segment_layout.hpp — header and section validation of the reference's on-disk segment images, HOST ONLY (no HIP, no
// device types): everything that reads UNTRUSTED bytes before a section is handed to the device.
//   flat segment     internal/segment/flat/format.go:11-165 (header), segment.go:105-300 (Open)
//   DiskANN segment  internal/segment/diskann/format.go:8-119 (header; HeaderSize const :49 = 160 bytes), segment.go:165-440, 1393-1408
// A parse either fails with the reference's error or returns a layout whose every section {off, bytes} lies inside the image:
// vg_segment.hip touches the image only through these sections.  Compiled twice: into the library (hipcc) and, by plain
// g++ -fsanitize=address,undefined, into tests/cpp/segment_fuzz.cpp, which mutates images and reads every byte a layout declares.
#pragma once

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>

#include "vecgo_hip.h"

namespace vg {
namespace seglayout {

inline uint32_t rd32(const uint8_t *p)
{
    return static_cast<uint32_t>(p[0]) | (static_cast<uint32_t>(p[1]) << 8) | (static_cast<uint32_t>(p[2]) << 16) |
           (static_cast<uint32_t>(p[3]) << 24);
}
inline uint16_t rd16(const uint8_t *p) { return static_cast<uint16_t>(p[0] | (p[1] << 8)); }
inline uint64_t rd64(const uint8_t *p) { return static_cast<uint64_t>(rd32(p)) | (static_cast<uint64_t>(rd32(p + 4)) << 32); }

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

inline uint32_t crc32c(const uint8_t *data, size_t n)
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

constexpr uint32_t kFlatMagic = 0x56454331;  // "VEC1" flat/format.go:12
constexpr uint32_t kDiskMagic = 0x4449534B;  // "DISK" diskann/format.go:9
constexpr size_t kFlatHeader = 152;          // flat/format.go:113
constexpr size_t kDiskHeader = 160;          // diskann/format.go:49

// off + a*b*c <= len with every quantity an untrusted 64-bit header field: checked by division, no product is
// ever formed before it is known to fit (a wrapped product passes an `x + y >= x` guard)
inline bool fits(uint64_t len, uint64_t off, uint64_t a, uint64_t b = 1, uint64_t c = 1)
{
    if (off > len) return false;
    if (a == 0 || b == 0 || c == 0) return true;
    const uint64_t room = len - off;
    if (a > room) return false;
    const uint64_t per_a = room / a;
    if (b > per_a) return false;
    return c <= per_a / b;
}

struct Section {
    uint64_t off = 0, bytes = 0;  // inside the image when bytes > 0 (checked by the parser that filled it in)
    bool present = false;
};

struct Error {
    int32_t status = VG_OK;
    std::string text;
};

inline int32_t fail(Error &e, int32_t status, const char *fmt, ...)
{
    char buf[256];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    e.status = status;
    e.text = buf;
    return status;
}

inline Section section(uint64_t off, uint64_t bytes)
{
    Section s;
    s.off = off;
    s.bytes = bytes;
    s.present = true;
    return s;
}

inline int32_t verify_body(const uint8_t *data, uint64_t size, size_t header, uint32_t want, Error &e)
{
    if (want == 0 || size <= header) return VG_OK;  // segment.go: `Checksum != 0`
    const uint32_t got = crc32c(data + header, static_cast<size_t>(size - header));
    if (got != want) return fail(e, VG_ERR_CHECKSUM, "checksum mismatch: expected %x, got %x", want, got);
    return VG_OK;
}

// ---- flat segment ------------------------------------------------------------------------------------------------------------
struct FlatLayout {
    uint64_t segment_id = 0;
    uint32_t rows = 0;
    int32_t dim = 0;
    int32_t metric = 0;
    uint32_t partitions = 0;
    int qtype = 0;              // format.go:22-26: 0 none, 1 SQ8, 2 PQ
    uint64_t pq_m = 0, pq_k = 0;
    Section sq_bounds;          // qtype 1: mins[dim] then maxs[dim] (fp32)
    Section pq_scales_offsets;  // qtype 2: scales[m] then offsets[m] (fp32)
    Section pq_codebooks;       // qtype 2: m * k * (dim / m) int8
    Section codes;              // qtype 1: rows * dim; qtype 2: rows * m
    Section vectors;            // rows * dim fp32
    Section centroids;          // partitions * dim fp32
    Section part_offsets;       // (partitions + 1) uint32
};

inline int32_t parse_flat(const uint8_t *data, uint64_t len, bool verify_checksum, FlatLayout &L, Error &e)
{
    if (len < kFlatHeader) return fail(e, VG_ERR_FORMAT, "buffer too small for header");  // format.go:138-140
    if (rd32(data) != kFlatMagic) return fail(e, VG_ERR_FORMAT, "invalid magic number");
    if (rd32(data + 4) != 1) return fail(e, VG_ERR_FORMAT, "unsupported version");
    L.segment_id = rd64(data + 8);
    L.rows = rd32(data + 16);
    L.dim = static_cast<int32_t>(rd32(data + 20));
    L.metric = data[24];
    L.partitions = rd32(data + 28);
    const uint64_t cent_off = rd64(data + 40), poff_off = rd64(data + 48);
    L.qtype = data[32];
    const uint64_t q_off = rd64(data + 56), codes_off = rd64(data + 64), vec_off = rd64(data + 72);
    const uint32_t checksum = rd32(data + 104);
    if (L.dim <= 0) return fail(e, VG_ERR_FORMAT, "flat segment: dimension 0");
    if (L.metric > VG_METRIC_DOT) return fail(e, VG_ERR_UNSUPPORTED, "flat segment: metric %d has no float32 kernels", L.metric);
    if (verify_checksum && verify_body(data, len, kFlatHeader, checksum, e) != VG_OK) return e.status;
    const uint64_t n = L.rows, dim = static_cast<uint64_t>(L.dim);
    if (L.qtype == 1) {  // segment.go:209-233: mins[dim] then maxs[dim], SetBounds, codes n*dim
        if (!fits(len, q_off, dim, 8)) return fail(e, VG_ERR_FORMAT, "file too short for quantization metadata");
        if (!fits(len, codes_off, n, dim)) return fail(e, VG_ERR_FORMAT, "file too short for codes");
        L.sq_bounds = section(q_off, dim * 8);
        L.codes = section(codes_off, n * dim);
    } else if (L.qtype == 2) {  // segment.go:234-281: m, k, scales[m], offsets[m], codebooks[m*k*dsub], codes n*m
        if (!fits(len, q_off, 8)) return fail(e, VG_ERR_FORMAT, "file too short for PQ metadata");
        const uint64_t m = rd32(data + q_off), k = rd32(data + q_off + 4);
        if (m == 0 || dim % m != 0)
            return fail(e, VG_ERR_FORMAT, "flat segment: %llu sub-quantizers do not divide dimension %llu",
                        static_cast<unsigned long long>(m), static_cast<unsigned long long>(dim));
        // (m <= dim < 2^31: 8 + m * 8 cannot wrap)
        if (!fits(len, q_off, 8 + m * 8) || !fits(len, q_off + 8 + m * 8, m, k, dim / m))
            return fail(e, VG_ERR_FORMAT, "file too short for PQ metadata");
        if (!fits(len, codes_off, n, m)) return fail(e, VG_ERR_FORMAT, "file too short for codes");
        L.pq_m = m;
        L.pq_k = k;
        L.pq_scales_offsets = section(q_off + 8, m * 8);
        L.pq_codebooks = section(q_off + 8 + m * 8, m * k * (dim / m));
        L.codes = section(codes_off, n * m);
    } else if (L.qtype != 0) {
        return fail(e, VG_ERR_FORMAT, "flat segment: unknown quantization type %d", L.qtype);
    }
    if (!fits(len, vec_off, n, dim, 4)) return fail(e, VG_ERR_FORMAT, "file too short for vectors");  // segment.go:283-289
    L.vectors = section(vec_off, n * dim * 4);
    if (L.partitions > 0) {  // segment.go:187-207: centroids [P*dim] fp32, partition offsets [P+1] uint32
        const uint64_t p = L.partitions;
        if (!fits(len, cent_off, p, dim, 4)) return fail(e, VG_ERR_FORMAT, "file too short for centroids");
        if (!fits(len, poff_off, p + 1, 4)) return fail(e, VG_ERR_FORMAT, "file too short for partition offsets");
        L.centroids = section(cent_off, p * dim * 4);
        L.part_offsets = section(poff_off, (p + 1) * 4);
    }
    return VG_OK;
}

// ---- DiskANN segment ---------------------------------------------------------------------------------------------------------
struct DiskLayout {
    uint64_t segment_id = 0;
    uint32_t rows = 0;
    int32_t dim = 0;
    int32_t metric = 0;
    uint32_t max_degree_raw = 0;
    int32_t search_list_size = 0;
    uint32_t entrypoint = 0;
    int qtype = 0;  // quantization.Type (types.go:6-14): 1 PQ, 5 RaBitQ, 6 INT4
    uint32_t pq_m = 0, pq_k = 0;
    Section vectors;            // rows * dim fp32
    Section graph;              // rows * max_degree uint32; absent when it does not fit or the degree is not 1..64 (see below)
    Section pq_codes;           // qtype 1: rows * m
    Section pq_scales_offsets;  // qtype 1: scales[m], offsets[m]
    Section pq_codebooks;       // qtype 1: m * k * (dim / m) int8
    Section rabitq_codes;       // qtype 5: rows * code bytes
    Section int4_params;        // qtype 6: min[dim], diff[dim] fp32 (behind the u32 dim)
    Section int4_codes;         // qtype 6: rows * ceil(dim / 2)
};

inline uint64_t rabitq_code_bytes(uint64_t dim) { return ((dim + 63) / 64) * 8 + 4; }  // rabitq.go:51-78

inline int32_t parse_diskann(const uint8_t *data, uint64_t len, bool verify_checksum, DiskLayout &L, Error &e)
{
    if (len < kDiskHeader) return fail(e, VG_ERR_FORMAT, "buffer too small for header");  // format.go:81-83
    if (rd32(data) != kDiskMagic) return fail(e, VG_ERR_FORMAT, "invalid magic number");
    const uint32_t version = rd32(data + 4);
    if (version != 2 && version != 1) return fail(e, VG_ERR_FORMAT, "unsupported version");
    L.segment_id = rd64(data + 8);
    L.rows = rd32(data + 16);
    L.dim = static_cast<int32_t>(rd32(data + 20));
    L.metric = data[24];
    L.max_degree_raw = rd32(data + 25);
    L.search_list_size = static_cast<int32_t>(rd32(data + 29));
    L.entrypoint = rd32(data + 33);
    L.qtype = data[37];
    L.pq_m = rd16(data + 38);
    L.pq_k = rd16(data + 40);
    // data[42] = CompressionType (format.go:32).  The writer records its option there (LZ4 by default,
    // writer.go:92,676) but streams every section raw (writer.go:697-740) and Open never consults the
    // field (segment.go:165-440), so neither does this reader: a default reference segment has 1 here.
    const uint64_t vec_off = rd64(data + 48), graph_off = rd64(data + 56), pq_codes_off = rd64(data + 64),
                   bq_codes_off = rd64(data + 72), cb_off = rd64(data + 80), pk_off = rd64(data + 88);
    const uint32_t checksum = rd32(data + 120);
    if (L.dim <= 0) return fail(e, VG_ERR_FORMAT, "diskann segment: dimension 0");
    if (L.metric > VG_METRIC_DOT) return fail(e, VG_ERR_UNSUPPORTED, "diskann segment: metric %d has no float32 kernels", L.metric);
    if (verify_checksum && verify_body(data, len, kDiskHeader, checksum, e) != VG_OK) return e.status;
    const uint64_t n = L.rows, dim = static_cast<uint64_t>(L.dim);
    if (!fits(len, pk_off, n, 8))  // segment.go:177-182 (the message's sum may wrap: it is only printed)
        return fail(e, VG_ERR_FORMAT, "file size too small: expected at least %llu, got %llu",
                    static_cast<unsigned long long>(pk_off + n * 8), static_cast<unsigned long long>(len));
    if (!fits(len, vec_off, n, dim, 4)) return fail(e, VG_ERR_FORMAT, "vector section out of bounds");
    L.vectors = section(vec_off, n * dim * 4);
    // the reference reads a node's neighbour list when a search visits it (segment.go:1376-1391) and never
    // checks the graph section in Open: a segment whose graph section does not fit (or whose degree this
    // library cannot walk) opens without a graph, and searching it fails then
    const uint64_t r = L.max_degree_raw;
    if (r >= 1 && r <= 64 && fits(len, graph_off, n, r, 4)) L.graph = section(graph_off, n * r * 4);
    if (L.qtype == 1) {  // segment.go:305-376 loadPQ: codes n*m; scales[m], offsets[m], codebooks[m*k*subDim]
        const uint64_t m = L.pq_m, k = L.pq_k;
        if (m == 0 || dim % m != 0)
            return fail(e, VG_ERR_FORMAT, "diskann segment: %llu sub-quantizers do not divide dimension %llu",
                        static_cast<unsigned long long>(m), static_cast<unsigned long long>(dim));
        if (!fits(len, pq_codes_off, n, m)) return fail(e, VG_ERR_FORMAT, "PQ codes section out of bounds");
        // (m < 2^16: m * 8 and cb_off + m * 8 — cb_off <= len after the first test — cannot wrap)
        if (!fits(len, cb_off, m * 8) || !fits(len, cb_off + m * 8, m, k, dim / m))
            return fail(e, VG_ERR_FORMAT, "failed to read PQ codebooks: out of bounds");
        L.pq_codes = section(pq_codes_off, n * m);
        L.pq_scales_offsets = section(cb_off, m * 8);
        L.pq_codebooks = section(cb_off + m * 8, m * k * (dim / m));
    } else if (L.qtype == 5) {  // segment.go:1393-1408 loadRaBitQ: n * (((dim+63)/64)*8 + 4) bytes
        const uint64_t per = rabitq_code_bytes(dim);
        if (!fits(len, bq_codes_off, n, per)) return fail(e, VG_ERR_FORMAT, "RaBitQ codes section out of bounds");
        L.rabitq_codes = section(bq_codes_off, n * per);
    } else if (L.qtype == 6) {  // segment.go:378-416 loadINT4: params = [dim u32][min f32 x dim][diff f32 x dim]
        if (cb_off == 0) return fail(e, VG_ERR_FORMAT, "missing INT4 params");
        if (pk_off <= cb_off) return fail(e, VG_ERR_FORMAT, "invalid INT4 params size");
        const uint64_t psize = pk_off - cb_off;
        if (!fits(len, cb_off, psize)) return fail(e, VG_ERR_FORMAT, "INT4 params out of bounds");
        if (psize < 4) return fail(e, VG_ERR_FORMAT, "data too short");  // int4.go:191-193
        if (rd32(data + cb_off) != dim || psize != 4 + dim * 8) return fail(e, VG_ERR_FORMAT, "data size mismatch");
        if (pq_codes_off == 0) return fail(e, VG_ERR_FORMAT, "missing INT4 codes");
        const uint64_t cs = (dim + 1) / 2;
        if (!fits(len, pq_codes_off, n, cs)) return fail(e, VG_ERR_FORMAT, "INT4 codes out of bounds");
        L.int4_params = section(cb_off + 4, dim * 8);
        L.int4_codes = section(pq_codes_off, n * cs);
    } else if (L.qtype != 0) {
        return fail(e, VG_ERR_UNSUPPORTED, "diskann segment: quantization type %d (OPQ / SQ8 / BQ) has no device scorer yet", L.qtype);
    }
    return VG_OK;
}

}  // namespace seglayout
}  // namespace vg
