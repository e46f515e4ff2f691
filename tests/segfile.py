"""Test-side WRITERS of the reference's segment file layouts, written from the format
definitions (internal/segment/flat/format.go:27-136, internal/segment/diskann/format.go:19-79 and
the section order the reference's Open() functions read: flat/segment.go:186-300,
diskann/segment.go:165-440,1393-1408).  No Go toolchain exists here, so these images are the
fixtures for vg_segment_open_*: parity of the reader with real vecgo files is pinned by the
format definitions only ("parity unpinned" for byte-level quirks of the reference writer)."""
import struct

import numpy as np

FLAT_MAGIC, FLAT_HEADER = 0x56454331, 152
DISK_MAGIC, DISK_HEADER = 0x4449534B, 160


def crc32c_py(data: bytes) -> int:
    """Bitwise CRC-32C (Castagnoli), independent of the library's table version."""
    table = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        table.append(c)
    c = 0xFFFFFFFF
    for b in data:
        c = table[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _pad8(buf: bytearray):
    while len(buf) % 8:
        buf.append(0)


def write_flat(vectors, metric=0, segment_id=7, sq=None, pq=None, codes=None, checksum=True, magic=FLAT_MAGIC,
               version=1, partitions=None):
    """sq = (mins, maxs); pq = (m, k, scales, offsets, codebooks int8); codes uint8 [n, dim] or [n, m];
    partitions = (centroids [P, dim] fp32, offsets [P + 1] uint32) with the rows already grouped by partition."""
    v = np.ascontiguousarray(vectors, np.float32)
    n, dim = v.shape
    body = bytearray()
    qtype = q_off = c_off = 0
    nparts = cent_off = poff_off = 0
    if partitions is not None:  # flat/writer.go: centroids, then the partition offsets
        cent = np.ascontiguousarray(partitions[0], np.float32).reshape(-1, dim)
        nparts = cent.shape[0]
        cent_off = FLAT_HEADER + len(body); body += cent.tobytes(); _pad8(body)
        poff_off = FLAT_HEADER + len(body); body += np.ascontiguousarray(partitions[1], np.uint32).tobytes(); _pad8(body)
    if sq is not None:
        qtype, q_off = 1, FLAT_HEADER + len(body)
        body += np.asarray(sq[0], np.float32).tobytes() + np.asarray(sq[1], np.float32).tobytes()
        _pad8(body)
        c_off = FLAT_HEADER + len(body)
        body += np.ascontiguousarray(codes, np.uint8).tobytes()
        _pad8(body)
    elif pq is not None:
        m, k, scales, offsets, cb = pq
        qtype, q_off = 2, FLAT_HEADER + len(body)
        body += struct.pack("<II", m, k) + np.asarray(scales, np.float32).tobytes() + \
            np.asarray(offsets, np.float32).tobytes() + np.asarray(cb, np.int8).tobytes()
        _pad8(body)
        c_off = FLAT_HEADER + len(body)
        body += np.ascontiguousarray(codes, np.uint8).tobytes()
        _pad8(body)
    v_off = FLAT_HEADER + len(body)
    body += v.tobytes()
    pk_off = FLAT_HEADER + len(body)
    body += np.arange(n, dtype=np.uint64).tobytes()
    h = bytearray(FLAT_HEADER)
    struct.pack_into("<IIQII", h, 0, magic, version, segment_id, n, dim)
    h[24] = metric
    struct.pack_into("<I", h, 28, nparts)  # NumPartitions
    h[32] = qtype
    struct.pack_into("<QQQQQQQQ", h, 40, cent_off, poff_off, q_off, c_off, v_off, pk_off, 0, 0)
    struct.pack_into("<I", h, 104, crc32c_py(bytes(body)) if checksum else 0)
    return bytes(h) + bytes(body)


def write_diskann(vectors, graph, entry, metric=0, segment_id=9, pq=None, pq_codes=None, rabitq_codes=None, int4=None,
                  search_list=100, checksum=True, version=2, compression=1, qtype=None):
    v = np.ascontiguousarray(vectors, np.float32)
    g = np.ascontiguousarray(graph, np.uint32)
    n, dim = v.shape
    r = g.shape[1] if n else 0
    body = bytearray()
    v_off = DISK_HEADER + len(body); body += v.tobytes(); _pad8(body)
    g_off = DISK_HEADER + len(body); body += g.tobytes(); _pad8(body)
    pqc_off = bq_off = cb_off = 0
    m = k = 0
    qt = 0
    if pq is not None:
        m, k, scales, offsets, cb = pq
        qt = 1
        pqc_off = DISK_HEADER + len(body); body += np.ascontiguousarray(pq_codes, np.uint8).tobytes(); _pad8(body)
        cb_off = DISK_HEADER + len(body)
        body += np.asarray(scales, np.float32).tobytes() + np.asarray(offsets, np.float32).tobytes() + \
            np.asarray(cb, np.int8).tobytes()
        _pad8(body)
    elif rabitq_codes is not None:
        qt = 5
        bq_off = DISK_HEADER + len(body); body += np.ascontiguousarray(rabitq_codes, np.uint8).tobytes(); _pad8(body)
    if int4 is not None:   # (min, diff, codes): MarshalBinary params end where the PK section starts (segment.go:384)
        mn, df, icodes = int4
        qt = 6
        pqc_off = DISK_HEADER + len(body); body += np.ascontiguousarray(icodes, np.uint8).tobytes(); _pad8(body)
        cb_off = DISK_HEADER + len(body)
        body += struct.pack("<I", dim) + np.asarray(mn, np.float32).tobytes() + np.asarray(df, np.float32).tobytes()
    if qtype is not None:
        qt = qtype
    pk_off = DISK_HEADER + len(body)
    body += np.arange(n, dtype=np.uint64).tobytes()
    h = bytearray(DISK_HEADER)
    struct.pack_into("<IIQII", h, 0, DISK_MAGIC, version, segment_id, n, dim)
    h[24] = metric
    struct.pack_into("<III", h, 25, r, search_list, entry)
    h[37] = qt
    struct.pack_into("<HH", h, 38, m, k)
    h[42] = compression
    struct.pack_into("<QQQQQQQQQ", h, 48, v_off, g_off, pqc_off, bq_off, cb_off, pk_off, 0, 0, 0)
    struct.pack_into("<I", h, 120, crc32c_py(bytes(body)) if checksum else 0)
    return bytes(h) + bytes(body)
