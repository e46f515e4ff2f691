"""Header layouts of the reference's segment files as (field, offset, width) triples, transcribed from the
PutUint* / buf[i] lines of FileHeader.Encode — flat: internal/segment/flat/format.go:110-133 (HeaderSize :108),
DiskANN: internal/segment/diskann/format.go:51-79 (HeaderSize :49) — and checked against BOTH sides of this repository
that were written from those files: the reader (vecgo_amd/csrc/vg_segment.hip: every rd16/rd32/rd64/data[i] of the
header) and the test-side writer (tests/segfile.py: an image with distinctive field values decoded at the table's
offsets).  Parity with real files the reference wrote stays unpinned (no Go toolchain here, no segment fixtures in the
reference's tree): what this pins is that reader and writer agree with the reference's Encode, field by field."""
import re
import struct
from pathlib import Path

import numpy as np

from tests import segfile

ROOT = Path(__file__).resolve().parents[1]

# (field, offset, width in bytes) — flat/format.go:110-133
FLAT = [
    ("Magic", 0, 4),                    # :112 PutUint32(buf[0:])
    ("Version", 4, 4),                  # :113
    ("SegmentID", 8, 8),                # :114 PutUint64(buf[8:])
    ("RowCount", 16, 4),                # :115
    ("Dim", 20, 4),                     # :116
    ("Metric", 24, 1),                  # :117 buf[24]
    ("NumPartitions", 28, 4),           # :118 (3 bytes of padding before it)
    ("QuantizationType", 32, 1),        # :119 buf[32]; padding [33:40]
    ("CentroidOffset", 40, 8),          # :121
    ("PartitionOffsetOffset", 48, 8),   # :122
    ("QuantizationOffset", 56, 8),      # :123
    ("CodesOffset", 64, 8),             # :124
    ("VectorOffset", 72, 8),            # :125
    ("PKOffset", 80, 8),                # :126
    ("MetadataOffset", 88, 8),          # :127
    ("BlockStatsOffset", 96, 8),        # :128
    ("Checksum", 104, 4),               # :129
]
FLAT_HEADER_SIZE = 4 + 4 + 8 + 4 + 4 + 1 + 3 + 4 + 1 + 7 + 8 + 8 + 8 + 8 + 8 + 8 + 8 + 8 + 48   # format.go:108

# diskann/format.go:51-79
DISK = [
    ("Magic", 0, 4),                    # :53
    ("Version", 4, 4),                  # :54
    ("SegmentID", 8, 8),                # :55
    ("RowCount", 16, 4),                # :56
    ("Dim", 20, 4),                     # :57
    ("Metric", 24, 1),                  # :58 buf[24]
    ("MaxDegree", 25, 4),               # :59 PutUint32(buf[25:]) — unaligned
    ("SearchListSize", 29, 4),          # :60
    ("Entrypoint", 33, 4),              # :61
    ("QuantizationType", 37, 1),        # :62 buf[37]
    ("PQSubvectors", 38, 2),            # :63 PutUint16
    ("PQCentroids", 40, 2),             # :64
    ("CompressionType", 42, 1),         # :65 buf[42]; padding [43:48]
    ("VectorOffset", 48, 8),            # :67
    ("GraphOffset", 56, 8),             # :68
    ("PQCodesOffset", 64, 8),           # :69
    ("BQCodesOffset", 72, 8),           # :70
    ("PQCodebookOffset", 80, 8),        # :71
    ("PKOffset", 88, 8),                # :72
    ("MetadataOffset", 96, 8),          # :73
    ("BlockStatsOffset", 104, 8),       # :74
    ("MetadataIndexOffset", 112, 8),    # :75
    ("Checksum", 120, 4),               # :76
]
DISK_HEADER_SIZE = 4 + 4 + 8 + 4 + 4 + 1 + 4 + 4 + 4 + 1 + 2 + 2 + 1 + 5 + 8 + 8 + 8 + 8 + 8 + 8 + 8 + 8 + 8 + 4 + 36  # format.go:49


def test_tables_are_consistent():
    for table, size in ((FLAT, FLAT_HEADER_SIZE), (DISK, DISK_HEADER_SIZE)):
        end = 0
        for name, off, width in table:
            assert off >= end, (name, off, end)           # fields do not overlap, ascending
            end = off + width
        assert end <= size
    assert FLAT_HEADER_SIZE == 152 == segfile.FLAT_HEADER and DISK_HEADER_SIZE == 160 == segfile.DISK_HEADER
    assert segfile.FLAT_MAGIC == 0x56454331 and segfile.DISK_MAGIC == 0x4449534B   # format.go:12 / :9


LAYOUT_HPP = ROOT / "vecgo_amd" / "csrc" / "vg_segment_layout.hpp"


def reader_accesses(func: str):
    """{offset: width} of every header access in seglayout::parse_<func> (vg_segment_layout.hpp, what vg_segment_open_<func>
    validates an image with): rdNN(data + K), rdNN(data), data[K]."""
    src = LAYOUT_HPP.read_text()
    body = src[src.index(f"inline int32_t parse_{func}("):]
    body = body[:body.index("\n}\n")]
    acc = {}
    for m in re.finditer(r"\brd(16|32|64)\(data(?: \+ (\d+))?\)", body):
        acc[int(m.group(2) or 0)] = int(m.group(1)) // 8
    for m in re.finditer(r"\bdata\[(\d+)\]", body):
        acc[int(m.group(1))] = 1
    return acc


def test_reader_reads_the_fields_where_the_reference_writes_them():
    for func, table, consts in (("flat", FLAT, ("kFlatHeader", 152, "kFlatMagic", 0x56454331)),
                                ("diskann", DISK, ("kDiskHeader", 160, "kDiskMagic", 0x4449534B))):
        by_off = {off: (name, width) for name, off, width in table}
        acc = reader_accesses(func)
        assert len(acc) >= 10, acc
        for off, width in acc.items():
            assert off in by_off, f"parse_{func} reads header offset {off}: not a field of the reference's header"
            assert by_off[off][1] == width, f"{func}: {by_off[off][0]} is {by_off[off][1]} bytes, read as {width}"
        src = LAYOUT_HPP.read_text()
        hname, hsize, mname, magic = consts
        assert re.search(rf"{hname}\s*=\s*{hsize}\b", src) and re.search(rf"{mname}\s*=\s*0x{magic:08X}", src, flags=re.I)
    # the fields the searches depend on are all read
    need_flat = {"SegmentID", "RowCount", "Dim", "Metric", "NumPartitions", "QuantizationType", "CentroidOffset",
                 "PartitionOffsetOffset", "QuantizationOffset", "CodesOffset", "VectorOffset", "Checksum"}
    got = {name for name, off, _ in FLAT if off in reader_accesses("flat")}
    assert need_flat <= got, need_flat - got
    need_disk = {"SegmentID", "RowCount", "Dim", "Metric", "MaxDegree", "SearchListSize", "Entrypoint", "QuantizationType",
                 "PQSubvectors", "PQCentroids", "VectorOffset", "GraphOffset", "PQCodesOffset", "BQCodesOffset",
                 "PQCodebookOffset", "PKOffset", "Checksum"}
    got = {name for name, off, _ in DISK if off in reader_accesses("diskann")}
    assert need_disk <= got, need_disk - got


def decode(image: bytes, table):
    out = {}
    for name, off, width in table:
        out[name] = int.from_bytes(image[off:off + width], "little")
    return out


def test_writer_puts_the_fields_where_the_reference_does():
    rng = np.random.default_rng(0)
    n, dim = 6, 16
    v = rng.standard_normal((n, dim)).astype(np.float32)
    cent = rng.standard_normal((2, dim)).astype(np.float32)
    codes = rng.integers(0, 256, (n, dim), dtype=np.uint8)
    img = segfile.write_flat(v, metric=2, segment_id=0x1122334455667788, sq=(np.zeros(dim), np.ones(dim)), codes=codes,
                             partitions=(cent, np.array([0, 3, 6], np.uint32)))
    h = decode(img, FLAT)
    assert (h["Magic"], h["Version"], h["SegmentID"], h["RowCount"], h["Dim"], h["Metric"], h["NumPartitions"],
            h["QuantizationType"]) == (0x56454331, 1, 0x1122334455667788, n, dim, 2, 2, 1)
    assert h["CentroidOffset"] == 152 and h["PartitionOffsetOffset"] == 152 + 2 * dim * 4
    assert 152 < h["QuantizationOffset"] < h["CodesOffset"] < h["VectorOffset"] < h["PKOffset"] <= len(img)
    assert img[h["VectorOffset"]:h["VectorOffset"] + n * dim * 4] == v.tobytes()
    assert h["Checksum"] == segfile.crc32c_py(img[152:])
    assert img[25:28] == bytes(3) and img[33:40] == bytes(7) and img[108:152] == bytes(44)   # the paddings stay zero

    g = rng.integers(0, n, (n, 4)).astype(np.uint32)
    rq = rng.integers(0, 256, (n, 12), dtype=np.uint8)
    img = segfile.write_diskann(v, g, entry=5, metric=1, segment_id=0x0102030405060708, rabitq_codes=rq, search_list=77)
    h = decode(img, DISK)
    assert (h["Magic"], h["Version"], h["SegmentID"], h["RowCount"], h["Dim"], h["Metric"], h["MaxDegree"], h["SearchListSize"],
            h["Entrypoint"], h["QuantizationType"], h["CompressionType"]) == \
        (0x4449534B, 2, 0x0102030405060708, n, dim, 1, 4, 77, 5, 5, 1)
    assert h["VectorOffset"] == 160 and h["GraphOffset"] == 160 + n * dim * 4
    assert img[h["GraphOffset"]:h["GraphOffset"] + n * 4 * 4] == g.tobytes()
    assert img[h["BQCodesOffset"]:h["BQCodesOffset"] + n * 12] == rq.tobytes()
    assert h["Checksum"] == segfile.crc32c_py(img[160:]) and img[43:48] == bytes(5) and img[124:160] == bytes(36)
    pq = (4, 256, np.ones(4, np.float32), np.zeros(4, np.float32), rng.integers(-128, 128, 4 * 256 * 4).astype(np.int8))
    img = segfile.write_diskann(v, g, entry=1, pq=pq, pq_codes=rng.integers(0, 256, (n, 4), dtype=np.uint8))
    h = decode(img, DISK)
    assert (h["QuantizationType"], h["PQSubvectors"], h["PQCentroids"]) == (1, 4, 256)   # quantization.Type, types.go:6-14
    assert h["GraphOffset"] < h["PQCodesOffset"] < h["PQCodebookOffset"] < h["PKOffset"]
