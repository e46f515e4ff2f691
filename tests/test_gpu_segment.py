"""Segment images (flat/format.go, diskann/format.go) opened straight onto the GPU
(SURVEY.md §8f rank 2): what comes out of the resident index must equal the oracle run on the
arrays that went into the file."""
import numpy as np
import pytest

from oracle import oracle as o
from tests import segfile
from tests.graphs import build_vamana

pytestmark = pytest.mark.gpu


def bits(x):
    return np.asarray(x, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def vg():
    import vecgo_amd
    return vecgo_amd


@pytest.fixture(scope="module")
def ctx(vg):
    return vg.Context(0)


def _pq_params(rng, dim, m, k=256):
    return (m, k, (rng.random(m) * 0.02 + 0.005).astype(np.float32), (rng.standard_normal(m) * 0.05).astype(np.float32),
            rng.integers(-128, 128, m * k * (dim // m)).astype(np.int8))


def test_flat_segment_plain_sq8_pq(vg, ctx):
    rng = np.random.default_rng(1)
    n, dim, nq, k = 3000, 64, 5, 10
    x = rng.standard_normal((n, dim)).astype(np.float32)
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    # --- no quantization: exact search -------------------------------------------------------
    seg = vg.Segment(ctx, segfile.write_flat(x, metric=0, segment_id=42))
    assert (seg.info.kind, seg.info.rows, seg.info.dim, seg.info.metric, seg.info.quantization,
            seg.info.segment_id) == (0, n, dim, 0, 0, 42)
    ids, sc = seg.index.search_flat(q, k)
    for i in range(nq):
        eid, esc = o.flat_search_f32(x, dim, q[i], k)
        assert np.array_equal(ids[i], eid) and np.array_equal(bits(sc[i]), bits(esc))
    seg.close()
    # --- SQ8: bounds from the file (SetBounds), codes scanned -----------------------------------
    sq = o.ScalarQuantizer(dim); sq.train(x)
    codes = sq.encode_batch(x)
    img = segfile.write_flat(x, sq=(sq.mins, sq.maxs), codes=codes)
    seg = vg.Segment(ctx, np.frombuffer(bytes(1) + img, np.uint8)[1:])   # byte-aligned only: the reader copies
    assert seg.info.quantization == 3
    ref = o.ScalarQuantizer(dim)        # SetBounds semantics (quantizer.go:52-78), not Train's
    ref.mins, ref.maxs = sq.mins.copy(), sq.maxs.copy()
    ref.inv_scales = ((ref.maxs - ref.mins) / np.float32(255.0)).astype(np.float32)
    ids, sc = seg.index.search_sq8(q, k)
    for i in range(nq):
        eid, esc = o.flat_search_sq8(ref, codes, q[i], k)
        assert np.array_equal(ids[i], eid) and np.array_equal(bits(sc[i]), bits(esc))
    eid, esc = o.flat_search_f32(x, dim, q[0], k)               # the fp32 rows are resident too (rerank)
    fid, fsc = seg.index.search_flat(q[:1], k)
    assert np.array_equal(fid[0], eid) and np.array_equal(bits(fsc[0]), bits(esc))
    seg.close()
    # --- PQ: codebooks + codes from the file -----------------------------------------------------
    m = 8
    pqp = _pq_params(rng, dim, m)
    opq = o.ProductQuantizer(dim, m, 256); opq.set_codebooks(pqp[4], pqp[2], pqp[3])
    pcodes = opq.encode_batch(x)
    seg = vg.Segment(ctx, segfile.write_flat(x, metric=0, pq=pqp, codes=pcodes))
    assert (seg.info.quantization, seg.info.pq_m, seg.info.pq_k) == (1, m, 256)
    ids, sc = seg.index.search_pq_adc(q, k)
    for i in range(nq):
        eid, esc = o.flat_search_pq(opq, pcodes, q[i], k)
        assert np.array_equal(ids[i], eid) and np.array_equal(bits(sc[i]), bits(esc))
    seg.close()


def test_diskann_segment_fp32_pq_rabitq(vg, ctx):
    rng = np.random.default_rng(2)
    n, dim, nq, k, r = 1500, 64, 6, 10, 16
    x = rng.standard_normal((n, dim)).astype(np.float32)
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    g, _ = build_vamana(x, r=r, seed=1)
    entry = 17
    m = 8
    pqp = _pq_params(rng, dim, m)
    opq = o.ProductQuantizer(dim, m, 256); opq.set_codebooks(pqp[4], pqp[2], pqp[3])
    pcodes = opq.encode_batch(x)
    rcodes = o.rabitq_encode_batch(x, dim)
    cases = [(dict(), 0, o.VamanaIndex(g, entry, dim, kind=0, base=x)),
             (dict(pq=pqp, pq_codes=pcodes), 1, o.VamanaIndex(g, entry, dim, kind=1, base=x, pq=opq, codes=pcodes)),
             (dict(rabitq_codes=rcodes), 2, o.VamanaIndex(g, entry, dim, kind=2, base=x, codes=rcodes))]
    for extra, kind, oracle_index in cases:
        seg = vg.Segment(ctx, segfile.write_diskann(x, g, entry, **extra), kind="diskann")
        assert (seg.info.kind, seg.info.max_degree, seg.info.entrypoint, seg.info.search_list_size) == (1, r, entry, 100)
        ids, sc, st = seg.index.search_vamana(q, k, kind=kind, stats=True)
        for i in range(nq):
            eid, esc, est = oracle_index.search(q[i], k)
            assert np.array_equal(ids[i, :eid.size], eid), (kind, i)
            assert np.array_equal(bits(sc[i, :eid.size]), bits(esc)), (kind, i)
            assert int(st[i, 1]) == est.distance_computations
        # Segment.search = diskann.Segment.Search: the distFn follows the file's quantization
        sid, ssc = seg.search(q, k)
        assert np.array_equal(sid, ids) and np.array_equal(bits(ssc), bits(sc))
        seg.close()


def test_segment_errors(vg, ctx):
    rng = np.random.default_rng(3)
    x = rng.standard_normal((50, 16)).astype(np.float32)
    good = segfile.write_flat(x)

    def msg(image, kind="flat", **kw):
        with pytest.raises(vg.VecgoHipError) as e:
            vg.Segment(ctx, image, kind=kind, **kw)
        return str(e.value)

    assert "invalid magic number" in msg(segfile.write_flat(x, magic=0x12345678))      # format.go:141-143
    assert "unsupported version" in msg(segfile.write_flat(x, version=9))              # format.go:145-147
    assert "buffer too small for header" in msg(good[:100])
    assert "checksum mismatch" in msg(good[:segfile.FLAT_HEADER + 100])                # verified before the sections
    assert "file too short for vectors" in msg(good[:segfile.FLAT_HEADER + 100], verify_checksum=False)
    corrupt = bytearray(good); corrupt[-1] ^= 0x40
    assert "checksum mismatch" in msg(bytes(corrupt))                                  # segment.go:170-180
    vg.Segment(ctx, bytes(corrupt), verify_checksum=False).close()                     # WithVerifyChecksum(false)
    vg.Segment(ctx, segfile.write_flat(x, checksum=False)).close()                     # Checksum == 0: not verified
    assert "invalid magic number" in msg(good, kind="diskann")
    g = np.zeros((50, 4), np.uint32)
    # the writer's default header says LZ4 (writer.go:92) over raw sections; Open ignores the byte
    vg.Segment(ctx, segfile.write_diskann(x, g, 0, compression=1), kind="diskann").close()
    assert "quantization type 4" in msg(segfile.write_diskann(x, g, 0, qtype=4), kind="diskann")  # BQ
    assert "missing INT4 params" in msg(segfile.write_diskann(x, g, 0, qtype=6), kind="diskann")
    d = segfile.write_diskann(x, g, 0)
    assert "file size too small" in msg(d[:segfile.DISK_HEADER + 64], kind="diskann", verify_checksum=False)
    empty = vg.Segment(ctx, segfile.write_flat(np.zeros((0, 16), np.float32)))
    ids, sc = empty.index.search_flat(np.zeros((1, 16), np.float32), 3)
    assert np.all(ids == 0xFFFFFFFF)


def test_hostile_headers_do_not_wrap(vg, ctx):
    """Header fields are untrusted 32/64-bit numbers: a row count x dimension x 4 that wraps around 2^64 (or an
    offset near 2^64) must be refused by the section checks, not pass them (`x + y >= x` does not catch a wrapped
    product)."""
    import struct
    rng = np.random.default_rng(4)
    x = rng.standard_normal((64, 16)).astype(np.float32)
    good = bytearray(segfile.write_flat(x, checksum=False))

    def refused(image, kind="flat"):
        with pytest.raises(vg.VecgoHipError) as e:
            vg.Segment(ctx, bytes(image), kind=kind, verify_checksum=False)
        assert e.value.status in (-10, -1), e.value      # a section check, or the index refusing 2^32 - 1 rows
        return str(e.value)

    img = bytearray(good)                       # rows * dim * 4 = 2^64 + small: the product wraps to a tiny number
    struct.pack_into("<I", img, 16, 0x80000000)  # rows = 2^31
    struct.pack_into("<I", img, 20, 0x40000000)  # dim  = 2^30: rows * dim * 4 = 2^63, + the offset
    assert "too short" in refused(img)
    img = bytearray(good)
    struct.pack_into("<I", img, 16, 0xFFFFFFFE)
    struct.pack_into("<I", img, 20, 0x7FFFFFFF)
    assert "too short" in refused(img)
    img = bytearray(good)
    struct.pack_into("<Q", img, 72, 0xFFFFFFFFFFFFFFF0)   # vector offset near 2^64: offset + bytes wraps
    assert "too short" in refused(img)
    g = np.zeros((64, 4), np.uint32)
    d = bytearray(segfile.write_diskann(x, g, 0, checksum=False))
    img = bytearray(d)
    struct.pack_into("<I", img, 25, 0x80000001)           # max degree: was cast to int32 and sign-extended
    seg = vg.Segment(ctx, bytes(img), kind="diskann", verify_checksum=False)   # opens (the reference checks no graph bounds in Open)
    with pytest.raises(vg.VecgoHipError) as e:                               # ... and has no graph to walk
        seg.index.search_vamana(np.zeros((1, 16), np.float32), 3)
    assert e.value.status == -9
    seg.close()
    img = bytearray(d)
    struct.pack_into("<Q", img, 56, 0xFFFFFFFFFFFFFF00)   # graph offset near 2^64
    seg = vg.Segment(ctx, bytes(img), kind="diskann", verify_checksum=False)
    with pytest.raises(vg.VecgoHipError):
        seg.index.search_vamana(np.zeros((1, 16), np.float32), 3)
    seg.close()
    img = bytearray(d)
    struct.pack_into("<I", img, 16, 0xFFFFFFFE)           # rows
    struct.pack_into("<I", img, 20, 0x7FFFFFFF)           # dim
    assert "too small" in refused(img, "diskann") or "out of bounds" in refused(img, "diskann")
