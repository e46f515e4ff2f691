"""INT4 on the GPU (int4.go, int4_avx512.c, diskann/segment.go:378-416,558-565) vs the oracle:
parameters, codes, decoded values, both distance orders and the Vamana search with INT4 node
scoring, bit-exact."""
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


@pytest.mark.parametrize("n,dim", [(400, 128), (200, 768), (257, 100), (64, 17), (1, 3), (300, 33), (500, 64)])
def test_train_encode_decode_distances_match_oracle(vg, ctx, n, dim):
    rng = np.random.default_rng(n + dim)
    x = rng.standard_normal((n, dim)).astype(np.float32)
    x[:, dim // 2] = -0.75   # constant dimension: diff 0 -> 1 (int4.go:53-58)
    iq = vg.Int4Quantizer(ctx, dim)
    assert not iq.is_trained()
    with pytest.raises(vg.VecgoHipError):
        iq.encode(x[:1])
    iq.train(x)
    ref = o.Int4Quantizer(dim); ref.train(x)
    mn, df, tb = iq.params()
    assert np.array_equal(bits(mn), bits(ref.min)) and np.array_equal(bits(df), bits(ref.diff))
    assert np.array_equal(bits(tb), bits(ref.table))
    y = np.vstack([x[:30], x[:2] * 9.0, x[:2] * -9.0]).astype(np.float32)   # incl. values to clamp
    codes = iq.encode(y)
    assert codes.shape == (y.shape[0], (dim + 1) // 2)
    assert np.array_equal(codes, ref.encode_batch(y))
    assert np.array_equal(bits(iq.decode(codes)), bits(np.stack([ref.decode(c) for c in codes])))
    q = rng.standard_normal(dim).astype(np.float32)
    assert np.array_equal(bits(iq.l2_distance_batch(q, codes)), bits(ref.l2_distance_batch(q, codes)))
    want = np.array([ref.l2_distance(q, c) for c in codes], np.float32)
    assert np.array_equal(bits(iq.l2_distance(q, codes)), bits(want))


# dim % 64 == 0: both distances run as streaming scans (int4_scan_kernel: rows turned through LDS, pair table);
# row pieces of 128 / 96 / 64 / 32 bytes, ragged last tile and workgroup
@pytest.mark.parametrize("n,dim", [(700, 192), (1000, 320), (513, 1024), (64, 64), (65, 256), (3000, 768)])
def test_int4_scan_kernels_match_oracle(vg, ctx, n, dim):
    rng = np.random.default_rng(n * 7 + dim)
    x = (rng.standard_normal((n, dim)) * rng.random(dim) * 3).astype(np.float32)
    iq = vg.Int4Quantizer(ctx, dim); iq.train(x)
    ref = o.Int4Quantizer(dim); ref.train(x)
    codes = iq.encode(x)
    assert np.array_equal(codes, ref.encode_batch(x))
    for qi in range(2):
        q = (rng.standard_normal(dim) * 2).astype(np.float32)
        assert np.array_equal(bits(iq.l2_distance_batch(q, codes)), bits(ref.l2_distance_batch(q, codes)))
        want = np.array([ref.l2_distance(q, c) for c in codes], np.float32)
        assert np.array_equal(bits(iq.l2_distance(q, codes)), bits(want))


@pytest.mark.parametrize("n,dim", [(400_003, 512), (250_001, 768), (230_000, 1024)])
def test_int4_scan_persistent_waves_many_tiles(vg, ctx, n, dim):
    """more tiles than the device holds waves (256 CUs x 12): every wave of int4_scan_tab_kernel walks several tiles —
    the next tile's first piece requested during the last piece of the current one, the scalar tile base advanced, a
    ragged last tile whose missing rows re-read row n - 1 — for both summation orders, every row compared"""
    rng = np.random.default_rng(n + dim)
    mn = (rng.standard_normal(dim) * 0.5).astype(np.float32); df = (rng.random(dim) * 3 + 0.1).astype(np.float32)
    iq = vg.Int4Quantizer(ctx, dim); iq.set_params(mn, df)
    ref = o.Int4Quantizer(dim); ref.set_params(mn, df)
    codes = rng.integers(0, 256, (n, dim // 2), dtype=np.uint8)
    q = (rng.standard_normal(dim) * 2).astype(np.float32)
    assert np.array_equal(bits(iq.l2_distance_batch(q, codes)), bits(ref.l2_distance_batch(q, codes)))
    got = iq.l2_distance(q, codes)
    want = ref.l2_distance_many(q, codes) if hasattr(ref, "l2_distance_many") else None
    if want is None:   # the per-code form row by row is slow in Python: every 97th row + the last tile
        pick = np.unique(np.concatenate([np.arange(0, n, 97), np.arange(max(0, n - 130), n)]))
        want = np.array([ref.l2_distance(q, codes[i]) for i in pick], np.float32)
        got = got[pick]
    assert np.array_equal(bits(got), bits(want))


def test_set_params_is_unmarshal_binary(vg, ctx):
    rng = np.random.default_rng(4)
    dim = 48
    mn = rng.standard_normal(dim).astype(np.float32); df = (rng.random(dim) + 0.2).astype(np.float32)
    iq = vg.Int4Quantizer(ctx, dim); iq.set_params(mn, df)
    ref = o.Int4Quantizer(dim); ref.set_params(mn, df)
    assert iq.is_trained()
    assert np.array_equal(bits(iq.params()[2]), bits(ref.table))


# dim % 32 == 0: the walk evaluates the table's terms in place (int4_l2_direct); other dims read the table
@pytest.mark.parametrize("dim", [96, 100, 32, 768, 47])
def test_vamana_search_with_int4_codes_and_segment_file(vg, ctx, dim):
    rng = np.random.default_rng(8)
    n, nq, k, r = 1200, 6, 10, 16
    x = rng.standard_normal((n, dim)).astype(np.float32)
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    g, entry = build_vamana(x, r=r, seed=2)
    ref = o.Int4Quantizer(dim); ref.train(x)
    codes = ref.encode_batch(x)
    oracle_index = o.VamanaIndex(g, entry, dim, kind=o.VAMANA_INT4, codes=codes, int4_table=ref.table)
    # through the API
    iq = vg.Int4Quantizer(ctx, dim); iq.train(x)
    idx = vg.Index(ctx, n, dim)
    idx.set_vamana_graph(g, entry)
    with pytest.raises(vg.VecgoHipError):
        idx.search_vamana(q, k, kind=3)                    # no INT4 codes attached yet
    idx.set_int4_codes(iq, codes)
    # and through a DiskANN segment image (quantization type 6, params before the PK section)
    seg = vg.Segment(ctx, segfile.write_diskann(x, g, entry, int4=(ref.min, ref.diff, codes)), kind="diskann")
    assert seg.info.quantization == 6
    for index in (idx, seg.index):
        ids, sc, st = index.search_vamana(q, k, kind=3, stats=True)
        for i in range(nq):
            eid, esc, est = oracle_index.search(q[i], k)
            assert np.array_equal(ids[i, :eid.size], eid), i
            assert np.array_equal(bits(sc[i, :eid.size]), bits(esc)), i
            assert int(st[i, 1]) == est.distance_computations
    seg.close()
    bad = segfile.write_diskann(x, g, entry, int4=(ref.min[:-1], ref.diff[:-1], codes))
    with pytest.raises(vg.VecgoHipError) as e:
        vg.Segment(ctx, bad, kind="diskann", verify_checksum=False)
    assert "data size mismatch" in str(e.value)            # int4.go:196-199


def test_simd_int4_reference_tests(vg, ctx, golden_dir):
    """internal/simd/int4_test.go (reference_kats.json simd_int4) through the C ABI: Int4L2DistanceBatch /
    Int4L2DistancePrecomputed of the reference's own codes, BuildInt4LookupTable's corners — and the same bits as the oracle"""
    import json
    g = json.loads((golden_dir / "reference_kats.json").read_text())["simd_int4"]
    for c in g["cases"]:
        dim = c["dim"]
        mn, df = np.array(c["min"], np.float32), np.array(c["diff"], np.float32)
        iq = vg.Int4Quantizer(ctx, dim); iq.set_params(mn, df)
        ref = o.Int4Quantizer(dim); ref.set_params(mn, df)
        q = np.array(c["query"], np.float32)
        codes = np.array(c["codes"], np.uint8)
        batch, one = iq.l2_distance_batch(q, codes), iq.l2_distance(q, codes)
        assert np.array_equal(bits(batch), bits(ref.l2_distance_batch(q, codes.reshape(-1))))
        assert np.array_equal(bits(one), bits(np.array([ref.l2_distance(q, code) for code in codes], np.float32)))
        if c.get("all_nonnegative"):
            assert np.all(batch >= 0) and np.all(one >= 0)
        for got in (batch, one):
            for v, e in zip(got, c.get("expected", [])):
                if e is not None:
                    assert abs(float(v) - e) <= c["tol"], (c["name"], v, e)
        if "table_len" in c:
            assert iq.params()[2].size == c["table_len"]
            assert np.all(np.abs(one - batch) <= c["precomputed_vs_direct_tol"])
    for t in g["lookup_table"]:
        iq = vg.Int4Quantizer(ctx, len(t["min"])); iq.set_params(np.array(t["min"], np.float32), np.array(t["diff"], np.float32))
        table = iq.params()[2]
        for e in t["expect"]:
            assert abs(float(table[e["dim"] * 16 + e["q"]]) - e["value"]) <= t["tol"], e
