"""PQ-ADC flat scan on the GPU vs the CPU oracle: bit-exact ids and scores.

Mirrors flat/pq_test.go + internal/simd/floats_test.go:328-447 (ADC) at the segment
level: every row scored with BuildDistanceTable + pqAdcLookupAvx512 order, top-k with
the (Score, RowID) tie-break."""
import numpy as np
import pytest

from oracle import oracle as o
from tests import hooks

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


def _random_pq(rng, dim, m, k=256):
    sd = dim // m
    opq = o.ProductQuantizer(dim, m, k)
    cb = rng.integers(-128, 128, m * k * sd).astype(np.int8)
    scales = (rng.random(m) * 0.02 + 0.005).astype(np.float32)
    offsets = ((rng.random(m) * 2 - 1) * 0.1).astype(np.float32)
    opq.set_codebooks(cb, scales, offsets)
    return opq


def _mk(vg, ctx, opq, codes, n):
    pq = vg.ProductQuantizer(ctx, opq.dim, opq.m, opq.k)
    pq.set_codebooks(opq.codebooks, opq.scales, opq.offsets)
    idx = vg.Index(ctx, n, opq.dim, vg.Metric.L2)
    idx.set_pq_codes(pq, codes)
    return pq, idx


@pytest.mark.parametrize("dim,m", [(768, 96), (128, 8), (128, 16), (200, 25), (64, 1), (272, 17),
                                   (512, 64)])
def test_build_distance_table_bit_exact(vg, ctx, dim, m):
    rng = np.random.default_rng(dim * 1000 + m)
    opq = _random_pq(rng, dim, m)
    pq = vg.ProductQuantizer(ctx, dim, m, 256)
    pq.set_codebooks(opq.codebooks, opq.scales, opq.offsets)
    q = rng.standard_normal((3, dim)).astype(np.float32)
    got = pq.build_distance_table(q)
    for i in range(3):
        assert np.array_equal(bits(got[i]), bits(opq.build_table(q[i])))


@pytest.mark.parametrize("n,dim,m,k,nq", [
    (10000, 768, 96, 10, 3),     # BASELINE shape, several tiles per wave
    (777, 768, 96, 10, 2),       # ragged last tile
    (64, 128, 16, 10, 1),        # exactly one tile
    (5, 128, 16, 10, 1),         # fewer rows than k
    (3000, 128, 8, 7, 2),        # m < 16: tail-only path
    (3000, 200, 25, 10, 2),      # one full group + 9 tail
    (3000, 272, 17, 1, 2),       # k = 1
    (20000, 128, 16, 200, 2),    # larger k
    (70000, 64, 4, 1000, 1),     # k near the fused limit, many compactions
    (9000, 1536, 192, 10, 3),    # d = 1536 at 8 dims per sub-quantizer: the table (192 KiB) is walked in two chunks
    (5000, 1024, 128, 10, 2),    # 8 groups: a 6-group chunk and a 2-group chunk
    (4000, 800, 100, 10, 2),     # 6 full groups + a 4-wide tail: past one LDS image with the top-k buffers
    (3000, 1600, 200, 64, 2),    # 12 groups + 8 tail, k = 64
    (130, 1536, 192, 10, 1),     # fewer tiles than a batch
])
def test_adc_scan_matches_oracle(vg, ctx, n, dim, m, k, nq):
    rng = np.random.default_rng(n + dim + m + k)
    opq = _random_pq(rng, dim, m)
    codes = rng.integers(0, 256, (n, m)).astype(np.uint8)
    pq, idx = _mk(vg, ctx, opq, codes, n)
    queries = rng.standard_normal((nq, dim)).astype(np.float32)
    ids, scores = idx.search_pq_adc(queries, k)
    assert ids.shape == (nq, k)
    for qi in range(nq):
        eid, esc = o.flat_search_pq(opq, codes, queries[qi], k)
        r = eid.size
        assert np.array_equal(ids[qi, :r], eid)
        assert np.array_equal(bits(scores[qi, :r]), bits(esc))
        assert np.all(ids[qi, r:] == 0xFFFFFFFF) and np.all(np.isinf(scores[qi, r:]))


def test_wide_table_limits(vg, ctx):
    rng = np.random.default_rng(5)
    opq = _random_pq(rng, 1536, 192)
    codes = rng.integers(0, 256, (500, 192)).astype(np.uint8)
    pq, idx = _mk(vg, ctx, opq, codes, 500)
    with pytest.raises(vg.VecgoHipError) as e:          # the chunked scan keeps its top-k in registers
        idx.search_pq_adc(rng.standard_normal((1, 1536)).astype(np.float32), 100)
    assert e.value.status == -5


def test_adc_ties_break_by_row_id(vg, ctx):
    """Duplicate codes → equal scores: the reference keeps the lower RowID
    (searcher/candidate_queue.go:12-23)."""
    rng = np.random.default_rng(11)
    dim, m, n, k = 128, 16, 4096, 16
    opq = _random_pq(rng, dim, m)
    codes = np.tile(rng.integers(0, 256, (8, m)).astype(np.uint8), (n // 8, 1))
    pq, idx = _mk(vg, ctx, opq, codes, n)
    q = rng.standard_normal((2, dim)).astype(np.float32)
    ids, scores = idx.search_pq_adc(q, k)
    for qi in range(2):
        eid, esc = o.flat_search_pq(opq, codes, q[qi], k)
        assert np.array_equal(ids[qi], eid)
        assert np.array_equal(bits(scores[qi]), bits(esc))


def test_adc_many_queries_share_slices(vg, ctx):
    rng = np.random.default_rng(12)
    dim, m, n, k, nq = 128, 16, 30000, 10, 300
    opq = _random_pq(rng, dim, m)
    codes = rng.integers(0, 256, (n, m)).astype(np.uint8)
    pq, idx = _mk(vg, ctx, opq, codes, n)
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    ids, scores = idx.search_pq_adc(q, k)
    for qi in range(0, nq, 17):
        eid, esc = o.flat_search_pq(opq, codes, q[qi], k)
        assert np.array_equal(ids[qi], eid)
        assert np.array_equal(bits(scores[qi]), bits(esc))


def test_adc_device_buffers(vg, ctx):
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(13)
    dim, m, n, k, nq = 768, 96, 20000, 10, 5
    opq = _random_pq(rng, dim, m)
    codes = rng.integers(0, 256, (n, m)).astype(np.uint8)
    pq = vg.ProductQuantizer(ctx, dim, m, 256)
    pq.set_codebooks(opq.codebooks, opq.scales, opq.offsets)
    idx = vg.Index(ctx, n, dim, vg.Metric.L2)
    idx.set_pq_codes(pq, torch.from_numpy(codes).cuda())
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    tq = torch.from_numpy(q).cuda()
    st = torch.cuda.current_stream()
    ids, scores = idx.search_pq_adc(tq, k, stream=st)
    torch.cuda.synchronize()
    ids = ids.cpu().numpy().view(np.uint32); scores = scores.cpu().numpy()
    for qi in range(nq):
        eid, esc = o.flat_search_pq(opq, codes, q[qi], k)
        assert np.array_equal(ids[qi], eid)
        assert np.array_equal(bits(scores[qi]), bits(esc))


def test_adc_big_k_proof_and_fallback(vg, ctx):
    """k > 64: per-wave 64-lists + proof.  Rows sorted by score put all the best rows into a few
    waves (their lists overflow), so the proof must fail and the exhaustive scan take over."""
    import os
    rng = np.random.default_rng(31)
    dim, m, n, k = 128, 16, 60000, 500
    opq = _random_pq(rng, dim, m)
    codes = rng.integers(0, 256, (n, m)).astype(np.uint8)
    q = rng.standard_normal((2, dim)).astype(np.float32)
    # sort rows by ADC distance to query 0: its top-500 are rows 0..499, i.e. ~8 tiles
    t = opq.build_table(q[0])
    d = np.array([o.adc(t, codes[i], m) for i in range(n)])
    codes = np.ascontiguousarray(codes[np.argsort(d, kind="stable")])
    pq, idx = _mk(vg, ctx, opq, codes, n)
    for env in (None, "1"):
        if env:
            hooks.set_hook("VG_ADC_BIGK_EXHAUSTIVE", env)
        try:
            ids, scores = idx.search_pq_adc(q, k)
        finally:
            hooks.set_hook("VG_ADC_BIGK_EXHAUSTIVE", 0)
        for qi in range(2):
            eid, esc = o.flat_search_pq(opq, codes, q[qi], k)
            assert np.array_equal(ids[qi], eid)
            assert np.array_equal(bits(scores[qi]), bits(esc))


def test_adc_errors(vg, ctx):
    opq = _random_pq(np.random.default_rng(1), 128, 16)
    pq = vg.ProductQuantizer(ctx, 128, 16, 256)
    idx = vg.Index(ctx, 10, 128, vg.Metric.L2)
    with pytest.raises(vg.VecgoHipError) as e:  # pq.go:148-150
        idx.set_pq_codes(pq, np.zeros((10, 16), np.uint8))
    assert e.value.status == -3 and "not trained" in e.value.message
    with pytest.raises(vg.VecgoHipError) as e:
        idx.search_pq_adc(np.zeros((1, 128), np.float32), 5)
    assert e.value.status == -9
    with pytest.raises(vg.VecgoHipError):  # pq.go:40-42
        vg.ProductQuantizer(ctx, 100, 7, 256)
    with pytest.raises(vg.VecgoHipError):  # pq.go:47-49
        vg.ProductQuantizer(ctx, 128, 8, 257)
    # zero queries / zero k succeed with empty outputs
    pq.set_codebooks(opq.codebooks, opq.scales, opq.offsets)
    idx.set_pq_codes(pq, np.zeros((10, 16), np.uint8))
    ids, sc = idx.search_pq_adc(np.zeros((0, 128), np.float32), 5)
    assert ids.shape == (0, 5)


# ---- batches through the bfloat16 nomination (vg_index_enable_pq_nomination) -------------------------------------------------
def _decoded(opq, codes):
    """ProductQuantizer.Decode of every row (pq.go:185-229): float32(int8) * scale, + offset, two rounded operations"""
    sd = opq.dim // opq.m
    cb = np.asarray(opq.codebooks, np.int8).reshape(opq.m, opq.k, sd).astype(np.float32)
    v = cb * np.asarray(opq.scales, np.float32)[:, None, None]
    v = (v + np.asarray(opq.offsets, np.float32)[:, None, None]).astype(np.float32)
    return v[np.arange(opq.m)[None, :], codes].reshape(codes.shape[0], opq.dim)


@pytest.fixture
def nominate_always():
    """the library nominates from 24M (query, row) pairs up; the tests' segments are smaller"""
    hooks.set_hook("VG_PQ_NOM_ALWAYS", 1)
    yield
    hooks.set_hook("VG_PQ_NOM_ALWAYS", 0)


def _clustered_codes(rng, n, m, clusters=40, flip=0.3):
    """codes around a few prototypes: neighbours exist (a uniform draw has none in 96 sub-spaces)"""
    proto = rng.integers(0, 256, (clusters, m)).astype(np.uint8)
    codes = proto[rng.integers(0, clusters, n)]
    noise = rng.integers(0, 256, (n, m)).astype(np.uint8)
    return np.where(rng.random((n, m)) < flip, noise, codes).astype(np.uint8)


@pytest.mark.parametrize("n,dim,m,nq,k", [(20000, 128, 16, 64, 10), (9000, 768, 96, 50, 10), (30000, 64, 8, 130, 48),
                                          (3000, 64, 4, 60, 5), (12000, 200, 25, 64, 10), (6000, 68, 17, 50, 100),
                                          (7000, 64, 1, 48, 10), (16000, 128, 16, 130, 256), (5000, 128, 16, 300, 49),
                                          (3000, 64, 8, 140, 10), (900, 128, 16, 200, 48)])
def test_batches_through_the_bf16_nomination(vg, ctx, nominate_always, n, dim, m, nq, k):
    """vg_index_enable_pq_nomination: a batch (queries x rows >= 24M; here: the test hook) is nominated by the bfloat16 GEMM over the DECODED rows, its 64 best
    (k > 48: everything below the threshold) re-scored from the CODES against the query's table in pqAdcLookupAvx512 order, the
    rest excluded by a proof — same ids and score bits as the table scan and as the oracle; duplicate codes (ties), image rows
    padded to 64 elements, fewer rows than the append budget, m below / not a multiple of 16 included."""
    rng = np.random.default_rng(n + dim + m)
    opq = _random_pq(rng, dim, m)
    codes = _clustered_codes(rng, n, m)
    codes[100:110] = codes[100]
    pq, idx = _mk(vg, ctx, opq, codes, n)
    x = _decoded(opq, codes)
    q = (x[rng.integers(0, n, nq)] + rng.standard_normal((nq, dim)).astype(np.float32) * 0.05).astype(np.float32)
    q[1] = x[100]
    plain = idx.search_pq_adc(q, k)
    idx.enable_pq_nomination(True)
    got = idx.search_pq_adc(q, k)
    assert np.array_equal(plain[0], got[0]) and np.array_equal(bits(plain[1]), bits(got[1]))
    for i in (0, 1, nq // 2, nq - 1):
        eid, esc = o.flat_search_pq(opq, codes, q[i], k)
        assert np.array_equal(got[0][i, :eid.size], eid) and np.array_equal(bits(got[1][i, :eid.size]), bits(esc))
    few = idx.search_pq_adc(q[:7], k)                  # a few queries (one GEMM tile, mostly padding)
    assert np.array_equal(few[0], plain[0][:7]) and np.array_equal(bits(few[1]), bits(plain[1][:7]))
    hooks.set_hook("VG_PQ_NOM_ALWAYS", 0)              # without the hook a batch this small keeps the scan
    assert np.array_equal(idx.search_pq_adc(q, k)[0], plain[0])
    idx.enable_pq_nomination(False)
    again = idx.search_pq_adc(q, k)
    assert np.array_equal(again[0], plain[0]) and np.array_equal(bits(again[1]), bits(plain[1]))


def test_nomination_with_far_queries_and_non_finite_values(vg, ctx, nominate_always):
    """queries far from every row (the proof's margin scales with |q|^2: it fails, the scan answers), NaN / Inf queries, a NaN scale
    in the quantizer (the norm bound becomes +Inf: every proof fails) — nomination on = off, bit for bit"""
    rng = np.random.default_rng(321)
    n, dim, m, nq, k = 12000, 128, 16, 70, 10
    opq = _random_pq(rng, dim, m)
    codes = _clustered_codes(rng, n, m)
    x = _decoded(opq, codes)
    q = (x[rng.integers(0, n, nq)] + rng.standard_normal((nq, dim)).astype(np.float32) * 0.05).astype(np.float32)
    q[2] = rng.standard_normal(dim).astype(np.float32) * 1000.0
    q[3, 5] = np.nan
    q[4, :] = np.inf
    q[5, 7] = -np.inf
    q[6] = 0.0
    for poison in (False, True):
        if poison:
            sc = np.array(opq.scales, np.float32)
            sc[3] = np.nan
            opq.set_codebooks(opq.codebooks, sc, opq.offsets)
        pq, idx = _mk(vg, ctx, opq, codes, n)
        plain = idx.search_pq_adc(q, k)
        idx.enable_pq_nomination(True)
        got = idx.search_pq_adc(q, k)
        assert np.array_equal(plain[0], got[0]) and np.array_equal(bits(plain[1]), bits(got[1])), poison
        if not poison:
            for i in (0, 2, 6, nq - 1):
                eid, esc = o.flat_search_pq(opq, codes, q[i], k)
                assert np.array_equal(got[0][i], eid) and np.array_equal(bits(got[1][i]), bits(esc))


def test_nomination_limits_and_lifetime(vg, ctx, nominate_always):
    """k beyond 256 and k >= rows keep the scan; new codes drop the image; K != 256 is refused as the scan refuses it; device
    buffers; an index without codes"""
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(77)
    n, dim, m, nq = 9000, 128, 16, 64
    opq = _random_pq(rng, dim, m)
    codes = _clustered_codes(rng, n, m)
    pq, idx = _mk(vg, ctx, opq, codes, n)
    x = _decoded(opq, codes)
    q = (x[rng.integers(0, n, nq)] + rng.standard_normal((nq, dim)).astype(np.float32) * 0.05).astype(np.float32)
    ref = {k: idx.search_pq_adc(q, k) for k in (10, 300)}
    idx.enable_pq_nomination(True)
    for k in (10, 300):
        got = idx.search_pq_adc(q, k)
        assert np.array_equal(ref[k][0], got[0]) and np.array_equal(bits(ref[k][1]), bits(got[1]))
    dq = torch.from_numpy(q).cuda()
    ids, sc = idx.search_pq_adc(dq, 10, stream=torch.cuda.current_stream())
    torch.cuda.synchronize()
    assert np.array_equal(ids.cpu().numpy().view(np.uint32), ref[10][0]) and np.array_equal(bits(sc.cpu().numpy()), bits(ref[10][1]))
    codes2 = _clustered_codes(rng, n, m)
    idx.set_pq_codes(pq, codes2)                       # the image belonged to the old codes
    got = idx.search_pq_adc(q, 10)
    for i in (0, nq - 1):
        eid, esc = o.flat_search_pq(opq, codes2, q[i], 10)
        assert np.array_equal(got[0][i], eid) and np.array_equal(bits(got[1][i]), bits(esc))
    idx.enable_pq_nomination(True)
    again = idx.search_pq_adc(q, 10)
    assert np.array_equal(got[0], again[0]) and np.array_equal(bits(got[1]), bits(again[1]))
    small = vg.Index(ctx, 40, dim, vg.Metric.L2)       # fewer rows than k: the scan's padding
    small.set_pq_codes(pq, codes[:40])
    small.enable_pq_nomination(True)
    r = small.search_pq_adc(q, 48)
    assert np.all(r[0][:, 40:] == 0xFFFFFFFF) and np.array_equal(r[0], (small.enable_pq_nomination(False), small.search_pq_adc(q, 48))[1][0])
    with pytest.raises(vg.VecgoHipError):
        vg.Index(ctx, 10, dim, vg.Metric.L2).enable_pq_nomination(True)
    o64 = _random_pq(rng, dim, m, k=64)
    p64 = vg.ProductQuantizer(ctx, dim, m, 64)
    p64.set_codebooks(o64.codebooks, o64.scales, o64.offsets)
    i64 = vg.Index(ctx, 100, dim, vg.Metric.L2)
    i64.set_pq_codes(p64, rng.integers(0, 64, (100, m)).astype(np.uint8))
    with pytest.raises(vg.VecgoHipError) as e:
        i64.enable_pq_nomination(True)
    assert e.value.status == -5
