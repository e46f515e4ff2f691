"""SQ8 on the GPU (quantizer.go, sq8_avx512.c, flat/segment.go:517-604) vs the oracle: codes,
decoded values, distances and search results bit-exact."""
import numpy as np
import pytest

from oracle import oracle as o

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


@pytest.mark.parametrize("n,dim", [(500, 128), (300, 768), (257, 100), (64, 17), (5, 3), (1000, 16)])
def test_train_encode_decode_match_oracle(vg, ctx, n, dim):
    rng = np.random.default_rng(n + dim)
    x = rng.standard_normal((n, dim)).astype(np.float32)
    x[:, dim // 2] = 1.25  # a constant dimension: max = min + 1e-6 (quantizer.go:168-170)
    sq = vg.ScalarQuantizer(ctx, dim)
    assert not sq.is_trained()
    with pytest.raises(vg.VecgoHipError):
        sq.encode(x[:1])
    sq.train(x)
    ref = o.ScalarQuantizer(dim); ref.train(x)
    mins, maxs, scales, inv = sq.params()
    for a, b in ((mins, ref.mins), (maxs, ref.maxs), (scales, ref.scales), (inv, ref.inv_scales)):
        assert np.array_equal(bits(a), bits(b))
    y = np.vstack([x[:40], x[:3] * 10.0, x[:3] * -10.0]).astype(np.float32)  # incl. values to clamp
    codes = sq.encode(y)
    assert np.array_equal(codes, ref.encode_batch(y))
    dec = sq.decode(codes)
    assert np.array_equal(bits(dec), bits(np.stack([ref.decode(c) for c in codes])))


def test_set_bounds_and_reference_kats(vg, ctx):
    sq = vg.ScalarQuantizer(ctx, 5)
    sq.set_bounds(np.full(5, -1.0, np.float32), np.full(5, 1.0, np.float32))
    code = sq.encode(np.array([[-1.0, -0.5, 0.0, 0.5, 1.0]], np.float32))[0]
    assert code[0] == 0 and code[4] == 255  # quantizer_test.go:39-86
    dec = sq.decode(code)[0]
    assert np.max(np.abs(dec - np.array([-1.0, -0.5, 0.0, 0.5, 1.0], np.float32))) <= (2.0 / 255.0) * 1.1
    sq3 = vg.ScalarQuantizer(ctx, 3)
    sq3.set_bounds(np.array([0, 0, 2], np.float32), np.array([1, 1, 2], np.float32))  # zero range: scales 0
    _, _, scales, inv = sq3.params()
    assert scales[2] == 0.0 and inv[2] == 0.0 and scales[0] == np.float32(255.0)
    d = sq3.decode(sq3.encode(np.array([[-1.0, 0.5, 2.0]], np.float32)))[0]
    assert d[0] >= -0.01 and d[1] <= 1.01  # clamping, quantizer_test.go:148-179


# dim % 128 == 0: rows turned through LDS (sq8_l2_batch_turn_kernel); n = 300: ragged last tile, 1000: several workgroups
@pytest.mark.parametrize("n", [300, 1000])
@pytest.mark.parametrize("dim", [1, 7, 8, 15, 16, 17, 31, 32, 33, 100, 128, 256, 768, 1024])
def test_l2_distance_batch_matches_oracle(vg, ctx, dim, n):
    rng = np.random.default_rng(dim + n)
    x = rng.standard_normal((n, dim)).astype(np.float32)
    sq = vg.ScalarQuantizer(ctx, dim); sq.train(x)
    ref = o.ScalarQuantizer(dim); ref.train(x)
    codes = sq.encode(x)
    q = rng.standard_normal(dim).astype(np.float32)
    got = sq.l2_distance_batch(q, codes)
    want = o.sq8u_l2_batch(q, codes, ref.mins, ref.inv_scales, dim)
    assert np.array_equal(bits(got), bits(want))


@pytest.mark.parametrize("n,dim,nq,k", [(5000, 128, 6, 10), (3000, 768, 4, 10), (2000, 100, 3, 5),
                                         (130, 24, 2, 64), (40, 16, 2, 10), (7, 5, 1, 10)])
def test_search_sq8_matches_oracle(vg, ctx, n, dim, nq, k):
    rng = np.random.default_rng(n + dim)
    x = rng.standard_normal((n, dim)).astype(np.float32)
    if n > 50:
        x[37] = x[12]  # identical codes: tie broken by RowID
    sq = vg.ScalarQuantizer(ctx, dim); sq.train(x)
    ref = o.ScalarQuantizer(dim); ref.train(x)
    codes = sq.encode(x)
    idx = vg.Index(ctx, n, dim)
    idx.set_sq8_codes(sq, codes)
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    q[0] = x[12 % n] + 0.001
    ids, sc = idx.search_sq8(q, k)
    for i in range(nq):
        eid, esc = o.flat_search_sq8(ref, codes, q[i], k)
        r = eid.size
        assert np.array_equal(ids[i, :r], eid), (i, ids[i], eid)
        assert np.array_equal(bits(sc[i, :r]), bits(esc))
        assert np.all(ids[i, r:] == 0xFFFFFFFF)


def test_sq8_errors(vg, ctx):
    sq = vg.ScalarQuantizer(ctx, 8)
    idx = vg.Index(ctx, 10, 8)
    with pytest.raises(vg.VecgoHipError):
        idx.set_sq8_codes(sq, np.zeros((10, 8), np.uint8))  # not trained
    sq.train(np.random.default_rng(0).standard_normal((20, 8)).astype(np.float32))
    with pytest.raises(vg.VecgoHipError):
        idx.search_sq8(np.zeros((1, 8), np.float32), 5)      # no codes attached
    with pytest.raises(vg.VecgoHipError):
        idx.search_sq8(np.zeros((1, 8), np.float32), 513)    # k <= 512


@pytest.mark.parametrize("n,dim,nq,k,metric", [(700, 128, 5, 10, 2), (300, 100, 3, 7, 1), (130, 17, 2, 64, 2)])
def test_sq8_dot_product_scan_matches_oracle(vg, ctx, n, dim, nq, k, metric):
    """Dot / Cosine SQ8 segments are scored with ScalarQuantizer.DotProduct (flat/segment.go:659-667,
    quantizer.go:109-119: one sequential fp32 sum per row), largest first."""
    rng = np.random.default_rng(n + dim + metric)
    x = rng.standard_normal((n, dim)).astype(np.float32)
    x[41] = x[9]
    sq = vg.ScalarQuantizer(ctx, dim); sq.train(x)
    ref = o.ScalarQuantizer(dim); ref.train(x)
    codes = sq.encode(x)
    idx = vg.Index(ctx, n, dim, vg.Metric(metric))
    idx.set_sq8_codes(sq, codes)
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    ids, sc = idx.search_sq8(q, k)
    seg = o.FlatSegment(x, dim, metric=metric, sq=ref, codes=codes)
    for i in range(nq):
        eid, esc = seg.search(q[i], k)
        assert np.array_equal(ids[i, :eid.size], eid), (i, ids[i], eid)
        assert np.array_equal(bits(sc[i, :eid.size]), bits(esc))


@pytest.mark.parametrize("n,dim,nq,k,metric", [(3000, 64, 5, 100, 0), (1500, 100, 1, 200, 0), (900, 48, 3, 65, 2),
                                               (90, 16, 2, 128, 0)])
def test_sq8_scan_pages_beyond_64_results(vg, ctx, n, dim, nq, k, metric):
    """k > 64: the scan runs once per page of 64 results, each page taking only keys after the previous
    page's last one; duplicates straddling a page boundary are the interesting case."""
    rng = np.random.default_rng(n + k)
    x = rng.standard_normal((n, dim)).astype(np.float32)
    x[n // 2:n // 2 + 70] = x[3]          # 71 identical codes: a run of equal scores longer than a page
    sq = vg.ScalarQuantizer(ctx, dim); sq.train(x)
    ref = o.ScalarQuantizer(dim); ref.train(x)
    codes = sq.encode(x)
    idx = vg.Index(ctx, n, dim, vg.Metric(metric))
    idx.set_sq8_codes(sq, codes)
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    q[0] = x[3]
    ids, sc = idx.search_sq8(q, k)
    seg = o.FlatSegment(x, dim, metric=metric, sq=ref, codes=codes)
    for i in range(nq):
        eid, esc = seg.search(q[i], k)
        r = eid.size
        assert np.array_equal(ids[i, :r], eid), (i, np.flatnonzero(ids[i, :r] != eid)[:5])
        assert np.array_equal(bits(sc[i, :r]), bits(esc))
        assert np.all(ids[i, r:] == 0xFFFFFFFF)


@pytest.mark.parametrize("n,dim,nq,k", [(20000, 128, 40, 10), (9000, 768, 24, 10), (30000, 64, 130, 48), (3000, 64, 20, 5),
                                        (12000, 100, 33, 10), (6000, 17, 20, 100), (7000, 300, 9, 10), (3000, 64, 140, 10), (900, 128, 200, 48)])
def test_batches_through_the_bf16_nomination(vg, ctx, n, dim, nq, k):
    """vg_index_enable_sq8_nomination: 5 queries up, an L2 batch is nominated by the bfloat16 GEMM over the dequantised rows, its
    64 best re-scored from the CODES (the reference's L2Distance), the rest excluded by a proof — same ids and score bits as the
    scan and as the oracle; duplicate codes (ties) and clustered rows included; small batches and Dot keep the scan."""
    rng = np.random.default_rng(n + dim)
    cent = rng.standard_normal((30, dim)).astype(np.float32) * 2
    x = (cent[rng.integers(0, 30, n)] + rng.standard_normal((n, dim)).astype(np.float32) * 0.5).astype(np.float32)
    x[100:110] = x[100]
    sq = vg.ScalarQuantizer(ctx, dim); sq.train(x)
    ref = o.ScalarQuantizer(dim); ref.train(x)
    codes = sq.encode(x)
    idx = vg.Index(ctx, n, dim)
    idx.set_sq8_codes(sq, codes)
    q = (cent[rng.integers(0, 30, nq)] + rng.standard_normal((nq, dim)).astype(np.float32) * 0.5).astype(np.float32)
    q[1] = x[100]
    plain = idx.search_sq8(q, k)
    idx.enable_sq8_nomination(True)
    got = idx.search_sq8(q, k)
    assert np.array_equal(plain[0], got[0]) and np.array_equal(bits(plain[1]), bits(got[1]))
    for i in (0, 1, nq // 2, nq - 1):
        eid, esc = o.flat_search_sq8(ref, codes, q[i], k)
        assert np.array_equal(got[0][i, :eid.size], eid) and np.array_equal(bits(got[1][i, :eid.size]), bits(esc))
    few = idx.search_sq8(q[:3], k)                     # below 5 queries: the scan
    assert np.array_equal(few[0], plain[0][:3])
    idx.enable_sq8_nomination(False)
    again = idx.search_sq8(q, k)
    assert np.array_equal(again[0], plain[0])


@pytest.mark.parametrize("n,nq", [(16000, 36), (3500, 140)])   # (the second: no threshold sample and two query tiles of 128+)
@pytest.mark.parametrize("metric", [0, 2])
def test_nomination_with_dot_metric_and_filters(vg, ctx, metric, n, nq):
    """the same for a Dot segment (DotProduct, the largest first) and for filtered batches over the whole segment (one filter for
    the batch, one per query, filters that leave fewer than k rows): nomination on = nomination off = the oracle"""
    rng = np.random.default_rng(90 + metric)
    dim, k = 128, 10
    cent = rng.standard_normal((20, dim)).astype(np.float32) * 2
    x = (cent[rng.integers(0, 20, n)] + rng.standard_normal((n, dim)).astype(np.float32) * 0.6).astype(np.float32)
    x[50:58] = x[50]
    sq = vg.ScalarQuantizer(ctx, dim); sq.train(x)
    ref = o.ScalarQuantizer(dim); ref.train(x)
    codes = sq.encode(x)
    idx = vg.Index(ctx, n, dim, vg.Metric(metric))
    idx.set_sq8_codes(sq, codes)
    seg = o.FlatSegment(x, dim, metric=metric, sq=ref, codes=codes)
    q = (cent[rng.integers(0, 20, nq)] + rng.standard_normal((nq, dim)).astype(np.float32) * 0.6).astype(np.float32)
    q[2] = x[50]
    masks = rng.random((nq, n)) < 0.3
    masks[4, :] = False
    masks[4, rng.integers(0, n, 4)] = True            # four rows pass
    masks[5, :] = False
    off = (idx.search_sq8(q, k), idx.search_flat_filtered(q, k, masks, 0, scan=idx.SCAN_SQ8),
           idx.search_flat_filtered(q, k, masks[0], 0, scan=idx.SCAN_SQ8))
    idx.enable_sq8_nomination(True)
    on = (idx.search_sq8(q, k), idx.search_flat_filtered(q, k, masks, 0, scan=idx.SCAN_SQ8),
          idx.search_flat_filtered(q, k, masks[0], 0, scan=idx.SCAN_SQ8))
    for a, b in zip(off, on):
        assert np.array_equal(a[0], b[0]) and np.array_equal(bits(a[1]), bits(b[1]))
    for i in (0, 2, 4, 5, nq - 1):
        for got, m in ((on[0], None), (on[1], masks[i]), (on[2], masks[0])):
            eid, esc = seg.search(q[i], k, mask=m)
            assert np.array_equal(got[0][i, :eid.size], eid) and np.array_equal(bits(got[1][i, :eid.size]), bits(esc))
            assert np.all(got[0][i, eid.size:] == 0xFFFFFFFF)


@pytest.mark.parametrize("dim", [64, 100])
@pytest.mark.parametrize("metric", [0, 2])
def test_probed_batches_through_the_grouped_nomination(vg, ctx, metric, dim):
    """a partitioned SQ8 segment, partitions probed by many queries each: the grouped bf16 nomination per (query, probe) pair +
    sq8_verify_kernel — nomination on = off = the oracle, with and without a filter"""
    from tests.test_gpu_probe import partitioned
    rng = np.random.default_rng(70 + metric)
    n, parts, nq, k = 12000, 6, 120, 10               # (dim 100: the bf16 image padded to 128)
    x, cent, off = partitioned(rng, n, dim, parts, metric)
    x[200:206] = x[200]
    sq = vg.ScalarQuantizer(ctx, dim); sq.train(x)
    ref = o.ScalarQuantizer(dim); ref.train(x)
    codes = sq.encode(x)
    idx = vg.Index(ctx, n, dim, vg.Metric(metric))
    idx.set_sq8_codes(sq, codes)
    idx.set_partitions(cent, off)
    seg = o.FlatSegment(x, dim, metric=metric, sq=ref, codes=codes, centroids=cent, part_offsets=off)
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    q[3] = x[200]
    masks = rng.random((nq, n)) < 0.4
    res = {}
    for on in (False, True):
        idx.enable_sq8_nomination(on)
        res[on] = [idx.search_flat_probed(q, k, np_, scan=idx.SCAN_SQ8) for np_ in (2, parts)] + \
                  [idx.search_flat_filtered(q, k, masks, 2, scan=idx.SCAN_SQ8), idx.search_flat_filtered(q, k, masks[0], 2, scan=idx.SCAN_SQ8)]
    for a, b in zip(res[False], res[True]):
        assert np.array_equal(a[0], b[0]) and np.array_equal(bits(a[1]), bits(b[1]))
    for i in (0, 3, 60, nq - 1):
        for got, np_, m in ((res[True][0], 2, None), (res[True][1], parts, None), (res[True][2], 2, masks[i]), (res[True][3], 2, masks[0])):
            eid, esc = seg.search(q[i], k, np_, mask=m)
            assert np.array_equal(got[0][i, :eid.size], eid) and np.array_equal(bits(got[1][i, :eid.size]), bits(esc))


@pytest.mark.parametrize("metric", [0, 2])
def test_nomination_with_k_beyond_the_64_candidate_budget(vg, ctx, metric):
    """48 < k <= 256 over the whole segment, up to 160 over probed partitions: every row below the (deeper) threshold is re-scored
    from the codes and sorted (sq8_verify_sort_kernel) — nomination on = off = the oracle, ties across the k-th place and
    filters included"""
    from tests.test_gpu_probe import partitioned
    rng = np.random.default_rng(170 + metric)
    n, dim, parts, nq = 16000, 64, 6, 130
    x, cent, off = partitioned(rng, n, dim, parts, metric)
    x[200:330] = x[200]
    sq = vg.ScalarQuantizer(ctx, dim); sq.train(x)
    ref = o.ScalarQuantizer(dim); ref.train(x)
    codes = sq.encode(x)
    flat = vg.Index(ctx, n, dim, vg.Metric(metric))    # one range; and the same rows partitioned
    flat.set_sq8_codes(sq, codes)
    idx = vg.Index(ctx, n, dim, vg.Metric(metric))
    idx.set_sq8_codes(sq, codes)
    idx.set_partitions(cent, off)
    whole = o.FlatSegment(x, dim, metric=metric, sq=ref, codes=codes)
    seg = o.FlatSegment(x, dim, metric=metric, sq=ref, codes=codes, centroids=cent, part_offsets=off)
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    q[3] = x[200]
    masks = rng.random((nq, n)) < 0.4

    def run(k):
        out = [flat.search_sq8(q, k), flat.search_flat_filtered(q, k, masks, 0, scan=flat.SCAN_SQ8)]
        if k <= 160:
            out += [idx.search_flat_probed(q, k, 2, scan=idx.SCAN_SQ8), idx.search_flat_filtered(q, k, masks[0], parts, scan=idx.SCAN_SQ8)]
        return out

    for k in (49, 100, 160, 256):
        for ix in (flat, idx):
            ix.enable_sq8_nomination(False)
        plain = run(k)
        for ix in (flat, idx):
            ix.enable_sq8_nomination(True)
        nom = run(k)
        for a, b in zip(plain, nom):
            assert np.array_equal(a[0], b[0]) and np.array_equal(bits(a[1]), bits(b[1])), k
        for i in (0, 3, nq - 1):
            cases = [(nom[0], whole, 0, None), (nom[1], whole, 0, masks[i])]
            if k <= 160:
                cases += [(nom[2], seg, 2, None), (nom[3], seg, parts, masks[0])]
            for got, s, np_, m in cases:
                eid, esc = s.search(q[i], k, np_, mask=m)
                assert np.array_equal(got[0][i, :eid.size], eid) and np.array_equal(bits(got[1][i, :eid.size]), bits(esc)), (k, i, np_)
