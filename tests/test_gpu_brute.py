"""a17's tail: vg_search_hnsw_brute = hnsw.BruteSearch + scanSegment (internal/hnsw/hnsw.go:2021-2101) and searchBitmap +
extraction (:2240-2263, :1732-1751) with the reference's PriorityQueue discipline for each: ids IN ORDER and score bits
equal the oracle's on tie-heavy integer grids (where the two loops return different ids, tests/test_oracle_brute.py),
for L2 / Cosine / Dot (HNSW distance convention: L2, 0.5*L2, -dot), with filters / bitmaps, per-query masks, ragged sizes
and k > n."""
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


def _oidx(base, metric):
    n, dim = base.shape
    return o.HnswIndex(base, dim, np.full((n, 2), 0xFFFFFFFF, np.uint32), metric=metric)


def _check(idx, oidx, q, k, mode, mask, per_query=False):
    ids, sc = idx.search_hnsw_brute(q, k, mode, mask)
    for qi in range(q.shape[0]):
        m = None if mask is None else (mask[qi] if per_query else mask)
        eid, esc = oidx.brute_search(q[qi], k, mode, m)
        r = eid.size
        assert np.array_equal(ids[qi, :r], eid), (qi, k, mode, ids[qi, :r], eid)
        assert np.array_equal(bits(sc[qi, :r]), bits(esc)), (qi, k, mode)
        assert np.all(ids[qi, r:] == 0xFFFFFFFF) and np.all(np.isposinf(sc[qi, r:]))


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("grid,k", [(3, 10), (3, 1), (2, 64), (5, 100), (40, 10), (3, 600)])
def test_tie_grids(vg, ctx, metric, mode, grid, k):
    rng = np.random.default_rng(17 + grid + k)
    n, dim = 5003, 8                                     # > one 4096-row step, n % 4 != 0
    base = rng.integers(0, grid, (n, dim)).astype(np.float32)
    if metric == 1:
        base[base.sum(1) == 0, 0] = 1
        base /= np.linalg.norm(base, axis=1, keepdims=True)
    q = rng.integers(0, grid, (12, dim)).astype(np.float32)
    if metric == 1:
        q[q.sum(1) == 0, 0] = 1
        q /= np.linalg.norm(q, axis=1, keepdims=True)
    idx = vg.Index(ctx, n, dim, vg.Metric(metric))
    idx.set_vectors(base)
    oidx = _oidx(base, metric)
    _check(idx, oidx, q, k, mode, None)
    _check(idx, oidx, q, k, mode, rng.random(n) < 0.5)            # filter / bitmap shared by the batch
    _check(idx, oidx, q, k, mode, rng.random(n) < 0.01)           # a selective bitmap (searchBitmap's regime, hnsw.go:1706-1716)
    _check(idx, oidx, q, k, mode, rng.random((12, n)) < 0.3, per_query=True)
    idx.close()


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("n,dim,k", [(1, 4, 3), (3, 7, 10), (4096, 16, 10), (4097, 33, 5), (20000, 128, 50), (64, 768, 64)])
def test_random_normal_and_ragged(vg, ctx, mode, n, dim, k):
    rng = np.random.default_rng(n + dim)
    base = rng.standard_normal((n, dim)).astype(np.float32)
    q = rng.standard_normal((5, dim)).astype(np.float32)
    for metric in (0, 2):
        idx = vg.Index(ctx, n, dim, vg.Metric(metric))
        idx.set_vectors(base)
        _check(idx, _oidx(base, metric), q, k, mode, None)
        idx.close()


@pytest.mark.parametrize("mode", [0, 1])
def test_adversarial_orders(vg, ctx, mode):
    """rows sorted by DEcreasing distance: every row beats the top (the replay's slow path, still the reference's
    answer); increasing: nothing after the first k does; all equal: nothing after the first k does, ties everywhere"""
    n, dim, k = 9000, 4, 16
    q = np.zeros((2, dim), np.float32)
    for order in ("dec", "inc", "equal", "two_values"):
        r = {"dec": np.arange(n, 0, -1), "inc": np.arange(1, n + 1), "equal": np.full(n, 3),
             "two_values": 1 + (np.arange(n) % 2)}[order].astype(np.float32)
        base = np.zeros((n, dim), np.float32)
        base[:, 0] = r
        idx = vg.Index(ctx, n, dim)
        idx.set_vectors(base)
        _check(idx, _oidx(base, 0), q, k, mode, None)
        idx.close()


def test_edges(vg, ctx):
    rng = np.random.default_rng(2)
    base = rng.standard_normal((50, 8)).astype(np.float32)
    q = rng.standard_normal((3, 8)).astype(np.float32)
    idx = vg.Index(ctx, 50, 8)
    idx.set_vectors(base)
    ids, sc = idx.search_hnsw_brute(q, 5, 0, np.zeros(50, bool))            # everything filtered
    assert np.all(ids == 0xFFFFFFFF) and np.all(np.isposinf(sc))
    ids, sc = idx.search_hnsw_brute(q[:0], 5, 0)                              # no queries
    assert ids.shape == (0, 5)
    with pytest.raises(vg.VecgoHipError):
        idx.search_hnsw_brute(q, 2000, 0)                                    # k beyond the LDS heap
    with pytest.raises(vg.VecgoHipError):
        idx.search_hnsw_brute(q, 5, 7)                                       # unknown mode
    # the (Score, RowID) order of vg_search_flat and this entry agree wherever distances are distinct
    fi, fs = idx.search_flat(q, 5)
    bi, bs = idx.search_hnsw_brute(q, 5, 0)
    assert np.array_equal(fi, bi) and np.array_equal(bits(fs), bits(bs))
    idx.close()


def test_device_buffers_and_batches(vg, ctx):
    """device-resident queries / outputs (torch), a batch larger than one chunk would need at 1M rows is covered by
    tests/test_gpu_fullsize.py; here: 300 queries, results equal one-query calls"""
    import torch
    rng = np.random.default_rng(4)
    n, dim = 30000, 32
    base = rng.integers(0, 4, (n, dim)).astype(np.float32)
    q = rng.integers(0, 4, (300, dim)).astype(np.float32)
    idx = vg.Index(ctx, n, dim)
    idx.set_vectors(base)
    ids, sc = idx.search_hnsw_brute(torch.from_numpy(q).cuda(), 10, 0)
    torch.cuda.synchronize()
    ids, sc = ids.cpu().numpy().view(np.uint32), sc.cpu().numpy()
    oidx = _oidx(base, 0)
    for qi in (0, 7, 150, 299):
        eid, esc = oidx.brute_search(q[qi], 10, 0)
        assert np.array_equal(ids[qi], eid) and np.array_equal(bits(sc[qi]), bits(esc))
    i1, s1 = idx.search_hnsw_brute(q[:3], 10, 0)
    assert np.array_equal(i1, ids[:3])
    idx.close()


def test_hnsw_dot_product_distance_ordering_kat(vg, ctx, golden_dir):
    """hnsw_test.go:104-159 through the C ABI: the graph built by vg_hnsw_build (M = 8, EF = 50), KNNSearch at EFSearch 100
    and BruteSearch both return rows [1, 0, 2] with distances [-2, -1, 1] under the Dot metric"""
    import json
    c = json.loads((golden_dir / "reference_kats.json").read_text())["hnsw_dot_ordering"]
    rows = np.array(c["rows"], np.float32); q = np.array(c["query"], np.float32)[None, :]
    idx = vg.Index(ctx, 3, 3, vg.Metric(2))
    idx.set_vectors(rows)
    idx.build_hnsw(m=c["m"], ef_construction=c["ef_construction"], max_batch=1, growth_div=1)
    want = np.array(c["expect_distances"], np.float32)
    for mode in (idx.BRUTE_SCAN, idx.BRUTE_BITMAP):
        ids, sc = idx.search_hnsw_brute(q, c["k"], mode)
        assert ids[0].tolist() == c["expect_ids"] and np.all(np.abs(sc[0] - want) <= c["tol"])
    ids, sc = idx.search_hnsw(q, c["k"], c["ef_search"])
    assert ids[0].tolist() == c["expect_ids"] and np.all(np.abs(sc[0] - want) <= c["tol"])
    idx.close()


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("mode", [0, 1])
def test_masked_batches_through_the_masked_nomination(vg, ctx, metric, mode):
    """8 queries up, a masked batch takes the flat search's matrix-core nomination with the mask in its epilogue (k + 1 best of
    the rows that take part), turned into the heap's answer when no two of them tie; ties and masks that leave fewer than
    k + 1 rows are replayed — a few queries one by one, most of a batch in one pass.  Same answers as the replay alone."""
    from tests import hooks
    rng = np.random.default_rng(50 + metric + 2 * mode)
    n, dim, nq, k = 20000, 32, 24, 10
    base = rng.standard_normal((n, dim)).astype(np.float32)
    if metric == 1:                                # Cosine: rows and queries normalised by the caller, distance 0.5 * L2
        base /= np.linalg.norm(base, axis=1, keepdims=True)
    base[100:120] = base[100]                      # ties
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    if metric == 1:
        q /= np.linalg.norm(q, axis=1, keepdims=True)
    q[2] = base[100]
    idx = vg.Index(ctx, n, dim, vg.Metric(metric))
    idx.set_vectors(base)
    oidx = _oidx(base, metric)
    for keep, few in ((0.5, 3), (0.02, 3), (0.5, 20)):
        mask = rng.random((nq, n)) < keep
        for j in range(few):                       # these queries' masks leave fewer than k + 1 rows (some: none)
            mask[j * 1] = False
            mask[j, rng.integers(0, n, j)] = True
        _check(idx, oidx, q, k, mode, mask, per_query=True)
        _check(idx, oidx, q, k, mode, mask[5])     # one mask for the batch
        hooks.set_hook("VG_BRUTE_NO_FLAT", "1")
        try:
            ref = idx.search_hnsw_brute(q, k, mode, mask)
        finally:
            hooks.set_hook("VG_BRUTE_NO_FLAT", 0)
        got = idx.search_hnsw_brute(q, k, mode, mask)
        assert np.array_equal(ref[0], got[0]) and np.array_equal(bits(ref[1]), bits(got[1]))
    _check(idx, oidx, q, k, mode, None)            # unmasked, every metric through the flat search
    _check(idx, oidx, q[:1], k, mode, None)
    idx.close()
