"""The oracle's hnsw.BruteSearch / searchBitmap restatement (oracle/vg_oracle.c vgo_hnsw_brute_search, after
internal/hnsw/hnsw.go:2021-2101, 2240-2263, 1732-1751) against a second reading of the same heap in plain Python
(tests/prioq_py.py) and against properties that hold whatever the tie order."""
import numpy as np
import pytest

from oracle import oracle as o
from tests import prioq_py


def _index(base, metric=0):
    n, dim = base.shape
    return o.HnswIndex(base, dim, np.full((n, 2), 0xFFFFFFFF, np.uint32), metric=metric)


def _dists(base, q, metric):
    if metric == 2:
        return np.array([-o.dot(v, q) for v in base], np.float32)
    d = np.array([o.l2(v, q) for v in base], np.float32)
    return d * np.float32(0.5) if metric == 1 else d


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("mode", [o.BRUTE_SCAN, o.BRUTE_BITMAP])
@pytest.mark.parametrize("grid,k", [(2, 10), (3, 1), (3, 17), (5, 64), (40, 10)])
def test_matches_python_heap(metric, mode, grid, k):
    rng = np.random.default_rng(grid * 100 + k + metric)
    n, dim = 700, 8
    base = rng.integers(0, grid, (n, dim)).astype(np.float32)
    h = _index(base, metric)
    for t in range(6):
        q = rng.integers(0, grid, dim).astype(np.float32)
        mask = None if t % 2 == 0 else rng.random(n) < (0.5 if t < 4 else 0.02)
        ids, sc = h.brute_search(q, k, mode, mask)
        eid, esc = prioq_py.brute_search(_dists(base, q, metric), k, mode, mask)
        assert np.array_equal(ids, eid), (metric, mode, grid, k, t)
        assert np.array_equal(sc.view(np.uint32), esc.view(np.uint32))


def test_the_two_disciplines_differ_on_ties():
    """PopItem + PushItem and replace-top + siftDown leave different layouts: on a tie-heavy corpus the two loops do
    not return the same ids in the same order — the reason vg_search_hnsw_brute replays each as written."""
    rng = np.random.default_rng(0)
    base = rng.integers(-2, 3, (500, 16)).astype(np.float32)
    h = _index(base)
    differ = 0
    for _ in range(20):
        q = rng.integers(-2, 3, 16).astype(np.float32)
        a, sa = h.brute_search(q, 10, o.BRUTE_SCAN)
        b, sb = h.brute_search(q, 10, o.BRUTE_BITMAP)
        assert np.array_equal(sa, sb)            # the distance multiset is the k smallest either way
        differ += not np.array_equal(a, b)
    assert differ > 0


@pytest.mark.parametrize("mode", [o.BRUTE_SCAN, o.BRUTE_BITMAP])
def test_edges(mode):
    rng = np.random.default_rng(3)
    base = rng.standard_normal((37, 5)).astype(np.float32)
    h = _index(base)
    q = rng.standard_normal(5).astype(np.float32)
    ids, sc = h.brute_search(q, 100, mode)                       # k > n: every row, ascending distance
    assert ids.size == 37 and np.all(np.diff(sc) >= 0) and sorted(ids.tolist()) == list(range(37))
    ids, sc = h.brute_search(q, 5, mode, np.zeros(37, bool))      # empty bitmap / everything filtered
    assert ids.size == 0
    one = np.zeros(37, bool); one[11] = True
    ids, sc = h.brute_search(q, 5, mode, one)
    assert ids.tolist() == [11] and sc[0] == o.l2(base[11], q)
    ids, _ = h.brute_search(q, 0, mode)
    assert ids.size == 0
    # distinct distances: both loops = plain sort
    ids, sc = h.brute_search(q, 7, mode)
    d = np.array([o.l2(v, q) for v in base], np.float32)
    assert np.array_equal(ids, np.argsort(d, kind="stable")[:7].astype(np.uint32))


def test_hnsw_dot_product_distance_ordering_kat(golden_dir):
    """hnsw_test.go:104-159: three rows under the Dot metric; BruteSearch and KNNSearch return [row 1, row 0, row 2] with
    distances -dot = [-2, -1, 1] — the HNSW distance convention vg_search_hnsw_brute reports (vg_search_flat reports dot
    products, largest first)."""
    import json
    c = json.loads((golden_dir / "reference_kats.json").read_text())["hnsw_dot_ordering"]
    rows = np.array(c["rows"], np.float32); q = np.array(c["query"], np.float32)
    l0, upper, entry = o.hnsw_build(rows, 3, metric=2, m=c["m"], ef=c["ef_construction"], max_batch=1, growth_div=1)
    h = o.HnswIndex(rows, 3, l0, upper, entry, metric=2, m=c["m"])
    for mode in (o.BRUTE_SCAN, o.BRUTE_BITMAP):
        ids, sc = h.brute_search(q, c["k"], mode)
        assert ids.tolist() == c["expect_ids"] and np.all(np.abs(sc - np.array(c["expect_distances"], np.float32)) <= c["tol"])
    ids, sc, _ = h.search(q, c["k"], c["ef_search"])
    assert ids.tolist() == c["expect_ids"] and np.all(np.abs(sc - np.array(c["expect_distances"], np.float32)) <= c["tol"])
