"""vg_search_hnsw_filtered = searchExecute with a filter and a selectivity hint above 0.3: searchLayerWithPostFilter
(internal/hnsw/hnsw.go:1159-1218) — the walk with an expanded ef, every result popped worst first, the passing rows pushed
back capped at ef, the usual extraction.  Ids, score bits and the walk's counters equal the oracle's
(vgo_hnsw_search_filtered) on random rows and on tie-heavy grids, for shared and per-query masks, LDS and split heaps."""
import numpy as np
import pytest

from oracle import oracle as o
from tests import graphs

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


def _stats_tuple(st):
    return (st.nodes_visited, st.distance_computations, st.distance_short_circuits, st.pops)


def run(vg, ctx, base, metric, m, q, k, ef, masks, selectivity, seed=1):
    n, dim = base.shape
    l0, upper, entry = graphs.build_hnsw(base, m=m, seed=seed)
    oidx = o.HnswIndex(base, dim, l0, upper, entry, metric=metric)
    idx = vg.Index(ctx, n, dim, vg.Metric(metric))
    idx.set_vectors(base)
    idx.set_hnsw_graph(l0, upper, entry, m=m)
    ids, sc, st = idx.search_hnsw_filtered(q, k, ef, masks, selectivity, stats=True)
    for qi in range(q.shape[0]):
        mq = masks if masks.ndim == 1 else masks[qi]
        eid, esc, est = oidx.search_filtered(q[qi], k, ef, mq, selectivity)
        r = eid.size
        assert np.array_equal(ids[qi, :r], eid), (qi, ids[qi], eid)
        assert np.array_equal(bits(sc[qi, :r]), bits(esc)), qi
        assert np.all(ids[qi, r:] == 0xFFFFFFFF)
        assert mq[eid].all()                                        # every returned row passes
        assert tuple(int(x) for x in st[qi]) == _stats_tuple(est), (qi, st[qi], _stats_tuple(est))


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("k,ef,sel", [(10, 64, 0.5), (5, 16, 0.9), (10, 128, 0.31), (70, 200, 0.7), (10, 400, 0.5),
                                      (10, 600, 0.6), (100, 1000, 0.8)])   # expanded ef: 80, 16, 172, 230, 500 (cap), 500, 500
def test_random_rows(vg, ctx, metric, k, ef, sel):
    rng = np.random.default_rng(100 * metric + ef)
    n, dim = 2500, 32
    base = rng.random((n, dim)).astype(np.float32)
    if metric:
        base /= np.linalg.norm(base, axis=1, keepdims=True)
    q = rng.random((10, dim)).astype(np.float32)
    if metric:
        q /= np.linalg.norm(q, axis=1, keepdims=True)
    shared = rng.random(n) < sel
    run(vg, ctx, base, metric, 8, q, k, ef, shared, sel)
    per_query = rng.random((10, n)) < sel
    run(vg, ctx, base, metric, 8, q, k, ef, per_query, sel)


@pytest.mark.parametrize("grid", [3, 30])
def test_tie_grids(vg, ctx, grid):
    """Equal distances: which rows survive the rebuild, and in which order they leave, is the heaps' doing."""
    rng = np.random.default_rng(grid)
    n = 1500
    pts = rng.integers(0, grid, (n, 8)).astype(np.float32)
    q = rng.integers(0, grid, (8, 8)).astype(np.float32)
    for k, ef, sel in ((10, 32, 0.5), (64, 64, 0.4), (20, 300, 0.9)):
        run(vg, ctx, pts, 0, 8, q, k, ef, rng.random(n) < sel, sel, seed=3)


def test_edges(vg, ctx):
    rng = np.random.default_rng(9)
    n, dim = 600, 16
    base = rng.random((n, dim)).astype(np.float32)
    q = rng.random((4, dim)).astype(np.float32)
    run(vg, ctx, base, 0, 8, q, 10, 50, np.zeros(n, bool), 0.5)            # nothing passes: no results
    run(vg, ctx, base, 0, 8, q, 10, 50, np.ones(n, bool), 1.0)             # everything passes: expanded ef = ef
    only = np.zeros(n, bool); only[::97] = True
    run(vg, ctx, base, 0, 8, q, 10, 50, only, 0.35)                        # fewer than k survivors
    l0, upper, entry = graphs.build_hnsw(base, m=8, seed=1)
    idx = vg.Index(ctx, n, dim)
    idx.set_vectors(base)
    idx.set_hnsw_graph(l0, upper, entry, m=8)
    # at or below 0.3 the same entry takes searchLayerPredicateAware (tests/test_gpu_hnsw_predicate.py)
    pid, psc = idx.search_hnsw_filtered(q, 10, 50, only, 0.2)
    eid, esc, _ = o.HnswIndex(base, dim, l0, upper, entry).search_predicate(q[0], 10, 50, only)
    assert np.array_equal(pid[0, :eid.size], eid) and np.array_equal(bits(psc[0, :eid.size]), bits(esc))
    with pytest.raises(ValueError):
        idx.search_hnsw_filtered(q, 10, 50, np.ones(n - 1, bool), 0.5)    # a short mask never reaches the library
    assert o.HnswIndex(base, dim, l0, upper, entry).search_filtered(q[0], 10, 50, np.ones(n, bool), 0.3) is None
