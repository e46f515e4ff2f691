"""vg_search_hnsw_predicate = searchExecute with a filter whose selectivity hint is at or below 0.3 or unknown:
searchLayerPredicateAware (hnsw.go:1406-1558) on the GPU vs the oracle's statement-by-statement restatement — ids, score bits
and the counters (NodesVisited, DistanceComputations, ExpansionsSkipped, pops) equal; cached edge distances uploaded or
recomputed from the rows; tombstones apart from the filter."""
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


def check(idx, oidx, q, k, ef, masks, deleted=None, l0_dist=None):
    ids, sc, st = idx.search_hnsw_predicate(q, k, ef, masks, deleted=deleted, stats=True)
    for qi in range(q.shape[0]):
        mq = masks if masks.ndim == 1 else masks[qi]
        eid, esc, est = oidx.search_predicate(q[qi], k, ef, mq, deleted=deleted, l0_dist=l0_dist)
        r = eid.size
        assert np.array_equal(ids[qi, :r], eid), (qi, ids[qi], eid)
        assert np.array_equal(bits(sc[qi, :r]), bits(esc)), qi
        assert np.all(ids[qi, r:] == 0xFFFFFFFF)
        assert mq[eid].all() and (deleted is None or not deleted[eid].any())
        assert tuple(int(x) for x in st[qi]) == (est.nodes_visited, est.distance_computations, est.distance_short_circuits,
                                                 est.pops), (qi, st[qi])


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("n,dim,m,k,ef", [(2000, 16, 8, 10, 64), (1500, 64, 16, 10, 128), (900, 768, 32, 10, 100),
                                          (1200, 33, 8, 5, 16), (2500, 32, 8, 70, 200), (600, 16, 4, 10, 1000), (300, 8, 4, 1, 1)])
def test_predicate_aware_walk_matches_oracle(vg, ctx, metric, n, dim, m, k, ef):
    rng = np.random.default_rng(n + dim + metric)
    base = rng.standard_normal((n, dim)).astype(np.float32)
    if metric:
        base /= np.linalg.norm(base, axis=1, keepdims=True)
    l0, upper, entry = graphs.build_hnsw(base, m=m, seed=3)
    oidx = o.HnswIndex(base, dim, l0, upper, entry, metric=metric)
    idx = vg.Index(ctx, n, dim, vg.Metric(metric))
    idx.set_vectors(base)
    idx.set_hnsw_graph(l0, upper, entry, m=m)
    q = rng.standard_normal((8, dim)).astype(np.float32)
    for keep in (0.3, 0.1, 0.02, 0.6):
        per_query = rng.random((8, n)) < keep
        per_query[1, entry] = False
        per_query[2, entry] = True
        per_query[3, :] = False                # nothing passes: the navigation queue drains
        per_query[4, :] = True                 # everything passes
        check(idx, oidx, q, k, ef, per_query)
        check(idx, oidx, q, k, ef, per_query[0])
    deleted = rng.random(n) < 0.2              # tombstones: passing rows that are dead reset the miss counter but stay out
    deleted[entry] = True
    check(idx, oidx, q, k, ef, rng.random((8, n)) < 0.25, deleted=deleted)


def test_uploaded_edge_distances_drive_the_navigation(vg, ctx):
    """Neighbor.Dist as the host holds it (here: scaled copies and zeros, so that a recomputed value would walk differently):
    `next.Dist > 0` picks the cached value, 0 falls back to distFunc; the 1.5 x worst gate reads it too."""
    rng = np.random.default_rng(12)
    n, dim, m = 1800, 24, 8
    base = rng.standard_normal((n, dim)).astype(np.float32)
    l0, upper, entry = graphs.build_hnsw(base, m=m, seed=5)
    oidx = o.HnswIndex(base, dim, l0, upper, entry)
    idx = vg.Index(ctx, n, dim)
    idx.set_vectors(base)
    idx.set_hnsw_graph(l0, upper, entry, m=m)
    m0 = l0.shape[1]
    nb = np.where(l0 == 0xFFFFFFFF, 0, l0).astype(np.int64)
    true = ((base[:, None, :] - base[nb]) ** 2).sum(-1).astype(np.float32)
    l0_dist = (true * rng.choice([0.5, 1.0, 3.0], size=true.shape)).astype(np.float32)
    l0_dist[rng.random(true.shape) < 0.2] = 0.0
    idx.set_hnsw_edge_distances(l0_dist)
    q = rng.standard_normal((6, dim)).astype(np.float32)
    masks = rng.random((6, n)) < 0.1
    check(idx, oidx, q, 10, 80, masks, l0_dist=l0_dist.reshape(n, m0))
    idx.set_hnsw_edge_distances(None)          # back to the recomputed ones
    check(idx, oidx, q, 10, 80, masks)


def test_predicate_errors_and_the_filtered_entry(vg, ctx):
    rng = np.random.default_rng(2)
    n, dim = 500, 16
    base = rng.standard_normal((n, dim)).astype(np.float32)
    l0, upper, entry = graphs.build_hnsw(base, m=8, seed=1)
    idx = vg.Index(ctx, n, dim)
    q = rng.standard_normal((3, dim)).astype(np.float32)
    with pytest.raises(vg.VecgoHipError):
        idx.search_hnsw_predicate(q, 5, 20, np.ones(n, bool))       # no graph
    idx.set_vectors(base)
    idx.set_hnsw_graph(l0, upper, entry, m=8)
    with pytest.raises(vg.VecgoHipError) as e:
        idx.search_hnsw_predicate(q, 5, 5000, np.ones(n, bool))     # ef beyond the LDS results heap
    assert e.value.status == -5
    mask = rng.random(n) < 0.15
    a = idx.search_hnsw_predicate(q, 5, 20, mask)
    b = idx.search_hnsw_filtered(q, 5, 20, mask, 0.15)              # selectivity <= 0.3 routes here
    c = idx.search_hnsw_filtered(q, 5, 20, mask, 0.0)               # unknown selectivity too
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[0], c[0]) and np.array_equal(bits(a[1]), bits(b[1]))


def test_a_queue_that_outgrows_its_first_pass_slots(vg, ctx):
    """The navigation queue is unbounded: a filter nothing passes walks the whole component on edge distances and queues every
    node.  More rows than the first pass's 131072 slots: the walk is run again with a slot per row — same answer and counters
    as the oracle; ordinary queries of the same batch are untouched."""
    rng = np.random.default_rng(31)
    n, dim, m = 140000, 8, 8
    base = rng.standard_normal((n, dim)).astype(np.float32)
    idx = vg.Index(ctx, n, dim)
    idx.set_vectors(base)
    idx.build_hnsw(m=m, ef_construction=40)
    l0, upper, entry = idx.get_hnsw_graph()
    oidx = o.HnswIndex(base, dim, l0, upper, entry, m=m)
    q = rng.standard_normal((3, dim)).astype(np.float32)
    masks = np.zeros((3, n), bool)
    masks[1] = rng.random(n) < 0.05
    masks[2, rng.integers(0, n, 3)] = True          # three rows pass: the queue grows until they are found, and beyond
    ids, sc, st = idx.search_hnsw_predicate(q, 5, 32, masks, stats=True)
    assert int(st[0][0]) > 131072                   # the whole component was visited (and queued)
    for qi in range(3):
        eid, esc, est = oidx.search_predicate(q[qi], 5, 32, masks[qi])
        assert np.array_equal(ids[qi, :eid.size], eid) and np.array_equal(bits(sc[qi, :eid.size]), bits(esc)), qi
        assert tuple(int(x) for x in st[qi]) == (est.nodes_visited, est.distance_computations, est.distance_short_circuits, est.pops), qi
