"""diskann.Segment.Search with `filter` set (diskann/segment.go:616-627) on the GPU vs the oracle: a row whose filter bit is
clear is walked through (traversal queue, visited set, counters) but never enters the result heap, and the stop test reads
the heap of matching rows — ids, scores and counters equal, for the fp32 / PQ / RaBitQ node scorers."""
import numpy as np
import pytest

from oracle import oracle as o
from tests import graphs
from tests.test_gpu_graph import bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vg():
    import vecgo_amd
    return vecgo_amd


@pytest.fixture(scope="module")
def ctx(vg):
    return vg.Context(0)


def check(idx, ov, q, k, kind, mask):
    ids, sc, st = idx.search_vamana_filtered(q, k, mask, kind=kind, stats=True)
    for qi in range(q.shape[0]):
        mi = mask if mask.ndim == 1 else mask[qi]
        eid, esc, est = ov.search(q[qi], k, mask=mi)
        r_ = eid.size
        assert np.array_equal(ids[qi, :r_], eid), (qi, ids[qi], eid)
        assert np.array_equal(bits(sc[qi, :r_]), bits(esc))
        assert np.all(ids[qi, r_:] == 0xFFFFFFFF)
        assert np.all(mi[ids[qi, :r_]])
        assert (int(st[qi][0]), int(st[qi][1]), int(st[qi][3])) == (est.nodes_visited, est.distance_computations, est.pops)


@pytest.mark.parametrize("kind", [0, 1, 2])
@pytest.mark.parametrize("n,dim,r,k", [(1500, 32, 16, 10), (800, 768, 32, 10), (1200, 32, 16, 100), (90, 16, 8, 128)])
def test_filtered_vamana_matches_oracle(vg, ctx, kind, n, dim, r, k):
    rng = np.random.default_rng(n + dim + kind)
    base = rng.standard_normal((n, dim)).astype(np.float32)
    g, entry = graphs.build_vamana(base, r=r, seed=n)
    idx = vg.Index(ctx, n, dim)
    idx.set_vamana_graph(g, entry)
    if kind == 0:
        ov = o.VamanaIndex(g, entry, dim, o.VAMANA_F32, base=base)
        idx.set_vectors(base)
    elif kind == 1:
        m = dim // 8
        opq = o.ProductQuantizer(dim, m, 256)
        opq.set_codebooks(rng.integers(-128, 128, m * 256 * 8).astype(np.int8),
                          (rng.random(m) * 0.02 + 0.005).astype(np.float32),
                          ((rng.random(m) * 2 - 1) * 0.1).astype(np.float32))
        codes = opq.encode_batch(base)
        ov = o.VamanaIndex(g, entry, dim, o.VAMANA_PQ, pq=opq, codes=codes)
        pq = vg.ProductQuantizer(ctx, dim, m, 256)
        pq.set_codebooks(opq.codebooks, opq.scales, opq.offsets)
        idx.set_pq_codes(pq, codes)
    else:
        codes = o.rabitq_encode_batch(base, dim)
        ov = o.VamanaIndex(g, entry, dim, o.VAMANA_RABITQ, codes=codes)
        idx.set_rabitq_codes(codes)
    q = rng.standard_normal((8, dim)).astype(np.float32)
    for keep in (0.7, 0.1, 0.01):
        per_query = rng.random((8, n)) < keep
        per_query[1, entry] = False          # the start node itself filtered out
        per_query[2, entry] = True
        per_query[3, :] = False              # nothing passes: the walk visits the whole component
        check(idx, ov, q, k, kind, per_query)
        check(idx, ov, q, k, kind, per_query[0])
    # everything passes = the unfiltered search, counters included
    ids, sc, st = idx.search_vamana_filtered(q, k, np.ones(n, bool), kind=kind, stats=True)
    uid, usc, ust = idx.search_vamana(q, k, kind=kind, stats=True)
    assert np.array_equal(ids, uid) and np.array_equal(bits(sc), bits(usc)) and np.array_equal(st, ust)


def test_filtered_vamana_dot_metric(vg, ctx):
    rng = np.random.default_rng(9)
    base = rng.standard_normal((600, 16)).astype(np.float32)
    g, entry = graphs.build_vamana(base, r=12, seed=1)
    ov = o.VamanaIndex(g, entry, 16, o.VAMANA_F32, metric=o.METRIC_DOT, base=base)
    idx = vg.Index(ctx, 600, 16, vg.Metric.DOT)
    idx.set_vectors(base); idx.set_vamana_graph(g, entry)
    q = rng.standard_normal((6, 16)).astype(np.float32)
    check(idx, ov, q, 5, 0, rng.random((6, 600)) < 0.3)


def test_selective_filter_on_a_large_segment_drops_nothing(vg, ctx):
    """with a selective filter the walk goes on until k MATCHING rows are in the heap — here there are fewer than k, so
    it visits the whole component — and the traversal queue outgrows the 65 536 items the unfiltered walk is given (every node is
    discovered long before it is popped: ~94 k queued at the peak).  The reference's queue is unbounded (diskann/segment.go:641-703):
    ids, scores and counters must still equal the oracle's, i.e. nothing may be dropped."""
    n, dim, r, k = 100_000, 8, 16, 50
    rng = np.random.default_rng(2025)
    base = rng.standard_normal((n, dim)).astype(np.float32)
    g = rng.integers(0, n, (n, r)).astype(np.uint32)      # a random regular digraph: any adjacency is a valid graph to walk
    ov = o.VamanaIndex(g, 0, dim, o.VAMANA_F32, base=base)
    idx = vg.Index(ctx, n, dim)
    idx.set_vectors(base)
    idx.set_vamana_graph(g, 0)
    q = rng.standard_normal((3, dim)).astype(np.float32)
    mask = rng.random(n) < 0.0003                          # ~30 matching rows < k
    assert 0 < mask.sum() < k
    ids, sc, st = idx.search_vamana_filtered(q, k, mask, kind=0, stats=True)
    for qi in range(q.shape[0]):
        eid, esc, est = ov.search(q[qi], k, mask=mask)
        assert est.nodes_visited > 90_000                  # the whole component was walked
        assert np.array_equal(ids[qi, :eid.size], eid) and np.array_equal(bits(sc[qi, :eid.size]), bits(esc))
        assert np.all(ids[qi, eid.size:] == 0xFFFFFFFF)
        assert (int(st[qi][0]), int(st[qi][1]), int(st[qi][2]), int(st[qi][3])) == (est.nodes_visited, est.distance_computations, 0, est.pops)
