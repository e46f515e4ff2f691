"""g.tombstones on the GPU vs the oracle (vg_index_set_hnsw_tombstones): a deleted node is walked through but never enters
the results (searchLayerUnfiltered hnsw.go:1381-1390, processEntryPointUnfiltered :1559-1565, the post-filter's re-filter
:1198, the predicate-aware walk :1485) — ids, score bits and counters equal for the fp32 walk (LDS and split heaps), the
PQ-scored walk, the post-filter walk and the predicate-aware walk."""
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


def same(got, want_fn, nq):
    ids, sc, st = got
    for qi in range(nq):
        eid, esc, est = want_fn(qi)
        r = eid.size
        assert np.array_equal(ids[qi, :r], eid), (qi, ids[qi], eid)
        assert np.array_equal(bits(sc[qi, :r]), bits(esc)), qi
        assert np.all(ids[qi, r:] == 0xFFFFFFFF)
        assert tuple(int(x) for x in st[qi]) == (est.nodes_visited, est.distance_computations, est.distance_short_circuits, est.pops), qi


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("n,dim,m,k,ef", [(2000, 16, 8, 10, 64), (1200, 768, 32, 10, 128), (1500, 33, 8, 5, 16), (1800, 32, 8, 10, 700)])
def test_walks_skip_deleted_nodes(vg, ctx, metric, n, dim, m, k, ef):
    rng = np.random.default_rng(n + dim + metric)
    base = rng.standard_normal((n, dim)).astype(np.float32)
    if metric:
        base /= np.linalg.norm(base, axis=1, keepdims=True)
    l0, upper, entry = graphs.build_hnsw(base, m=m, seed=3)
    oidx = o.HnswIndex(base, dim, l0, upper, entry, metric=metric)
    idx = vg.Index(ctx, n, dim, vg.Metric(metric))
    idx.set_vectors(base)
    idx.set_hnsw_graph(l0, upper, entry, m=m)
    q = rng.standard_normal((6, dim)).astype(np.float32)
    clean = idx.search_hnsw(q, k, ef, stats=True)
    for frac in (0.1, 0.6):
        dead = rng.random(n) < frac
        dead[entry] = True                                    # the entry point itself deleted
        oidx.set_tombstones(dead)
        idx.set_hnsw_tombstones(dead)
        got = idx.search_hnsw(q, k, ef, stats=True)
        same(got, lambda qi: oidx.search(q[qi], k, ef), 6)
        assert not dead[got[0][got[0] != 0xFFFFFFFF]].any()
        mask = rng.random((6, n)) < 0.6                       # post-filter walk: filter.Matches && !tombstones (hnsw.go:1198)
        same(idx.search_hnsw_filtered(q, k, ef, mask, 0.6, stats=True), lambda qi: oidx.search_filtered(q[qi], k, ef, mask[qi], 0.6), 6)
        if ef <= 4096:                                        # the predicate-aware walk reads the same bitmap
            sel = rng.random((6, n)) < 0.2
            same(idx.search_hnsw_predicate(q, k, ef, sel, stats=True), lambda qi: oidx.search_predicate(q[qi], k, ef, sel[qi]), 6)
    oidx.set_tombstones(None)
    idx.set_hnsw_tombstones(None)                             # cleared: the tuned walk again, same answers as before
    again = idx.search_hnsw(q, k, ef, stats=True)
    assert np.array_equal(again[0], clean[0]) and np.array_equal(bits(again[1]), bits(clean[1])) and np.array_equal(again[2], clean[2])


def test_pq_scored_walk_skips_deleted_nodes(vg, ctx):
    rng = np.random.default_rng(4)
    n, dim, m = 1500, 64, 8
    base = rng.standard_normal((n, dim)).astype(np.float32)
    l0, upper, entry = graphs.build_hnsw(base, m=m, seed=2)
    opq = o.ProductQuantizer(dim, dim // 8, 256); opq.train(base, iters=3, seed=1)
    codes = opq.encode_batch(base)
    oidx = o.HnswIndex(base, dim, l0, upper, entry, pq=opq, codes=codes)
    pq = vg.ProductQuantizer(ctx, dim, dim // 8, 256)
    pq.set_codebooks(opq.codebooks, opq.scales, opq.offsets)
    idx = vg.Index(ctx, n, dim)
    idx.set_vectors(base)
    idx.set_pq_codes(pq, codes)
    idx.set_hnsw_graph(l0, upper, entry, m=m)
    dead = rng.random(n) < 0.4
    oidx.set_tombstones(dead)
    idx.set_hnsw_tombstones(dead)
    q = rng.standard_normal((5, dim)).astype(np.float32)
    for ef in (64, 600):
        same(idx.search_hnsw_pq(q, 10, ef, stats=True), lambda qi: oidx.search(q[qi], 10, ef), 5)
