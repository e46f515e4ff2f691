"""HNSW construction on the GPU (vg_hnsw_build) vs the oracle's restatement of hnsw.go's insert path
(oracle/vg_oracle_hnsw_build.c): the same graph, list order included.  The oracle recomputes every pair
distance and pops every candidate order out of the reference's heap; the GPU keeps a bit matrix per row and
ranks by distance, replaying the heap only on ties — the two must not differ anywhere."""
import numpy as np
import pytest

from oracle import oracle as o

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vg():
    import vecgo_amd
    return vecgo_amd


@pytest.fixture(scope="module")
def ctx(vg):
    return vg.Context(0)


def _same_graph(a, b):
    l0a, ua, ea = a
    l0b, ub, eb = b
    assert ea == eb
    assert len(ua) == len(ub)
    bad = np.nonzero((l0a != l0b).any(1))[0]
    assert bad.size == 0, (bad[:5], l0a[bad[0]], l0b[bad[0]])
    for (sa, aa), (sb, ab) in zip(ua, ub):
        assert np.array_equal(sa, sb)
        assert np.array_equal(aa, ab)


@pytest.mark.parametrize("n,dim,m,ef,metric,max_batch,growth_div,kind", [
    (600, 16, 4, 32, 0, 1, 32, "uniform"),       # sequential = the reference's Insert loop; M0 = 8 rows prune constantly
    (1500, 16, 8, 200, 0, 1, 32, "uniform"),     # hnsw_test.go:43-60 shape (M = 8, EF = 200), sequential
    (2000, 32, 8, 64, 0, 64, 16, "normal"),      # batches
    (1200, 64, 16, 100, 0, 128, 8, "normal"),    # d = 64: bounded-kernel path inside the insert search
    (700, 768, 32, 300, 0, 64, 16, "normal"),    # BASELINE shape of a row: M = 32, M0 = 64, EF = 300
    (1000, 100, 6, 48, 0, 32, 16, "normal"),     # ragged dim, M0 = 12
    (1500, 24, 8, 64, 2, 64, 16, "unit"),        # Dot
    (1500, 24, 8, 64, 1, 64, 16, "unit"),        # Cosine
    (1200, 8, 4, 40, 0, 32, 8, "grid"),          # integer grid: equal distances everywhere (heap-order ties)
    (900, 16, 8, 64, 0, 48, 16, "dups"),         # duplicated rows: zero distances and ties
    (40, 8, 8, 16, 0, 8, 4, "normal"),           # fewer nodes than M0: rows never fill
    (3000, 16, 2, 24, 0, 256, 16, "normal"),     # M = 2: many levels
])
def test_build_matches_oracle(vg, ctx, n, dim, m, ef, metric, max_batch, growth_div, kind):
    rng = np.random.default_rng(n * 7 + dim + m)
    if kind == "uniform":
        base = rng.random((n, dim)).astype(np.float32)
    elif kind == "grid":
        base = rng.integers(0, 3, (n, dim)).astype(np.float32)
    elif kind == "dups":
        base = rng.standard_normal((n // 3, dim)).astype(np.float32)[rng.integers(0, n // 3, n)]
    else:
        base = rng.standard_normal((n, dim)).astype(np.float32)
        if kind == "unit":
            base /= np.linalg.norm(base, axis=1, keepdims=True)
    want = o.hnsw_build(base, dim, m=m, ef=ef, metric=metric, max_batch=max_batch, growth_div=growth_div)
    idx = vg.Index(ctx, n, dim, vg.Metric(metric))
    idx.set_vectors(base)
    idx.build_hnsw(m=m, ef_construction=ef, max_batch=max_batch, growth_div=growth_div)
    got = idx.get_hnsw_graph()
    _same_graph(got, want)
    # and the graph is searchable where it was built: same answers as the oracle's search over the oracle's graph
    q = rng.standard_normal((8, dim)).astype(np.float32)
    ids, sc = idx.search_hnsw(q, 5, 32)
    oidx = o.HnswIndex(base, dim, *want, metric=metric, m=m)
    for qi in range(8):
        eid, esc, _ = oidx.search(q[qi], 5, 32)
        assert np.array_equal(ids[qi, :eid.size], eid)
        assert np.array_equal(sc[qi, :eid.size].view(np.uint32), esc.view(np.uint32))


def test_levels_match_oracle(vg):
    lib = vg._lib.load()
    for m in (2, 8, 16, 32):
        lv, _, _ = o.hnsw_layout(20000, m)
        got = np.array([lib.vg_hnsw_level_for_id(i, m) for i in range(0, 20000, 7)])
        assert np.array_equal(got, lv[::7])


def test_built_graph_recall(vg, ctx):
    """hnsw_test.go:43-103: precision against brute force >= 0.99 at (1000 x 16, M = 8, EF = 200)."""
    rng = np.random.default_rng(4711)
    n, dim = 1000, 16
    base = rng.random((n, dim)).astype(np.float32)
    idx = vg.Index(ctx, n, dim)
    idx.set_vectors(base)
    idx.build_hnsw(m=8, ef_construction=200, max_batch=16, growth_div=32)
    q = rng.random((100, dim)).astype(np.float32)
    ids, _ = idx.search_hnsw(q, 10, 200)
    gt, _ = idx.search_flat(q, 10)
    hit = sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(ids, gt))
    assert hit / 1000 >= 0.99
