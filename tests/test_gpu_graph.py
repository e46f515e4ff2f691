"""HNSW and Vamana search on the GPU vs the oracle on the same graph: ids, scores AND the
per-query counters (FilterGateStats) must be identical — the traversal is a restatement of the
reference's heap semantics, not an approximation."""
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


@pytest.mark.parametrize("n,dim,m,metric,k,ef,nq", [
    (2000, 16, 8, 0, 10, 128, 24),     # hnsw_test.go:76-84 shape (2000x16, EF=128)
    (1500, 64, 16, 0, 10, 64, 12),     # d = 64: one full AVX-512 block, bounded path
    (1200, 768, 32, 0, 10, 128, 8),    # BASELINE dims, M0 = 64
    (1000, 100, 8, 0, 5, 40, 10),      # ragged dim: 8-wide and scalar tails of the bounded kernel
    (1500, 32, 8, 2, 10, 100, 12),     # Dot: -dot, no short-circuit
    (1500, 32, 8, 1, 10, 100, 12),     # Cosine: 0.5*L2
    (300, 16, 4, 0, 10, 300, 6),       # ef >= n: everything explored
    (2000, 16, 8, 0, 1, 16, 8),        # tiny ef: adaptive cap shrink paths
    (50, 8, 4, 0, 10, 8, 4),           # ef < k is raised to k (determineEF)
])
def test_hnsw_matches_oracle(vg, ctx, n, dim, m, metric, k, ef, nq):
    rng = np.random.default_rng(n + dim + m + metric)
    base = rng.random((n, dim)).astype(np.float32)
    if metric:
        base /= np.linalg.norm(base, axis=1, keepdims=True)
    l0, upper, entry = graphs.build_hnsw(base, m=m, seed=n)
    oidx = o.HnswIndex(base, dim, l0, upper, entry, metric=metric)
    idx = vg.Index(ctx, n, dim, vg.Metric(metric))
    idx.set_vectors(base)
    idx.set_hnsw_graph(l0, upper, entry, m=m)
    q = rng.random((nq, dim)).astype(np.float32)
    if metric:
        q /= np.linalg.norm(q, axis=1, keepdims=True)
    ids, sc, st = idx.search_hnsw(q, k, ef, stats=True)
    for qi in range(nq):
        eid, esc, est = oidx.search(q[qi], k, ef)
        r = eid.size
        assert np.array_equal(ids[qi, :r], eid), (qi, ids[qi], eid)
        assert np.array_equal(bits(sc[qi, :r]), bits(esc))
        assert tuple(int(x) for x in st[qi]) == _stats_tuple(est), (qi, st[qi], _stats_tuple(est))
        assert np.all(ids[qi, r:] == 0xFFFFFFFF)


# (k, ef): the extraction takes the register selection (k < 64), the LDS sort (k >= 64, heaps in LDS) or the pops (a tie
# among the k + 1 closest; k >= 64 with split heaps) — ef > 512 (fp32) / 448 (PQ) = split heaps
@pytest.mark.parametrize("k,ef", [(10, 32), (5, 200), (64, 64), (100, 300), (10, 600), (70, 700), (600, 600)])
@pytest.mark.parametrize("grid", [3, 40])   # 3: nearly every distance tied; 40: a tie here and there
def test_hnsw_duplicate_distances(vg, ctx, k, ef, grid):
    """Many equal distances: heap tie behaviour (strict comparisons, queue.go:161-290) decides
    which ids survive and in which order they come out."""
    rng = np.random.default_rng(5 + grid)
    n = 1500
    pts = rng.integers(0, grid, (n, 8)).astype(np.float32)   # integer grid: ties
    l0, upper, entry = graphs.build_hnsw(pts, m=8, seed=3)
    oidx = o.HnswIndex(pts, 8, l0, upper, entry)
    idx = vg.Index(ctx, n, 8)
    idx.set_vectors(pts)
    idx.set_hnsw_graph(l0, upper, entry, m=8)
    q = rng.integers(0, grid, (16, 8)).astype(np.float32)
    ids, sc, st = idx.search_hnsw(q, k, ef, stats=True)
    for qi in range(16):
        eid, esc, est = oidx.search(q[qi], k, ef)
        r = eid.size
        assert np.array_equal(ids[qi, :r], eid) and np.array_equal(bits(sc[qi, :r]), bits(esc)), (qi, k, ef)
        assert np.all(ids[qi, r:] == 0xFFFFFFFF)
        assert tuple(int(x) for x in st[qi]) == _stats_tuple(est)
    # the same walk scored from PQ codes (sub-dimension 8, one sub-quantizer: quantised sums tie even more)
    cb = rng.integers(-128, 128, 256 * 8).astype(np.int8)
    scales = np.array([0.02], np.float32); offsets = np.array([0.5], np.float32)
    opq = o.ProductQuantizer(8, 1, 256); opq.set_codebooks(cb, scales, offsets)
    pq = vg.ProductQuantizer(ctx, 8, 1, 256); pq.set_codebooks(cb, scales, offsets)
    codes = opq.encode_batch(pts)
    idx.set_pq_codes(pq, codes)
    oq = o.HnswIndex(pts, 8, l0, upper, entry, m=8, pq=opq, codes=codes)
    kk = min(k, ef)
    ids, sc = idx.search_hnsw_pq(q, kk, ef)
    for qi in range(16):
        eid, esc = oq.search(q[qi], kk, ef)[:2]
        r = eid.size
        assert np.array_equal(ids[qi, :r], eid) and np.array_equal(bits(sc[qi, :r]), bits(esc)), ("pq", qi, k, ef)


@pytest.mark.parametrize("kind", [0, 1, 2])
@pytest.mark.parametrize("n,dim,r,k", [(1500, 32, 16, 10), (800, 768, 32, 10), (400, 96, 12, 3),
                                       (1200, 32, 16, 100), (600, 64, 16, 300), (90, 16, 8, 128)])  # k > 64: LDS result list
def test_vamana_matches_oracle(vg, ctx, kind, n, dim, r, k):
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
    q = rng.standard_normal((10, dim)).astype(np.float32)
    ids, sc, st = idx.search_vamana(q, k, kind=kind, stats=True)
    for qi in range(10):
        eid, esc, est = ov.search(q[qi], k)
        r_ = eid.size
        assert np.array_equal(ids[qi, :r_], eid), (qi, ids[qi], eid)
        assert np.array_equal(bits(sc[qi, :r_]), bits(esc))
        assert (int(st[qi][0]), int(st[qi][1]), int(st[qi][3])) == (est.nodes_visited, est.distance_computations, est.pops)


def test_vamana_dot_metric_is_restated_as_written(vg, ctx):
    """Metric Dot: the reference keeps a min-heap on the raw dot product and a descending result
    heap (diskann/segment.go:597, :655-668) — restated faithfully, not 'fixed'."""
    rng = np.random.default_rng(9)
    base = rng.standard_normal((600, 16)).astype(np.float32)
    g, entry = graphs.build_vamana(base, r=12, seed=1)
    ov = o.VamanaIndex(g, entry, 16, o.VAMANA_F32, metric=o.METRIC_DOT, base=base)
    idx = vg.Index(ctx, 600, 16, vg.Metric.DOT)
    idx.set_vectors(base); idx.set_vamana_graph(g, entry)
    q = rng.standard_normal((6, 16)).astype(np.float32)
    ids, sc = idx.search_vamana(q, 5, kind=0)
    for qi in range(6):
        eid, esc, _ = ov.search(q[qi], 5)
        assert np.array_equal(ids[qi, :eid.size], eid) and np.array_equal(bits(sc[qi, :eid.size]), bits(esc))


def test_graph_errors(vg, ctx):
    idx = vg.Index(ctx, 10, 8)
    with pytest.raises(vg.VecgoHipError) as e:
        idx.search_hnsw(np.zeros((1, 8), np.float32), 3, 10)
    assert e.value.status == -9
    with pytest.raises(vg.VecgoHipError) as e:
        idx.search_vamana(np.zeros((1, 8), np.float32), 3)
    assert e.value.status == -9
    with pytest.raises(vg.VecgoHipError):
        idx.set_vamana_graph(np.zeros((10, 65), np.uint32), 0)


@pytest.mark.parametrize("n,dim,m,ef,k", [(3000, 32, 8, 600, 10), (2500, 64, 16, 1500, 100), (1200, 16, 8, 4000, 10)])
def test_hnsw_large_ef_heaps_in_hbm(vg, ctx, n, dim, m, ef, k):
    """ef > 512: the two heaps of a query live in HBM scratch; same ids, scores and counters."""
    rng = np.random.default_rng(ef)
    base = rng.standard_normal((n, dim)).astype(np.float32)
    l0, upper, entry = graphs.build_hnsw(base, m=m, seed=2)
    oidx = o.HnswIndex(base, dim, l0, upper, entry)
    idx = vg.Index(ctx, n, dim)
    idx.set_vectors(base)
    idx.set_hnsw_graph(l0, upper, entry, m=m)
    q = rng.standard_normal((6, dim)).astype(np.float32)
    ids, sc, st = idx.search_hnsw(q, k, ef, stats=True)
    for qi in range(6):
        eid, esc, est = oidx.search(q[qi], k, ef)
        assert np.array_equal(ids[qi, :eid.size], eid)
        assert np.array_equal(bits(sc[qi, :eid.size]), bits(esc))
        assert tuple(int(x) for x in st[qi]) == _stats_tuple(est)


@pytest.mark.parametrize("n,dim,m,ef,k", [(2000, 32, 8, 64, 10), (1500, 768, 16, 128, 128), (1000, 96, 8, 700, 50)])
def test_hnsw_pq_scored_matches_oracle(vg, ctx, n, dim, m, ef, k):
    """vg_search_hnsw_pq: the layer walk with distFunc = ComputeAsymmetricDistance over PQ codes."""
    rng = np.random.default_rng(n + ef)
    base = rng.standard_normal((n, dim)).astype(np.float32)
    l0, upper, entry = graphs.build_hnsw(base, m=m, seed=4)
    pm = dim // 8
    opq = o.ProductQuantizer(dim, pm, 256)
    opq.set_codebooks(rng.integers(-128, 128, pm * 256 * 8).astype(np.int8),
                      (rng.random(pm) * 0.02 + 0.005).astype(np.float32),
                      ((rng.random(pm) * 2 - 1) * 0.1).astype(np.float32))
    codes = opq.encode_batch(base)
    oidx = o.HnswIndex(base, dim, l0, upper, entry, pq=opq, codes=codes)
    pq = vg.ProductQuantizer(ctx, dim, pm, 256)
    pq.set_codebooks(opq.codebooks, opq.scales, opq.offsets)
    idx = vg.Index(ctx, n, dim)
    idx.set_vectors(base)
    idx.set_pq_codes(pq, codes)
    idx.set_hnsw_graph(l0, upper, entry, m=m)
    q = rng.standard_normal((8, dim)).astype(np.float32)
    ids, sc, st = idx.search_hnsw_pq(q, k, ef, stats=True)
    for qi in range(8):
        eid, esc, est = oidx.search(q[qi], k, ef)
        assert np.array_equal(ids[qi, :eid.size], eid), (qi, ids[qi], eid)
        assert np.array_equal(bits(sc[qi, :eid.size]), bits(esc))
        assert tuple(int(x) for x in st[qi]) == _stats_tuple(est)


def test_adjacency_ids_are_validated(vg, ctx):
    """A neighbour id that is neither a row nor VG_INVALID_ID would index the visited bitmap and the row
    arrays out of bounds: refused when the graph is set."""
    rng = np.random.default_rng(1)
    base = rng.standard_normal((100, 8)).astype(np.float32)
    idx = vg.Index(ctx, 100, 8); idx.set_vectors(base)
    l0 = rng.integers(0, 100, (100, 8)).astype(np.uint32)
    l0[17, 3] = 100
    with pytest.raises(vg.VecgoHipError) as e:
        idx.set_hnsw_graph(l0, (), 0, m=4)
    assert e.value.status == -1 and "neighbour ids" in e.value.message
    with pytest.raises(vg.VecgoHipError):      # and nothing half-set is searchable
        idx.search_hnsw(base[:1], 3, 10)
    with pytest.raises(vg.VecgoHipError) as e:
        idx.set_vamana_graph(l0, 0)
    assert "neighbour ids" in e.value.message
    l0[17, 3] = 0xFFFFFFFF
    idx.set_hnsw_graph(l0, (), 0, m=4); idx.set_vamana_graph(l0, 0)
    idx.search_hnsw(base[:1], 3, 10); idx.search_vamana(base[:1], 3)
