"""CPU tests of the oracle's heaps and search loops against the properties the reference's own
tests assert (searcher/queue_test.go, candidate_queue_test.go; hnsw/hnsw_test.go:43-103 recall
vs brute force; diskann/extra_test.go:254-325)."""
import numpy as np
import pytest

from oracle import oracle as o
from tests import graphs


def test_flat_search_is_sorted_and_tie_broken():
    rng = np.random.default_rng(1)
    base = rng.standard_normal((500, 32)).astype(np.float32)
    base[100] = base[7]; base[300] = base[7]   # equal scores → RowID ascending
    q = base[7].copy()
    ids, sc = o.flat_search_f32(base, 32, q, 5)
    assert list(ids[:3]) == [7, 100, 300] and np.all(sc[:3] == 0)
    assert np.all(np.diff(sc) >= 0)
    ids_d, sc_d = o.flat_search_f32(base, 32, q, 5, o.METRIC_DOT)
    assert np.all(np.diff(sc_d) <= 0)


@pytest.mark.parametrize("metric", [o.METRIC_L2, o.METRIC_DOT, o.METRIC_COSINE])
def test_hnsw_oracle_recall_vs_brute_force(metric):
    """hnsw_test.go:43-103 style: precision vs brute force on a well-connected graph."""
    rng = np.random.default_rng(4711)
    n, dim, k = 1000, 16, 10
    base = rng.random((n, dim)).astype(np.float32)
    if metric != o.METRIC_L2:
        base /= np.linalg.norm(base, axis=1, keepdims=True)
    l0, upper, entry = graphs.build_hnsw(base, m=8, seed=1, ragged=False)
    idx = o.HnswIndex(base, dim, l0, upper, entry, metric=metric)
    hits = 0
    for t in range(30):
        q = rng.random(dim).astype(np.float32)
        if metric != o.METRIC_L2:
            q /= np.linalg.norm(q)
        ids, sc, st = idx.search(q, k, 200)
        assert ids.size == k and st.distance_computations > 0 and st.pops > 0
        d = ((base - q) ** 2).sum(1) if metric != o.METRIC_DOT else -(base @ q)
        hits += len(set(ids.tolist()) & set(np.argsort(d, kind="stable")[:k].tolist()))
        assert np.all(np.diff(sc) >= 0)
    assert hits / (30 * k) >= 0.95


def test_vamana_oracle_finds_neighbours():
    rng = np.random.default_rng(3)
    n, dim, k = 800, 24, 10
    base = rng.standard_normal((n, dim)).astype(np.float32)
    g, entry = graphs.build_vamana(base, r=16, seed=2)
    v = o.VamanaIndex(g, entry, dim, o.VAMANA_F32, base=base)
    hits = 0
    for t in range(20):
        q = rng.standard_normal(dim).astype(np.float32)
        ids, sc, st = v.search(q, k)
        assert ids.size == k and np.all(np.diff(sc) >= 0)
        d = ((base - q) ** 2).sum(1)
        hits += len(set(ids.tolist()) & set(np.argsort(d)[:k].tolist()))
    assert hits / 200 >= 0.6
    # PQ and RaBitQ distance functions return k results (diskann/extra_test.go:254-325)
    opq = o.ProductQuantizer(dim, 6, 256); opq.train(base, iters=5, seed=1)
    codes = opq.encode_batch(base)
    ids, sc, _ = o.VamanaIndex(g, entry, dim, o.VAMANA_PQ, pq=opq, codes=codes).search(base[5], k)
    assert ids.size == k
    rc = o.rabitq_encode_batch(base, dim)
    ids, sc, _ = o.VamanaIndex(g, entry, dim, o.VAMANA_RABITQ, codes=rc).search(base[5], k)
    assert ids.size == k and np.all(sc >= 0)


def test_kmeans_reference_properties():
    """kmeans/kmeans_test.go:12-93: two obvious clusters separate; n<k → nil; bad metric → error;
    FindClosestCentroids ordering."""
    rng = np.random.default_rng(0)
    a = rng.standard_normal((50, 4)).astype(np.float32) * 0.1
    b = rng.standard_normal((50, 4)).astype(np.float32) * 0.1 + 10
    x = np.concatenate([a, b])
    c = o.kmeans_train(x, 4, 2, max_iter=10, seed=3).reshape(2, 4)
    assert sorted(np.round(c.mean(1)).tolist()) == [0.0, 10.0]
    assert o.kmeans_train(x[:1], 4, 2) is None
    with pytest.raises(ValueError):
        o.kmeans_train(x, 4, 2, metric=o.METRIC_HAMMING)
    cents = np.array([[0, 0], [1, 1], [5, 5], [10, 10]], np.float32)
    assert list(o.find_closest_centroids(np.array([0.9, 0.9], np.float32), cents, 2, 2)) == [1, 0]
    assert o.assign_partition(np.array([9, 9], np.float32), cents, 2) == 3


def test_pq_oracle_reference_assertions():
    """pq_test.go:84-128: ADC ~= L2(q, Decode(code)) within 1e-3; LUT entries are the a8 terms."""
    rng = np.random.default_rng(0)
    x = rng.standard_normal((1000, 128)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    pq = o.ProductQuantizer(128, 8, 256); pq.train(x, iters=20, seed=1)
    q = x[1]; code = pq.encode(x[2])
    dec = pq.decode(code)
    assert float(np.mean((x[2] - dec) ** 2)) < 0.5
    assert abs(float(pq.asym_distance(q, code)) - float(np.sum((q - dec) ** 2, dtype=np.float32))) <= 1e-3
    table = pq.build_table(q)
    assert abs(float(o.adc(table, code, 8)) - float(pq.asym_distance(q, code))) <= 1e-4


def test_cpu_twin_of_the_hnsw_pq_rerank_pipeline():
    """oracle.BENCH_HNSW_PQ_RERANK (bench.py's CPU twin of the metric's named pipeline, vg_cpu_bench.c): the graph
    walked on PQ codes for ef candidates, Segment.Rerank's exact distances, best k by (Score, RowID)
    (engine/search.go:914-965) — equals the same pipeline composed by hand from the oracle's pieces."""
    from tests import graphs
    rng = np.random.default_rng(1)
    n, dim, m = 3000, 32, 4
    base = rng.standard_normal((n, dim)).astype(np.float32)
    l0, upper, entry = graphs.build_hnsw(base, m=8, seed=1)
    pq = o.ProductQuantizer(dim, m, 256)
    pq.train(base[:1500], iters=4, seed=3)
    codes = pq.encode_batch(base)
    h = o.HnswIndex(base, dim, l0, upper, entry, m=8, pq=pq, codes=codes)
    q = rng.standard_normal((16, dim)).astype(np.float32)
    r = o.bench_run(o.BENCH_HNSW_PQ_RERANK, q, 10, 16, 0.0, hnsw=h, ef=64, want_ids=True)
    for i in range(16):
        cid, _, _ = h.search(q[i], 64, 64)
        sc = np.array([o.l2(q[i], base[c]) for c in cid], np.float32)
        order = sorted(range(len(cid)), key=lambda j: (sc[j], cid[j]))[:10]
        assert np.array_equal(r["ids"][i], cid[order])
        assert np.array_equal(r["scores"][i].view(np.uint32), sc[order].view(np.uint32))


def test_flat_segment_reference_tests():
    """flat/{segment,quantization,partitioned}_test.go (reference_kats.json flat_segment_search) against the oracle's
    flat.Segment.Search / Rerank restatement"""
    from tests import flat_segment_kats

    def search(rows, q, k):
        return o.FlatSegment(rows, rows.shape[1]).search(q, k)

    def search_sq8_rerank(rows, q, k):
        sq = o.ScalarQuantizer(rows.shape[1]); sq.train(rows)
        codes = np.stack([sq.encode(r) for r in rows])
        ids, _ = o.FlatSegment(rows, rows.shape[1], sq=sq, codes=codes).search(q, k)
        return ids, o.rerank_f32(rows, rows.shape[1], q, ids)

    def partition(rows, parts):
        cent = o.kmeans_train(rows, rows.shape[1], parts, max_iter=10, seed=1)
        assign = np.array([o.assign_partition(r, cent, rows.shape[1]) for r in rows])
        order = np.argsort(assign, kind="stable")
        off = np.concatenate([[0], np.cumsum(np.bincount(assign, minlength=parts))]).astype(np.uint32)
        return cent.reshape(parts, -1), off, rows[order]

    def search_probed(rows, cent, off, q, k, nprobes):
        return o.FlatSegment(rows, rows.shape[1], centroids=cent, part_offsets=off).search(q, k, nprobes)

    flat_segment_kats.run(search, search_sq8_rerank, partition, search_probed)


def test_vamana_oracle_filter_only_touches_the_result_heap():
    """diskann/segment.go:616-627: a filtered-out row still feeds the traversal; the results are matching rows, and with
    every row matching the search is the unfiltered one, counters included.  A filter that rejects everything walks the whole
    component (the stop test never fires: the result heap stays empty)."""
    rng = np.random.default_rng(4)
    n, dim, k = 700, 16, 10
    base = rng.standard_normal((n, dim)).astype(np.float32)
    g, entry = graphs.build_vamana(base, r=12, seed=2)
    v = o.VamanaIndex(g, entry, dim, o.VAMANA_F32, base=base)
    q = rng.standard_normal(dim).astype(np.float32)
    uid, usc, ust = v.search(q, k)
    aid, asc, ast = v.search(q, k, mask=np.ones(n, bool))
    assert np.array_equal(uid, aid) and np.array_equal(usc, asc)
    assert (ust.nodes_visited, ust.pops) == (ast.nodes_visited, ast.pops)
    mask = rng.random(n) < 0.2
    fid, fsc, fst = v.search(q, k, mask=mask)
    assert fid.size == k and np.all(mask[fid]) and np.all(np.diff(fsc) >= 0)
    assert fst.nodes_visited >= ust.nodes_visited                 # the stop test waits for k MATCHING rows
    nid, nsc, nst = v.search(q, k, mask=np.zeros(n, bool))
    assert nid.size == 0 and nst.nodes_visited >= fst.nodes_visited


def test_predicate_aware_oracle_properties():
    """searchLayerPredicateAware restated (hnsw.go:1406-1558): every result passes the filter and is alive; with every row
    passing no expansion is skipped and each visited node is scored once; tombstoned rows never surface; cached edge distances
    change the navigation (a rejected node is not scored while results < ef/2), not the eligibility."""
    rng = np.random.default_rng(6)
    n, dim, k, ef = 1500, 16, 10, 64
    base = rng.standard_normal((n, dim)).astype(np.float32)
    l0, upper, entry = graphs.build_hnsw(base, m=8, seed=2)
    h = o.HnswIndex(base, dim, l0, upper, entry)
    q = rng.standard_normal(dim).astype(np.float32)
    ids, sc, st = h.search_predicate(q, k, ef, np.ones(n, bool))
    assert ids.size == k and np.all(np.diff(sc) >= 0)
    assert st.distance_short_circuits == 0 and st.distance_computations == st.nodes_visited
    mask = rng.random(n) < 0.15
    dead = rng.random(n) < 0.3
    ids, sc, st = h.search_predicate(q, k, ef, mask, deleted=dead)
    assert np.all(mask[ids]) and not np.any(dead[ids]) and np.all(np.diff(sc) >= 0)
    assert st.distance_computations < st.nodes_visited          # rejected nodes navigated by edge distance or skipped
    d = ((base - q) ** 2).sum(1)
    d[~mask | dead] = np.inf
    assert len(set(ids.tolist()) & set(np.argsort(d)[:k].tolist())) >= 5
    zero = np.zeros((n, l0.shape[1]), np.float32)                # no cached distances: every navigated node is scored
    _, _, st0 = h.search_predicate(q, k, ef, mask, l0_dist=zero)
    assert st0.distance_computations > st.distance_computations
