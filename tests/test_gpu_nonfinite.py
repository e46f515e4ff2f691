"""NaN scores in the exhaustive searches.  A NaN score is neither better nor worse than anything for the reference's heaps
(candidate_queue.go:12-38: `a.Score != b.Score`, then both `<` and `>` false), so what a search answers is decided by the heap's
layout: a NaN that enters while the heap fills stays, at the root it is never replaced, rows far better than everything kept are
turned away.  The outcome is DEFINED — the loops are sequential — and the library reproduces it: a query whose inputs could
produce a NaN score (non-finite query values or index data, dot products that can overflow both ways) is answered by a replay of
the reference's heap with float comparisons (vg_cand_replay.hpp) instead of the 64-bit keys of the scans.  Results: what the engine
takes out of the heap, Pop() until empty (engine/search.go:859-862), best first.  Ids equal the oracle's; score bits too, with
NaN == NaN (the sign / payload of a NaN is the instruction set's, not the algorithm's)."""
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


def same_scores(a, b):
    a = np.asarray(a, np.float32); b = np.asarray(b, np.float32)
    return a.shape == b.shape and bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))))


def check(got, want_of, nq, k):
    ids, sc = got
    for i in range(nq):
        eid, esc = want_of(i)
        r = eid.size
        assert np.array_equal(ids[i, :r], eid), (i, ids[i, :r][:12], eid[:12])
        assert same_scores(sc[i, :r], esc), (i, sc[i, :r][:12], esc[:12])
        assert np.all(ids[i, r:] == 0xFFFFFFFF)


def poisoned_queries(rng, x, nq):
    """finite queries first (they matter when the ROWS are poisoned), then NaN / +Inf / -Inf / mixed / huge ones"""
    dim = x.shape[1]
    q = (x[rng.integers(0, x.shape[0], nq)] + rng.standard_normal((nq, dim)).astype(np.float32) * 0.1).astype(np.float32)
    q[3, dim // 2] = np.nan
    q[4, :] = np.nan
    q[5, 1] = np.inf
    q[6, 2] = -np.inf
    q[7, 0] = np.inf; q[7, 1] = -np.inf
    q[8, :] = np.inf
    q[9] = q[9] * np.float32(1e25)                       # finite; dot products overflow both ways
    q[10] = 0.0
    return q


@pytest.mark.parametrize("metric", [0, 1])
@pytest.mark.parametrize("n,dim,k", [(3000, 64, 10), (2500, 100, 1), (5000, 30, 70), (70, 64, 100), (20000, 128, 10)])
def test_flat_fp32_with_non_finite_queries(vg, ctx, metric, n, dim, k):
    """vg_search_flat, clean rows: queries 0-2 and 11+ take the usual paths, the poisoned ones the replay"""
    rng = np.random.default_rng(n + dim + k + metric)
    x = rng.standard_normal((n, dim)).astype(np.float32)
    x[40:44] = x[40]                                      # ties
    nq = 14
    q = poisoned_queries(rng, x, nq)
    idx = vg.Index(ctx, n, dim, vg.Metric(metric))
    idx.set_vectors(x)
    check(idx.search_flat(q, k), lambda i: o.flat_search_f32(x, dim, q[i], k, metric=metric), nq, k)
    idx.enable_bf16_filter(True)
    check(idx.search_flat(q, k), lambda i: o.flat_search_f32(x, dim, q[i], k, metric=metric), nq, k)
    big = np.tile(q, (10, 1))                             # 140 queries: the GEMM nomination for the clean ones
    check(idx.search_flat(big, k), lambda i: o.flat_search_f32(x, dim, big[i], k, metric=metric), big.shape[0], k)


@pytest.mark.parametrize("metric", [0, 1])
@pytest.mark.parametrize("where", ["first", "late", "root", "many"])
def test_flat_fp32_with_non_finite_rows(vg, ctx, metric, where):
    """NaN / Inf in the ROWS: every query is replayed.  `first`: a NaN row among the first k (it enters the filling heap and stays);
    `late`: only beyond the first k (never enters: Better(NaN, top) is false); `root`: row 0 is a NaN and k = 1 — the root is never
    replaced, the answer is row 0 whatever else the segment holds; `many`: a third of the rows, NaN and +/-Inf mixed."""
    rng = np.random.default_rng(5 + metric)
    n, dim = 4000, 64
    k = 1 if where == "root" else 10
    x = rng.standard_normal((n, dim)).astype(np.float32)
    if where == "first":
        x[3, 7] = np.nan; x[6, :] = np.inf
    elif where == "late":
        x[500, 0] = np.nan; x[2000, 5] = -np.inf
    elif where == "root":
        x[0, 0] = np.nan
    else:
        bad = rng.random(n) < 0.33
        kind = rng.integers(0, 3, n)
        col = rng.integers(0, dim, n)
        for i in np.flatnonzero(bad):
            x[i, col[i]] = (np.nan, np.inf, -np.inf)[kind[i]]
    nq = 12
    q = (x[rng.integers(0, n, nq)] + rng.standard_normal((nq, dim)).astype(np.float32) * 0.1).astype(np.float32)
    q = np.nan_to_num(q, nan=0.5, posinf=1.0, neginf=-1.0).astype(np.float32)
    q[5, 3] = np.nan
    idx = vg.Index(ctx, n, dim, vg.Metric(metric))
    idx.set_vectors(x)
    check(idx.search_flat(q, k), lambda i: o.flat_search_f32(x, dim, q[i], k, metric=metric), nq, k)
    if where == "root":
        assert np.all(idx.search_flat(q, 1)[0][:, 0] == 0)


def _random_pq(rng, dim, m):
    sd = dim // m
    opq = o.ProductQuantizer(dim, m, 256)
    cb = rng.integers(-128, 128, m * 256 * sd).astype(np.int8)
    scales = (rng.random(m) * 0.02 + 0.005).astype(np.float32)
    offsets = ((rng.random(m) * 2 - 1) * 0.1).astype(np.float32)
    opq.set_codebooks(cb, scales, offsets)
    return opq


@pytest.mark.parametrize("n,dim,m,k", [(3000, 128, 16, 10), (2000, 200, 25, 1), (5000, 64, 4, 70), (50, 128, 16, 100), (9000, 768, 96, 10)])
def test_pq_adc_with_non_finite_queries(vg, ctx, n, dim, m, k):
    """vg_search_pq_adc: a NaN / Inf query value makes table entries NaN / +Inf — every row's sum holds them (NaN: all sums NaN, the
    first k rows stay; +Inf: ties, by row id; both: by the heap).  With and without the bf16 nomination; and a NaN offset in the
    quantizer (every query is at risk)."""
    from tests import hooks
    rng = np.random.default_rng(n + dim + m)
    opq = _random_pq(rng, dim, m)
    codes = rng.integers(0, 256, (n, m)).astype(np.uint8)
    codes[10:14] = codes[10]
    pq = vg.ProductQuantizer(ctx, dim, m, 256)
    pq.set_codebooks(opq.codebooks, opq.scales, opq.offsets)
    idx = vg.Index(ctx, n, dim, vg.Metric.L2)
    idx.set_pq_codes(pq, codes)
    nq = 14
    q = poisoned_queries(rng, rng.standard_normal((n, dim)).astype(np.float32), nq)
    want = lambda qq: (lambda i: o.flat_search_pq(opq, codes, qq[i], k))
    check(idx.search_pq_adc(q, k), want(q), nq, k)
    big = np.tile(q, (10, 1))
    hooks.set_hook("VG_PQ_NOM_ALWAYS", 1)
    try:
        idx.enable_pq_nomination(True)
        check(idx.search_pq_adc(big, k), want(big), big.shape[0], k)
    finally:
        hooks.set_hook("VG_PQ_NOM_ALWAYS", 0)
    of = np.array(opq.offsets, np.float32)
    of[m // 2] = np.nan
    opq.set_codebooks(opq.codebooks, opq.scales, of)
    pq2 = vg.ProductQuantizer(ctx, dim, m, 256)
    pq2.set_codebooks(opq.codebooks, opq.scales, opq.offsets)
    idx2 = vg.Index(ctx, n, dim, vg.Metric.L2)
    idx2.set_pq_codes(pq2, codes)
    check(idx2.search_pq_adc(q, k), want(q), nq, k)


@pytest.mark.parametrize("metric", [0, 2])
@pytest.mark.parametrize("n,dim,k", [(3000, 64, 10), (2500, 100, 1), (5000, 30, 70), (60, 64, 100)])
def test_sq8_with_non_finite_queries_and_bounds(vg, ctx, metric, n, dim, k):
    """vg_search_sq8 (L2Distance / DotProduct by the metric): poisoned queries; then a quantizer whose bounds hold a NaN"""
    rng = np.random.default_rng(n + dim + k + metric)
    x = rng.standard_normal((n, dim)).astype(np.float32)
    x[40:44] = x[40]
    nq = 14
    q = poisoned_queries(rng, x, nq)
    for poison in (False, True):
        ref = o.ScalarQuantizer(dim)
        sq = vg.ScalarQuantizer(ctx, dim)
        if poison:
            mins, maxs = x.min(0).copy(), x.max(0).copy()
            mins[dim // 3] = np.nan
            maxs[dim // 2] = np.inf
            sq.set_bounds(mins, maxs)
            codes = rng.integers(0, 256, (n, dim)).astype(np.uint8)
        else:
            sq.train(x)
            codes = sq.encode(x)
        for dst, src in zip((ref.mins, ref.maxs, ref.scales, ref.inv_scales), sq.params()):   # (SetBounds / Train parity: test_gpu_sq8.py)
            dst[:] = src
        ref.trained = True
        idx = vg.Index(ctx, n, dim, vg.Metric(metric))
        idx.set_sq8_codes(sq, codes)
        seg = o.FlatSegment(x, dim, metric=metric, sq=ref, codes=codes)
        check(idx.search_sq8(q, k), lambda i: seg.search(q[i], k), nq, k)
        if not poison:
            idx.enable_sq8_nomination(True)
            big = np.tile(q, (10, 1))
            check(idx.search_sq8(big, k), lambda i: seg.search(big[i], k), big.shape[0], k)


@pytest.mark.parametrize("n,dim,k", [(3000, 64, 10), (2500, 100, 1), (5000, 200, 70), (60, 64, 100)])
def test_rabitq_with_non_finite_queries_and_norms(vg, ctx, n, dim, k):
    """vg_search_rabitq: a non-finite query gives a non-finite query norm (rabitq.go:119-176) — every distance a NaN or +Inf; a row
    whose stored norm is a NaN / Inf puts every query at risk; 4 |q| |y| overflowing next to a Hamming distance of 0 is a NaN"""
    rng = np.random.default_rng(n + dim + k)
    x = rng.standard_normal((n, dim)).astype(np.float32)
    x[40:44] = x[40]
    nq = 14
    q = poisoned_queries(rng, x, nq)
    q[11] = x[7] * np.float32(3e18)                       # |q| ~ 2e19 finite, same sign bits as row 7: Hamming 0
    codes = o.rabitq_encode_batch(x, dim)
    cb = o.rabitq_code_bytes(dim)
    for poison in (False, True):
        c = codes.copy().reshape(n, cb)
        if poison:
            c[3, cb - 4:] = np.frombuffer(np.float32(np.nan).tobytes(), np.uint8)
            c[n // 2, cb - 4:] = np.frombuffer(np.float32(np.inf).tobytes(), np.uint8)
            c[7, cb - 4:] = np.frombuffer(np.float32(3e19).tobytes(), np.uint8)
        idx = vg.Index(ctx, n, dim, vg.Metric.L2)
        idx.set_rabitq_codes(c)
        check(idx.search_rabitq(q, k), lambda i: o.flat_search_rabitq(c, dim, q[i], k), nq, k)
        check(idx.search_rabitq(q[:1], k), lambda i: o.flat_search_rabitq(c, dim, q[i], k), 1, k)


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("where", ["queries", "rows"])
def test_hnsw_brute_with_nan_distances(vg, ctx, metric, mode, where):
    """vg_search_hnsw_brute (hnsw.BruteSearch + scanSegment / searchBitmap + extraction): searcher.PriorityQueue with the tests of
    hnsw.go:2089-2097 (`d < top.Distance`) / queue.go:199-203 (`d >= top.Distance` rejects — a NaN is NOT rejected there: it
    replaces the root).  With a NaN inside the heap its top can rise, which the batched replay's flagging assumes it never does:
    queries at risk are decided row by row against the live top.  Masks, per-query masks, k > n, ties."""
    rng = np.random.default_rng(31 + metric + 7 * mode)
    n, dim = 5003, 32
    x = rng.integers(0, 4, (n, dim)).astype(np.float32)          # a grid: ties
    nq = 14
    if where == "rows":
        bad = rng.random(n) < 0.05
        col = rng.integers(0, dim, n)
        kind = rng.integers(0, 3, n)
        for i in np.flatnonzero(bad):
            x[i, col[i]] = (np.nan, np.inf, -np.inf)[kind[i]]
        x[2, 0] = np.nan                                          # one among the first k
        q = rng.integers(0, 4, (nq, dim)).astype(np.float32)
    else:
        q = poisoned_queries(rng, x, nq)
    idx = vg.Index(ctx, n, dim, vg.Metric(metric))
    idx.set_vectors(x)
    oidx = o.HnswIndex(x, dim, np.full((n, 2), 0xFFFFFFFF, np.uint32), metric=metric)
    masks = rng.random((nq, n)) < 0.3
    for k in (1, 10, 70):
        for mask, per_query in ((None, False), (rng.random(n) < 0.5, False), (masks, True)):
            ids, sc = idx.search_hnsw_brute(q, k, mode, mask)
            for i in range(nq):
                m = None if mask is None else (mask[i] if per_query else mask)
                eid, esc = oidx.brute_search(q[i], k, mode, m)
                r = eid.size
                assert np.array_equal(ids[i, :r], eid), (k, per_query, i, ids[i, :r][:10], eid[:10])
                assert same_scores(sc[i, :r], esc), (k, per_query, i)
                assert np.all(ids[i, r:] == 0xFFFFFFFF)
    one = idx.search_hnsw_brute(q[3:4], 10, mode, None)           # one query (its own fast path)
    eid, esc = oidx.brute_search(q[3], 10, mode, None)
    assert np.array_equal(one[0][0, :eid.size], eid) and same_scores(one[1][0, :eid.size], esc)


@pytest.mark.parametrize("kind", [0, 1, 2])
@pytest.mark.parametrize("metric", [0, 2])
@pytest.mark.parametrize("where", ["queries", "data"])
def test_vamana_beam_with_nan_distances(vg, ctx, kind, metric, where):
    """diskann.Segment.Search (segment.go:503-706): the traversal queue is a PriorityQueue (float comparisons in the kernel too), the
    results a CandidateHeap whose root the stop test reads (`c.dist > heap[0].score`).  With NaN distances the heap's layout decides
    its contents, its root — and so where the walk stops: ids, scores AND the walk's counters equal the oracle's.  fp32 / PQ / RaBitQ
    node scorers, L2 and Dot, with and without a filter, k below and above 64."""
    from tests import graphs
    if metric == 2 and kind != 0:
        pytest.skip("the code scorers have one distance")
    rng = np.random.default_rng(77 + kind + metric)
    n, dim, r = 1500, 32, 16
    base = rng.standard_normal((n, dim)).astype(np.float32)
    g, entry = graphs.build_vamana(base, r=r, seed=5)
    nq = 14
    q = poisoned_queries(rng, base, nq) if where == "queries" else rng.standard_normal((nq, dim)).astype(np.float32)
    idx = vg.Index(ctx, n, dim, vg.Metric(metric))
    idx.set_vamana_graph(g, entry)
    if kind == 0:
        rows = base.copy()
        if where == "data":
            hubs = np.bincount(g[g != 0xFFFFFFFF].astype(np.int64), minlength=n).argsort()[-4:]
            rows[hubs[0], 3] = np.nan; rows[hubs[1], 0] = np.inf; rows[hubs[2], 5] = -np.inf
            rows[entry, 1] = np.nan
        ov = o.VamanaIndex(g, entry, dim, o.VAMANA_F32, metric=metric, base=rows)
        idx.set_vectors(rows)
    elif kind == 1:
        m = dim // 8
        opq = o.ProductQuantizer(dim, m, 256)
        sc = (rng.random(m) * 0.02 + 0.005).astype(np.float32)
        if where == "data":
            sc[1] = np.nan
        opq.set_codebooks(rng.integers(-128, 128, m * 256 * 8).astype(np.int8), sc, ((rng.random(m) * 2 - 1) * 0.1).astype(np.float32))
        codes = rng.integers(0, 256, (n, m)).astype(np.uint8)
        ov = o.VamanaIndex(g, entry, dim, o.VAMANA_PQ, pq=opq, codes=codes)
        pq = vg.ProductQuantizer(ctx, dim, m, 256)
        pq.set_codebooks(opq.codebooks, opq.scales, opq.offsets)
        idx.set_pq_codes(pq, codes)
    else:
        codes = o.rabitq_encode_batch(base, dim)
        if where == "data":
            cb = codes.shape[1]
            hubs = np.bincount(g[g != 0xFFFFFFFF].astype(np.int64), minlength=n).argsort()[-3:]
            codes[hubs[0], cb - 4:] = np.frombuffer(np.float32(np.nan).tobytes(), np.uint8)
            codes[hubs[1], cb - 4:] = np.frombuffer(np.float32(np.inf).tobytes(), np.uint8)
        ov = o.VamanaIndex(g, entry, dim, o.VAMANA_RABITQ, codes=codes)
        idx.set_rabitq_codes(codes)
    masks = rng.random((nq, n)) < 0.4
    for k in (1, 10, 100):
        for mask in (None, masks):
            if mask is None:
                ids, sc, st = idx.search_vamana(q, k, kind=kind, stats=True)
            else:
                ids, sc, st = idx.search_vamana_filtered(q, k, mask, kind=kind, stats=True)
            for i in range(nq):
                eid, esc, est = ov.search(q[i], k, mask=None if mask is None else mask[i])
                r_ = eid.size
                assert np.array_equal(ids[i, :r_], eid), (k, mask is not None, i, ids[i, :r_][:10], eid[:10])
                assert same_scores(sc[i, :r_], esc), (k, i)
                assert np.all(ids[i, r_:] == 0xFFFFFFFF)
                assert (int(st[i][0]), int(st[i][1]), int(st[i][3])) == (est.nodes_visited, est.distance_computations, est.pops), (k, i)


@pytest.mark.parametrize("scan", ["f32", "sq8", "pq"])
@pytest.mark.parametrize("metric", [0, 2])
@pytest.mark.parametrize("where", ["queries", "data"])
def test_partition_probed_and_filtered_scans(vg, ctx, scan, metric, where):
    """vg_search_flat_probed / vg_search_flat_filtered (flat/segment.go:727-749): the probed partitions' row ranges go into ONE
    heap, in FindClosestCentroids' order — with NaN scores the order matters (without it does not: the key order is total).  fp32,
    SQ8 and PQ scans; nprobes 1 / 3 / all; no filter, one for the batch, one per query; an unpartitioned segment with a filter."""
    from tests.test_gpu_probe import partitioned
    rng = np.random.default_rng(400 + metric + len(scan))
    n, dim, parts, nq, k = 6000, 32, 6, 14, 10
    x, cent, off = partitioned(rng, n, dim, parts, metric)
    x[50:54] = x[50]
    q = poisoned_queries(rng, x, nq) if where == "queries" else rng.standard_normal((nq, dim)).astype(np.float32)
    rows = x
    if where == "data" and scan == "f32":
        rows = x.copy()
        bad = rng.integers(0, n, 40)
        rows[bad, rng.integers(0, dim, 40)] = np.array([np.nan, np.inf, -np.inf, 3e38], np.float32)[rng.integers(0, 4, 40)]
        rows[off[2], 0] = np.nan                                  # the first row of a partition
    masks = rng.random((nq, n)) < 0.4

    def build(partitioned_):
        idx = vg.Index(ctx, n, dim, vg.Metric(metric))
        kw = dict(centroids=cent, part_offsets=off) if partitioned_ else {}
        if scan == "f32":
            idx.set_vectors(rows)
            seg = o.FlatSegment(rows, dim, metric=metric, **kw)
            code = idx.SCAN_F32
        elif scan == "sq8":
            sq = vg.ScalarQuantizer(ctx, dim)
            mins, maxs = x.min(0).copy(), x.max(0).copy() + 1e-3
            if where == "data":
                mins[3] = np.nan
            sq.set_bounds(mins, maxs)
            ref = o.ScalarQuantizer(dim)
            for dst, src in zip((ref.mins, ref.maxs, ref.scales, ref.inv_scales), sq.params()):
                dst[:] = src
            ref.trained = True
            codes = rng.integers(0, 256, (n, dim)).astype(np.uint8)
            idx.set_sq8_codes(sq, codes)
            seg = o.FlatSegment(x, dim, metric=metric, sq=ref, codes=codes, **kw)
            code = idx.SCAN_SQ8
        else:
            m = dim // 8
            opq = o.ProductQuantizer(dim, m, 256)
            offsets = ((rng.random(m) * 2 - 1) * 0.1).astype(np.float32)
            if where == "data":
                offsets[1] = np.inf
            opq.set_codebooks(rng.integers(-128, 128, m * 256 * 8).astype(np.int8), (rng.random(m) * 0.02 + 0.005).astype(np.float32), offsets)
            codes = rng.integers(0, 256, (n, m)).astype(np.uint8)
            pq = vg.ProductQuantizer(ctx, dim, m, 256)
            pq.set_codebooks(opq.codebooks, opq.scales, opq.offsets)
            idx.set_pq_codes(pq, codes)
            seg = o.FlatSegment(x, dim, metric=metric, pq=opq, codes=codes, **kw)
            code = idx.SCAN_PQ
        if partitioned_:
            idx.set_partitions(cent, off)
        return idx, seg, code

    idx, seg, code = build(True)
    for nprobes in (1, 3, parts):
        check(idx.search_flat_probed(q, k, nprobes, scan=code), lambda i: seg.search(q[i], k, nprobes), nq, k)
        check(idx.search_flat_filtered(q, k, masks, nprobes, scan=code), lambda i: seg.search(q[i], k, nprobes, mask=masks[i]), nq, k)
        check(idx.search_flat_filtered(q, k, masks[0], nprobes, scan=code), lambda i: seg.search(q[i], k, nprobes, mask=masks[0]), nq, k)
    idx, seg, code = build(False)
    check(idx.search_flat_filtered(q, k, masks, 0, scan=code), lambda i: seg.search(q[i], k, mask=masks[i]), nq, k)
    check(idx.search_flat_filtered(q[:3], 70, masks[1], 0, scan=code), lambda i: seg.search(q[i], 70, mask=masks[1]), 3, 70)


@pytest.mark.parametrize("metric", [0, 2])
def test_merge_topk_with_nan_scores(vg, ctx, metric):
    """vg_merge_topk / _packed = the engine's fan-in (engine/search.go:904-918): every segment's candidates, in the order they were
    popped from its heap (worst first = a best-first list read backwards), into one CandidateHeap with TryPushBounded(k), then
    popped.  With a NaN in a list the heap's layout decides; without one it is the key merge.  Ragged lists (padding), ties,
    global ids through the per-list offsets."""
    rng = np.random.default_rng(60 + metric)
    lists, nq, k = 5, 12, 16
    desc = metric != 0
    ids = np.stack([rng.permutation(1000)[:nq * k].reshape(nq, k) for _ in range(lists)]).astype(np.uint32)
    sc = rng.standard_normal((lists, nq, k)).astype(np.float32)
    sc = np.sort(sc, axis=2)
    if desc:
        sc = sc[:, :, ::-1].copy()
    sc[1, 2, 3] = np.nan; sc[0, 3, 0] = np.nan; sc[4, 3, k - 1] = np.nan; sc[2, 4, :] = np.nan
    # (the lists stay best-first where no NaN is involved: that is the entry point's contract)
    sc[3, 5, k - 1 if not desc else 0] = np.inf; sc[0, 6, 0 if not desc else k - 1] = -np.inf; sc[1, 7, 4] = sc[1, 7, 5]
    ids[2, 8, 10:] = 0xFFFFFFFF                                   # a short list
    ids[:, 9, :] = 0xFFFFFFFF                                     # nothing at all for query 9
    sc[0, 10, 0] = np.nan; ids[1:, 10, :] = 0xFFFFFFFF            # one list, one NaN
    off = (np.arange(lists) * 1000).astype(np.uint32)
    got_i, got_s = vg.merge_topk(ctx, ids, sc, k, metric=metric, id_offsets=off)
    for q in range(nq):
        h = o.CandidateHeap(desc)
        for l in range(lists):
            valid = int(np.sum(ids[l, q] != 0xFFFFFFFF))
            for i in range(valid - 1, -1, -1):
                h.push(np.float32(sc[l, q, i]), l, int(ids[l, q, i]) + int(off[l]), k=k)
        want = []
        while True:
            e = h.pop()
            if e is None:
                break
            want.append(e)
        h.close()
        want = want[::-1]
        r = len(want)
        assert list(got_i[q, :r]) == [e[2] for e in want], (q, got_i[q], [e[2] for e in want])
        assert same_scores(got_s[q, :r], np.array([e[0] for e in want], np.float32)), q
        assert np.all(got_i[q, r:] == 0xFFFFFFFF)
