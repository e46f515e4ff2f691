"""Oracle checks for the partitioned flat.Segment.Search restatement (flat/segment.go:447-751) and
ScalarQuantizer.DotProduct (quantizer.go:109-119).  No GPU."""
import numpy as np

from oracle import oracle as o


def bits(x):
    return np.asarray(x, np.float32).view(np.uint32)


def grouped(rng, n, dim, parts):
    x = rng.standard_normal((n, dim)).astype(np.float32)
    cent = rng.standard_normal((parts, dim)).astype(np.float32)
    a = np.array([o.assign_partition(x[i], cent, dim) for i in range(n)])
    order = np.argsort(a, kind="stable")
    x, a = x[order], a[order]
    return x, cent, np.searchsorted(a, np.arange(parts + 1)).astype(np.uint32)


def test_unpartitioned_segment_equals_the_plain_scans():
    rng = np.random.default_rng(1)
    n, dim, k = 400, 32, 10
    x = rng.standard_normal((n, dim)).astype(np.float32)
    q = rng.standard_normal(dim).astype(np.float32)
    for metric in (0, 2):
        a = o.FlatSegment(x, dim, metric=metric).search(q, k, nprobes=3)
        b = o.flat_search_f32(x, dim, q, k, metric)
        assert np.array_equal(a[0], b[0]) and np.array_equal(bits(a[1]), bits(b[1]))
    sq = o.ScalarQuantizer(dim); sq.train(x)
    codes = sq.encode_batch(x)
    a = o.FlatSegment(x, dim, sq=sq, codes=codes).search(q, k)
    b = o.flat_search_sq8(sq, codes, q, k)
    assert np.array_equal(a[0], b[0]) and np.array_equal(bits(a[1]), bits(b[1]))
    pq = o.ProductQuantizer(dim, 4, 256); pq.train(x, iters=3, seed=1)
    pc = np.stack([pq.encode(r) for r in x])
    a = o.FlatSegment(x, dim, pq=pq, codes=pc).search(q, k)
    b = o.flat_search_pq(pq, pc, q, k)
    assert np.array_equal(a[0], b[0]) and np.array_equal(bits(a[1]), bits(b[1]))


def test_probing_scans_exactly_the_closest_partitions():
    rng = np.random.default_rng(2)
    n, dim, parts, k = 900, 24, 6, 8
    x, cent, off = grouped(rng, n, dim, parts)
    seg = o.FlatSegment(x, dim, centroids=cent, part_offsets=off)
    full = o.FlatSegment(x, dim)
    for t in range(5):
        q = rng.standard_normal(dim).astype(np.float32)
        ids_all, sc_all = seg.search(q, k, nprobes=parts)            # every partition = exhaustive
        fid, fsc = full.search(q, k)
        assert np.array_equal(ids_all, fid) and np.array_equal(bits(sc_all), bits(fsc))
        for nprobes in (0, 1, 3):
            want = max(nprobes, 1)                                   # segment.go:728-731
            probed = o.find_closest_centroids(q, cent, dim, want)
            rows = np.concatenate([np.arange(off[p], off[p + 1]) for p in probed])
            ids, sc = seg.search(q, k, nprobes)
            assert set(ids) <= set(rows)
            sub_ids, sub_sc = o.flat_search_f32(x[rows], dim, q, k)  # brute force over just those rows
            assert np.array_equal(rows[sub_ids], ids) and np.array_equal(bits(sub_sc), bits(sc))


def test_sq8_dot_product_is_the_sequential_go_loop():
    rng = np.random.default_rng(3)
    for dim in (1, 7, 16, 100):
        q = rng.standard_normal(dim).astype(np.float32)
        mins = rng.standard_normal(dim).astype(np.float32)
        inv = (rng.random(dim).astype(np.float32) * np.float32(0.02)).astype(np.float32)
        code = rng.integers(0, 256, dim).astype(np.uint8)
        dot = np.float32(0.0)
        for i in range(dim):  # quantizer.go:114-117, every operation rounded to float32
            val = np.float32(mins[i] + np.float32(np.float32(code[i]) * inv[i]))
            dot = np.float32(dot + np.float32(q[i] * val))
        got = o.lib.vgo_sq8_dot(q.ctypes.data_as(o._f32p), code.ctypes.data_as(o._u8p), dim,
                                mins.ctypes.data_as(o._f32p), inv.ctypes.data_as(o._f32p))
        assert bits(got) == bits(dot)


def test_sq8_segments_follow_the_metric():
    """L2: sq8u batch order, ascending; Dot: DotProduct, descending (segment.go:449,659-667)."""
    rng = np.random.default_rng(4)
    n, dim, k = 300, 20, 5
    x = rng.standard_normal((n, dim)).astype(np.float32)
    sq = o.ScalarQuantizer(dim); sq.train(x)
    codes = sq.encode_batch(x)
    q = rng.standard_normal(dim).astype(np.float32)
    ids, sc = o.FlatSegment(x, dim, metric=2, sq=sq, codes=codes).search(q, k)
    dots = np.array([o.lib.vgo_sq8_dot(q.ctypes.data_as(o._f32p), codes[i].ctypes.data_as(o._u8p), dim,
                                       sq.mins.ctypes.data_as(o._f32p), sq.inv_scales.ctypes.data_as(o._f32p))
                     for i in range(n)], np.float32)
    order = np.lexsort((np.arange(n), -dots))[:k]
    assert np.array_equal(ids, order.astype(np.uint32)) and np.array_equal(bits(sc), bits(dots[order]))


def test_filtered_search_is_the_search_of_the_matching_rows():
    """segment.go:631-635: a row whose filter.Matches is false is skipped — the result is the unfiltered search of a segment
    holding only the matching rows, ids mapped back (the (score, row id) order keeps ascending ids among ties)."""
    rng = np.random.default_rng(3)
    n, dim, parts, k = 900, 24, 6, 12
    x, cent, off = grouped(rng, n, dim, parts)
    x[100:140] = x[100]                                              # ties across the filter
    seg = o.FlatSegment(x, dim, centroids=cent, part_offsets=off)
    full = o.FlatSegment(x, dim)
    for t in range(4):
        q = rng.standard_normal(dim).astype(np.float32) if t else x[100].copy()
        mask = rng.random(n) < (0.5, 0.5, 0.05, 0.0)[t]
        rows = np.flatnonzero(mask)
        ids, sc = full.search(q, k, mask=mask)
        if rows.size == 0:
            assert ids.size == 0
            continue
        sid, ssc = o.flat_search_f32(x[rows], dim, q, k)
        assert np.array_equal(rows[sid], ids) and np.array_equal(bits(ssc), bits(sc))
        probed = o.find_closest_centroids(q, cent, dim, 2)
        in_probed = np.zeros(n, bool)
        for p in probed:
            in_probed[off[p]:off[p + 1]] = True
        rows = np.flatnonzero(mask & in_probed)
        ids, sc = seg.search(q, k, 2, mask=mask)
        sid, ssc = o.flat_search_f32(x[rows], dim, q, k)
        assert np.array_equal(rows[sid], ids) and np.array_equal(bits(ssc), bits(sc))
    all_ids, all_sc = full.search(q, k, mask=np.ones(n, bool))
    nid, nsc = full.search(q, k)
    assert np.array_equal(all_ids, nid) and np.array_equal(bits(all_sc), bits(nsc))


def test_selection_loop_breaks_equal_distances_by_position_not_by_id():
    """kmeans.go:255-269 (n <= k/4 && n < 16): each step takes the FIRST minimum by position, then swaps it with position i — which
    moves the entry that stood there behind others of its own distance.  Worked by hand from the Go loop:
    distances [5, 5, 1, 9 x 9] (12 centroids), n = 2: step 0 picks position 2 and swaps -> [1, 5(id 1), 5(id 0), ...]; step 1 scans from
    position 1 and keeps the first 5 it meets: id 1, not id 0.  The full sort (n > k/4) is by (distance, id) in the oracle."""
    dim = 1
    # L2 distance of a query at 0 to a centroid at sqrt(d) is d
    d = np.array([5, 5, 1] + [9] * 9, np.float32)
    cent = np.sqrt(d).reshape(-1, dim).astype(np.float32)
    q = np.zeros(dim, np.float32)
    assert list(o.find_closest_centroids(q, cent, dim, 2)) == [2, 1]
    # position 0 holds the minimum already: nothing moves, the tie goes to the lower position
    d2 = np.array([1, 5, 5] + [9] * 9, np.float32)
    assert list(o.find_closest_centroids(q, np.sqrt(d2).reshape(-1, dim).astype(np.float32), dim, 2)) == [0, 1]
    # three steps: [5, 3, 5, 1, 9...] -> pick 3 (swap with 0: [1,3,5,5(id 0)]), pick 1, then the first 5 by position: id 2
    d3 = np.array([5, 3, 5, 1] + [9] * 8, np.float32)
    assert list(o.find_closest_centroids(q, np.sqrt(d3).reshape(-1, dim).astype(np.float32), dim, 3)) == [3, 1, 2]
    # the same distances through the full sort (n = 4 > 12 / 4): by (distance, id)
    assert list(o.find_closest_centroids(q, np.sqrt(d3).reshape(-1, dim).astype(np.float32), dim, 4)) == [3, 1, 0, 2]
