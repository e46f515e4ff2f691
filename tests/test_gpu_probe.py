"""flat.Segment.Search over IVF partitions (flat/segment.go:727-749) on the GPU vs the oracle: the
nprobes closest centroids' row ranges, scanned with the segment's scan type, ids and scores bit-exact."""
import os

import numpy as np
import pytest

from oracle import oracle as o
from tests import hooks
from tests import segfile

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


def partitioned(rng, n, dim, parts, metric=0, empty=()):
    """Rows grouped by their closest centroid, the way flat/writer.go lays a partitioned segment out;
    the partitions listed in `empty` get no rows (their rows go to the next one)."""
    x = rng.standard_normal((n, dim)).astype(np.float32)
    cent = (rng.standard_normal((parts, dim)) * 0.7).astype(np.float32)
    a = np.array([o.assign_partition(x[i], cent, dim, metric) for i in range(n)], np.int64)
    for e in empty:
        a[a == e] = (e + 1) % parts
    order = np.argsort(a, kind="stable")
    x, a = x[order], a[order]
    off = np.searchsorted(a, np.arange(parts + 1)).astype(np.uint32)
    return x, cent, off


def check(ids, sc, seg, q, k, nprobes):
    for i in range(q.shape[0]):
        eid, esc = seg.search(q[i], k, nprobes)
        r = eid.size
        assert np.array_equal(ids[i, :r], eid), (i, nprobes, ids[i], eid)
        assert np.array_equal(bits(sc[i, :r]), bits(esc)), (i, nprobes)
        assert np.all(ids[i, r:] == 0xFFFFFFFF)


@pytest.mark.parametrize("n,dim,parts,metric", [(3000, 64, 7, 0), (2000, 100, 12, 2), (5000, 768, 5, 0),
                                                (900, 24, 40, 1), (400, 8, 3, 0)])
def test_probed_fp32_scan(vg, ctx, n, dim, parts, metric):
    rng = np.random.default_rng(n + dim + parts)
    x, cent, off = partitioned(rng, n, dim, parts, metric, empty=(1,) if parts > 4 else ())
    idx = vg.Index(ctx, n, dim, vg.Metric(metric))
    idx.set_vectors(x)
    idx.set_partitions(cent, off)
    seg = o.FlatSegment(x, dim, metric=metric, centroids=cent, part_offsets=off)
    q = rng.standard_normal((9, dim)).astype(np.float32)
    for nprobes, k in ((0, 10), (1, 1), (3, 10), (parts, 64), (parts + 5, 17)):
        ids, sc = idx.search_flat_probed(q, k, nprobes, scan=idx.SCAN_F32)
        check(ids, sc, seg, q, k, nprobes)
    # a batch large enough for full groups of queries per partition; the grouped scan (rows read once
    # per group of 8 queries) and the pair-by-pair scan give the same lists
    many = rng.standard_normal((150, dim)).astype(np.float32)
    ids, sc = idx.search_flat_probed(many, 10, 2, scan=idx.SCAN_F32)
    check(ids[:6], sc[:6], seg, many[:6], 10, 2)
    hooks.set_hook("VG_PROBE_NO_GROUP", "1")
    try:
        pid, psc = idx.search_flat_probed(many, 10, 2, scan=idx.SCAN_F32)
    finally:
        hooks.set_hook("VG_PROBE_NO_GROUP", 0)
    assert np.array_equal(ids, pid) and np.array_equal(bits(sc), bits(psc))
    # every partition probed = the exhaustive search
    ids, sc = idx.search_flat_probed(q, 10, parts, scan=idx.SCAN_F32)
    fid, fsc = idx.search_flat(q, 10)
    assert np.array_equal(ids, fid) and np.array_equal(bits(sc), bits(fsc))
    idx.set_partitions(None, [])
    ids, sc = idx.search_flat_probed(q, 10, 1, scan=idx.SCAN_F32)  # no partitions: one range, the whole segment
    assert np.array_equal(ids, fid) and np.array_equal(bits(sc), bits(fsc))


@pytest.mark.parametrize("n,dim,m,parts", [(4000, 64, 8, 9), (3000, 96, 96, 6), (1500, 40, 20, 4)])
def test_probed_pq_scan(vg, ctx, n, dim, m, parts):
    rng = np.random.default_rng(n + m)
    x, cent, off = partitioned(rng, n, dim, parts)
    pq = vg.ProductQuantizer(ctx, dim, m, 256)
    pq.train(x, iters=4, seed=2)
    codes = pq.encode(x)
    cb, scales, offsets = pq.codebooks()
    opq = o.ProductQuantizer(dim, m, 256); opq.set_codebooks(cb, scales, offsets)
    idx = vg.Index(ctx, n, dim)
    idx.set_pq_codes(pq, codes)
    idx.set_partitions(cent, off)
    seg = o.FlatSegment(x, dim, pq=opq, codes=codes, centroids=cent, part_offsets=off)
    q = rng.standard_normal((6, dim)).astype(np.float32)
    for nprobes, k in ((1, 10), (2, 64), (parts, 10)):
        ids, sc = idx.search_flat_probed(q, k, nprobes, scan=idx.SCAN_PQ)
        check(ids, sc, seg, q, k, nprobes)
    many = rng.standard_normal((300, dim)).astype(np.float32)  # more queries than compute units: split = 1
    ids, sc = idx.search_flat_probed(many, 10, 3, scan=idx.SCAN_PQ)
    check(ids[:8], sc[:8], seg, many[:8], 10, 3)


@pytest.mark.parametrize("metric", [2, 1])
def test_probed_pq_scan_dot_and_cosine_keep_the_largest(vg, ctx, metric):
    """flat/segment.go:449: the result heap's direction follows the segment metric for every scan type, while
    AdcDistance is always a squared L2 — a Dot / Cosine PQ segment keeps its k LARGEST table-lookup distances.
    Matched as written (the oracle restates it), not 'fixed'."""
    rng = np.random.default_rng(70 + metric)
    n, dim, m, parts = 3000, 64, 8, 5
    x, cent, off = partitioned(rng, n, dim, parts)
    pq = vg.ProductQuantizer(ctx, dim, m, 256)
    pq.train(x, iters=4, seed=2)
    codes = pq.encode(x)
    cb, scales, offsets = pq.codebooks()
    opq = o.ProductQuantizer(dim, m, 256); opq.set_codebooks(cb, scales, offsets)
    idx = vg.Index(ctx, n, dim, vg.Metric(metric))
    idx.set_pq_codes(pq, codes)
    idx.set_partitions(cent, off)
    seg = o.FlatSegment(x, dim, metric=metric, pq=opq, codes=codes, centroids=cent, part_offsets=off)
    q = rng.standard_normal((5, dim)).astype(np.float32)
    for nprobes, k in ((1, 10), (parts, 70)):
        ids, sc = idx.search_flat_probed(q, k, nprobes, scan=idx.SCAN_PQ)
        check(ids, sc, seg, q, k, nprobes)
        assert np.all(np.diff(sc[:, :min(k, 10)], axis=1) <= 0)      # largest first


@pytest.mark.parametrize("n,dim,parts,metric", [(3000, 64, 6, 0), (2500, 100, 10, 0), (1200, 17, 4, 2)])
def test_probed_sq8_scan(vg, ctx, n, dim, parts, metric):
    rng = np.random.default_rng(n + dim)
    x, cent, off = partitioned(rng, n, dim, parts, metric)
    sq = vg.ScalarQuantizer(ctx, dim); sq.train(x)
    ref = o.ScalarQuantizer(dim); ref.train(x)
    codes = sq.encode(x)
    idx = vg.Index(ctx, n, dim, vg.Metric(metric))
    idx.set_sq8_codes(sq, codes)
    idx.set_partitions(cent, off)
    seg = o.FlatSegment(x, dim, metric=metric, sq=ref, codes=codes, centroids=cent, part_offsets=off)
    q = rng.standard_normal((5, dim)).astype(np.float32)
    for nprobes, k in ((1, 10), (3, 33), (parts, 64)):
        ids, sc = idx.search_flat_probed(q, k, nprobes, scan=idx.SCAN_SQ8)
        check(ids, sc, seg, q, k, nprobes)
    # a batch with full groups of queries per partition: codes decoded once per group of 8 queries
    many = rng.standard_normal((120, dim)).astype(np.float32)
    ids, sc = idx.search_flat_probed(many, 10, 2, scan=idx.SCAN_SQ8)
    check(ids[:6], sc[:6], seg, many[:6], 10, 2)
    hooks.set_hook("VG_PROBE_NO_GROUP", "1")
    try:
        pid, psc = idx.search_flat_probed(many, 10, 2, scan=idx.SCAN_SQ8)
    finally:
        hooks.set_hook("VG_PROBE_NO_GROUP", 0)
    assert np.array_equal(ids, pid) and np.array_equal(bits(sc), bits(psc))


def test_partitioned_segment_file(vg, ctx):
    """A flat segment image with partitions: Segment.search = flat.Segment.Search with NProbes."""
    rng = np.random.default_rng(77)
    n, dim, parts = 2500, 48, 8
    x, cent, off = partitioned(rng, n, dim, parts)
    q = rng.standard_normal((4, dim)).astype(np.float32)
    seg = vg.Segment(ctx, segfile.write_flat(x, partitions=(cent, off)))
    assert seg.info.num_partitions == parts
    ref = o.FlatSegment(x, dim, centroids=cent, part_offsets=off)
    for nprobes in (0, 2, parts):
        ids, sc = seg.search(q, 10, nprobes)
        check(ids, sc, ref, q, 10, nprobes)
    seg.close()
    sq = o.ScalarQuantizer(dim); sq.train(x)
    codes = sq.encode_batch(x)
    seg = vg.Segment(ctx, segfile.write_flat(x, sq=(sq.mins, sq.maxs), codes=codes, partitions=(cent, off)))
    ref = o.FlatSegment(x, dim, sq=sq, codes=codes, centroids=cent, part_offsets=off)
    ids, sc = seg.search(q, 10, 3)
    check(ids, sc, ref, q, 10, 3)
    seg.close()
    plain = vg.Segment(ctx, segfile.write_flat(x))              # no partitions: the whole segment
    ids, sc = plain.search(q, 10, 5)
    check(ids, sc, o.FlatSegment(x, dim), q, 10, 5)
    plain.close()


def test_probe_errors(vg, ctx):
    idx = vg.Index(ctx, 100, 8)
    idx.set_vectors(np.zeros((100, 8), np.float32))
    with pytest.raises(vg.VecgoHipError):
        idx.set_partitions(np.zeros((2, 8), np.float32), [0, 60, 50])   # offsets decrease
    with pytest.raises(vg.VecgoHipError):
        idx.set_partitions(np.zeros((2, 8), np.float32), [0, 60, 101])  # past the last row
    idx.set_partitions(np.zeros((2, 8), np.float32), [0, 60, 100])
    with pytest.raises(vg.VecgoHipError):
        idx.search_flat_probed(np.zeros((1, 8), np.float32), 513, 1)    # k <= 512
    with pytest.raises(vg.VecgoHipError):
        idx.search_flat_probed(np.zeros((1, 8), np.float32), 5, 1, scan=idx.SCAN_PQ)  # no PQ codes


@pytest.mark.parametrize("scan_name", ["f32", "sq8", "pq"])
def test_probed_scans_page_beyond_64_results(vg, ctx, scan_name):
    """k > 64 on the probed path: pages of 64 results, each page a scan of the same probed ranges for the keys
    after the previous page's last one — pair-by-pair and grouped kernels, every scan type."""
    rng = np.random.default_rng(99)
    n, dim, parts = 4000, 64, 5
    x, cent, off = partitioned(rng, n, dim, parts)
    x[off[2] + 5:off[2] + 80] = x[off[2] + 1]      # a run of equal scores longer than a page, inside one partition
    idx = vg.Index(ctx, n, dim)
    kw = {}
    if scan_name == "f32":
        idx.set_vectors(x); scan = idx.SCAN_F32
    elif scan_name == "sq8":
        sq = vg.ScalarQuantizer(ctx, dim); sq.train(x)
        ref = o.ScalarQuantizer(dim); ref.train(x)
        codes = sq.encode(x)
        idx.set_sq8_codes(sq, codes); scan = idx.SCAN_SQ8
        kw = dict(sq=ref, codes=codes)
    else:
        pq = vg.ProductQuantizer(ctx, dim, 8, 256); pq.train(x, iters=3, seed=4)
        codes = pq.encode(x)
        cb, scales, offsets = pq.codebooks()
        opq = o.ProductQuantizer(dim, 8, 256); opq.set_codebooks(cb, scales, offsets)
        idx.set_pq_codes(pq, codes); scan = idx.SCAN_PQ
        kw = dict(pq=opq, codes=codes)
    idx.set_partitions(cent, off)
    seg = o.FlatSegment(x, dim, centroids=cent, part_offsets=off, **kw)
    few = rng.standard_normal((3, dim)).astype(np.float32)       # 3 x 2 pairs: the pair-by-pair kernels
    few[0] = x[off[2] + 1]
    many = rng.standard_normal((40, dim)).astype(np.float32)     # 40 x 3 pairs: the grouped kernels
    many[0] = x[off[2] + 1]
    for q, nprobes, k in ((few, 2, 100), (many, 3, 200), (few, parts, 512)):
        ids, sc = idx.search_flat_probed(q, k, nprobes, scan=scan)
        check(ids[:5], sc[:5], seg, q[:5], k, nprobes)


def test_probed_fp32_batches_with_the_bf16_filter(vg, ctx):
    """vg_index_enable_bf16_filter on a partitioned segment: the grouped nomination runs on the bfloat16 copy of the rows, the exact
    re-score and the widened proof keep ids and scores — equal to the search without the filter and to the oracle"""
    rng = np.random.default_rng(91)
    n, dim, parts, nq = 14000, 64, 6, 150
    x, cent, off = partitioned(rng, n, dim, parts)
    x[300:306] = x[300]
    idx = vg.Index(ctx, n, dim)
    idx.set_vectors(x)
    idx.set_partitions(cent, off)
    seg = o.FlatSegment(x, dim, centroids=cent, part_offsets=off)
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    q[5] = x[300]
    plain = [idx.search_flat_probed(q, 10, np_, scan=idx.SCAN_F32) for np_ in (2, parts)]
    idx.enable_bf16_filter(True)
    filt = [idx.search_flat_probed(q, 10, np_, scan=idx.SCAN_F32) for np_ in (2, parts)]
    for a, b in zip(plain, filt):
        assert np.array_equal(a[0], b[0]) and np.array_equal(bits(a[1]), bits(b[1]))
    check(filt[0][0][:8], filt[0][1][:8], seg, q[:8], 10, 2)


@pytest.mark.parametrize("metric", [0, 2])
def test_probed_fp32_batches_with_k_beyond_the_64_candidate_budget(vg, ctx, metric):
    """the grouped nomination for 48 < k <= 160: a deeper threshold per (query, probe) pair, EVERY row below it re-scored
    (flat_verify_all / _sort), flagged queries searched again as a subset — ids and score bits of the oracle, duplicates (ties at
    the k-th place), a filter per query, and partitions smaller than the threshold's sample included"""
    rng = np.random.default_rng(130 + metric)
    n, dim, parts, nq = 16000, 64, 6, 160
    x, cent, off = partitioned(rng, n, dim, parts, metric)
    x[300:420] = x[300]                                # 120 equal rows: a tie across the k-th place for one query
    idx = vg.Index(ctx, n, dim, vg.Metric(metric))
    idx.set_vectors(x)
    idx.set_partitions(cent, off)
    seg = o.FlatSegment(x, dim, metric=metric, centroids=cent, part_offsets=off)
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    q[5] = x[300]
    masks = rng.random((nq, n)) < 0.5
    for k in (49, 64, 100, 160):
        for nprobes in (2, parts):
            ids, sc = idx.search_flat_probed(q, k, nprobes, scan=idx.SCAN_F32)
            check(ids[:8], sc[:8], seg, q[:8], k, nprobes)
            check(ids[-2:], sc[-2:], seg, q[-2:], k, nprobes)
        ids, sc = idx.search_flat_filtered(q, k, masks, 2, scan=idx.SCAN_F32)
        for i in (0, 5, nq - 1):
            eid, esc = seg.search(q[i], k, 2, mask=masks[i])
            assert np.array_equal(ids[i, :eid.size], eid) and np.array_equal(bits(sc[i, :eid.size]), bits(esc))
    # small partitions: the sample holds fewer rows than thresholds are kept, every row passes
    n2, parts2 = 3000, 5
    x2, cent2, off2 = partitioned(rng, n2, dim, parts2, metric)
    idx2 = vg.Index(ctx, n2, dim, vg.Metric(metric))
    idx2.set_vectors(x2)
    idx2.set_partitions(cent2, off2)
    seg2 = o.FlatSegment(x2, dim, metric=metric, centroids=cent2, part_offsets=off2)
    ids, sc = idx2.search_flat_probed(q, 100, 3, scan=idx2.SCAN_F32)
    check(ids[:6], sc[:6], seg2, q[:6], 100, 3)


@pytest.mark.parametrize("metric", [0, 2])
@pytest.mark.parametrize("scan_name", ["f32", "sq8"])
def test_probes_with_equal_centroid_distances_follow_the_selection_loop(vg, ctx, metric, scan_name):
    """duplicated centroids: FindClosestCentroids' selection loop (kmeans.go:255-269, nprobes <= parts/4) takes the first minimum
    by POSITION after its swaps — not by centroid id — and the probed partitions are exactly the oracle's; the full-sort shape
    (nprobes > parts/4) beside it"""
    rng = np.random.default_rng(400 + metric)
    n, dim, parts = 2400, 32, 12
    x = rng.standard_normal((n, dim)).astype(np.float32)
    cent = (rng.standard_normal((parts, dim)) * 0.7).astype(np.float32)
    cent[5] = cent[1]; cent[9] = cent[1]; cent[7] = cent[2]; cent[10] = cent[0]; cent[11] = cent[0]
    off = (np.arange(parts + 1) * (n // parts)).astype(np.uint32)       # rows in every partition, duplicates included
    idx = vg.Index(ctx, n, dim, vg.Metric(metric))
    kw = {}
    if scan_name == "sq8":
        sq = vg.ScalarQuantizer(ctx, dim); sq.train(x)
        ref = o.ScalarQuantizer(dim); ref.train(x)
        codes = sq.encode(x)
        idx.set_sq8_codes(sq, codes)
        kw = dict(sq=ref, codes=codes)
        scan = idx.SCAN_SQ8
    else:
        idx.set_vectors(x)
        scan = idx.SCAN_F32
    idx.set_partitions(cent, off)
    seg = o.FlatSegment(x, dim, metric=metric, centroids=cent, part_offsets=off, **kw)
    for nq in (3, 60):
        q = rng.standard_normal((nq, dim)).astype(np.float32)
        q[0] = cent[1]; q[1] = cent[0] * 1.5; q[2] = cent[2]
        for nprobes in (1, 2, 3, 4):
            ids, sc = idx.search_flat_probed(q, 10, nprobes, scan=scan)
            check(ids, sc, seg, q, 10, nprobes)


def test_probes_with_nan_centroid_distances_follow_the_selection_loop(vg, ctx):
    """A centroid with a NaN coordinate has a NaN distance to every query: `dists[j].dist < dists[minIdx].dist` (kmeans.go:261) is
    false either way, so the selection loop TAKES a NaN entry when it stands at position i and never moves to one behind it —
    the probed partitions (and with them ids and scores) equal the oracle's, which runs the loop as written."""
    rng = np.random.default_rng(515)
    n, dim, parts = 2400, 32, 12
    x = rng.standard_normal((n, dim)).astype(np.float32)
    off = (np.arange(parts + 1) * (n // parts)).astype(np.uint32)
    q = rng.standard_normal((20, dim)).astype(np.float32)
    for nan_at in ((0,), (1,), (0, 2), (11,)):
        cent = (rng.standard_normal((parts, dim)) * 0.7).astype(np.float32)
        for c in nan_at:
            cent[c, 3] = np.nan
        idx = vg.Index(ctx, n, dim)
        idx.set_vectors(x)
        idx.set_partitions(cent, off)
        seg = o.FlatSegment(x, dim, metric=0, centroids=cent, part_offsets=off)
        for nprobes in (1, 2, 3):
            ids, sc = idx.search_flat_probed(q, 10, nprobes, scan=idx.SCAN_F32)
            check(ids, sc, seg, q, 10, nprobes)


@pytest.mark.parametrize("scan_name", ["f32", "sq8"])
def test_probed_batches_beyond_65535_pairs_run_in_chunks(vg, ctx, scan_name):
    """8200 queries x 8 probes: more (query, probe) pairs than one grouped nomination takes — the batch is cut into chunks of
    queries that fit; the queries either side of the cut and a filter per query included"""
    rng = np.random.default_rng(77)
    n, dim, parts, nq, k = 4000, 64, 12, 8200, 10
    x, cent, off = partitioned(rng, n, dim, parts)
    idx = vg.Index(ctx, n, dim)
    kw = {}
    if scan_name == "sq8":
        sq = vg.ScalarQuantizer(ctx, dim); sq.train(x)
        ref = o.ScalarQuantizer(dim); ref.train(x)
        codes = sq.encode(x)
        idx.set_sq8_codes(sq, codes)
        idx.enable_sq8_nomination(True)
        kw = dict(sq=ref, codes=codes)
        scan = idx.SCAN_SQ8
    else:
        idx.set_vectors(x)
        scan = idx.SCAN_F32
    idx.set_partitions(cent, off)
    seg = o.FlatSegment(x, dim, centroids=cent, part_offsets=off, **kw)
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    ids, sc = idx.search_flat_probed(q, k, 8, scan=scan)
    cut = 65535 // 8
    rows = [0, cut - 1, cut, cut + 1, nq - 1]
    check(ids[rows], sc[rows], seg, q[rows], k, 8)
    masks = np.packbits(rng.random((nq, n)) < 0.5, axis=1, bitorder="little")
    ids, sc = idx.search_flat_filtered(q, k, masks, 8, scan=scan)
    for i in rows:
        eid, esc = seg.search(q[i], k, 8, mask=np.unpackbits(masks[i], bitorder="little")[:n].astype(bool))
        assert np.array_equal(ids[i, :eid.size], eid) and np.array_equal(bits(sc[i, :eid.size]), bits(esc)), i
