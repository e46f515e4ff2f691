"""flat.Segment.Search with `filter segment.Filter` set (flat/segment.go:631-635, :559-561) on the GPU vs the oracle: rows
whose filter bit is clear never become candidates; the k best (score, row id) of the rest, over the probed partitions or the
whole segment — fp32, PQ and SQ8 scans, one filter for the batch or one per query, ids and scores bit-exact."""
import numpy as np
import pytest

from oracle import oracle as o
from tests import hooks
from tests.test_gpu_probe import bits, partitioned

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vg():
    import vecgo_amd
    return vecgo_amd


@pytest.fixture(scope="module")
def ctx(vg):
    return vg.Context(0)


def masks(rng, nq, n, keep):
    m = rng.random((nq, n)) < keep
    m[0, :] = rng.random(n) < keep
    return m


def check(ids, sc, seg, q, k, nprobes, mask):
    for i in range(q.shape[0]):
        mi = mask if mask.ndim == 1 else mask[i]
        eid, esc = seg.search(q[i], k, nprobes, mask=mi)
        r = eid.size
        assert np.array_equal(ids[i, :r], eid), (i, nprobes, ids[i], eid)
        assert np.array_equal(bits(sc[i, :r]), bits(esc)), (i, nprobes)
        assert np.all(ids[i, r:] == 0xFFFFFFFF)
        assert np.all(mi[ids[i, :r]])


def both_ways(search, q, many, seg, nprobes_list, rng, n):
    """one filter for the batch and one per query; few queries (one pass per pair) and many (grouped by partition)"""
    for keep in (0.5, 0.03):
        per_query = masks(rng, q.shape[0], n, keep)
        for nprobes, k in nprobes_list:
            ids, sc = search(q, k, per_query, nprobes)
            check(ids, sc, seg, q, k, nprobes, per_query)
            ids, sc = search(q, k, per_query[0], nprobes)
            check(ids, sc, seg, q, k, nprobes, per_query[0])
    shared = rng.random(n) < 0.3
    nprobes, k = nprobes_list[0]
    ids, sc = search(many, k, shared, nprobes)
    check(ids[:5], sc[:5], seg, many[:5], k, nprobes, shared)
    hooks.set_hook("VG_PROBE_NO_GROUP", "1")
    try:
        pid, psc = search(many, k, shared, nprobes)
    finally:
        hooks.set_hook("VG_PROBE_NO_GROUP", 0)
    assert np.array_equal(ids, pid) and np.array_equal(bits(sc), bits(psc))
    per_query = masks(rng, many.shape[0], n, 0.2)
    ids, sc = search(many, k, per_query, nprobes)
    sel = np.arange(0, many.shape[0], max(1, many.shape[0] // 6))
    check(ids[sel], sc[sel], seg, many[sel], k, nprobes, per_query[sel])


@pytest.mark.parametrize("n,dim,parts,metric", [(3000, 64, 7, 0), (2000, 100, 0, 2), (9000, 32, 0, 0), (900, 24, 12, 1),
                                                (70000, 16, 0, 0)])
def test_filtered_fp32_scan(vg, ctx, n, dim, parts, metric):
    rng = np.random.default_rng(n + dim + parts)
    if parts:
        x, cent, off = partitioned(rng, n, dim, parts, metric)
    else:
        x, cent, off = rng.standard_normal((n, dim)).astype(np.float32), None, None
    idx = vg.Index(ctx, n, dim, vg.Metric(metric))
    idx.set_vectors(x)
    if parts:
        idx.set_partitions(cent, off)
    seg = o.FlatSegment(x, dim, metric=metric, centroids=cent, part_offsets=off)
    q = rng.standard_normal((5, dim)).astype(np.float32)
    many = rng.standard_normal((90, dim)).astype(np.float32)

    def search(qq, k, mask, nprobes):
        return idx.search_flat_filtered(qq, k, mask, nprobes, scan=idx.SCAN_F32)

    both_ways(search, q, many, seg, ((2, 10), (max(parts, 1), 64), (1, 130)), rng, n)
    # every bit set = the unfiltered search; no bit set = nothing
    ids, sc = search(q, 10, np.ones(n, bool), 2)
    pid, psc = idx.search_flat_probed(q, 10, 2, scan=idx.SCAN_F32)
    assert np.array_equal(ids, pid) and np.array_equal(bits(sc), bits(psc))
    ids, sc = search(q, 10, np.zeros(n, bool), 2)
    assert np.all(ids == 0xFFFFFFFF)
    ids, sc = idx.search_flat_filtered(q, 10, None, 2, scan=idx.SCAN_F32)
    assert np.array_equal(ids, pid)


def test_filtered_ties_follow_row_id(vg, ctx):
    """Equal scores: CandidateHeap orders by (score, row id) (flat/segment.go:714-721) — duplicated rows, some filtered out"""
    rng = np.random.default_rng(5)
    n, dim = 6000, 32
    x = np.repeat(rng.standard_normal((n // 8, dim)).astype(np.float32), 8, axis=0)
    idx = vg.Index(ctx, n, dim)
    idx.set_vectors(x)
    seg = o.FlatSegment(x, dim)
    q = x[::700][:6] + 0.0
    mask = rng.random((6, n)) < 0.6
    ids, sc = idx.search_flat_filtered(q, 20, mask, 0, scan=idx.SCAN_F32)
    check(ids, sc, seg, q, 20, 0, mask)


@pytest.mark.parametrize("n,dim,m,parts,metric", [(4000, 64, 8, 9, 0), (9000, 96, 96, 0, 0), (3000, 40, 20, 0, 2)])
def test_filtered_pq_scan(vg, ctx, n, dim, m, parts, metric):
    rng = np.random.default_rng(n + m)
    if parts:
        x, cent, off = partitioned(rng, n, dim, parts, metric)
    else:
        x, cent, off = rng.standard_normal((n, dim)).astype(np.float32), None, None
    pq = vg.ProductQuantizer(ctx, dim, m, 256)
    pq.train(x, iters=3, seed=2)
    codes = pq.encode(x)
    cb, scales, offsets = pq.codebooks()
    opq = o.ProductQuantizer(dim, m, 256); opq.set_codebooks(cb, scales, offsets)
    idx = vg.Index(ctx, n, dim, vg.Metric(metric))
    idx.set_pq_codes(pq, codes)
    if parts:
        idx.set_partitions(cent, off)
    seg = o.FlatSegment(x, dim, metric=metric, pq=opq, codes=codes, centroids=cent, part_offsets=off)
    q = rng.standard_normal((5, dim)).astype(np.float32)
    many = rng.standard_normal((300, dim)).astype(np.float32)

    def search(qq, k, mask, nprobes):
        return idx.search_flat_filtered(qq, k, mask, nprobes, scan=idx.SCAN_PQ)

    both_ways(search, q, many, seg, ((2, 10), (max(parts, 1), 64), (1, 100)), rng, n)


@pytest.mark.parametrize("n,dim,parts,metric", [(3000, 64, 6, 0), (9000, 100, 0, 0), (1200, 17, 0, 2)])
def test_filtered_sq8_scan(vg, ctx, n, dim, parts, metric):
    rng = np.random.default_rng(n + dim)
    if parts:
        x, cent, off = partitioned(rng, n, dim, parts, metric)
    else:
        x, cent, off = rng.standard_normal((n, dim)).astype(np.float32), None, None
    sq = vg.ScalarQuantizer(ctx, dim); sq.train(x)
    ref = o.ScalarQuantizer(dim); ref.train(x)
    codes = sq.encode(x)
    idx = vg.Index(ctx, n, dim, vg.Metric(metric))
    idx.set_sq8_codes(sq, codes)
    if parts:
        idx.set_partitions(cent, off)
    seg = o.FlatSegment(x, dim, metric=metric, sq=ref, codes=codes, centroids=cent, part_offsets=off)
    q = rng.standard_normal((5, dim)).astype(np.float32)
    many = rng.standard_normal((120, dim)).astype(np.float32)

    def search(qq, k, mask, nprobes):
        return idx.search_flat_filtered(qq, k, mask, nprobes, scan=idx.SCAN_SQ8)

    both_ways(search, q, many, seg, ((2, 10), (max(parts, 1), 64), (1, 70)), rng, n)


def test_filtered_device_buffers_and_bad_masks(vg, ctx):
    import torch
    rng = np.random.default_rng(11)
    n, dim = 5000, 48
    x = rng.standard_normal((n, dim)).astype(np.float32)
    idx = vg.Index(ctx, n, dim)
    idx.set_vectors(x)
    q = rng.standard_normal((4, dim)).astype(np.float32)
    mask = rng.random(n) < 0.4
    hid, hsc = idx.search_flat_filtered(q, 10, mask, 0)
    did, dsc = idx.search_flat_filtered(torch.from_numpy(q).cuda(), 10, mask, 0)
    assert np.array_equal(hid, did.cpu().numpy()) and np.array_equal(bits(hsc), bits(dsc.cpu().numpy()))
    with pytest.raises(ValueError):
        idx.search_flat_filtered(q, 10, mask[:-9], 0)
    with pytest.raises(ValueError):
        idx.search_flat_filtered(q, 10, np.zeros((3, n), bool), 0)


@pytest.mark.parametrize("metric", [0, 2])
def test_filtered_batches_on_the_matrix_cores(vg, ctx, metric):
    """8 queries up, a filtered fp32 search of an unpartitioned segment is nominated by the masked GEMM (k_flat.hip): thin
    filters (fewer than k rows pass), k in the three proof regimes (<= 48, <= 64, pages beyond), the bf16 filter on top."""
    rng = np.random.default_rng(40 + metric)
    n, dim, nq = 20000, 64, 40
    x = rng.standard_normal((n, dim)).astype(np.float32)
    x[500:520] = x[500]                                     # ties inside and across the filter
    idx = vg.Index(ctx, n, dim, vg.Metric(metric))
    idx.set_vectors(x)
    seg = o.FlatSegment(x, dim, metric=metric)
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    q[3] = x[500]
    sel = np.array([0, 3, 7, 21, 39])
    for keep in (0.9, 0.2, 0.01, 0.0004):
        per_query = rng.random((nq, n)) < keep
        per_query[7, :] = False                             # a query whose filter rejects everything
        for k in (10, 60, 150):
            ids, sc = idx.search_flat_filtered(q, k, per_query, 0)
            check(ids[sel], sc[sel], seg, q[sel], k, 0, per_query[sel])
            ids, sc = idx.search_flat_filtered(q, k, per_query[0], 0)
            check(ids[sel], sc[sel], seg, q[sel], k, 0, per_query[0])
    idx.enable_bf16_filter(True)
    per_query = rng.random((nq, n)) < 0.3
    ids, sc = idx.search_flat_filtered(q, 10, per_query, 0)
    check(ids[sel], sc[sel], seg, q[sel], 10, 0, per_query[sel])
    # the unfiltered search is untouched by the mask plumbing
    ids, sc = idx.search_flat(q, 10)
    for i in sel:
        eid, esc = seg.search(q[i], 10)
        assert np.array_equal(ids[i], eid) and np.array_equal(bits(sc[i]), bits(esc))


def test_segment_files_with_a_filter(vg, ctx):
    """vg_segment_search_filtered: a flat image (partitions, SQ8 codes) and a DiskANN image, each through the filtered search of
    what the file holds."""
    from tests import graphs, segfile
    rng = np.random.default_rng(78)
    n, dim, parts = 2500, 48, 8
    x, cent, off = partitioned(rng, n, dim, parts)
    q = rng.standard_normal((4, dim)).astype(np.float32)
    mask = rng.random((4, n)) < 0.3
    seg = vg.Segment(ctx, segfile.write_flat(x, partitions=(cent, off)))
    ref = o.FlatSegment(x, dim, centroids=cent, part_offsets=off)
    for nprobes in (0, 2, parts):
        ids, sc = seg.search_filtered(q, 10, mask, nprobes)
        check(ids, sc, ref, q, 10, nprobes, mask)
    a = seg.search_filtered(q, 10, None, 2)
    b = seg.search(q, 10, 2)
    assert np.array_equal(a[0], b[0])
    seg.close()
    sq = o.ScalarQuantizer(dim); sq.train(x)
    codes = sq.encode_batch(x)
    seg = vg.Segment(ctx, segfile.write_flat(x, sq=(sq.mins, sq.maxs), codes=codes))
    ids, sc = seg.search_filtered(q, 10, mask[0], 0)
    check(ids, sc, o.FlatSegment(x, dim, sq=sq, codes=codes), q, 10, 0, mask[0])
    seg.close()
    g, entry = graphs.build_vamana(x, r=16, seed=2)
    seg = vg.Segment(ctx, segfile.write_diskann(x, g, entry), kind="diskann")
    ov = o.VamanaIndex(g, entry, dim, o.VAMANA_F32, base=x)
    ids, sc = seg.search_filtered(q, 10, mask)
    for qi in range(4):
        eid, esc, _ = ov.search(q[qi], 10, mask=mask[qi])
        assert np.array_equal(ids[qi, :eid.size], eid) and np.array_equal(bits(sc[qi, :eid.size]), bits(esc))
    seg.close()
