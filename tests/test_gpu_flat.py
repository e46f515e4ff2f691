"""Exact brute-force search on the GPU (MFMA GEMM candidates + exact re-score + proof) vs
the oracle's flat scan (flat/segment.go:691-721): ids and scores bit-exact."""
import os

import numpy as np
import pytest

from oracle import oracle as o
from tests import hooks

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


def check(vg, ctx, n, dim, nq, k, metric, rng, base=None):
    if base is None:
        base = rng.standard_normal((n, dim)).astype(np.float32)
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    idx = vg.Index(ctx, n, dim, vg.Metric(metric))
    idx.set_vectors(base)
    ids, scores = idx.search_flat(q, k)
    for qi in range(nq):
        eid, esc = o.flat_search_f32(base, dim, q[qi], k, metric)
        r = eid.size
        assert np.array_equal(ids[qi, :r], eid), (qi, ids[qi], eid)
        assert np.array_equal(bits(scores[qi, :r]), bits(esc))
        assert np.all(ids[qi, r:] == 0xFFFFFFFF)
    return idx, base, q


@pytest.mark.parametrize("n,dim,nq,k,metric", [
    (10000, 128, 20, 10, 0),    # BASELINE config 1: flat exact L2 10k x 128
    (20000, 768, 9, 10, 0),
    (5000, 100, 5, 10, 0),      # dim not a multiple of 32: ragged K edge of the GEMM
    (3000, 777, 3, 5, 0),       # unaligned rows
    (300, 64, 130, 32, 0),      # more than one query tile, k = 32
    (40, 64, 3, 10, 0),         # fewer rows than candidates
    (7, 32, 2, 10, 0),          # fewer rows than k
    (10000, 128, 20, 10, 2),    # Dot: descending
    (4000, 768, 4, 10, 1),      # Cosine -> Dot provider (distance.go:91-106)
])
def test_flat_matches_oracle(vg, ctx, n, dim, nq, k, metric):
    check(vg, ctx, n, dim, nq, k, metric, np.random.default_rng(n + dim + nq))


@pytest.mark.parametrize("dim,metric,nodma", [(768, 0, False), (128, 0, False), (100, 0, False), (36, 2, False),
                                              (768, 0, True), (100, 2, True), (777, 0, False)])
def test_flat_fast_path_answers_random_data(vg, ctx, dim, metric, nodma):
    """On random data the GEMM candidates + proof must answer (nearly) every query: a wrong GEMM
    (bad LDS image, bad fragment pairing) would still be *correct* through the exhaustive fallback,
    so this checks the counter.  Covers the LDS-DMA kernel (dim % 4 == 0, incl. a ragged K edge)
    and the register-staged one (dim % 4 != 0, or forced)."""
    if nodma:
        hooks.set_hook("VG_FLAT_NO_DMA", "1")
    try:
        idx, _, _ = check(vg, ctx, 6000, dim, 140, 10, metric, np.random.default_rng(dim + metric))
        searched, exhaustive = idx.flat_stats()
        assert searched == 140
        assert exhaustive <= 2, exhaustive
    finally:
        hooks.set_hook("VG_FLAT_NO_DMA", 0)


@pytest.mark.parametrize("n,dim,nq,k,metric", [(30000, 768, 1, 10, 0), (9000, 128, 8, 32, 0), (5000, 1024, 3, 10, 2),
                                                 (3000, 100, 7, 10, 0), (100, 64, 5, 10, 0), (20000, 256, 2, 10, 1),
                                                 (20000, 768, 33, 10, 0), (7000, 100, 64, 10, 2), (6000, 36, 50, 10, 0),
                                                 (20000, 128, 32, 10, 0)])
def test_flat_small_batch_paths_match_oracle(vg, ctx, n, dim, nq, k, metric):
    """Small batches: nq <= 4 is answered by the HBM-bound multi-query exact scan (no GEMM, no proof),
    5..64 by the 32- / 64-query GEMM tiles; the same inputs through the other GEMM path (VG_FLAT_NO_SCAN=1)
    give the same bits."""
    rng = np.random.default_rng(n + dim + nq)
    idx, base, q = check(vg, ctx, n, dim, nq, k, metric, rng)
    assert idx.flat_stats()[0] == nq and idx.flat_stats()[1] <= 1
    ids, sc = idx.search_flat(q, k)
    hooks.set_hook("VG_FLAT_NO_SCAN", "1")
    hooks.set_hook("VG_FLAT_NO_SMALL_TILE", "1")
    try:
        ids2, sc2 = idx.search_flat(q, k)   # the 128-query tile
    finally:
        hooks.set_hook("VG_FLAT_NO_SCAN", 0)
        hooks.set_hook("VG_FLAT_NO_SMALL_TILE", 0)
    assert np.array_equal(ids, ids2) and np.array_equal(bits(sc), bits(sc2))


def test_flat_more_queries_than_one_chunk(vg, ctx):
    """nq > 4096: the host loop walks the batch in chunks of whole query tiles; every 37th query is
    checked against the oracle, all of them against a second call in a different chunking (two
    halves)."""
    rng = np.random.default_rng(77)
    n, dim, nq, k = 3000, 32, 4300, 5
    base = rng.standard_normal((n, dim)).astype(np.float32)
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    idx = vg.Index(ctx, n, dim); idx.set_vectors(base)
    ids, sc = idx.search_flat(q, k)
    for qi in range(0, nq, 37):
        eid, esc = o.flat_search_f32(base, dim, q[qi], k)
        assert np.array_equal(ids[qi], eid) and np.array_equal(bits(sc[qi]), bits(esc)), qi
    a = idx.search_flat(q[:2150], k); b = idx.search_flat(q[2150:], k)
    assert np.array_equal(np.vstack([a[0], b[0]]), ids) and np.array_equal(bits(np.vstack([a[1], b[1]])), bits(sc))
    assert idx.flat_stats()[0] == 2 * nq


def test_flat_duplicates_and_near_ties(vg, ctx):
    """Rows that differ in the last bits and exact duplicates: the proof step must either
    accept or fall back, and ties resolve by RowID."""
    rng = np.random.default_rng(2)
    base = rng.standard_normal((2000, 128)).astype(np.float32)
    base[1000:1010] = base[5]           # exact duplicates of one row
    base[1500] = base[7] * np.float32(1.0000001)
    check(vg, ctx, 2000, 128, 6, 12, 0, rng, base=base)


def test_flat_forced_exhaustive_path(vg, ctx):
    """The fallback kernel (step 4) alone must give the same answer."""
    hooks.set_hook("VG_FLAT_FORCE_EXACT", "1")
    try:
        idx, _, _ = check(vg, ctx, 6000, 128, 4, 10, 0, np.random.default_rng(8))
        assert idx.flat_stats() == (4, 4)
        check(vg, ctx, 3000, 96, 3, 10, 2, np.random.default_rng(9))
    finally:
        hooks.set_hook("VG_FLAT_FORCE_EXACT", 0)


def test_flat_unfused_path_matches_too(vg, ctx):
    """The score-matrix variant (GEMM -> select) stays available behind VG_FLAT_UNFUSED=1."""
    hooks.set_hook("VG_FLAT_UNFUSED", "1")
    try:
        check(vg, ctx, 20000, 128, 6, 10, 0, np.random.default_rng(18))
        check(vg, ctx, 3000, 100, 3, 10, 2, np.random.default_rng(19))
    finally:
        hooks.set_hook("VG_FLAT_UNFUSED", 0)


def test_flat_sorted_rows_and_candidate_overflow(vg, ctx):
    """Rows sorted by distance to the query defeat the sampled threshold on purpose (the sample is
    not representative, far more than 4096 rows pass): the overflow is detected and the
    exhaustive kernel answers; results stay exact."""
    rng = np.random.default_rng(23)
    n, dim = 40000, 64
    base = rng.standard_normal((n, dim)).astype(np.float32)
    q = rng.standard_normal((2, dim)).astype(np.float32)
    order = np.argsort(-((base - q[0]) ** 2).sum(1))   # farthest first: the sampled tiles are all far
    base = np.ascontiguousarray(base[order])
    idx = vg.Index(ctx, n, dim); idx.set_vectors(base)
    ids, sc = idx.search_flat(q, 10)
    for qi in range(2):
        eid, esc = o.flat_search_f32(base, dim, q[qi], 10)
        assert np.array_equal(ids[qi], eid) and np.array_equal(bits(sc[qi]), bits(esc))


def test_flat_clustered_data_triggers_fallback_safely(vg, ctx):
    """All rows within a tiny ball: GEMM-form scores cannot separate them, so the proof fails
    and the exhaustive kernel must take over; results still exact."""
    rng = np.random.default_rng(4)
    center = rng.standard_normal(128).astype(np.float32) * 10
    base = (center + rng.standard_normal((3000, 128)).astype(np.float32) * 1e-4).astype(np.float32)
    check(vg, ctx, 3000, 128, 3, 10, 0, rng, base=base)


def test_flat_errors(vg, ctx):
    idx = vg.Index(ctx, 10, 16)
    with pytest.raises(vg.VecgoHipError) as e:
        idx.search_flat(np.zeros((1, 16), np.float32), 3)
    assert e.value.status == -9
    with pytest.raises(vg.VecgoHipError):
        idx.search_flat(np.zeros((1, 15), np.float32), 3)
    with pytest.raises(vg.VecgoHipError) as e:  # distance.go:103-105
        vg.Index(ctx, 10, 16, vg.Metric.HAMMING).set_vectors(np.zeros((10, 16), np.float32))
    assert e.value.status == -5


@pytest.mark.parametrize("n,dim,nq,k,metric", [(3000, 128, 20, 64, 0), (2500, 768, 9, 40, 2), (1500, 100, 6, 50, 0),
                                               (700, 30, 3, 33, 1), (40, 64, 5, 64, 0)])
def test_k_above_the_gemm_budget(vg, ctx, n, dim, nq, k, metric):
    """k up to 64: above k = 48 the 64 nominated candidates leave the proof no margin, so every row the
    GEMM appended (score under the query's threshold) is re-scored exactly and the proof is made against
    the threshold itself; results match the oracle like every other k."""
    rng = np.random.default_rng(n + dim + k)
    x = rng.standard_normal((n, dim)).astype(np.float32)
    x[n // 3] = x[5]
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    idx = vg.Index(ctx, n, dim, vg.Metric(metric))
    idx.set_vectors(x)
    ids, sc = idx.search_flat(q, k)
    for i in range(nq):
        eid, esc = o.flat_search_f32(x, dim, q[i], k, metric)
        r = eid.size
        assert np.array_equal(ids[i, :r], eid), (i, ids[i], eid)
        assert np.array_equal(sc[i, :r].view(np.uint32), esc.view(np.uint32))
        assert np.all(ids[i, r:] == 0xFFFFFFFF)
    with pytest.raises(vg.VecgoHipError):
        idx.search_flat(q, 513)


@pytest.mark.parametrize("n,dim,nq,k,metric", [(6000, 128, 7, 100, 0), (9000, 768, 5, 256, 2), (5000, 100, 3, 512, 0),
                                               (300, 64, 4, 400, 0), (20000, 32, 70, 65, 1), (4500, 30, 2, 130, 0)])
def test_k_above_64(vg, ctx, n, dim, nq, k, metric):
    """64 < k <= 512: the threshold is taken deeper in the sample (~3k rows pass it), every appended row is
    re-scored exactly and sorted in LDS, the proof is against the threshold.  The fall-back of a failed proof
    (forced here through VG_FLAT_FORCE_EXACT) pages through the exhaustive kernel 64 results at a time."""
    rng = np.random.default_rng(n + dim + k)
    x = rng.standard_normal((n, dim)).astype(np.float32)
    x[n // 3] = x[5]
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    idx = vg.Index(ctx, n, dim, vg.Metric(metric))
    idx.set_vectors(x)
    exp = [o.flat_search_f32(x, dim, q[i], k, metric) for i in range(nq)]

    def check(ids, sc):
        for i in range(nq):
            eid, esc = exp[i]
            r = eid.size
            assert np.array_equal(ids[i, :r], eid), (i, np.flatnonzero(ids[i, :r] != eid)[:5])
            assert np.array_equal(sc[i, :r].view(np.uint32), esc.view(np.uint32))
            assert np.all(ids[i, r:] == 0xFFFFFFFF)

    ids, sc = idx.search_flat(q, k)
    check(ids, sc)
    hooks.set_hook("VG_FLAT_FORCE_EXACT", "1")
    try:
        ids, sc = idx.search_flat(q, k)
    finally:
        hooks.set_hook("VG_FLAT_FORCE_EXACT", 0)
    check(ids, sc)


def test_flat_segment_reference_tests(vg, ctx):
    """internal/segment/flat/{segment,quantization,partitioned}_test.go (reference_kats.json flat_segment_search)
    through the C ABI: fp32 Segment.Search, SQ8 Segment.Search + Rerank, and the partition-probed search over k-means
    partitions trained and assigned on the GPU"""
    from tests import flat_segment_kats

    def search(rows, q, k):
        idx = vg.Index(ctx, rows.shape[0], rows.shape[1]); idx.set_vectors(rows)
        ids, sc = idx.search_flat(q[None, :], k)
        idx.close()
        keep = ids[0] != 0xFFFFFFFF
        return ids[0][keep], sc[0][keep]

    def search_sq8_rerank(rows, q, k):
        idx = vg.Index(ctx, rows.shape[0], rows.shape[1]); idx.set_vectors(rows)
        sq = vg.ScalarQuantizer(ctx, rows.shape[1]); sq.train(rows)
        idx.set_sq8_codes(sq, sq.encode(rows))
        cand, _ = idx.search_sq8(q[None, :], k)
        ids, sc = idx.rerank(q[None, :], cand, k)
        idx.close(); sq.close()
        keep = ids[0] != 0xFFFFFFFF
        return ids[0][keep], sc[0][keep]

    def partition(rows, parts):
        dim = rows.shape[1]
        cent = np.asarray(vg.kmeans_train(ctx, rows, dim, parts, max_iter=10, seed=1)).reshape(parts, dim)
        assign = np.asarray(vg.kmeans_assign(ctx, rows, cent, dim)).astype(np.int64)
        order = np.argsort(assign, kind="stable")
        off = np.concatenate([[0], np.cumsum(np.bincount(assign, minlength=parts))]).astype(np.uint32)
        return cent, off, rows[order]

    def search_probed(rows, cent, off, q, k, nprobes):
        idx = vg.Index(ctx, rows.shape[0], rows.shape[1]); idx.set_vectors(rows)
        idx.set_partitions(cent, off)
        ids, sc = idx.search_flat_probed(q[None, :], k, nprobes, scan=idx.SCAN_F32)
        idx.close()
        keep = ids[0] != 0xFFFFFFFF
        return ids[0][keep], sc[0][keep]

    flat_segment_kats.run(search, search_sq8_rerank, partition, search_probed)
