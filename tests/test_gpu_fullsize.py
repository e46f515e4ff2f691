"""Parity at BASELINE.json's FULL sizes, where the oracle cannot scan the corpus in seconds:
size-independent properties checked on the GPU results, plus the oracle on exactly the rows the
GPU reported.

  * every reported (row, score) pair is re-scored by the oracle on that row: bit-exact;
  * results are ordered by (score, RowID) — searcher/candidate_queue.go:12-23;
  * partition property: top-k of the whole corpus == merge of the top-k of its two halves
    (engine/search.go:904-908 fan-in), i.e. nothing depends on how rows are tiled;
  * two independent device paths agree (GEMM candidates + proof vs the exhaustive exact kernel);
  * a random sample of rows holds no row that beats the reported k-th;
  * idempotence: the same call twice gives the same bits;
  * WHOLE queries replayed by the oracle over the full corpus (oracle.replay: the oracle's scan / search
    loops, one C thread per query, the reference's compiled AVX-512 kernels when oracle/_ref is present):
    ids and scores of two queries per scan, so a row missed anywhere in the corpus cannot pass.
"""
import os

import numpy as np
import pytest

from oracle import oracle as o
from tests import hooks

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def bits(x):
    return np.asarray(x, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def vg():
    import vecgo_amd
    return vecgo_amd


@pytest.fixture(scope="module")
def ctx(vg):
    return vg.Context(0)


def np_(t):
    return t.cpu().numpy() if hasattr(t, "cpu") else np.asarray(t)


def assert_ordered(ids, sc, descending=False):
    ids = np_(ids).view(np.uint32).astype(np.int64); sc = np_(sc)
    for i in range(ids.shape[0]):
        key = list(zip((-sc[i] if descending else sc[i]).tolist(), ids[i].tolist()))
        assert key == sorted(key), i


def test_flat_exact_1m_x_768(vg, ctx):
    """BASELINE configs[1]: 1M x 768 fp32, exact top-10."""
    n, dim, nq, k = 1_000_000, 768, 96, 10
    g = torch.Generator(device="cuda"); g.manual_seed(20260130)
    base = torch.randn(n, dim, device="cuda", generator=g)
    q = torch.randn(nq, dim, device="cuda", generator=g)
    idx = vg.Index(ctx, n, dim); idx.set_vectors(base)
    ids, sc = idx.search_flat(q, k)
    ids2, sc2 = idx.search_flat(q, k)
    assert torch.equal(ids, ids2) and torch.equal(sc.view(torch.int32), sc2.view(torch.int32))  # idempotent
    searched, exhaustive = idx.flat_stats()
    assert searched == 2 * nq and exhaustive == 0          # answered by GEMM candidates + proof
    assert_ordered(ids, sc)
    # (1) oracle on the reported rows
    hid = np_(ids).view(np.uint32).astype(np.int64)
    rows = base[torch.from_numpy(hid.reshape(-1)).cuda()].cpu().numpy().reshape(nq, k, dim)
    hq = q.cpu().numpy()
    for i in range(nq):
        want = np.array([o.l2(hq[i], rows[i, j]) for j in range(k)], np.float32)
        assert np.array_equal(bits(np_(sc)[i]), bits(want)), i
    # (1b) two whole queries replayed by the oracle over all 1M rows
    hbase = base.cpu().numpy()
    rid, rsc = o.replay(o.BENCH_FLAT, hq[:2], k, base=hbase)
    assert np.array_equal(rid, np_(ids).view(np.uint32)[:2]) and np.array_equal(bits(rsc), bits(np_(sc)[:2]))
    del hbase
    # (2) the exhaustive exact kernel (no GEMM, no proof) gives the same bits
    hooks.set_hook("VG_FLAT_FORCE_EXACT", "1")
    try:
        eids, esc = idx.search_flat(q[:32], k)
    finally:
        hooks.set_hook("VG_FLAT_FORCE_EXACT", 0)
    assert torch.equal(eids, ids[:32]) and torch.equal(esc.view(torch.int32), sc[:32].view(torch.int32))
    # (2b) the bfloat16 nomination filter: the same ids and fp32 score bits, still without a fall-back
    _, e_before = idx.flat_stats()
    idx.enable_bf16_filter(True)
    fids, fsc = idx.search_flat(q, k)
    _, e_after = idx.flat_stats()
    idx.enable_bf16_filter(False)
    assert torch.equal(fids, ids) and torch.equal(fsc.view(torch.int32), sc.view(torch.int32))
    assert e_after == e_before
    # (3) partition property
    half = n // 2
    a = vg.Index(ctx, half, dim); a.set_vectors(base[:half])
    b = vg.Index(ctx, n - half, dim); b.set_vectors(base[half:])
    ia, sa = a.search_flat(q, k); ib, sb = b.search_flat(q, k)
    off = torch.tensor([0, half], dtype=torch.int32, device="cuda")
    mi, ms = vg.merge_topk(ctx, torch.stack((ia, ib)), torch.stack((sa, sb)), k, metric=0, id_offsets=off)
    assert torch.equal(mi, ids) and torch.equal(ms.view(torch.int32), sc.view(torch.int32))
    # (4) no sampled row beats the k-th
    sample = torch.randint(0, n, (20000,), device="cuda", generator=g)
    d = torch.cdist(q.double(), base[sample].double()) ** 2
    assert bool((d.min(dim=1).values.float() >= sc[:, 0] * (1 - 1e-5)).all())
    srt = d.sort(dim=1).values
    assert bool((srt[:, 0].float() * (1 + 1e-5) >= sc[:, 0]).all())


def _train_small_pq(vg, ctx, dim, m):
    rng = np.random.default_rng(3)
    pq = vg.ProductQuantizer(ctx, dim, m, 256)
    pq.set_codebooks(rng.integers(-128, 128, m * 256 * (dim // m)).astype(np.int8),
                     (rng.random(m) * 0.02 + 0.005).astype(np.float32),
                     (rng.standard_normal(m) * 0.1).astype(np.float32))
    return pq


def test_pq_adc_scan_10m_x_96(vg, ctx):
    """BASELINE configs[3]: 10M rows of m=96 PQ codes, LUT scan with fused top-k."""
    n, dim, m, nq, k = 10_000_000, 768, 96, 6, 10
    g = torch.Generator(device="cuda"); g.manual_seed(7)
    codes = torch.randint(0, 256, (n, m), dtype=torch.uint8, device="cuda", generator=g)
    codes[123_456] = codes[9_000_001]      # a tie between far-apart rows: RowID decides
    pq = _train_small_pq(vg, ctx, dim, m)
    idx = vg.Index(ctx, n, dim); idx.set_pq_codes(pq, codes)
    q = torch.randn(nq, dim, device="cuda", generator=g)
    ids, sc = idx.search_pq_adc(q, k)
    ids2, sc2 = idx.search_pq_adc(q, k)
    assert torch.equal(ids, ids2) and torch.equal(sc.view(torch.int32), sc2.view(torch.int32))
    assert_ordered(ids, sc)
    # oracle on the reported rows: BuildDistanceTable + pqAdcLookupAvx512 order
    cb, s_, of_ = pq.codebooks()
    opq = o.ProductQuantizer(dim, m, 256); opq.set_codebooks(cb, s_, of_)
    hid = np_(ids).view(np.uint32).astype(np.int64)
    hq = q.cpu().numpy()
    for i in range(nq):
        table = opq.build_table(hq[i])
        rc = codes[torch.from_numpy(hid[i]).cuda()].cpu().numpy()
        want = np.array([o.adc(table, rc[j], m) for j in range(k)], np.float32)
        assert np.array_equal(bits(np_(sc)[i]), bits(want)), i
    # two whole queries replayed by the oracle over all 10M codes
    rid, rsc = o.replay(o.BENCH_ADC, hq[:2], k, pq=opq, codes=codes.cpu().numpy(), n=n)
    assert np.array_equal(rid, np_(ids).view(np.uint32)[:2]) and np.array_equal(bits(rsc), bits(np_(sc)[:2]))
    # partition property
    half = n // 2
    a = vg.Index(ctx, half, dim); a.set_pq_codes(pq, codes[:half])
    b = vg.Index(ctx, n - half, dim); b.set_pq_codes(pq, codes[half:])
    ia, sa = a.search_pq_adc(q, k); ib, sb = b.search_pq_adc(q, k)
    off = torch.tensor([0, half], dtype=torch.int32, device="cuda")
    mi, ms = vg.merge_topk(ctx, torch.stack((ia, ib)), torch.stack((sa, sb)), k, metric=0, id_offsets=off)
    assert torch.equal(mi, ids) and torch.equal(ms.view(torch.int32), sc.view(torch.int32))
    # big-k path (64 < k <= 1024) contains the small-k answer as its prefix
    i100, s100 = idx.search_pq_adc(q, 100)
    assert torch.equal(i100[:, :k], ids) and torch.equal(s100[:, :k].view(torch.int32), sc.view(torch.int32))
    assert_ordered(i100, s100)
    # a sample of rows scored by the batch entry point holds nothing better than the k-th
    table = pq.build_distance_table(q[:1])
    lo = 4_000_000
    d = vg.pq_adc_lookup_batch(ctx, table, codes[lo:lo + 500_000], m)
    kth = float(np_(sc)[0, k - 1]); worst_id = int(hid[0, k - 1])
    better = torch.nonzero(d < kth).flatten() + lo
    assert set(np_(better).tolist()) <= set(hid[0].tolist()), (better, worst_id)


def test_pq_batches_through_the_bf16_nomination_1m_x_96(vg, ctx):
    """vg_index_enable_pq_nomination at the bench's size: 1024 queries x 1M rows of m = 96 codes ENCODED from random-normal rows
    (the nomination's bf16 image is the decoded rows), duplicates included.  Nominated batch == one table scan per query, bit for
    bit; the oracle's table sums on the reported rows; two whole queries replayed by the oracle over all the codes; k = 100 (every
    row below the threshold re-scored and sorted)."""
    n, dim, m, nq, k = 1_000_000, 768, 96, 1024, 10
    g = torch.Generator(device="cuda"); g.manual_seed(11)
    rows = torch.randn(n, dim, device="cuda", generator=g)
    pq = _train_small_pq(vg, ctx, dim, m)
    codes = pq.encode(rows)
    codes[123_456] = codes[900_001]
    del rows
    idx = vg.Index(ctx, n, dim); idx.set_pq_codes(pq, codes)
    q = torch.randn(nq, dim, device="cuda", generator=g)
    ids, sc = idx.search_pq_adc(q, k)
    i100, s100 = idx.search_pq_adc(q[:256], 100)
    idx.enable_pq_nomination(True)
    nid, nsc = idx.search_pq_adc(q, k)
    assert torch.equal(ids, nid) and torch.equal(sc.view(torch.int32), nsc.view(torch.int32))
    n100, t100 = idx.search_pq_adc(q[:256], 100)
    assert torch.equal(i100, n100) and torch.equal(s100.view(torch.int32), t100.view(torch.int32))
    assert_ordered(nid, nsc)
    cb, s_, of_ = pq.codebooks()
    opq = o.ProductQuantizer(dim, m, 256); opq.set_codebooks(cb, s_, of_)
    hid = np_(nid).view(np.uint32).astype(np.int64)
    hq = q.cpu().numpy()
    for i in (0, 1, 511, 1023):
        table = opq.build_table(hq[i])
        rc = codes[torch.from_numpy(hid[i]).cuda()].cpu().numpy()
        want = np.array([o.adc(table, rc[j], m) for j in range(k)], np.float32)
        assert np.array_equal(bits(np_(nsc)[i]), bits(want)), i
    rid, rsc = o.replay(o.BENCH_ADC, hq[:2], k, pq=opq, codes=codes.cpu().numpy(), n=n)
    assert np.array_equal(rid, np_(nid).view(np.uint32)[:2]) and np.array_equal(bits(rsc), bits(np_(nsc)[:2]))


def test_rabitq_scan_10m_x_768(vg, ctx):
    """BASELINE configs[4] on one GPU: 10M RaBitQ codes of 100 bytes."""
    n, dim, nq, k = 10_000_000, 768, 4, 10
    cb = (dim + 63) // 64 * 8 + 4
    g = torch.Generator(device="cuda"); g.manual_seed(11)
    codes = torch.randint(0, 256, (n, cb), dtype=torch.uint8, device="cuda", generator=g)
    codes[:, cb - 4:] = (torch.rand(n, device="cuda", generator=g) * 5 + 25).view(torch.uint8).reshape(n, 4)
    idx = vg.Index(ctx, n, dim); idx.set_rabitq_codes(codes)
    q = torch.randn(nq, dim, device="cuda", generator=g)
    ids, sc = idx.search_rabitq(q, k)
    assert_ordered(ids, sc)
    hid = np_(ids).view(np.uint32).astype(np.int64)
    hq = q.cpu().numpy()
    for i in range(nq):
        rc = codes[torch.from_numpy(hid[i]).cuda()].cpu().numpy()
        want = np.array([o.rabitq_distance(hq[i], rc[j]) for j in range(k)], np.float32)
        assert np.array_equal(bits(np_(sc)[i]), bits(want)), i
    # two whole queries replayed by the oracle over all 10M codes
    rid, rsc = o.replay(o.BENCH_RABITQ, hq[:2], k, codes=codes.cpu().numpy(), n=n, dim=dim)
    assert np.array_equal(rid, np_(ids).view(np.uint32)[:2]) and np.array_equal(bits(rsc), bits(np_(sc)[:2]))
    # nq >= 2 runs the query-blocked kernel (16 queries per workgroup pass), nq == 1 the one-query kernel: a ragged
    # batch of 37 must give every query what its own pass gives
    qb = torch.randn(37, dim, device="cuda", generator=g)
    bi, bs = idx.search_rabitq(qb, k)
    for i in (0, 15, 16, 36):
        i1, s1 = idx.search_rabitq(qb[i:i + 1], k)
        assert torch.equal(i1[0], bi[i]) and torch.equal(s1[0].view(torch.int32), bs[i].view(torch.int32)), i
    half = n // 2
    a = vg.Index(ctx, half, dim); a.set_rabitq_codes(codes[:half])
    b = vg.Index(ctx, n - half, dim); b.set_rabitq_codes(codes[half:])
    ia, sa = a.search_rabitq(q, k); ib, sb = b.search_rabitq(q, k)
    off = torch.tensor([0, half], dtype=torch.int32, device="cuda")
    mi, ms = vg.merge_topk(ctx, torch.stack((ia, ib)), torch.stack((sa, sb)), k, metric=0, id_offsets=off)
    assert torch.equal(mi, ids) and torch.equal(ms.view(torch.int32), sc.view(torch.int32))
    # rabitq scores are heavily tied: the order must still be (score, RowID)
    rq = vg.RaBitQuantizer(ctx, dim)
    d = rq.distance(q[:1], codes[:300_000])
    kth = float(np_(sc)[0, k - 1])
    better = torch.nonzero(d < kth).flatten()
    assert set(np_(better).tolist()) <= set(hid[0].tolist())


def test_hnsw_layer0_1m_x_768(vg, ctx):
    """BASELINE configs[2]: layer-0 search, ef = 128, over a 1M x 768 corpus.  The graph is a
    random 32-regular graph (search parity does not depend on graph quality; building a real one
    takes minutes); 6 queries are replayed by the oracle over the same graph: ids, scores and
    counters bit-exact."""
    n, dim, m0, nq, k, ef = 1_000_000, 768, 32, 64, 10, 128
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    base = torch.randn(n, dim, device="cuda", generator=g)
    l0 = torch.randint(0, n, (n, m0), dtype=torch.int32, device="cuda", generator=g)
    q = torch.randn(nq, dim, device="cuda", generator=g)
    idx = vg.Index(ctx, n, dim); idx.set_vectors(base)
    hl0 = l0.cpu().numpy().view(np.uint32)
    idx.set_hnsw_graph(hl0, (), entry_point=12345, m=m0 // 2)
    ids, sc, stats = idx.search_hnsw(q, k, ef, stats=True)
    ids2, sc2 = idx.search_hnsw(q, k, ef)
    assert torch.equal(ids, ids2) and torch.equal(sc.view(torch.int32), sc2.view(torch.int32))
    assert_ordered(ids, sc)
    hbase = base.cpu().numpy(); hq = q.cpu().numpy()
    graph = o.HnswIndex(hbase, dim, hl0, (), entry_point=12345, m=m0 // 2)
    hid = np_(ids).view(np.uint32); hsc = np_(sc)
    for i in range(6):
        eid, esc, est = graph.search(hq[i], k, ef)
        assert np.array_equal(hid[i, :eid.size], eid), i
        assert np.array_equal(bits(hsc[i, :eid.size]), bits(esc)), i
        assert tuple(int(x) for x in stats[i][:3]) == \
               (est.nodes_visited, est.distance_computations, est.distance_short_circuits), i


def test_sq8_scan_10m_x_768(vg, ctx):
    """SURVEY.md §8f rank 3 at the config-4 row count: 10M x 768 SQ8 codes (7.7 GB), exhaustive scan."""
    n, dim, nq, k = 10_000_000, 768, 3, 10
    g = torch.Generator(device="cuda"); g.manual_seed(13)
    codes = torch.randint(0, 256, (n, dim), dtype=torch.uint8, device="cuda", generator=g)
    codes[6_000_123] = codes[42]
    rng = np.random.default_rng(1)
    mins = (rng.standard_normal(dim) - 4).astype(np.float32); maxs = (mins + 8 + rng.random(dim)).astype(np.float32)
    sq = vg.ScalarQuantizer(ctx, dim); sq.set_bounds(mins, maxs)
    idx = vg.Index(ctx, n, dim); idx.set_sq8_codes(sq, codes)
    q = torch.randn(nq, dim, device="cuda", generator=g)
    ids, sc = idx.search_sq8(q, k)
    assert_ordered(ids, sc)
    _, _, _, inv = sq.params()
    hid = np_(ids).view(np.uint32).astype(np.int64); hq = q.cpu().numpy()
    for i in range(nq):
        rc = codes[torch.from_numpy(hid[i]).cuda()].cpu().numpy()
        want = o.sq8u_l2_batch(hq[i], rc, mins, inv, dim)
        assert np.array_equal(bits(np_(sc)[i]), bits(want)), i
    # two whole queries replayed by the oracle over all 10M codes
    rid, rsc = o.replay(o.BENCH_SQ8, hq[:2], k, codes=codes.cpu().numpy(), n=n, dim=dim, sq_mins=mins, sq_inv_scales=inv)
    assert np.array_equal(rid, np_(ids).view(np.uint32)[:2]) and np.array_equal(bits(rsc), bits(np_(sc)[:2]))
    half = n // 2
    a = vg.Index(ctx, half, dim); a.set_sq8_codes(sq, codes[:half])
    b = vg.Index(ctx, n - half, dim); b.set_sq8_codes(sq, codes[half:])
    ia, sa = a.search_sq8(q, k); ib, sb = b.search_sq8(q, k)
    off = torch.tensor([0, half], dtype=torch.int32, device="cuda")
    mi, ms = vg.merge_topk(ctx, torch.stack((ia, ib)), torch.stack((sa, sb)), k, metric=0, id_offsets=off)
    assert torch.equal(mi, ids) and torch.equal(ms.view(torch.int32), sc.view(torch.int32))
    d = sq.l2_distance_batch(q[:1], codes[2_000_000:2_200_000])
    better = torch.nonzero(d < float(np_(sc)[0, k - 1])).flatten() + 2_000_000
    assert set(np_(better).tolist()) <= set(hid[0].tolist())


def test_partition_probed_flat_1m_x_768(vg, ctx):
    """flat.Segment.Search over a compacted-size flat segment (flat/segment.go:727-749): 1M x 768 in
    rows/8192 = 122 k-means partitions (trained, assigned and grouped on the GPU), 96 queries.
    The oracle replays whole queries over just the probed partitions (2 x ~8k rows each), the rest is
    checked through properties: probed rows only, ordered, idempotent, grouped == pair-by-pair scan."""
    n, dim, nq, k, nprobes = 1_000_000, 768, 96, 10, 2
    g = torch.Generator(device="cuda"); g.manual_seed(20260131)
    base = torch.randn(n, dim, device="cuda", generator=g)
    q = torch.randn(nq, dim, device="cuda", generator=g)
    parts = n // 8192
    cent = vg.kmeans_train(ctx, base, dim, parts, max_iter=4, seed=5)
    assign = vg.kmeans_assign(ctx, base, cent, dim).to(torch.int64)
    order = torch.argsort(assign, stable=True)
    base = base[order].contiguous()
    off = np.concatenate([[0], np.cumsum(torch.bincount(assign, minlength=parts).cpu().numpy())]).astype(np.uint32)
    cent_h = cent.cpu().numpy()
    idx = vg.Index(ctx, n, dim); idx.set_vectors(base); idx.set_partitions(cent_h, off)
    ids, sc = idx.search_flat_probed(q, k, nprobes, scan=idx.SCAN_F32)
    ids2, sc2 = idx.search_flat_probed(q, k, nprobes, scan=idx.SCAN_F32)
    assert torch.equal(ids, ids2) and torch.equal(sc.view(torch.int32), sc2.view(torch.int32))
    assert_ordered(ids, sc)
    hooks.set_hook("VG_PROBE_NO_GROUP", "1")
    try:
        pid, psc = idx.search_flat_probed(q, k, nprobes, scan=idx.SCAN_F32)
    finally:
        hooks.set_hook("VG_PROBE_NO_GROUP", 0)
    assert torch.equal(ids, pid) and torch.equal(sc.view(torch.int32), psc.view(torch.int32))
    ids_h = np_(ids).view(np.uint32); sc_h = np_(sc); q_h = np_(q)
    for i in range(0, nq, 12):  # whole-query replay by the oracle on the probed partitions' rows
        probed = o.find_closest_centroids(q_h[i], cent_h, dim, nprobes)
        rows = np.concatenate([np.arange(off[p], off[p + 1]) for p in probed])
        assert set(ids_h[i].tolist()) <= set(rows.tolist())
        sub = np_(base[torch.as_tensor(rows, device="cuda")])
        eid, esc = o.flat_search_f32(sub, dim, q_h[i], k)
        assert np.array_equal(rows[eid], ids_h[i]) and np.array_equal(bits(esc), bits(sc_h[i]))
    # one probe of every query = the reference's default (NProbes <= 0 -> 1)
    d_ids, d_sc = idx.search_flat_probed(q, k, 0, scan=idx.SCAN_F32)
    o_ids, o_sc = idx.search_flat_probed(q, k, 1, scan=idx.SCAN_F32)
    assert torch.equal(d_ids, o_ids) and torch.equal(d_sc.view(torch.int32), o_sc.view(torch.int32))
    idx.close()


def test_built_hnsw_graph_1m_x_768(vg, ctx):
    """The bench's own configuration: HNSW built by vg_hnsw_build (M = 32, M0 = 64, EF = 300) on 1M x 768, then
    searched at ef = 128 (heaps in LDS) and ef = 1024 / 4096 (heaps split between LDS and HBM scratch), on fp32 rows and
    on PQ codes (ef = 256 and 2048), and
    walked as a Vamana graph (layer 0, R = 64) with PQ and RaBitQ scoring.  Whole queries are replayed by the
    oracle over the same graph: ids and scores bit-exact."""
    n, dim, k = 1_000_000, 768, 10
    g = torch.Generator(device="cuda"); g.manual_seed(20260130)
    base = torch.randn(n, dim, device="cuda", generator=g)
    q = torch.randn(16, dim, device="cuda", generator=g)
    idx = vg.Index(ctx, n, dim); idx.set_vectors(base)
    idx.build_hnsw(m=32, ef_construction=300, max_batch=8192, growth_div=32)
    l0, upper, entry = idx.get_hnsw_graph()
    deg = (l0 != 0xFFFFFFFF).sum(1)
    assert deg.min() >= 1 and deg.mean() > 48          # layer-0 rows fill up (M0 = 64)
    lv = np.array([vg._lib.load().vg_hnsw_level_for_id(i, 32) for i in range(0, n, 997)])
    assert len(upper) >= lv.max()                        # the top level is at least what the sample saw
    hbase = base.cpu().numpy(); hq = q.cpu().numpy()
    oidx = o.HnswIndex(hbase, dim, l0, upper, entry, m=32)
    for ef, nrep in ((128, 4), (1024, 2), (4096, 1)):
        ids, sc = idx.search_hnsw(q, k, ef)
        rid, rsc = o.replay(o.BENCH_HNSW, hq[:nrep], k, hnsw=oidx, ef=ef)
        assert np.array_equal(rid, np_(ids).view(np.uint32)[:nrep]), ef
        assert np.array_equal(bits(rsc), bits(np_(sc)[:nrep])), ef
    # PQ codes: graph walk on ComputeAsymmetricDistance, and the Vamana beam over layer 0
    pq = _train_small_pq(vg, ctx, dim, 96)
    codes = pq.encode(base)
    idx.set_pq_codes(pq, codes)
    cb, s_, of_ = pq.codebooks()
    opq = o.ProductQuantizer(dim, 96, 256); opq.set_codebooks(cb, s_, of_)
    hcodes = codes.cpu().numpy()
    ids, sc = idx.search_hnsw_pq(q, 64, 256)
    opidx = o.HnswIndex(hbase, dim, l0, upper, entry, m=32, pq=opq, codes=hcodes)
    rid, rsc = o.replay(o.BENCH_HNSW, hq[:2], 64, hnsw=opidx, ef=256)
    assert np.array_equal(rid, np_(ids).view(np.uint32)[:2]) and np.array_equal(bits(rsc), bits(np_(sc)[:2]))
    ids, sc = idx.search_hnsw_pq(q, 64, 2048)
    rid, rsc = o.replay(o.BENCH_HNSW, hq[:1], 64, hnsw=opidx, ef=2048)
    assert np.array_equal(rid, np_(ids).view(np.uint32)[:1]) and np.array_equal(bits(rsc), bits(np_(sc)[:1]))
    idx.set_vamana_graph(l0, entry)
    ids, sc = idx.search_vamana(q, k, kind=1)
    ov = o.VamanaIndex(l0, entry, dim, o.VAMANA_PQ, pq=opq, codes=hcodes)
    rid, rsc = o.replay(o.BENCH_VAMANA, hq[:4], k, vamana=ov)
    assert np.array_equal(rid, np_(ids).view(np.uint32)[:4]) and np.array_equal(bits(rsc), bits(np_(sc)[:4]))
    rcodes = vg.RaBitQuantizer(ctx, dim).encode(base)
    idx.set_rabitq_codes(rcodes)
    ids, sc = idx.search_vamana(q, k, kind=2)
    ov = o.VamanaIndex(l0, entry, dim, o.VAMANA_RABITQ, codes=rcodes.cpu().numpy())
    rid, rsc = o.replay(o.BENCH_VAMANA, hq[:4], k, vamana=ov)
    assert np.array_equal(rid, np_(ids).view(np.uint32)[:4]) and np.array_equal(bits(rsc), bits(np_(sc)[:4]))
    idx.close()


def test_vamana_10m_nodes_r64(vg, ctx):
    """BASELINE configs[3]: Vamana + PQ (m = 96, 256 centroids) at 10M nodes, R = 64 — and the RaBitQ flavour.
    The graph is a random 64-regular graph with holes (search parity does not depend on graph quality); the codes
    are random bytes.  Four whole queries per flavour are replayed by the oracle: ids, scores, counters."""
    n, dim, r, k = 10_000_000, 768, 64, 10
    rng = np.random.default_rng(99)
    graph = rng.integers(0, n, (n, r), dtype=np.uint32)
    graph[rng.integers(0, n, n // 50), rng.integers(0, r, n // 50)] = 0xFFFFFFFF   # empty slots (segment.go:671-681)
    g = torch.Generator(device="cuda"); g.manual_seed(31)
    q = torch.randn(32, dim, device="cuda", generator=g)
    hq = q.cpu().numpy()
    idx = vg.Index(ctx, n, dim)
    idx.set_vamana_graph(graph, 4242)
    pq = _train_small_pq(vg, ctx, dim, 96)
    codes = torch.randint(0, 256, (n, 96), dtype=torch.uint8, device="cuda", generator=g)
    idx.set_pq_codes(pq, codes)
    cb, s_, of_ = pq.codebooks()
    opq = o.ProductQuantizer(dim, 96, 256); opq.set_codebooks(cb, s_, of_)
    ids, sc, st = idx.search_vamana(q, k, kind=1, stats=True)
    ov = o.VamanaIndex(graph, 4242, dim, o.VAMANA_PQ, pq=opq, codes=codes.cpu().numpy())
    for i in range(4):
        eid, esc, est = ov.search(hq[i], k)
        assert np.array_equal(np_(ids).view(np.uint32)[i, :eid.size], eid), i
        assert np.array_equal(bits(np_(sc)[i, :eid.size]), bits(esc)), i
        assert (int(st[i][0]), int(st[i][1]), int(st[i][3])) == (est.nodes_visited, est.distance_computations, est.pops)
    del codes
    cbytes = (dim + 63) // 64 * 8 + 4
    rcodes = torch.randint(0, 256, (n, cbytes), dtype=torch.uint8, device="cuda", generator=g)
    rcodes[:, cbytes - 4:] = (torch.rand(n, device="cuda", generator=g) * 5 + 25).view(torch.uint8).reshape(n, 4)
    idx.set_rabitq_codes(rcodes)
    ids, sc, st = idx.search_vamana(q, k, kind=2, stats=True)
    ov = o.VamanaIndex(graph, 4242, dim, o.VAMANA_RABITQ, codes=rcodes.cpu().numpy())
    for i in range(4):
        eid, esc, est = ov.search(hq[i], k)
        assert np.array_equal(np_(ids).view(np.uint32)[i, :eid.size], eid), i
        assert np.array_equal(bits(np_(sc)[i, :eid.size]), bits(esc)), i
        assert (int(st[i][0]), int(st[i][1]), int(st[i][3])) == (est.nodes_visited, est.distance_computations, est.pops)
    idx.close()


def test_hnsw_brute_1m_x_768(vg, ctx):
    """hnsw.BruteSearch / searchBitmap (hnsw.go:2021-2101, 2240-2263) at 1M x 768: random-normal rows have no tied
    distances, so both heap disciplines must return what flat.Segment.Search returns (ids, order, score bits — L2), the
    replay of 245 steps of 4096 rows per query may not lose or reorder anything, a selective bitmap answers with its
    own rows only, and two whole queries are re-run by the oracle over all rows."""
    n, dim, nq, k = 1_000_000, 768, 24, 10
    g = torch.Generator(device="cuda"); g.manual_seed(20260133)
    base = torch.randn(n, dim, device="cuda", generator=g)
    q = torch.randn(nq, dim, device="cuda", generator=g)
    idx = vg.Index(ctx, n, dim); idx.set_vectors(base)
    fi, fs = idx.search_flat(q, k)
    for mode in (idx.BRUTE_SCAN, idx.BRUTE_BITMAP):
        bi, bs = idx.search_hnsw_brute(q, k, mode)
        assert torch.equal(bi, fi) and torch.equal(bs.view(torch.int32), fs.view(torch.int32)), mode
    # a 1 % bitmap (searchBitmap's regime): results lie in the bitmap and equal a flat search over just those rows
    rng = np.random.default_rng(5)
    member = rng.random(n) < 0.01
    mi, ms = idx.search_hnsw_brute(q, k, idx.BRUTE_BITMAP, member)
    rows = np.nonzero(member)[0]
    sub = vg.Index(ctx, rows.size, dim); sub.set_vectors(base[torch.from_numpy(rows).cuda()].contiguous())
    si, ss = sub.search_flat(q, k)
    assert np.array_equal(rows[np_(si).view(np.uint32).astype(np.int64)], np_(mi).view(np.uint32).astype(np.int64))
    assert torch.equal(ss.view(torch.int32), ms.view(torch.int32))
    sub.close()
    # two whole queries by the oracle (its own heap, the reference's compiled distance kernel when present)
    hbase = base.cpu().numpy(); hq = q.cpu().numpy()
    oidx = o.HnswIndex(hbase, dim, np.full((n, 1), 0xFFFFFFFF, np.uint32))
    was = o.use_reference_kernels(True)
    try:
        for qi in (0, nq - 1):
            eid, esc = oidx.brute_search(hq[qi], k, o.BRUTE_SCAN)
            assert np.array_equal(eid, np_(bi).view(np.uint32)[qi]) and np.array_equal(bits(esc), bits(np_(bs)[qi]))
    finally:
        o.use_reference_kernels(False)
    idx.close()
