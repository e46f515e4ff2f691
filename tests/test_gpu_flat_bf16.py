"""vg_index_enable_bf16_filter: the flat search with its nomination GEMMs in bfloat16 must return the same ids and the
same fp32 scores as without the filter — i.e. the reference's (flat/segment.go:691-721 order and tie-break), which is
what the oracle gives.  The filter only changes WHICH rows are nominated; the exact re-score and the widened proof (or
the exhaustive fallback) decide the result."""
import numpy as np
import pytest

from oracle import oracle as o

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


@pytest.mark.parametrize("n,dim,nq,k,metric", [
    (20000, 768, 130, 10, 0),     # BASELINE shape, two query tiles (one ragged)
    (20000, 768, 70, 1, 0),
    (5000, 128, 200, 10, 0),
    (9000, 1024, 65, 32, 0),
    (20000, 768, 130, 10, 2),     # Dot
    (20000, 768, 130, 10, 1),     # Cosine (the reference scores it as Dot on the rows as given)
    (3000, 64, 129, 64, 0),       # k = the candidate budget
    (30000, 192, 100, 100, 0),    # k > 64: every appended row is re-scored
    (20000, 768, 5, 10, 0),       # one block of 32 queries (the HBM-bound tile)
    (20000, 768, 33, 10, 0),      # two blocks
    (20000, 768, 64, 10, 2),
    (15000, 100, 130, 10, 0),     # dim % 64 != 0: the copies are zero-padded to whole K steps (128)
    (15000, 300, 40, 10, 2),
    (8000, 17, 70, 100, 0),       # dim % 4 != 0 too
    (6000, 3, 33, 10, 0),
    (3000, 64, 140, 10, 2),       # no threshold sample (<= 4096 rows) and more than one query tile: the persistent 256 x 256 tile starts
    (4000, 128, 300, 10, 1),      # from a threshold that stands for +Inf — r06 answered Dot segments of this shape wrongly
    (1000, 64, 200, 48, 2),       # (flat_thr_cap_kernel)
    (3000, 64, 140, 10, 0),
])
def test_filter_is_bit_identical(vg, ctx, n, dim, nq, k, metric):
    rng = np.random.default_rng(n + dim + nq + k + metric)
    base = rng.standard_normal((n, dim)).astype(np.float32)
    base[n // 3] = base[7]                      # duplicates: ties broken by row id
    base[n // 2] = base[7]
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    q[3] = base[7]                              # a query that is a corpus row
    idx = vg.Index(ctx, n, dim, vg.Metric(metric))
    idx.set_vectors(base)
    ids0, sc0 = idx.search_flat(q, k)
    idx.enable_bf16_filter(True)
    ids1, sc1 = idx.search_flat(q, k)
    assert np.array_equal(ids0, ids1) and np.array_equal(bits(sc0), bits(sc1))
    for qi in (0, 3, nq - 1):
        eid, esc = o.flat_search_f32(base, dim, q[qi], k, metric)
        assert np.array_equal(ids1[qi], eid) and np.array_equal(bits(sc1[qi]), bits(esc)), qi
    idx.enable_bf16_filter(False)
    ids2, sc2 = idx.search_flat(q, k)
    assert np.array_equal(ids0, ids2) and np.array_equal(bits(sc0), bits(sc2))
    idx.close()


def test_filter_survives_values_bf16_cannot_tell_apart(vg, ctx):
    """Rows that differ only below bfloat16's 8 bits of mantissa get equal filter scores; the exact re-score must still
    order them, and clusters larger than the candidate budget must fall back to the exhaustive kernel."""
    rng = np.random.default_rng(5)
    n, dim, nq, k = 8192, 128, 96, 10
    centre = rng.standard_normal(dim).astype(np.float32)
    base = (centre + 1e-4 * rng.standard_normal((n, dim))).astype(np.float32)     # one tight cluster: all rows alike in bf16
    q = (centre + 1e-4 * rng.standard_normal((nq, dim))).astype(np.float32)
    idx = vg.Index(ctx, n, dim)
    idx.set_vectors(base)
    idx.enable_bf16_filter(True)
    ids, sc = idx.search_flat(q, k)
    for qi in range(0, nq, 13):
        eid, esc = o.flat_search_f32(base, dim, q[qi], k, 0)
        assert np.array_equal(ids[qi], eid) and np.array_equal(bits(sc[qi]), bits(esc)), qi
    searched, exhaustive = idx.flat_stats()
    assert searched == nq and exhaustive > 0          # the proof cannot pass here: the fallback did the work
    idx.close()


def test_filter_needs_rows_and_follows_them(vg, ctx):
    idx = vg.Index(ctx, 100, 64)
    with pytest.raises(vg.VecgoHipError) as e:
        idx.enable_bf16_filter(True)
    assert e.value.status == -9
    rng = np.random.default_rng(1)
    a = rng.standard_normal((100, 64)).astype(np.float32)
    b = rng.standard_normal((100, 64)).astype(np.float32)
    q = rng.standard_normal((70, 64)).astype(np.float32)
    idx.set_vectors(a)
    idx.enable_bf16_filter(True)
    idx.set_vectors(b)                                 # drops the copy of the old rows
    ids, sc = idx.search_flat(q, 5)
    eid, esc = o.flat_search_f32(b, 64, q[0], 5, 0)
    assert np.array_equal(ids[0], eid) and np.array_equal(bits(sc[0]), bits(esc))
    idx.close()


def test_filter_worst_case_rounding(vg, ctx):
    """the proof margin of the bf16 filter covered the rounding of a DOT product; an L2 score
    (|x|^2 - 2 q.x) moves by twice that.  Components just under a bfloat16 rounding boundary (v = 1 + 2^-8 - 2^-20
    rounds DOWN to 1.0) make it bite: the bf16 dot product of the query (all v) with a row of mostly v is short by 2^-7
    of its value, so that row's nomination score is too large by 2^-6 * dim = 12, while rows of exactly representable
    components (1.0 / 2.0) are off by only 6.  The three rows nearest to the query (true distances 3, 4, 5) then look
    FARTHER (15, 16, 17) than eight thousand rows at true distance 6 or 7 (13 and 14), are not nominated, and with the
    r02 margin (6.2 here) the proof of the nominated ones passed: 6.0 < 13 - 6.2.  With the margin L2 needs (12.3)
    it fails, the exhaustive kernel runs, and the result is the reference's."""
    dim, n, k = 768, 8192, 10
    v = np.float32(1.0 + 2.0 ** -8 - 2.0 ** -20)
    rng = np.random.default_rng(5)
    base = np.ones((n, dim), np.float32)
    for r in range(n):                      # C rows: p components 2.0: true distance p + 0.012
        base[r, rng.choice(dim, 6 if r % 273 == 1 else 7, replace=False)] = np.float32(2.0)
    near = {}
    for j, row_id in ((3, 4000), (4, 17), (5, 8000)):   # B rows: all v, j components v + 1: true distance j
        row = np.full(dim, v, np.float32)
        row[rng.choice(dim, j, replace=False)] += np.float32(1.0)
        base[row_id] = row
        near[j] = row_id
    queries = np.tile(np.full(dim, v, np.float32), (8, 1))
    idx = vg.Index(ctx, n, dim, vg.Metric(0))
    idx.set_vectors(base)
    ids0, sc0 = idx.search_flat(queries, k)
    idx.enable_bf16_filter(True)
    ids1, sc1 = idx.search_flat(queries, k)
    eid, esc = o.flat_search_f32(base, dim, queries[0], k, 0)
    assert eid[:3].tolist() == [near[3], near[4], near[5]]
    for qi in range(8):
        assert np.array_equal(ids0[qi], eid) and np.array_equal(bits(sc0[qi]), bits(esc)), qi
        assert np.array_equal(ids1[qi], eid) and np.array_equal(bits(sc1[qi]), bits(esc)), ("bf16 filter", qi, ids1[qi], eid)
    idx.close()


def test_rows_with_inf_next_to_a_ragged_query_tile(vg, ctx):
    """The case tools/fuzz_nonfinite.py caught in r06 (kept as found: tests/golden/r06_bf16_tile_inf_rows.npz — 1000 x 16 rows with
    NaN / Inf / 3e38 among the first 64, 140 queries, Dot): the persistent bf16 tile gave the padding queries of its ragged query
    tile -1e30 to start from, a row holding an Inf made their dot product +Inf, and the "passing" element was appended to a
    candidate list that does not exist — a memory fault.  Now: no fault, and every query equals the oracle (the rows put every
    query at risk: the heap replay answers, NaN scores compare equal)."""
    from pathlib import Path
    d = np.load(Path(__file__).resolve().parent / "golden" / "r06_bf16_tile_inf_rows.npz", allow_pickle=True)
    x, q = d["xr"], d["q"]
    n, dim = x.shape
    for metric in (1, 0):
        idx = vg.Index(ctx, n, dim, vg.Metric(metric))
        idx.set_vectors(x)
        idx.enable_bf16_filter(True)
        ids, sc = idx.search_flat(q, 64)
        for qi in range(0, q.shape[0], 7):
            eid, esc = o.flat_search_f32(x, dim, q[qi], 64, metric)
            assert np.array_equal(ids[qi, :eid.size], eid), (metric, qi)
            a, b = sc[qi, :eid.size], esc
            assert np.all((bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))), (metric, qi)
        idx.close()


@pytest.mark.parametrize("metric", [0, 2])
@pytest.mark.parametrize("nq", [63, 64, 65, 95, 96, 97, 127, 128, 129])
def test_query_tile_boundaries(vg, ctx, nq, metric):
    """the flat search picks its query tile by the batch size — 1 .. 3 blocks of 32 rows up to 96 queries (fp32; up to 4 blocks = 128
    queries on the bf16 image), the 128-query tile, the persistent 256 x 256 tile above 128 with the bf16 image: every boundary,
    fp32 and bf16 nomination, L2 and Dot, against the oracle"""
    rng = np.random.default_rng(700 + nq + metric)
    n, dim, k = 9000, 128, 10
    base = rng.standard_normal((n, dim)).astype(np.float32)
    base[4000] = base[11]
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    q[nq - 1] = base[11]
    idx = vg.Index(ctx, n, dim, vg.Metric(metric))
    idx.set_vectors(base)
    plain = idx.search_flat(q, k)
    idx.enable_bf16_filter(True)
    filt = idx.search_flat(q, k)
    assert np.array_equal(plain[0], filt[0]) and np.array_equal(bits(plain[1]), bits(filt[1]))
    for qi in (0, nq // 2, nq - 1):
        eid, esc = o.flat_search_f32(base, dim, q[qi], k, metric)
        assert np.array_equal(plain[0][qi], eid) and np.array_equal(bits(plain[1][qi]), bits(esc)), qi
    idx.close()
