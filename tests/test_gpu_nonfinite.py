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
