"""a20 on the GPU: csrc/vg_heap.hpp (the heap every graph search and vg_search_hnsw_brute run on) against the
reference's OWN queue tests (internal/searcher/queue_test.go:12-184 as data in tests/golden/reference_kats.json),
through vg_debug_heap_replay — a one-wave kernel that replays a script of PriorityQueue operations — and against the
oracle's heap on seeded random scripts, for the float sifts and the unsigned-key sifts the walks use.  CandidateHeap
(candidate_queue_test.go) is a total order on (score, segment, row): the GPU realises it as sort keys, checked through
vg_merge_topk."""
import numpy as np
import pytest

from oracle import oracle as o
from tests import heap_kats

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vg():
    import vecgo_amd
    return vecgo_amd


@pytest.fixture(scope="module")
def ctx(vg):
    return vg.Context(0)


def _gpu_replay(vg, ctx, uk):
    def run(is_max, script):
        out, nodes, dists = vg.heap_replay(ctx, is_max, o.heap_script_array(script), unsigned_keys=uk, cap=4096)
        return out, (nodes, dists)
    return run


@pytest.mark.parametrize("uk", [False, True])
def test_priority_queue_reference_tests(vg, ctx, uk):
    heap_kats.check_priority_queue_kats(_gpu_replay(vg, ctx, uk))


@pytest.mark.parametrize("uk", [False, True])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_scripts_equal_oracle(vg, ctx, uk, seed):
    """flags, popped items AND the final heap array, item for item: equal distances sit where queue.go puts them"""
    run = _gpu_replay(vg, ctx, uk)
    for is_max, script in heap_kats.random_scripts(seed):
        assert heap_kats.same(run(is_max, script), o.prioq_replay(is_max, script)), (seed, is_max, len(script))


def test_negative_distances_float_sifts(vg, ctx):
    """-dot distances (metric Dot) take the float sifts: negative and mixed-sign keys"""
    rng = np.random.default_rng(9)
    run = _gpu_replay(vg, ctx, False)
    for is_max in (False, True):
        script = [(0, i, float(rng.integers(-4, 5)), 0) for i in range(300)]
        script += [(3, 1000 + i, float(rng.integers(-4, 5)), 300) for i in range(200)] + [(1, 0, 0.0, 0)] * 300
        assert heap_kats.same(run(is_max, script), o.prioq_replay(is_max, script))


def test_replay_rejects_overflow(vg, ctx):
    with pytest.raises(vg.VecgoHipError):
        vg.heap_replay(ctx, True, o.heap_script_array([(0, i, 1.0, 0) for i in range(20)]), cap=8)


def _best_first(vg, ctx, items, descending):
    """(score, row) candidates -> best-first order by the GPU's CandidateHeap key: every candidate is its own one-entry
    list (vg_merge_topk merges SORTED lists), padded to k with empty slots"""
    n = len(items)
    ids = np.full((n, 1, n), 0xFFFFFFFF, np.uint32)
    sc = np.full((n, 1, n), -np.inf if descending else np.inf, np.float32)
    for j, (score, row) in enumerate(items):
        ids[j, 0, 0], sc[j, 0, 0] = row, score
    bi, bs = vg.merge_topk(ctx, ids, sc, n, metric=2 if descending else 0)
    return [(float(bs[0, j]), int(bi[0, j])) for j in range(n)]


def test_candidate_heap_reference_tests_as_sort_keys(vg, ctx):
    g = heap_kats.KATS["searcher_candidate_heap"]
    for c in g["cases"]:
        if "push" not in c:
            continue
        # the reference's cases leave RowID 0 everywhere; a row can only be listed once per merge here, so a case
        # whose candidates share (segment, row) gets the push index as its row — scores decide those cases anyway
        rows = [x["row_id"] for x in c["push"]]
        uniq = len(set(rows)) == len(rows)
        items = [(x["score"], x["row_id"] if uniq else j) for j, x in enumerate(c["push"])]
        for r in c.get("replace_top", []):   # ReplaceTop = drop the worst (last of best-first), add the new one
            items = _best_first(vg, ctx, items, c["descending"])[:-1] + [(r["with"]["score"], 100 + len(items))]
            assert _best_first(vg, ctx, items, c["descending"])[-1][0] == np.float32(r["expect_top_score"]), (c["name"], r)
        order = _best_first(vg, ctx, items, c["descending"])
        if "expect_top_score" in c:
            assert order[-1][0] == np.float32(c["expect_top_score"]), c["name"]          # heap top = the worst
        if "expect_top_row" in c:
            assert order[-1][1] == c["expect_top_row"], c["name"]
        if "expect_pop_scores" in c:          # pops = worst first = the reverse of best-first
            assert [x[0] for x in order[::-1]] == [np.float32(x) for x in c["expect_pop_scores"]]
    # InternalCandidateBetter's truth table through a 2-candidate merge (one segment: rows of different shards differ
    # by their id offset — see vg_merge_topk)
    for b in g["better"]:
        if b["a"]["segment_id"] != b["b"]["segment_id"]:
            continue
        ra, rb = b["a"]["row_id"], b["b"]["row_id"]
        if ra == rb:
            ra, rb = 1, 2     # scores differ in these cases: the rows do not decide
        order = _best_first(vg, ctx, [(b["a"]["score"], ra), (b["b"]["score"], rb)], b["descending"])
        assert (order[0] == (np.float32(b["a"]["score"]), ra)) == b["expected"], b
