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


def test_candidate_heap_reference_tests_as_sort_keys(vg, ctx):
    g = heap_kats.KATS["searcher_candidate_heap"]
    for c in g["cases"]:
        if "push" not in c:
            continue
        items = [(x["score"], x["row_id"]) for x in c["push"]]
        for r in c.get("replace_top", []):   # ReplaceTop = drop the worst, add the new one
            ids = np.array([[[i[1] for i in items]]], np.uint32); sc = np.array([[[i[0] for i in items]]], np.float32)
            bi, bs = vg.merge_topk(ctx, ids, sc, len(items), metric=2 if c["descending"] else 0)
            items = [(float(bs[0, j]), int(bi[0, j])) for j in range(len(items) - 1)] + [(r["with"]["score"], r["with"]["row_id"])]
            ids = np.array([[[i[1] for i in items]]], np.uint32); sc = np.array([[[i[0] for i in items]]], np.float32)
            bi, bs = vg.merge_topk(ctx, ids, sc, len(items), metric=2 if c["descending"] else 0)
            assert bs[0, -1] == np.float32(r["expect_top_score"]), (c["name"], r)    # worst = last of best-first
        ids = np.array([[[i[1] for i in items]]], np.uint32); sc = np.array([[[i[0] for i in items]]], np.float32)
        bi, bs = vg.merge_topk(ctx, ids, sc, len(items), metric=2 if c["descending"] else 0)
        if "expect_top_score" in c and "replace_top" not in c:
            assert bs[0, -1] == np.float32(c["expect_top_score"]), c["name"]
        if "expect_top_row" in c:
            assert bi[0, -1] == c["expect_top_row"], c["name"]
        if "expect_pop_scores" in c:          # pops = worst first = the reverse of best-first
            assert bs[0, ::-1].tolist() == [np.float32(x) for x in c["expect_pop_scores"]]
    # InternalCandidateBetter's truth table through a 2-candidate merge (one segment: SegmentID ties are the engine's,
    # rows of different shards differ by their id offset — see vg_merge_topk)
    for b in g["better"]:
        if b["a"]["segment_id"] != b["b"]["segment_id"]:
            continue
        if (b["a"]["score"], b["a"]["row_id"]) == (b["b"]["score"], b["b"]["row_id"]):
            continue
        ids = np.array([[[b["a"]["row_id"], b["b"]["row_id"]]]], np.uint32)
        sc = np.array([[[b["a"]["score"], b["b"]["score"]]]], np.float32)
        bi, bs = vg.merge_topk(ctx, ids, sc, 2, metric=2 if b["descending"] else 0)
        a_first = (bs[0, 0], bi[0, 0]) == (np.float32(b["a"]["score"]), b["a"]["row_id"])
        assert a_first == b["expected"], b
