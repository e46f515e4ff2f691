"""a20 on the CPU: the oracle's searcher.PriorityQueue and CandidateHeap against the reference's OWN tests
(internal/searcher/queue_test.go:12-184, candidate_queue_test.go:10-252, transcribed as data into
tests/golden/reference_kats.json) and against a second reading of queue.go in plain Python on seeded random scripts."""
import numpy as np

from oracle import oracle as o
from tests import heap_kats


def test_priority_queue_reference_tests_oracle():
    heap_kats.check_priority_queue_kats(o.prioq_replay)


def test_priority_queue_reference_tests_python_heap():
    heap_kats.check_priority_queue_kats(heap_kats.python_replay)


def test_priority_queue_random_scripts_oracle_equals_python_heap():
    for is_max, script in heap_kats.random_scripts(7):
        assert heap_kats.same(o.prioq_replay(is_max, script), heap_kats.python_replay(is_max, script))


def test_candidate_heap_reference_tests():
    g = heap_kats.KATS["searcher_candidate_heap"]
    for c in g["cases"]:
        if "push" not in c:
            continue
        h = o.CandidateHeap(c["descending"])
        for x in c["push"]:
            h.push(x["score"], x["segment_id"], x["row_id"])
        if "expect_len" in c:
            assert len(h) == c["expect_len"], c["name"]
        if "expect_top_score" in c:
            assert h.top()[0] == np.float32(c["expect_top_score"]), c["name"]
        if "expect_top_row" in c:
            assert h.top()[2] == c["expect_top_row"], c["name"]
        for r in c.get("replace_top", []):
            assert h.replace_top(r["with"]["score"], r["with"]["segment_id"], r["with"]["row_id"])
            assert h.top()[0] == np.float32(r["expect_top_score"]), (c["name"], r)
        if "expect_pop_scores" in c:
            assert [h.pop()[0] for _ in c["expect_pop_scores"]] == [np.float32(x) for x in c["expect_pop_scores"]]
            assert h.pop() is None
        h.close()
    for b in g["better"]:
        t = lambda x: (x["score"], x["segment_id"], x["row_id"])
        assert o.cand_better(t(b["a"]), t(b["b"]), b["descending"]) == b["expected"], b
    # Ascending (candidate_queue_test.go:10-41): pops come worst first (seeded here; the property is the KAT)
    rng = np.random.default_rng(20260402)
    h = o.CandidateHeap(False)
    for i, s in enumerate(rng.random(100, dtype=np.float32)):
        h.push(float(s), 1, i)
    assert len(h) == 100
    pops = [h.pop()[0] for _ in range(100)]
    assert all(pops[i + 1] <= pops[i] for i in range(99))
    h.close()


def test_candidate_heap_determinism():
    """candidate_queue_test.go:146-183: many duplicate (score, segment, row) triples, two runs pop identically — and,
    (score, segment, row) being a total order, the pop sequence is the sorted sequence whatever the heap layout."""
    rng = np.random.default_rng(42)
    cands = [(float(rng.integers(0, 10)), int(rng.integers(0, 5)), int(rng.integers(0, 100))) for _ in range(1000)]
    runs = []
    for _ in range(2):
        h = o.CandidateHeap(False, cap=1000)
        for c in cands:
            h.push(*c)
        runs.append([h.pop() for _ in range(1000)])
        h.close()
    assert runs[0] == runs[1]
    assert runs[0] == sorted(cands, reverse=True)


def test_visited_set_reference_tests():
    """a20's third structure: searcher.VisitedSet (visited.go:12-129) restated literally (uint8 epochs, growth, clear on
    wrap-around) against the reference's own visited_test.go, as data in reference_kats.json.  On the device the set is a
    per-query bitmap with an atomic test-and-set (vg_hnsw_layer.hpp); its observable effect — which nodes a walk scores,
    `nodes_visited` — is pinned by the stats parity of tests/test_gpu_graph.py."""
    g = heap_kats.KATS["searcher_visited_set"]
    for c in g["cases"]:
        if "script" in c:
            assert o.visited_replay(c["capacity"], [tuple(s) for s in c["script"]]) == c["expect"], c["name"]
    rng = np.random.default_rng(20260404)
    ids = rng.integers(0, 5000, 100).tolist()
    script = [("visit", i) for i in ids] + [("visited", i) for i in range(5000)] + [("reset", 0)] + [("visited", i) for i in range(5000)]
    out = o.visited_replay(10, script)
    member = set(ids)
    assert out[100:5100] == [int(i in member) for i in range(5000)]
    assert not any(out[5101:])
    # the epoch wraps after 255 resets: everything is cleared, earlier visits do not come back (visited.go:102-109)
    script = [("visit", 7)] + [("reset", 0)] * 255 + [("visited", 7), ("visit", 3), ("visited", 3), ("check_and_visit", 3), ("check_and_visit", 4),
                                                       ("check_and_visit", 4)]
    assert o.visited_replay(16, script)[-6:] == [0, 0, 1, 1, 0, 1]
