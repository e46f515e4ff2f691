"""searcher.CandidateHeap with NaN scores (candidate_queue.go:12-38): worked by hand from the reference's loops, against the oracle's
restatement — the semantics the GPU replay (vg_cand_replay.hpp) and tests/test_gpu_nonfinite.py are held to."""
import numpy as np

from oracle import oracle as o


def test_the_heap_by_hand():
    """k = 2, scores NaN, 1, 0.5, 3, NaN: the NaN enters the empty heap and is its root; 1 is appended (not worse than its parent:
    stays a leaf); 0.5 is not Better than a NaN root — turned away, like everything after it.  Pop() gives the root (the NaN)
    first, then 1: best first = [row 1, row 0].  (A sort on (score, row) would answer [row 2, row 1].)"""
    h = o.CandidateHeap(False)
    for row, s in enumerate([np.nan, 1.0, 0.5, 3.0, np.nan]):
        h.push(np.float32(s), 0, row, k=2)
    popped = [h.pop(), h.pop()]
    assert h.pop() is None
    assert popped[0][2] == 0 and np.isnan(popped[0][0]) and popped[1][2] == 1 and popped[1][0] == 1.0
    h.close()


def test_flat_search_pops_the_heap():
    """the oracle's flat searches report Pop() order reversed (engine/search.go:859-862): with a NaN row among the first k the
    answer keeps it, and rows better than everything kept are turned away"""
    x = np.array([[np.nan], [1.0], [0.0], [3.0]], np.float32)      # L2 to q = 0: NaN, 1, 0, 9
    ids, sc = o.flat_search_f32(x, 1, np.zeros(1, np.float32), 2)
    assert list(ids) == [1, 0] and sc[0] == 1.0 and np.isnan(sc[1])
    x[0, 0] = 5.0                                                   # no NaN: the sort everyone expects
    ids, sc = o.flat_search_f32(x, 1, np.zeros(1, np.float32), 2)
    assert list(ids) == [2, 1] and list(sc) == [0.0, 1.0]
