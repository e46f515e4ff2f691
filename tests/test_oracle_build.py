"""Oracle restatements of the construction-time neighbour selections (diskann/writer.go:571-625,
hnsw.go:1009-1106): behaviour checks on hand-made geometry (the reference has no KATs for them
and its builders are not reproducible)."""
import numpy as np

from oracle import oracle as o


def test_robust_prune_drops_occluded_candidates():
    # points on a line: from node 0 at x=0, candidate 1 (x=1) occludes candidate 2 (x=2) for alpha=1:
    # dist(2,1)=1 < dist(2,0)=4; candidate 3 on the other side (x=-1.5) survives
    base = np.zeros((5, 4), np.float32)
    base[:, 0] = [0.0, 1.0, 2.0, -1.5, 0.0]
    kept = o.robust_prune(base, 4, 0, [2, 1, 3, 1, 0, 0xFFFFFFFF, 77], r=8, alpha=1.0)
    assert list(kept) == [1, 3]                      # sorted by distance: 1 (1.0), 3 (2.25), 2 (4.0, occluded)
    kept = o.robust_prune(base, 4, 0, [2, 1, 3], r=8, alpha=5.0)
    assert list(kept) == [1, 3, 2]                   # alpha * dist(2,1) = 5 >= 4: kept
    assert list(o.robust_prune(base, 4, 0, [2, 1, 3], r=1, alpha=5.0)) == [1]
    assert list(o.robust_prune(base, 4, 0, [4], r=4, alpha=1.2)) == [4]   # a duplicate point: distance 0, kept


def test_hnsw_select_neighbors_heuristic_and_fill_up():
    base = np.zeros((6, 2), np.float32)
    base[:, 0] = [0.0, 1.0, 2.0, 3.0, -1.0, 10.0]
    src = base[0]
    ids = np.array([1, 4, 2, 3, 5], np.uint32)       # nearest first from the source at x=0
    d = np.array([np.sum((base[i] - src) ** 2) for i in ids], np.float32)
    # few candidates: all kept
    assert list(o.hnsw_select_neighbors(base, 2, ids[:2], d[:2], m=3)) == [1, 4]
    # heuristic keeps 1 and 4 (opposite sides); 2 and 3 are closer to 1 than to the source; 5 too.
    # m = 3: fill-up appends the first not-yet-kept candidate in order: 2
    assert list(o.hnsw_select_neighbors(base, 2, ids, d, m=3)) == [1, 4, 2]
    assert list(o.hnsw_select_neighbors(base, 2, ids, d, m=2)) == [1, 4]
