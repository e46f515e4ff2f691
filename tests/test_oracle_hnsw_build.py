"""The oracle's HNSW construction (oracle/vg_oracle_hnsw_build.c), pinned the way the reference pins its own
builder: structure invariants plus the recall thresholds of internal/hnsw/hnsw_test.go."""
import numpy as np

from oracle import oracle as o

INVALID = 0xFFFFFFFF


def _brute(base, q, k):
    d = ((base[None, :, :] - q[:, None, :]) ** 2).sum(-1)
    return np.argsort(d, axis=1, kind="stable")[:, :k]


def test_levels_distribution_and_determinism():
    lv, rows, top = o.hnsw_layout(100000, 32)
    lv2, _, _ = o.hnsw_layout(100000, 32)
    assert np.array_equal(lv, lv2)
    # P(level >= 1) = 1/M (layerMultiplier = 1/ln M, hnsw.go:218)
    assert abs((lv >= 1).mean() - 1 / 32) < 0.004
    assert rows[0] == (lv >= 1).sum() and top == lv.max()
    # splitmix64 finaliser of id (hnsw.go:2103-2116): spot values computed by hand from the formula
    import math
    def ref(i, m):
        x = (i + 0x9e3779b97f4a7c15) & (2**64 - 1)
        x = ((x ^ (x >> 30)) * 0xbf58476d1ce4e5b9) & (2**64 - 1)
        x = ((x ^ (x >> 27)) * 0x94d049bb133111eb) & (2**64 - 1)
        x ^= x >> 31
        r = (x >> 11) / float(1 << 53) or 1.0 / (1 << 53)
        return int(math.floor(-math.log(r) / math.log(m)))
    for i in (0, 1, 2, 12345, 99999):
        assert lv[i] == ref(i, 32)


def test_structure():
    rng = np.random.default_rng(3)
    n, dim, m = 800, 12, 6
    base = rng.standard_normal((n, dim)).astype(np.float32)
    l0, upper, entry = o.hnsw_build(base, dim, m=m, ef=40, max_batch=16, growth_div=8)
    lv, rows, top = o.hnsw_layout(n, m)
    assert len(upper) == top and lv[entry] == top
    assert entry == int(np.argmax(lv))  # the first node that reaches the top level
    for i in range(n):
        row = l0[i]
        k = int((row != INVALID).sum())
        assert (row[:k] != INVALID).all() and (row[k:] == INVALID).all()   # terminated, no holes
        assert len(set(row[:k].tolist())) == k and i not in row[:k]       # no duplicates, no self loop
    assert (l0 != INVALID).sum(1).min() >= 1
    for l, (slot, adj) in enumerate(upper):
        members = np.nonzero(lv >= l + 1)[0]
        assert np.array_equal(np.nonzero(slot != INVALID)[0], members)
        ids = adj[adj != INVALID]
        assert np.isin(ids, members).all()    # a level's links stay inside the level


def test_sequential_recall_reference_thresholds():
    # hnsw_test.go:43-60: 1000 x 16 uniform, M = 8, EF = 200 -> precision >= 0.99
    rng = np.random.default_rng(4711)
    base = rng.random((1000, 16)).astype(np.float32)
    g = o.hnsw_build(base, 16, m=8, ef=200, max_batch=1)
    h = o.HnswIndex(base, 16, *g, m=8)
    q = rng.random((100, 16)).astype(np.float32)
    gt = _brute(base, q, 10)
    hit = 0
    for qi in range(100):
        ids, _, _ = h.search(q[qi], 10, 200)
        hit += len(set(ids.tolist()) & set(gt[qi].tolist()))
    assert hit / 1000 >= 0.99


def test_batched_recall_close_to_sequential():
    rng = np.random.default_rng(11)
    base = rng.random((2000, 16)).astype(np.float32)
    q = rng.random((100, 16)).astype(np.float32)
    gt = _brute(base, q, 10)
    rec = []
    for mb in (1, 64):
        g = o.hnsw_build(base, 16, m=16, ef=128, max_batch=mb, growth_div=32)
        h = o.HnswIndex(base, 16, *g, m=16)
        hit = 0
        for qi in range(100):
            ids, _, _ = h.search(q[qi], 10, 128)
            hit += len(set(ids.tolist()) & set(gt[qi].tolist()))
        rec.append(hit / 1000)
    assert rec[0] >= 0.99 and rec[1] >= 0.99   # hnsw_test.go:76-84 (2000 x 16, M = 16, EF = 128)
