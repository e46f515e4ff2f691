"""Small deterministic graphs for the search-loop tests (host-side numpy; graph CONSTRUCTION is
out of scope and non-deterministic in the reference — parity is defined on scoring + heap
semantics over a given graph, SURVEY.md §7 'Hard parts')."""
import numpy as np

INVALID = np.uint32(0xFFFFFFFF)


def knn(base, ids, deg):
    """Exact deg nearest neighbours (L2) of every node in `ids` among `ids` (excluding itself)."""
    sub = base[ids].astype(np.float64)
    d = ((sub[:, None, :] - sub[None, :, :]) ** 2).sum(-1) if len(ids) <= 1500 else None
    if d is None:
        sq = (sub ** 2).sum(1)
        d = sq[:, None] + sq[None, :] - 2 * sub @ sub.T
    np.fill_diagonal(d, np.inf)
    order = np.argsort(d, axis=1, kind="stable")[:, :deg]
    return ids[order]


def build_hnsw(base, m=8, seed=0, ragged=True):
    """HNSW-shaped adjacency: layer 0 = 2m exact nearest neighbours (some lists cut short to
    exercise the terminator), upper layers over geometric random subsets."""
    n = base.shape[0]
    rng = np.random.default_rng(seed)
    m0 = 2 * m
    all_ids = np.arange(n)
    l0 = np.full((n, m0), INVALID, np.uint32)
    nb = knn(base, all_ids, min(m0, n - 1))
    l0[:, :nb.shape[1]] = nb
    if ragged:
        for i in rng.choice(n, size=max(1, n // 10), replace=False):
            l0[i, rng.integers(1, m0):] = INVALID
    levels = np.minimum((-np.log(rng.random(n)) * (1.0 / np.log(m))).astype(int), 4)
    upper = []
    max_level = int(levels.max())
    for L in range(1, max_level + 1):
        members = all_ids[levels >= L]
        slot = np.full(n, INVALID, np.uint32)
        slot[members] = np.arange(len(members), dtype=np.uint32)
        adj = np.full((len(members), m), INVALID, np.uint32)
        if len(members) > 1:
            nbm = knn(base, members, min(m, len(members) - 1))
            adj[:, :nbm.shape[1]] = nbm
        upper.append((slot, adj))
    entry = int(np.argmax(levels))
    return l0, upper, entry


def build_vamana(base, r=16, seed=0):
    n = base.shape[0]
    rng = np.random.default_rng(seed)
    g = np.full((n, r), INVALID, np.uint32)
    near = knn(base, np.arange(n), min(r - 4, n - 1))
    g[:, :near.shape[1]] = near
    for i in range(n):  # a few random long edges, and holes in the middle of some lists
        g[i, near.shape[1]:near.shape[1] + 2] = rng.choice(n, 2, replace=False)
        if i % 7 == 0:
            g[i, rng.integers(0, near.shape[1])] = INVALID
    sq = ((base - base.mean(0)) ** 2).sum(1)
    return g, int(np.argmin(sq))
