"""Construction-time neighbour selection on the GPU (robustPrune, selectNeighborsHeuristic) vs the
oracle: kept ids in order and counts, for every node of a batch."""
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


def _knn(base, ids, k):
    d = ((base[ids][:, None, :] - base[None, :, :]) ** 2).sum(-1)
    return np.argsort(d, axis=1)[:, :k].astype(np.uint32)


@pytest.mark.parametrize("n,dim,nc,r,alpha,metric", [(600, 64, 80, 16, 1.2, 0), (400, 100, 200, 32, 1.0, 0),
                                                      (300, 768, 70, 64, 1.2, 0), (500, 48, 33, 8, 2.0, 2),
                                                      (200, 17, 5, 4, 1.2, 0)])
def test_robust_prune_matches_oracle(vg, ctx, n, dim, nc, r, alpha, metric):
    rng = np.random.default_rng(n + dim + nc)
    base = rng.standard_normal((n, dim)).astype(np.float32)
    base[7] = base[3]                                    # duplicate vectors: distance ties, id decides
    nodes = rng.choice(n, 40, replace=False).astype(np.uint32)
    near = _knn(base, nodes, max(2, nc // 2))
    cands = np.full((nodes.size, nc), 0xFFFFFFFF, np.uint32)
    for i in range(nodes.size):
        row = np.concatenate([near[i], rng.integers(0, n, nc - near.shape[1] - 2).astype(np.uint32),
                              [nodes[i]], [near[i][0]]])  # incl. the node itself and a duplicate
        rng.shuffle(row)
        cands[i, :row.size] = row[:nc]
    idx = vg.Index(ctx, n, dim, vg.Metric(metric)); idx.set_vectors(base)
    kept, cnt = idx.robust_prune(nodes, cands, r, alpha)
    for i in range(nodes.size):
        want = o.robust_prune(base, dim, int(nodes[i]), cands[i], r, alpha, metric)
        assert cnt[i] == want.size, i
        assert np.array_equal(kept[i, :cnt[i]], want), (i, kept[i], want)
        assert np.all(kept[i, cnt[i]:] == 0xFFFFFFFF)


@pytest.mark.parametrize("n,dim,nc,m,metric", [(600, 64, 60, 16, 0), (300, 768, 40, 32, 0), (500, 32, 10, 16, 0),
                                               (400, 96, 50, 8, 2), (400, 64, 45, 12, 1)])
def test_hnsw_select_neighbors_matches_oracle(vg, ctx, n, dim, nc, m, metric):
    rng = np.random.default_rng(n + dim + nc + metric)
    base = rng.standard_normal((n, dim)).astype(np.float32)
    if metric == 1:
        base /= np.linalg.norm(base, axis=1, keepdims=True)
    nn = 30
    src = rng.choice(n, nn, replace=False)
    ids = np.full((nn, nc), 0xFFFFFFFF, np.uint32)
    dists = np.zeros((nn, nc), np.float32)
    for i in range(nn):
        cand = rng.choice(n, nc - (i % 3), replace=False)           # some lists end in padding
        if metric == 2:
            d = np.array([-o.dot(base[c], base[src[i]]) for c in cand], np.float32)
        else:
            d = np.array([o.l2(base[c], base[src[i]]) for c in cand], np.float32)
            if metric == 1:
                d = (np.float32(0.5) * d).astype(np.float32)
        order = np.lexsort((cand, d))
        ids[i, :cand.size] = cand[order]; dists[i, :cand.size] = d[order]
    idx = vg.Index(ctx, n, dim, vg.Metric(metric)); idx.set_vectors(base)
    kept, cnt = idx.hnsw_select_neighbors(ids, dists, m)
    for i in range(nn):
        ln = int(np.sum(ids[i] != 0xFFFFFFFF))
        want = o.hnsw_select_neighbors(base, dim, ids[i, :ln], dists[i, :ln], m, metric)
        assert cnt[i] == want.size, i
        assert np.array_equal(kept[i, :cnt[i]], want), (i, kept[i], want)


def test_build_limits(vg, ctx):
    idx = vg.Index(ctx, 10, 8); idx.set_vectors(np.zeros((10, 8), np.float32))
    with pytest.raises(vg.VecgoHipError):
        idx.robust_prune(np.zeros(1, np.uint32), np.zeros((1, 2000), np.uint32), 8)
    with pytest.raises(vg.VecgoHipError):
        idx.hnsw_select_neighbors(np.zeros((1, 4), np.uint32), np.zeros((1, 4), np.float32), 1000)
