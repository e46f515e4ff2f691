"""kmeans.AssignPartition / TrainKMeans through the MFMA nomination + exact decision (k_kmeans.hip: km_gemm_kernel,
km_decide_kernel, km_assign_*_kernel<LIST>): every assignment equals the oracle's (squaredL2BatchAvx512 / dotBatchAvx512
order, strict compare, lowest index on ties) — on random data, where the matrix scores decide nearly everything, and on
inputs built so that they cannot (exact ties, near ties inside the error bound, duplicate centroids, non-finite values)."""
import numpy as np
import pytest

from oracle import oracle as o
from tests.hooks import set_hook

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vg():
    import vecgo_amd
    return vecgo_amd


@pytest.fixture(scope="module")
def ctx(vg):
    return vg.Context(0)


def check(vg, ctx, x, c, metric=0):
    got = vg.kmeans_assign(ctx, x, c, x.shape[1], metric)
    want = o.assign_partition_batch(x, c, metric)
    bad = np.nonzero(np.asarray(got) != want)[0]
    assert bad.size == 0, (bad[:8], np.asarray(got)[bad[:8]], want[bad[:8]])


@pytest.mark.parametrize("n,dim,k", [(8192, 768, 122), (5000, 128, 37), (4097, 96, 5), (6000, 100, 300), (9000, 64, 129),
                                     (4100, 36, 2), (5000, 1024, 128), (4096, 32, 257)])
@pytest.mark.parametrize("metric", [0, 2])
def test_random_rows(vg, ctx, n, dim, k, metric):
    rng = np.random.default_rng(n + dim + k + metric)
    x = rng.standard_normal((n, dim)).astype(np.float32)
    # centroids as a Lloyd iteration leaves them (means of many points: small norms) and as the first iteration has them
    # (rows of the corpus: norms like the points')
    for c in (x[rng.choice(n, k, replace=False)].copy(),
              (rng.standard_normal((k, dim)) * 0.05).astype(np.float32)):
        check(vg, ctx, x, c, metric)


def test_exact_ties_pick_the_lowest_index(vg, ctx):
    rng = np.random.default_rng(5)
    n, dim, k = 6000, 64, 40
    x = rng.standard_normal((n, dim)).astype(np.float32)
    c = rng.standard_normal((k, dim)).astype(np.float32)
    c[7] = c[31]                   # duplicate centroids: every point nearest to them is an exact tie
    c[12] = c[3]
    c[39] = c[0]
    for metric in (0, 1):
        check(vg, ctx, x, c, metric)
    # integer grid: many points at equal distance from several centroids
    xi = rng.integers(-2, 3, (n, 32)).astype(np.float32)
    ci = rng.integers(-2, 3, (20, 32)).astype(np.float32)
    for metric in (0, 2):
        check(vg, ctx, xi, ci, metric)


def test_near_ties_inside_the_error_bound(vg, ctx):
    """Centroid pairs one or a few ulps apart: the matrix scores cannot separate them, the reference's order does."""
    rng = np.random.default_rng(6)
    n, dim, k = 8000, 256, 24
    x = rng.standard_normal((n, dim)).astype(np.float32)
    c = rng.standard_normal((k, dim)).astype(np.float32) * 0.3
    for a, b, ulps in ((0, 1, 1), (2, 3, 3), (5, 4, 1), (10, 20, 17), (23, 22, 2)):
        c[b] = c[a]
        j = rng.integers(0, dim, 5)
        c[b, j] = (c[a, j].view(np.int32) + ulps).view(np.float32)
    for metric in (0, 2):
        check(vg, ctx, x, c, metric)
    # points ON a centroid, and points midway between two
    x[:24] = c
    x[24:48] = (c + np.roll(c, 1, axis=0)) * 0.5
    check(vg, ctx, x, c, 0)


def test_scales_and_degenerate_inputs(vg, ctx):
    rng = np.random.default_rng(7)
    n, dim, k = 5000, 128, 16
    for scale in (1e-20, 1e-6, 1e4, 1e15):
        x = (rng.standard_normal((n, dim)) * scale).astype(np.float32)
        c = x[rng.choice(n, k, replace=False)].copy()
        check(vg, ctx, x, c, 0)
        check(vg, ctx, x, c, 2)
    x = rng.standard_normal((n, dim)).astype(np.float32)
    c = x[:k].copy()
    check(vg, ctx, np.zeros((n, dim), np.float32), c, 0)       # every point the same
    check(vg, ctx, x, np.zeros((k, dim), np.float32), 0)       # every centroid the same: all ties -> 0
    x2 = x.copy()
    x2[17, 3] = np.nan
    x2[99, 0] = np.inf
    x2[100] = 3e19                                             # |x|^2 overflows
    check(vg, ctx, x2, c, 0)
    check(vg, ctx, x2, c, 2)
    c2 = c.copy()
    c2[5, 9] = np.nan                                          # a NaN centroid: never chosen (unless index 0)
    check(vg, ctx, x, c2, 0)
    c2 = c.copy()
    c2[0, 0] = np.nan                                          # NaN at index 0: the reference keeps 0 for every row
    check(vg, ctx, x, c2, 0)
    c2 = c.copy()
    c2[3] = np.inf
    check(vg, ctx, x, c2, 2)


def test_forced_paths_agree(vg, ctx):
    """The LIST kernels over every point (VG_KM_LIST_ALL) and the reference-order kernels alone (VG_KM_NO_MFMA)."""
    rng = np.random.default_rng(8)
    for n, dim, k in ((7000, 768, 50), (4500, 96, 9)):
        x = rng.standard_normal((n, dim)).astype(np.float32)
        c = x[rng.choice(n, k, replace=False)].copy()
        want = o.assign_partition_batch(x, c, 0)
        for hook in ("VG_KM_LIST_ALL", "VG_KM_NO_MFMA"):
            set_hook(hook, 1)
            try:
                assert np.array_equal(vg.kmeans_assign(ctx, x, c, dim, 0), want), hook
            finally:
                set_hook(hook, 0)


@pytest.mark.parametrize("hook", ["VG_KM_BF16"])
def test_bfloat16_split_passes(vg, ctx, hook):
    """The assignment pass on bfloat16 splits ([hi | lo | hi] x [hi | hi | lo]: what a training run of three or more
    iterations uses), forced for single assignments: the adversarial inputs above, unchanged answers."""
    set_hook(hook, 1)
    try:
        rng = np.random.default_rng(12)
        for n, dim, k in ((8192, 768, 122), (5000, 128, 37), (4200, 64, 300)):
            x = rng.standard_normal((n, dim)).astype(np.float32)
            for c in (x[rng.choice(n, k, replace=False)].copy(), (rng.standard_normal((k, dim)) * 0.05).astype(np.float32)):
                check(vg, ctx, x, c, 0)
                check(vg, ctx, x, c, 2)
        test_exact_ties_pick_the_lowest_index(vg, ctx)
        test_near_ties_inside_the_error_bound(vg, ctx)
        test_scales_and_degenerate_inputs(vg, ctx)
    finally:
        set_hook(hook, 0)


def test_train_at_a_size_the_matrix_path_serves(vg, ctx):
    rng = np.random.default_rng(9)
    for n, dim, k, metric in ((20000, 64, 16, 0), (9000, 128, 130, 2)):
        x = rng.standard_normal((n, dim)).astype(np.float32)
        exp = o.kmeans_train(x, dim, k, metric, 5, seed=4)
        got = vg.kmeans_train(ctx, x, dim, k, metric, 5, seed=4)
        assert np.array_equal(np.asarray(got).reshape(-1).view(np.uint32), np.asarray(exp, np.float32).view(np.uint32))


def test_device_rows_one_million(vg, ctx):
    """The bench's shape on device-resident rows; a sample of the rows against the oracle."""
    import torch
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    n, dim, k = 1_000_000, 768, 122
    x = torch.randn((n, dim), dtype=torch.float32, device="cuda", generator=g)
    c = vg.kmeans_train(ctx, x, dim, k, max_iter=2, seed=1)
    a = vg.kmeans_assign(ctx, x, c, dim).cpu().numpy()
    idx = np.random.default_rng(1).choice(n, 20000, replace=False)
    want = o.assign_partition_batch(x[torch.from_numpy(idx).cuda()].cpu().numpy(), c.cpu().numpy(), 0)
    assert np.array_equal(a[idx], want)
