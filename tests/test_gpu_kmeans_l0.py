"""internal/kmeans and the remaining batched L0 seams on the GPU vs the oracle / the compiled
reference golden vectors: bit-exact."""
import json

import numpy as np
import pytest

from oracle import oracle as o

pytestmark = pytest.mark.gpu


def bits(x):
    return np.asarray(x, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def vg():
    import vecgo_amd
    return vecgo_amd


@pytest.fixture(scope="module")
def ctx(vg):
    return vg.Context(0)


@pytest.mark.parametrize("n,dim,k,metric,iters", [(400, 16, 8, 0, 10), (1000, 128, 12, 0, 10),
                                                  (300, 100, 5, 0, 6), (500, 64, 7, 2, 10),
                                                  (500, 768, 4, 1, 5), (64, 8, 64, 0, 3),
                                                  # several 2048-point parts of the member sort, ragged centroid tile,
                                                  # odd n for the two-rows-per-group assignment
                                                  (5001, 128, 37, 0, 4), (4500, 1024, 9, 2, 3), (2500, 64, 130, 0, 3),
                                                  # more clusters than LDS counters: the walk-all member kernels
                                                  (6000, 8, 5000, 0, 2)])
def test_kmeans_train_matches_oracle(vg, ctx, n, dim, k, metric, iters):
    rng = np.random.default_rng(n + dim + k)
    x = rng.standard_normal((n, dim)).astype(np.float32)
    exp = o.kmeans_train(x, dim, k, metric, iters, seed=11)
    got = vg.kmeans_train(ctx, x, dim, k, metric, iters, seed=11)
    assert np.array_equal(bits(got.reshape(-1)), bits(exp))
    a = vg.kmeans_assign(ctx, x, got, dim, metric)
    assert np.array_equal(a, np.array([o.assign_partition(x[i], exp, dim, metric) for i in range(n)], np.int32))


def test_kmeans_reference_behaviour(vg, ctx):
    """kmeans_test.go:12-93."""
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal((50, 4)) * 0.1, rng.standard_normal((50, 4)) * 0.1 + 10]).astype(np.float32)
    c = vg.kmeans_train(ctx, x, 4, 2, max_iter=10, seed=3)
    assert sorted(np.round(c.mean(1)).tolist()) == [0.0, 10.0]
    assert vg.kmeans_train(ctx, x[:1], 4, 2) is None               # n < k -> (nil, nil)
    with pytest.raises(vg.VecgoHipError) as e:                      # bad metric -> error
        vg.kmeans_train(ctx, x, 4, 2, metric=3)
    assert e.value.status == -5
    cents = np.array([[0, 0], [1, 1], [5, 5], [10, 10]], np.float32)
    assert list(vg.find_closest_centroids(ctx, np.array([0.9, 0.9], np.float32), cents, 2, 2)) == [1, 0]
    for nprobe in (1, 3, 4, 9):
        rng2 = np.random.default_rng(nprobe)
        cc = rng2.standard_normal((40, 16)).astype(np.float32); q = rng2.standard_normal(16).astype(np.float32)
        for metric in (0, 2):
            assert list(vg.find_closest_centroids(ctx, q, cc, 16, nprobe, metric)) == \
                list(o.find_closest_centroids(q, cc, 16, nprobe, metric))


def test_bounded_batch_golden_and_kats(vg, ctx, golden_dir):
    g = np.load(golden_dir / "l0_ref.npz")   # outputs of the compiled bounded_l2_avx512.c
    off = g["pair_off"]; bi = 0
    for i in range(len(off) - 1):
        a = g["pair_a"][off[i]:off[i + 1]]; b = g["pair_b"][off[i]:off[i + 1]]
        for _ in range(4):
            if a.size:
                bound = float(g["bounded_bound"][bi])
                d, e = vg.squared_l2_bounded_batch(ctx, a, b, a.size, bound)
                assert bool(e[0]) == bool(g["bounded_exceeded"][bi])
                # the reference's value in both cases: the full sum, or the partial total of the block where it exits
                assert bits(d[0]) == bits(g["bounded_dist"][bi]), (a.size, bound)
                if e[0]:
                    assert d[0] >= bound   # floats_test.go:158-161
            bi += 1
    kats = json.loads((golden_dir / "reference_kats.json").read_text())
    for c in kats["squared_l2_bounded"]["cases"]:
        if "n" in c:
            a = np.full(c["n"], c["fill_a"], np.float32); b = np.full(c["n"], c["fill_b"], np.float32)
        else:
            a = np.array(c["a"], np.float32); b = np.array(c["b"], np.float32)
        d, e = vg.squared_l2_bounded_batch(ctx, a, b, a.size, c["bound"])
        assert bool(e[0]) == c["exceeded"]
        if not e[0]:
            assert abs(float(d[0]) - c["dist"]) <= 1e-4
    rng = np.random.default_rng(3)
    q = rng.standard_normal(768).astype(np.float32); t = rng.standard_normal((50, 768)).astype(np.float32)
    bounds = (rng.random(50) * 3000).astype(np.float32)
    d, e = vg.squared_l2_bounded_batch(ctx, q, t, 768, bounds)
    partials = 0
    for i in range(50):
        full, _ = o.l2_bounded(q, t[i], 1e30)
        want, exc = o.l2_bounded(q, t[i], float(bounds[i]))
        assert bits(d[i]) == bits(want) and bool(e[i]) == bool(exc) == bool(full > bounds[i])
        partials += bits(want) != bits(full)
    assert partials > 5      # the early exit's partial sums are exercised, per-row bounds
    # ragged dims and rows that are not 16-byte aligned, shared bound
    for dim in (67, 100, 130, 200):
        q = rng.standard_normal(dim).astype(np.float32); t = rng.standard_normal((33, dim)).astype(np.float32)
        for bound in (0.5 * dim, 1.5 * dim, 2.5 * dim):
            d, e = vg.squared_l2_bounded_batch(ctx, q, t, dim, bound)
            for i in range(33):
                want, exc = o.l2_bounded(q, t[i], bound)
                assert bits(d[i]) == bits(want) and bool(e[i]) == bool(exc), (dim, bound, i)


# m % 16 == 0: table in LDS, rows turned through LDS (adc_lookup_batch_lds_kernel; m = 96 its own instance, m = 128
# fewer waves); other m: one lane per row.  Ragged last tile, several workgroups
@pytest.mark.parametrize("m", [16, 32, 96, 112, 128, 100, 144])
@pytest.mark.parametrize("n", [1, 63, 65, 1500])
def test_adc_lookup_batch_matches_oracle(vg, ctx, m, n):
    rng = np.random.default_rng(m * 31 + n)
    table = rng.standard_normal(m * 256).astype(np.float32)
    codes = rng.integers(0, 256, (n, m), dtype=np.uint8)
    got = vg.pq_adc_lookup_batch(ctx, table, codes, m)
    want = np.array([o.adc(table, codes[i], m) for i in range(n)], np.float32)
    assert np.array_equal(bits(got), bits(want))


def test_adc_lookup_batch_golden_and_kats(vg, ctx, golden_dir):
    g = np.load(golden_dir / "l0_ref.npz")   # outputs of the compiled pqAdcLookupAvx512
    to = co = 0
    for i, m in enumerate(g["adc_m"]):
        m = int(m)
        got = vg.pq_adc_lookup_batch(ctx, g["adc_table"][to:to + m * 256], g["adc_codes"][co:co + m], m)
        assert bits(got[0]) == bits(g["adc_out"][i]), f"m={m}"
        to += m * 256; co += m
    kats = json.loads((golden_dir / "reference_kats.json").read_text())
    for c in kats["pq_adc"]["cases"]:
        m = c["m"]
        t = np.zeros(m * 256, np.float32)
        for i in range(m):
            for j in range(256):
                t[i * 256 + j] = (i * 256 + j) % 256 if c["table_rule"] == "i_mod_256" else i * 1000 + j
        assert vg.pq_adc_lookup_batch(ctx, t, np.array(c["codes"], np.uint8), m)[0] == np.float32(c["expected"])
