"""Exact fp32 kernels on the GPU vs the oracle (= the reference's AVX-512 order): bit-exact.

Mirrors internal/simd/floats_test.go:195-276 (boundary sizes, batch dims) with the
tolerance tightened to bit equality, plus Segment.Rerank."""
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


@pytest.mark.parametrize("dim", [1, 3, 4, 7, 8, 15, 16, 17, 31, 32, 33, 63, 64, 65, 80, 100, 128,
                                 200, 768, 777, 1536])
def test_batch_kernels_bit_exact(vg, ctx, dim):
    rng = np.random.default_rng(dim)
    for n in (1, 5, 17, 200):
        q = rng.standard_normal(dim).astype(np.float32)
        t = rng.standard_normal(n * dim).astype(np.float32)
        assert np.array_equal(bits(vg.squared_l2_batch(ctx, q, t, dim)), bits(o.l2_batch(q, t, dim)))
        assert np.array_equal(bits(vg.dot_batch(ctx, q, t, dim)), bits(o.dot_batch(q, t, dim)))


def test_batch_kernels_golden_fixture(vg, ctx, golden_dir):
    """Directly against outputs of the reference's compiled batch_avx512.c."""
    g = np.load(golden_dir / "l0_ref.npz")
    qo = to = oo = 0
    for dim, n in zip(g["batch_dim"], g["batch_n"]):
        dim, n = int(dim), int(n)
        q = g["batch_q"][qo:qo + dim]; t = g["batch_t"][to:to + n * dim]
        assert np.array_equal(bits(vg.squared_l2_batch(ctx, q, t, dim)), bits(g["batch_l2"][oo:oo + n]))
        assert np.array_equal(bits(vg.dot_batch(ctx, q, t, dim)), bits(g["batch_dot"][oo:oo + n]))
        qo += dim; to += n * dim; oo += n


def test_batch_kats(vg, ctx, golden_dir):
    kats = json.loads((golden_dir / "reference_kats.json").read_text())
    for c in kats["squared_l2"]["cases"]:
        a = np.array(c["a"], np.float32); b = np.array(c["b"], np.float32)
        assert vg.squared_l2_batch(ctx, a, b, a.size)[0] == np.float32(c["expected"])
    for c in kats["dot"]["cases"]:
        a = np.array(c["a"], np.float32); b = np.array(c["b"], np.float32)
        if a.size:
            assert vg.dot_batch(ctx, a, b, a.size)[0] == np.float32(c["expected"])
    # empty input succeeds (kernels_amd64.go:291-297)
    assert vg.squared_l2_batch(ctx, np.zeros(4, np.float32), np.zeros(0, np.float32), 4).size == 0


@pytest.mark.parametrize("dim,metric", [(768, 0), (128, 0), (100, 0), (777, 0), (768, 2), (65, 2)])
def test_score_candidates_and_rerank(vg, ctx, dim, metric):
    rng = np.random.default_rng(dim + metric)
    n, nq, nc, k = 3000, 7, 150, 10
    base = rng.standard_normal((n, dim)).astype(np.float32)
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    idx = vg.Index(ctx, n, dim, vg.Metric(metric))
    idx.set_vectors(base)
    cand = np.stack([rng.permutation(n)[:nc] for _ in range(nq)]).astype(np.uint32)
    cand[0, 5] = 0xFFFFFFFF  # invalid ids are skipped
    sc = idx.score_candidates(q, cand)
    ids, scores = idx.rerank(q, cand, k)
    for qi in range(nq):
        valid = cand[qi] != 0xFFFFFFFF
        exp = o.rerank_f32(base, dim, q[qi], cand[qi][valid], metric)
        assert np.array_equal(bits(sc[qi][valid]), bits(exp))
        order = sorted(range(exp.size), key=lambda i: ((-exp[i] if metric else exp[i]), cand[qi][valid][i]))[:k]
        assert np.array_equal(ids[qi], cand[qi][valid][order])
        assert np.array_equal(bits(scores[qi]), bits(exp[order]))


def test_rerank_fewer_candidates_than_k(vg, ctx):
    rng = np.random.default_rng(5)
    base = rng.standard_normal((50, 64)).astype(np.float32)
    idx = vg.Index(ctx, 50, 64)
    idx.set_vectors(base)
    q = rng.standard_normal((1, 64)).astype(np.float32)
    ids, sc = idx.rerank(q, np.array([[3, 9, 0xFFFFFFFF, 20]], np.uint32), 10)
    assert set(ids[0, :3].tolist()) == {3, 9, 20} and np.all(ids[0, 3:] == 0xFFFFFFFF)
    assert np.all(np.isinf(sc[0, 3:]))
    with pytest.raises(vg.VecgoHipError) as e:
        vg.Index(ctx, 5, 64).rerank(q, np.zeros((1, 4), np.uint32), 2)
    assert e.value.status == -9


@pytest.mark.parametrize("dim,metric,nc,k", [(128, 0, 700, 100), (96, 2, 300, 256), (64, 0, 90, 128)])
def test_rerank_more_than_64_results(vg, ctx, dim, metric, nc, k):
    """k > 64: the exact keys are sorted in LDS; a row listed twice is reported twice (the reference's loop
    re-scores whatever it is handed, flat/segment.go:757-779)."""
    rng = np.random.default_rng(dim + k)
    n, nq = 2000, 4
    base = rng.standard_normal((n, dim)).astype(np.float32)
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    idx = vg.Index(ctx, n, dim, vg.Metric(metric))
    idx.set_vectors(base)
    cand = np.stack([rng.permutation(n)[:nc] for _ in range(nq)]).astype(np.uint32)
    cand[:, 7] = cand[:, 3]          # a duplicate candidate
    cand[1, 11] = 0xFFFFFFFF
    ids, scores = idx.rerank(q, cand, k)
    for qi in range(nq):
        uniq = cand[qi][cand[qi] != 0xFFFFFFFF]
        exp = o.rerank_f32(base, dim, q[qi], uniq, metric)
        order = sorted(range(exp.size), key=lambda i: ((-exp[i] if metric else exp[i]), uniq[i]))[:k]
        r = len(order)
        assert np.array_equal(ids[qi, :r], uniq[order])
        assert np.array_equal(bits(scores[qi, :r]), bits(exp[order]))
        assert np.all(ids[qi, r:] == 0xFFFFFFFF)
