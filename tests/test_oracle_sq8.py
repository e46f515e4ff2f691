"""SQ8 oracle pinned against the reference: golden vectors minted from
the compiled sq8_avx512.c (tests/golden/sq8_ref.npz), the live objects when oracle/_ref is
present, and the reference's own ScalarQuantizer tests (quantizer_test.go) restated."""
from pathlib import Path

import numpy as np
import pytest

from oracle import oracle as o

G = np.load(Path(__file__).parent / "golden" / "sq8_ref.npz")


def bits(x):
    return np.asarray(x, np.float32).view(np.uint32)


def test_sq8u_batch_matches_reference_objects_golden():
    qo = co = oo = 0
    for dim, n in zip(G["dim"], G["n"]):
        dim, n = int(dim), int(n)
        q = G["q"][qo:qo + dim]; mn = G["mins"][qo:qo + dim]; iv = G["inv"][qo:qo + dim]
        codes = G["codes"][co:co + n * dim]
        want = G["out"][oo:oo + n]
        got = o.sq8u_l2_batch(q, codes, mn, iv, dim)
        assert np.array_equal(bits(got), bits(want)), (dim, n)
        qo += dim; co += n * dim; oo += n


def test_sq8u_batch_live_against_ref_objects():
    ref = o.Ref()
    if not ref.ok or not hasattr(ref.lib, "sq8uL2BatchPerDimensionAvx512"):
        pytest.skip("oracle/_ref not built here")
    rng = np.random.default_rng(5)
    bad = 0
    for _ in range(400):
        dim = int(rng.integers(1, 900)); n = int(rng.integers(1, 6))
        q = rng.standard_normal(dim).astype(np.float32)
        mn = rng.standard_normal(dim).astype(np.float32)
        iv = (rng.random(dim) * 0.05).astype(np.float32)
        codes = rng.integers(0, 256, n * dim).astype(np.uint8)
        bad += not np.array_equal(bits(o.sq8u_l2_batch(q, codes, mn, iv, dim)),
                                  bits(ref.sq8u_l2_batch(q, codes, mn, iv, dim)))
    assert bad == 0


def test_scalar_quantizer_train_kat():  # quantizer_test.go:8-37
    sq = o.ScalarQuantizer(3)
    sq.train(np.array([[-1.0, 0.0, 1.0], [-0.5, 0.5, 2.0], [-2.0, 1.0, 3.0]], np.float32))
    assert sq.mins[0] == -2.0 and sq.maxs[0] == -0.5 and sq.mins[2] == 1.0 and sq.maxs[2] == 3.0


def _manual(dim, lo, hi):
    sq = o.ScalarQuantizer(dim)
    sq.mins[:] = lo; sq.maxs[:] = hi
    sq.scales[:] = np.float32(255.0) / np.float32(hi - lo)
    sq.inv_scales[:] = np.float32(hi - lo) / np.float32(255.0)
    sq.trained = True
    return sq


def test_scalar_quantizer_encode_decode_error_bound():  # quantizer_test.go:39-86
    sq = _manual(5, -1.0, 1.0)
    orig = np.array([-1.0, -0.5, 0.0, 0.5, 1.0], np.float32)
    code = sq.encode(orig)
    assert code[0] == 0 and code[4] == 255
    dec = sq.decode(code)
    assert np.max(np.abs(dec - orig)) <= (2.0 / 255.0) * 1.1


def test_scalar_quantizer_uniform_and_clamping():  # quantizer_test.go:113-179
    sq = o.ScalarQuantizer(3)
    sq.train(np.full((2, 3), 5.0, np.float32))
    assert np.all(sq.maxs > sq.mins)
    assert np.all(np.abs(sq.decode(sq.encode(np.full(3, 5.0, np.float32))) - 5.0) < 0.01)
    sq = _manual(3, 0.0, 1.0)
    dec = sq.decode(sq.encode(np.array([-1.0, 0.5, 2.0], np.float32)))
    assert dec[0] >= -0.01 and dec[2] <= 1.01
    assert sq.encode(np.array([-1.0, 0.5, 2.0], np.float32))[0] == 0


def test_scalar_quantizer_l2_distance_batch():  # quantizer_test.go:230-271
    sq = _manual(4, 0.0, 10.0)
    q = np.array([1, 2, 3, 4], np.float32)
    codes = np.concatenate([sq.encode(q), sq.encode(np.array([2, 3, 4, 5], np.float32))])
    out = o.sq8u_l2_batch(q, codes, sq.mins, sq.inv_scales, 4)
    assert out[0] <= 0.1 and abs(out[1] - 4.0) <= 0.2


def test_sq8u_matches_generic_within_reference_tolerance():  # floats_test.go:471-500 (5e-2)
    rng = np.random.default_rng(9)
    for dim in [1, 7, 8, 15, 16, 17, 31, 32, 33]:
        q = (rng.random(dim) * 2 - 1).astype(np.float32)
        mn = (rng.random(dim) * 2 - 1).astype(np.float32)
        iv = (rng.random(dim) * 2 - 1).astype(np.float32)
        for n in [1, 2, 5]:
            codes = rng.integers(0, 256, n * dim).astype(np.uint8)
            got = o.sq8u_l2_batch(q, codes, mn, iv, dim)
            c = codes.reshape(n, dim).astype(np.float32)
            want = np.sum((q - (mn + c * iv)) ** 2, axis=1)
            assert np.all(np.abs(got - want) <= 5e-2 * np.maximum(1.0, np.abs(want)))


def test_flat_search_sq8_orders_by_score_then_row():
    rng = np.random.default_rng(3)
    dim, n, k = 24, 300, 7
    x = rng.standard_normal((n, dim)).astype(np.float32)
    x[40] = x[11]  # identical codes: tie resolved by RowID
    sq = o.ScalarQuantizer(dim); sq.train(x)
    codes = sq.encode_batch(x)
    q = x[11] + 0.01
    ids, sc = o.flat_search_sq8(sq, codes, q, k)
    d = o.sq8u_l2_batch(q, codes, sq.mins, sq.inv_scales, dim)
    order = np.lexsort((np.arange(n), d))[:k]
    assert np.array_equal(ids, order.astype(np.uint32)) and np.array_equal(bits(sc), bits(d[order]))
    assert list(ids[:2]) == [11, 40]
