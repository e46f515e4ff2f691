"""distance/distance_test.go, internal/quantization/int4_test.go and internal/segment/flat/pq_test.go as data
(tests/extra_kats.py) through the C ABI — and, beyond the reference's tolerances, bit-equal to the oracle."""
import numpy as np
import pytest

from oracle import oracle as o
from tests import extra_kats

pytestmark = pytest.mark.gpu
bits = lambda x: np.asarray(x, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def vg():
    import vecgo_amd
    return vecgo_amd


@pytest.fixture(scope="module")
def ctx(vg):
    return vg.Context(0)


def test_distance_package_tests(vg, ctx):
    # the batch forms are what the C ABI has (one query against n rows); no elements: the reference returns 0 without
    # calling a kernel (kernels_amd64.go:291-297) and the batch form has nothing to return
    def dot(a, b):
        return vg.dot_batch(ctx, a, b, a.size)[0] if a.size else None

    def l2(a, b):
        return vg.squared_l2_batch(ctx, a, b, a.size)[0] if a.size else None
    assert extra_kats.run_distance(dot, l2) == 9

    def normalize(v):
        if v.size == 0:
            return None
        rows = v.copy()[None, :]
        ok = vg.normalize_l2(ctx, rows, v.size)
        assert np.array_equal(bits(rows[0]), bits(o.normalize_l2(v)[0]))
        return rows[0], bool(ok[0])
    assert extra_kats.run_normalize(normalize) == 4


def test_int4_quantizer_test(vg, ctx):
    class Q:
        def __init__(self, dim):
            self.q = vg.Int4Quantizer(ctx, dim); self.ref = o.Int4Quantizer(dim); self.dim = dim

        def train(self, rows):
            self.q.train(rows); self.ref.train(rows)

        def encode(self, vec):
            code = np.asarray(self.q.encode(vec[None, :]))[0]
            assert np.array_equal(code, self.ref.encode(vec))
            return code

        def decode(self, code):
            dec = np.asarray(self.q.decode(np.asarray(code, np.uint8)[None, :]))[0]
            assert np.array_equal(bits(dec), bits(self.ref.decode(code)))
            return dec
    extra_kats.run_int4(Q)


def test_flat_pq_segment_test(vg, ctx):
    def pq_search(rows, dim, m, kc, q, k):
        pq = vg.ProductQuantizer(ctx, dim, m, kc); pq.train(rows, iters=20, seed=1)
        ref = o.ProductQuantizer(dim, m, kc); ref.train(rows, iters=20, seed=1)
        codes = np.asarray(pq.encode(rows))
        assert np.array_equal(codes, ref.encode_batch(rows))
        idx = vg.Index(ctx, rows.shape[0], dim); idx.set_vectors(rows); idx.set_pq_codes(pq, codes)
        ids, sc = idx.search_pq_adc(q[None, :], k)
        eids, esc = o.flat_search_pq(ref, codes, q, k)
        assert np.array_equal(ids[0], eids) and np.array_equal(bits(sc[0]), bits(esc))
        return ids[0], sc[0]
    extra_kats.run_pq_segment(pq_search)
