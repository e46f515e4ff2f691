"""distance/distance_test.go, internal/quantization/int4_test.go and internal/segment/flat/pq_test.go as data
(tests/extra_kats.py) against the oracle."""
import numpy as np

from oracle import oracle as o
from tests import extra_kats


def test_distance_package_tests():
    assert extra_kats.run_distance(o.dot, o.l2) == 11      # Empty included: 0 for no elements (kernels_amd64.go:291-297)
    assert extra_kats.run_normalize(o.normalize_l2) == 5


def test_int4_quantizer_test():
    extra_kats.run_int4(o.Int4Quantizer)


def test_flat_pq_segment_test():
    def pq_search(rows, dim, m, kc, q, k):
        pq = o.ProductQuantizer(dim, m, kc); pq.train(rows, iters=20, seed=1)
        return o.flat_search_pq(pq, pq.encode_batch(rows), q, k)
    extra_kats.run_pq_segment(pq_search)
