"""BinaryQuantizer / NormalizeL2InPlace restatements against the reference tests' own known answers
(internal/quantization/binary_test.go, distance/distance_test.go)."""
import numpy as np

from oracle import oracle as o


def test_train_mean_kat():
    # binary_test.go:41-63: mean of 1..8 = 4.5
    v = np.arange(1, 9, dtype=np.float32).reshape(2, 4)
    assert o.binary_train(v, 4) == np.float32(4.5)


def test_threshold_kat_and_decode():
    # binary_test.go:64-123: threshold 0.5 over [0.1, 0.9, 0.5, 0.4, 0.6, 0.3, 0.8, 0.2] -> bits where v >= 0.5
    v = np.array([0.1, 0.9, 0.5, 0.4, 0.6, 0.3, 0.8, 0.2], np.float32)
    w = o.binary_encode_u64(v, 0.5)
    want = sum(1 << i for i, x in enumerate(v) if x >= np.float32(0.5))
    assert int(w[0]) == want
    dec = o.binary_decode(w.view(np.uint8), 8, 0.5)
    assert np.array_equal(dec, np.where(v >= 0.5, np.float32(1.0), np.float32(0.0)))   # threshold +- 0.5
    # a code shorter than the dimension decodes its missing bits as 0 (binary.go:179)
    assert np.array_equal(o.binary_decode(np.zeros(0, np.uint8), 3, 2.0), np.full(3, 1.5, np.float32))


def test_normalize_kats():
    # distance_test.go: [3, 4] -> [0.6, 0.8]; zero vector -> false and untouched
    out, ok = o.normalize_l2([3.0, 4.0])
    assert ok and np.allclose(out, [0.6, 0.8], atol=1e-7)
    out, ok = o.normalize_l2([0.0, 0.0, 0.0])
    assert not ok and np.array_equal(out, np.zeros(3, np.float32))
    out, ok = o.normalize_l2(np.zeros(0, np.float32))
    assert not ok
    rng = np.random.default_rng(0)
    v = rng.standard_normal(768).astype(np.float32)
    out, ok = o.normalize_l2(v)
    assert ok and abs(float(np.dot(out.astype(np.float64), out.astype(np.float64))) - 1.0) < 1e-6
    # the steps as written: inv = 1 / float32(sqrt(float64(dot))), v *= inv
    inv = np.float32(1.0) / np.float32(np.sqrt(np.float64(o.dot(v, v))))
    assert np.array_equal(out, v * inv)
