"""ProductQuantizer on the GPU vs the oracle: Train (same seeded stream → identical
codebooks), Encode, Decode, ComputeAsymmetricDistance, and the reference's own
quantizer-level assertions (internal/quantization/pq_test.go:10-128)."""
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


def unit_vectors(rng, n, dim):
    v = rng.standard_normal((n, dim)).astype(np.float32)
    return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)


@pytest.mark.parametrize("n,dim,m,k,iters", [
    (1000, 128, 8, 256, 20),    # pq_test.go:11-17 shape
    (600, 64, 8, 16, 20),       # small k: converges, exercises the early `break`
    (300, 32, 4, 256, 5),       # n barely above k
    (100, 32, 4, 256, 5),       # n < k: pq.go:285-291 path
    (500, 768, 96, 64, 3),      # BASELINE sub-dim 8, many sub-quantizers
    (400, 256, 2, 32, 4),       # sub-dim 128 >= 64: 4x16 accumulator order in training
    (600, 192, 2, 256, 3),      # sub-dim 96 x 256 centroids: 96 KiB of centroids in LDS (generic kernels)
    (9001, 32, 4, 64, 3),       # several 4096-point chunks of the k-means++ sum chain, ragged tail
    (8192, 24, 2, 32, 2),       # whole chunks only; sub-dim 12 takes the generic kernels
])
def test_train_matches_oracle_bit_for_bit(vg, ctx, n, dim, m, k, iters):
    rng = np.random.default_rng(n + dim + m)
    x = unit_vectors(rng, n, dim)
    opq = o.ProductQuantizer(dim, m, k)
    opq.train(x, iters=iters, seed=42)
    pq = vg.ProductQuantizer(ctx, dim, m, k)
    assert not pq.is_trained()
    pq.train(x, iters=iters, seed=42)
    assert pq.is_trained()
    cb, sc, of = pq.codebooks()
    assert np.array_equal(bits(sc), bits(opq.scales))
    assert np.array_equal(bits(of), bits(opq.offsets))
    assert np.array_equal(cb, opq.codebooks)


def test_train_handles_duplicates_and_empty_clusters(vg, ctx):
    """Many identical points → zero running sum (pq.go:306-310) and empty clusters (:408-411)."""
    rng = np.random.default_rng(3)
    x = np.repeat(rng.standard_normal((5, 32)).astype(np.float32), 40, axis=0)
    opq = o.ProductQuantizer(32, 4, 16)
    opq.train(x, iters=6, seed=9)
    pq = vg.ProductQuantizer(ctx, 32, 4, 16)
    pq.train(x, iters=6, seed=9)
    cb, sc, of = pq.codebooks()
    assert np.array_equal(cb, opq.codebooks) and np.array_equal(bits(sc), bits(opq.scales))
    assert np.array_equal(bits(of), bits(opq.offsets))


@pytest.mark.parametrize("dim,m,k", [(128, 8, 256), (768, 96, 256), (100, 25, 256), (64, 1, 256),
                                     (96, 3, 17)])
def test_encode_decode_asym_bit_exact(vg, ctx, dim, m, k):
    rng = np.random.default_rng(dim + m + k)
    sd = dim // m
    opq = o.ProductQuantizer(dim, m, k)
    opq.set_codebooks(rng.integers(-128, 128, m * k * sd).astype(np.int8),
                      (rng.random(m) * 0.02 + 0.005).astype(np.float32),
                      ((rng.random(m) * 2 - 1) * 0.1).astype(np.float32))
    pq = vg.ProductQuantizer(ctx, dim, m, k)
    pq.set_codebooks(opq.codebooks, opq.scales, opq.offsets)
    x = rng.standard_normal((300, dim)).astype(np.float32)
    codes = pq.encode(x)
    assert np.array_equal(codes, opq.encode_batch(x))
    dec = pq.decode(codes)
    for i in range(0, 300, 37):
        assert np.array_equal(bits(dec[i]), bits(opq.decode(codes[i])))
    q = rng.standard_normal(dim).astype(np.float32)
    asym = pq.asymmetric_distance(q, codes)
    exp = np.array([opq.asym_distance(q, codes[i]) for i in range(300)], np.float32)
    assert np.array_equal(bits(asym), bits(exp))


def test_reference_quantizer_assertions(vg, ctx):
    """pq_test.go:10-128: MSE < 0.5 on unit vectors d=128 M=8; ADC(q, code) within 1e-3 of
    L2(q, Decode(code)); untrained quantizer errors."""
    rng = np.random.default_rng(0)
    x = unit_vectors(rng, 1000, 128)
    pq = vg.ProductQuantizer(ctx, 128, 8, 256)
    with pytest.raises(vg.VecgoHipError) as e:
        pq.encode(x[:1])
    assert e.value.status == -3
    pq.train(x, iters=20, seed=1)
    t = unit_vectors(rng, 2, 128)
    codes = pq.encode(t[:1])
    assert codes.shape == (1, 8)
    rec = pq.decode(codes)
    assert rec.shape == (1, 128)
    assert float(np.mean((t[0] - rec[0]) ** 2)) < 0.5
    adc = float(pq.asymmetric_distance(t[1], codes)[0])
    full = float(np.sum((t[1].astype(np.float32) - rec[0]) ** 2, dtype=np.float32))
    assert abs(adc - full) <= 1e-3
    with pytest.raises(vg.VecgoHipError):
        pq.encode(np.zeros(100, np.float32))  # dimension mismatch


def test_pq_train_by_subquantizer_range_equals_full_training(vg, ctx):
    """vg_pq_train_subset (the multi-GPU partition of Train, pq.go:83-138): training [0,5) and
    [5,12) separately and assembling gives the codebooks of one vg_pq_train call, bit for bit."""
    rng = np.random.default_rng(77)
    dim, m, k = 48, 12, 64
    x = rng.standard_normal((1500, dim)).astype(np.float32)
    full = vg.ProductQuantizer(ctx, dim, m, k)
    full.train(x, iters=8, seed=3)
    fcb, fsc, fof = full.codebooks()
    part = vg.ProductQuantizer(ctx, dim, m, k)
    part.train_subset(x, 5, 7, iters=8, seed=3)
    assert not part.is_trained()
    part.train_subset(x, 0, 5, iters=8, seed=3)
    cb, sc, of = part.codebooks_range(0, m)
    assert np.array_equal(cb, fcb)
    assert np.array_equal(sc.view(np.uint32), fsc.view(np.uint32))
    assert np.array_equal(of.view(np.uint32), fof.view(np.uint32))
    part.set_codebooks(cb, sc, of)
    assert part.is_trained()
    with pytest.raises(vg.VecgoHipError):
        part.train_subset(x, 10, 5)
