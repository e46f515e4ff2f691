"""OptimizedProductQuantizer on the GPU vs the oracle's restatement of opq.go / svd.go: rotation, encode,
decode, asymmetric distance and the whole of Train (rotations and codebooks) bit for bit."""
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


def _random_rotations(rng, nblocks, block):
    return np.stack([o.procrustes(rng.standard_normal((block, block))) for _ in range(nblocks)])


@pytest.mark.parametrize("dim,m,k", [(768, 96, 256), (128, 16, 64), (64, 8, 16), (48, 6, 16), (96, 12, 256)])
def test_rotate_encode_decode_distance(vg, ctx, dim, m, k):
    rng = np.random.default_rng(dim + m)
    opq = vg.OptimizedProductQuantizer(ctx, dim, m, k, num_iterations=1)
    ref = o.OptimizedProductQuantizer(dim, m, k, num_iterations=1)
    assert (opq.block, opq.nblocks) == (ref.block, ref.nblocks)
    rot = _random_rotations(rng, ref.nblocks, ref.block)
    sd = dim // m
    cb = rng.integers(-128, 128, m * k * sd).astype(np.int8)
    sc = (rng.random(m) * 0.02 + 0.005).astype(np.float32); of = (rng.standard_normal(m) * 0.1).astype(np.float32)
    ref.rotations = rot; ref.pq.set_codebooks(cb, sc, of)
    with pytest.raises(vg.VecgoHipError) as e:
        opq.encode(np.zeros((1, dim), np.float32))
    assert "not trained" in e.value.message
    opq.set_rotations(rot); opq.pq.set_codebooks(cb, sc, of)
    x = rng.standard_normal((40, dim)).astype(np.float32)
    r = opq.rotate(x)
    codes = opq.encode(x)
    dec = opq.decode(codes)
    q = rng.standard_normal(dim).astype(np.float32)
    d = opq.asymmetric_distance(q, codes)
    for i in range(40):
        assert np.array_equal(bits(r[i]), bits(ref.rotate(x[i]))), i
        assert np.array_equal(codes[i], ref.encode(x[i])), i
        assert np.array_equal(bits(dec[i]), bits(ref.decode(codes[i]))), i
        assert bits(d[i]) == bits(ref.asym_distance(q, codes[i])), i


@pytest.mark.parametrize("n,dim,m,k,iters", [(700, 32, 4, 16, 3), (900, 128, 16, 32, 2), (1200, 768, 96, 64, 2)])
def test_train_matches_oracle(vg, ctx, n, dim, m, k, iters):
    rng = np.random.default_rng(n)
    mix = rng.standard_normal((dim, dim)).astype(np.float32) / np.sqrt(dim)
    x = (rng.standard_normal((n, dim)).astype(np.float32) * np.linspace(2, 0.2, dim, dtype=np.float32)) @ mix
    ref = o.OptimizedProductQuantizer(dim, m, k, num_iterations=iters)
    ref.train(x, pq_iters=6, seed=11)
    opq = vg.OptimizedProductQuantizer(ctx, dim, m, k, num_iterations=iters)
    opq.train(x, pq_iters=6, seed=11)
    assert opq.is_trained
    assert np.array_equal(bits(opq.rotations()), bits(ref.rotations))
    cb, sc, of = opq.pq.codebooks()
    assert np.array_equal(cb, ref.pq.codebooks) and np.array_equal(bits(sc), bits(ref.pq.scales)) and np.array_equal(bits(of), bits(ref.pq.offsets))
    for b in range(opq.nblocks):
        rb = opq.rotations()[b]
        assert np.allclose(rb @ rb.T, np.eye(opq.block), atol=1e-3)     # opq_test.go:56-99 orthogonality
    # the reference's own reconstruction bar (opq_test.go:133-192): OPQ error <= 2 x PQ error
    pq = vg.ProductQuantizer(ctx, dim, m, k); pq.train(x, iters=6, seed=11)
    e_pq = float(((pq.decode(pq.encode(x)) - x) ** 2).sum(1).mean())
    e_opq = float(((opq.decode(opq.encode(x)) - x) ** 2).sum(1).mean())
    assert e_opq <= 2.0 * e_pq
