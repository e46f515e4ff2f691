"""BinaryQuantizer and NormalizeL2InPlace on the GPU vs the oracle: codes, decodes, Hamming distances and
normalized rows bit for bit; the trained threshold equal on every input here (see k_binary.hip on why the
contract for Train is 1 ulp)."""
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


@pytest.mark.parametrize("n,dim", [(1, 8), (37, 128), (200, 768), (64, 100), (5, 1), (33, 1536), (1000, 65)])
def test_binary_quantizer(vg, ctx, n, dim):
    rng = np.random.default_rng(n + dim)
    x = (rng.standard_normal((n, dim)) * 2 + 0.3).astype(np.float32)
    bq = vg.BinaryQuantizer(ctx, dim)
    bq.train(x)
    want_th = o.binary_train(x, dim)
    assert abs(int(bq.threshold.view(np.int32)) - int(want_th.view(np.int32))) <= 1
    assert bq.threshold == want_th
    codes = bq.encode(x)
    nw = (dim + 63) // 64
    assert codes.shape == (n, nw * 8)
    for i in range(n):
        assert np.array_equal(codes[i].view(np.uint64), o.binary_encode_u64(x[i], float(want_th))), i
    dec = bq.decode(codes)
    for i in range(0, n, max(1, n // 7)):
        assert np.array_equal(dec[i], o.binary_decode(codes[i], dim, float(want_th)))
    q = rng.standard_normal(dim).astype(np.float32)
    h = bq.compute_hamming_distance(q, codes)
    qc = o.binary_encode_u64(q, float(want_th)).view(np.uint8)
    for i in range(n):
        assert int(h[i]) == o.hamming(qc, codes[i])


def test_binary_reference_kats(vg, ctx):
    # binary_test.go:9-39: alternating signs -> 0x5555...; :64-123 threshold example
    v = np.where(np.arange(128) % 2 == 0, 1.0, -1.0).astype(np.float32)
    bq = vg.BinaryQuantizer(ctx, 128)
    w = bq.encode(v[None]).view(np.uint64)[0]
    assert w[0] == 0x5555555555555555 and w[1] == 0x5555555555555555
    bq4 = vg.BinaryQuantizer(ctx, 4)
    bq4.train(np.arange(1, 9, dtype=np.float32).reshape(2, 4))
    assert bq4.threshold == np.float32(4.5) and bq4.trained
    with pytest.raises(ValueError):
        vg.BinaryQuantizer(ctx, 4).train(np.zeros((0, 4), np.float32))


@pytest.mark.parametrize("n,dim", [(1, 2), (50, 768), (17, 100), (300, 64), (9, 3)])
def test_normalize_l2(vg, ctx, n, dim):
    rng = np.random.default_rng(dim)
    x = (rng.standard_normal((n, dim)) * 3).astype(np.float32)
    if n > 4:
        x[3] = 0.0                      # zero norm: reported, left untouched
    got = x.copy()
    ok = vg.normalize_l2(ctx, got, dim)
    for i in range(n):
        want, wok = o.normalize_l2(x[i])
        assert bool(ok[i]) == wok
        assert np.array_equal(got[i].view(np.uint32), want.view(np.uint32)), i
    # device buffers are normalized in place
    import torch
    t = torch.from_numpy(x).cuda()
    ok2 = vg.normalize_l2(ctx, t, dim)
    torch.cuda.synchronize()
    assert np.array_equal(t.cpu().numpy().view(np.uint32), got.view(np.uint32))
    assert np.array_equal(ok2.cpu().numpy(), np.asarray(ok))
