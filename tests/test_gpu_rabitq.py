"""RaBitQ + Hamming on the GPU vs the oracle (rabitq.go:51-176, popcount_avx512.c:25-46):
codes byte-identical, distances and scan results bit-exact; reference KATs for Hamming."""
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


@pytest.mark.parametrize("dim", [8, 64, 100, 128, 768, 777, 1536])
def test_encode_and_distance_bit_exact(vg, ctx, dim):
    rng = np.random.default_rng(dim)
    rq = vg.RaBitQuantizer(ctx, dim)
    assert rq.bytes_total() == o.rabitq_code_bytes(dim) == ((dim + 63) // 64) * 8 + 4
    x = rng.standard_normal((200, dim)).astype(np.float32)
    x[3, :5] = 0.0          # zeros count as >= 0 (rabitq.go:67)
    x[4, 0] = -0.0
    codes = rq.encode(x)
    assert np.array_equal(codes, o.rabitq_encode_batch(x, dim))
    q = rng.standard_normal(dim).astype(np.float32)
    d = rq.distance(q, codes)
    exp = np.array([o.rabitq_distance(q, codes[i]) for i in range(200)], np.float32)
    assert np.array_equal(bits(d), bits(exp))
    assert np.all(d >= 0)   # rabitq_test.go:62-97


@pytest.mark.parametrize("n,dim,nq,k", [(10000, 768, 4, 10), (777, 128, 3, 10), (50, 64, 2, 64),
                                        (5, 100, 1, 10), (30000, 768, 300, 10)])
def test_rabitq_scan_matches_oracle(vg, ctx, n, dim, nq, k):
    rng = np.random.default_rng(n + dim)
    x = rng.standard_normal((n, dim)).astype(np.float32)
    codes = o.rabitq_encode_batch(x, dim)
    idx = vg.Index(ctx, n, dim)
    idx.set_rabitq_codes(codes)
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    ids, sc = idx.search_rabitq(q, k)
    for qi in range(0, nq, max(1, nq // 7)):
        eid, esc = o.flat_search_rabitq(codes, dim, q[qi], k)
        r = eid.size
        assert np.array_equal(ids[qi, :r], eid)
        assert np.array_equal(bits(sc[qi, :r]), bits(esc))
        assert np.all(ids[qi, r:] == 0xFFFFFFFF)


def test_hamming_kats_and_random(vg, ctx, golden_dir):
    kats = json.loads((golden_dir / "reference_kats.json").read_text())
    for c in kats["hamming_bytes"]["cases"]:
        if c["a"]:
            got = vg.hamming_batch(ctx, np.array(c["a"], np.uint8), np.array(c["b"], np.uint8))
            assert int(got[0]) == c["expected"]
    g = np.load(golden_dir / "l0_ref.npz")   # outputs of the compiled popcount_avx512.c
    off = 0
    for i, n in enumerate(g["ham_n"]):
        got = vg.hamming_batch(ctx, g["ham_a"][off:off + n], g["ham_b"][off:off + n])
        assert int(got[0]) == int(g["ham_out"][i])
        off += n
    rng = np.random.default_rng(1)
    a = rng.integers(0, 256, 96).astype(np.uint8); codes = rng.integers(0, 256, (500, 96)).astype(np.uint8)
    got = vg.hamming_batch(ctx, a, codes)
    assert np.array_equal(got, np.array([o.hamming(a, codes[i]) for i in range(500)], np.int32))


def test_rabitq_errors(vg, ctx):
    idx = vg.Index(ctx, 10, 64)
    with pytest.raises(vg.VecgoHipError) as e:
        idx.search_rabitq(np.zeros((1, 64), np.float32), 3)
    assert e.value.status == -9
    rq = vg.RaBitQuantizer(ctx, 64)
    with pytest.raises(vg.VecgoHipError):  # rabitq.go:52-54
        rq.encode(np.zeros(63, np.float32))
    with pytest.raises(vg.VecgoHipError):  # rabitq.go:123-125
        rq.distance(np.zeros(64, np.float32), np.zeros(11, np.uint8))


@pytest.mark.parametrize("n,dim,nq,k", [(3000, 128, 4, 100), (700, 768, 2, 300), (100, 64, 3, 128)])
def test_scan_pages_beyond_64_results(vg, ctx, n, dim, nq, k):
    """k > 64: one scan per page of 64 results, each page after the previous page's last key.  RaBitQ
    distances collide a lot (few distinct hamming counts x norms), so pages regularly split a run of ties."""
    rng = np.random.default_rng(n + k)
    x = rng.standard_normal((n, dim)).astype(np.float32)
    x[10:90] = x[5]                                  # 81 identical codes: one tie run longer than a page
    codes = vg.RaBitQuantizer(ctx, dim).encode(x)
    idx = vg.Index(ctx, n, dim)
    idx.set_rabitq_codes(codes)
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    q[0] = x[5]
    ids, sc = idx.search_rabitq(q, k)
    for i in range(nq):
        eid, esc = o.flat_search_rabitq(codes, dim, q[i], k)
        r = eid.size
        assert np.array_equal(ids[i, :r], eid), (i, np.flatnonzero(ids[i, :r] != eid)[:5])
        assert np.array_equal(np.asarray(sc[i, :r]).view(np.uint32), esc.view(np.uint32))
        assert np.all(ids[i, r:] == 0xFFFFFFFF)
    with pytest.raises(vg.VecgoHipError):
        idx.search_rabitq(q, 513)
