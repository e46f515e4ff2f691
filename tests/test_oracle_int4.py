"""INT4 oracle pinned against the reference: golden vectors minted from the
compiled int4_avx512.c (tests/golden/int4_ref.npz), the live objects when oracle/_ref is present,
and the reference's own Int4Quantizer test (int4_test.go) restated."""
from pathlib import Path

import numpy as np
import pytest

from oracle import oracle as o

G = np.load(Path(__file__).parent / "golden" / "int4_ref.npz")


def bits(x):
    return np.asarray(x, np.float32).view(np.uint32)


def test_int4_kernels_match_reference_objects_golden():
    qo = co = to = oo = 0
    for i, (dim, n) in enumerate(zip(G["dim"], G["n"])):
        dim, n = int(dim), int(n)
        cs = (dim + 1) // 2
        q = G["q"][qo:qo + dim]; mn = G["min"][qo:qo + dim]; df = G["diff"][qo:qo + dim]
        codes = G["codes"][co:co + n * cs]; table = G["table"][to:to + dim * 16]
        iq = o.Int4Quantizer(dim); iq.set_params(mn, df)
        assert np.array_equal(bits(iq.table), bits(table)), dim
        assert np.array_equal(bits(iq.l2_distance_batch(q, codes)), bits(G["batch"][oo:oo + n])), dim
        assert bits(iq.l2_distance(q, codes[:cs])) == bits(G["precomputed"][i]), dim
        assert bits(iq.l2_distance_batch(q, codes[:cs]))[0] == bits(G["single"][i]), dim  # single == batch kernel
        qo += dim; co += n * cs; to += dim * 16; oo += n


def test_int4_live_against_ref_objects():
    ref = o.Ref()
    if not ref.ok or not hasattr(ref.lib, "int4L2DistanceBatchAvx512"):
        pytest.skip("oracle/_ref not built here")
    rng = np.random.default_rng(6)
    bad = 0
    for _ in range(300):
        dim = int(rng.integers(1, 900))
        q = rng.standard_normal(dim).astype(np.float32)
        mn = rng.standard_normal(dim).astype(np.float32); df = (rng.random(dim) * 2 + 0.05).astype(np.float32)
        iq = o.Int4Quantizer(dim); iq.set_params(mn, df)
        codes = rng.integers(0, 256, 2 * iq.code_size).astype(np.uint8)
        bad += not np.array_equal(bits(iq.l2_distance_batch(q, codes)), bits(ref.int4_l2_batch(q, codes, dim, mn, df)))
        bad += bits(iq.l2_distance(q, codes[:iq.code_size])) != bits(ref.int4_l2_precomputed(q, codes[:iq.code_size], iq.table))
    assert bad == 0


def test_int4_quantizer_reference_test():  # int4_test.go:11-68
    rng = np.random.default_rng(1)
    dim = 128
    x = rng.random((100, dim)).astype(np.float32)
    iq = o.Int4Quantizer(dim); iq.train(x)
    code = iq.encode(x[0])
    assert code.size == (dim + 1) // 2
    dec = iq.decode(code)
    assert np.mean((x[0] - dec) ** 2) < 0.01
    odd = o.Int4Quantizer(3); odd.train(np.array([[0.1, 0.5, 0.9]], np.float32))
    assert np.all(odd.diff == 1.0)                      # a single vector: diff 0 -> 1 (int4.go:53-58)
    c = odd.encode(np.array([0.1, 0.5, 0.9], np.float32))
    assert c.size == 2 and (c[1] & 0x0F) == 0           # the unused low nibble is zero
    d = odd.decode(c)
    assert np.all(np.abs(d - np.array([0.1, 0.5, 0.9], np.float32)) <= 0.1)


def test_int4_encode_layout_and_rounding():
    iq = o.Int4Quantizer(4); iq.set_params(np.zeros(4, np.float32), np.ones(4, np.float32))
    c = iq.encode(np.array([0.0, 1.0, 0.5, 2.0], np.float32))   # 0, 15, round(7.5)=8 (half away), clamp -> 15
    assert list(c) == [0x0F, 0x8F]
    assert list(iq.encode(np.array([-3.0, 0.1, 1 / 30, 0.0], np.float32))) == [0x02, 0x10]  # 0, round(1.5)=2 | round(0.5)=1, 0


def test_int4_distance_close_to_decoded_l2():
    rng = np.random.default_rng(2)
    dim = 100
    x = rng.standard_normal((50, dim)).astype(np.float32)
    iq = o.Int4Quantizer(dim); iq.train(x)
    q = rng.standard_normal(dim).astype(np.float32)
    codes = iq.encode_batch(x)
    d_batch = iq.l2_distance_batch(q, codes)
    for i in range(50):
        want = float(np.sum((q.astype(np.float64) - iq.decode(codes[i]).astype(np.float64)) ** 2))
        assert abs(d_batch[i] - want) <= 1e-4 * max(1.0, want)
        assert abs(iq.l2_distance(q, codes[i]) - want) <= 1e-4 * max(1.0, want)


def test_simd_int4_reference_tests(golden_dir):
    """internal/simd/int4_test.go as data (tests/golden/reference_kats.json simd_int4) against the oracle's three INT4
    distance forms and BuildInt4LookupTable."""
    import json
    g = json.loads((golden_dir / "reference_kats.json").read_text())["simd_int4"]
    for c in g["cases"]:
        dim = c["dim"]
        iq = o.Int4Quantizer(dim); iq.set_params(np.array(c["min"], np.float32), np.array(c["diff"], np.float32))
        q = np.array(c["query"], np.float32)
        codes = np.array(c["codes"], np.uint8)
        batch = iq.l2_distance_batch(q, codes.reshape(-1))
        one = np.array([iq.l2_distance(q, code) for code in codes], np.float32)
        if c.get("all_nonnegative"):
            assert np.all(batch >= 0) and np.all(one >= 0)
        for got in (batch, one):
            for v, e in zip(got, c.get("expected", [])):
                if e is not None:
                    assert abs(float(v) - e) <= c["tol"], (c["name"], v, e)
        if "table_len" in c:
            assert iq.table.size == c["table_len"]
            assert np.all(np.abs(one - batch) <= c["precomputed_vs_direct_tol"])
    for t in g["lookup_table"]:
        iq = o.Int4Quantizer(len(t["min"])); iq.set_params(np.array(t["min"], np.float32), np.array(t["diff"], np.float32))
        for e in t["expect"]:
            assert abs(float(iq.table[e["dim"] * 16 + e["q"]]) - e["value"]) <= t["tol"], e
