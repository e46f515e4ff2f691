"""More of the reference's own tests as data (tests/golden/reference_kats.json: distance_dot, distance_squared_l2,
distance_normalize_l2 = distance/distance_test.go:11-117; quantization_int4_quantizer = internal/quantization/
int4_test.go:11-68; flat_pq_segment = internal/segment/flat/pq_test.go:17-93), run against any implementation of
  dot(a, b) -> float, l2(a, b) -> float           (None: the implementation has no one-pair form for that input)
  normalize(v) -> (vector, ok)
  int4(dim) -> object with train(rows), encode(vec) -> code bytes, decode(code) -> vector
  pq_search(rows, dim, m, k_centroids, query, k) -> (ids, scores)   PQ trained on `rows`, ADC search of the codes."""
import json
from pathlib import Path

import numpy as np

KATS = json.loads((Path(__file__).resolve().parent / "golden" / "reference_kats.json").read_text())


def _vec(c, key):
    if "fill" in c:
        return np.full(c["fill"]["n"], c["fill"][key], np.float32)
    return np.array(c[key], np.float32)


def run_distance(dot, l2):
    seen = 0
    for group, fn in (("distance_dot", dot), ("distance_squared_l2", l2)):
        g = KATS[group]
        for c in g["cases"]:
            got = fn(_vec(c, "a"), _vec(c, "b"))
            if got is None:
                continue
            assert abs(float(got) - c["expected"]) <= g["tol"], (group, c["name"], got)
            seen += 1
    return seen


def run_normalize(normalize):
    g = KATS["distance_normalize_l2"]
    seen = 0
    for c in g["cases"]:
        r = normalize(np.array(c["v"], np.float32))
        if r is None:
            continue
        v, ok = r
        assert bool(ok) == c["ok"], c["name"]
        if "expected" in c:
            assert np.all(np.abs(v - np.array(c["expected"], np.float32)) <= g["tol"]), (c["name"], v)
            assert abs(float(np.sqrt(np.float64(v[0] * v[0] + v[1] * v[1]))) - 1.0) <= g["tol"]   # :92
        if "expected_exact" in c:
            assert np.array_equal(v, np.array(c["expected_exact"], np.float32)), (c["name"], v)
        seen += 1
    return seen


def run_int4(int4):
    for c in KATS["quantization_int4_quantizer"]["cases"]:
        if c["name"] == "EncodeDecode":
            rows = np.random.default_rng(20260404).random((c["rows"], c["dim"]), np.float32)
            q = int4(c["dim"]); q.train(rows)
            code = np.asarray(q.encode(rows[0])).ravel()
            assert code.size == c["expect_code_len"] == (c["dim"] + 1) // 2
            dec = np.asarray(q.decode(code)).ravel()
            assert dec.size == c["dim"]
            diff = rows[0] - dec
            assert float(np.sum(diff * diff) / np.float32(c["dim"])) < c["expect_mse_below"]
        else:
            rows = np.array(c["train"], np.float32)
            q = int4(c["dim"]); q.train(rows)
            code = np.asarray(q.encode(rows[0])).ravel()
            assert code.size == c["expect_code_len"]
            dec = np.asarray(q.decode(code)).ravel()
            assert dec.size == c["dim"] and np.all(np.abs(dec - rows[0]) <= c["decode_tol"]), dec


def pq_segment_rows():
    g = KATS["flat_pq_segment"]
    rows = np.zeros((301, g["dim"]), np.float32)
    for i in range(300):
        for j in range(g["dim"]):
            rows[1 + i, j] = np.float32(i + j) * np.float32(0.01)
    return rows


def run_pq_segment(pq_search):
    g = KATS["flat_pq_segment"]
    rows = pq_segment_rows()
    ids, sc = pq_search(rows, g["dim"], g["m"], g["k_centroids"], np.zeros(g["dim"], np.float32), g["k"])
    ids, sc = np.asarray(ids), np.asarray(sc)
    assert ids.size == g["k"]
    best = int(np.argmin(sc))
    assert ids[best] == g["expect_best_row"], (ids, sc)
    assert sc[best] < g["expect_best_score_below"]
