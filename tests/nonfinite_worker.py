"""Worker of tests/test_gpu_nan.py::test_no_entry_point_faults_on_non_finite_input: every search entry point with NaN / Inf
in queries and in rows.  A memory fault kills this process (and would kill a pytest session), so it runs as a child; for
each entry point it prints `name ok` after checking that the FINITE queries of the batch still get the oracle's answer on
the clean corpus (a query's answer does not depend on its batch)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import vecgo_amd as vg
from oracle import oracle as o
from tests import graphs

ctx = vg.Context(0)
rng = np.random.default_rng(5)
n, dim, k = 3000, 64, 10
base = rng.standard_normal((n, dim)).astype(np.float32)
q = rng.standard_normal((16, dim)).astype(np.float32)
odd = [1, 4, 7, 9, 12]
q[1, 3] = np.nan
q[4] = np.nan
q[7, 0] = np.inf
q[9, 5] = -np.inf
q[12, 0] = 3e38
fin = [i for i in range(16) if i not in odd]


def bits(x):
    return np.asarray(x, np.float32).view(np.uint32)


def finite_rows_match(name, got, want_fn):
    ids, sc = got[0], got[1]
    for qi in fin:
        eid, esc = want_fn(q[qi])
        assert np.array_equal(np.asarray(ids)[qi, :len(eid)], eid), (name, qi)
        assert np.array_equal(bits(np.asarray(sc)[qi, :len(eid)]), bits(esc)), (name, qi)
    print(name, "ok", flush=True)


idx = vg.Index(ctx, n, dim)
idx.set_vectors(base)
finite_rows_match("flat", idx.search_flat(q, k), lambda v: o.flat_search_f32(base, dim, v, k))
finite_rows_match("flat_one_query", (np.concatenate([idx.search_flat(q[i:i + 1], k)[0] for i in range(16)]),
                                     np.concatenate([idx.search_flat(q[i:i + 1], k)[1] for i in range(16)])),
                  lambda v: o.flat_search_f32(base, dim, v, k))
finite_rows_match("flat_k100", idx.search_flat(q, 100), lambda v: o.flat_search_f32(base, dim, v, 100))

opq = o.ProductQuantizer(dim, 8, 256)
opq.train(base, iters=2, seed=3)
codes = opq.encode_batch(base)
pq = vg.ProductQuantizer(ctx, dim, 8, 256)
pq.set_codebooks(opq.codebooks, opq.scales, opq.offsets)
idx.set_pq_codes(pq, codes)
finite_rows_match("pq_adc", idx.search_pq_adc(q, k), lambda v: o.flat_search_pq(opq, codes, v, k))

rcodes = o.rabitq_encode_batch(base, dim)
idx.set_rabitq_codes(rcodes)
finite_rows_match("rabitq", idx.search_rabitq(q, k), lambda v: o.flat_search_rabitq(rcodes, dim, v, k))

sq = vg.ScalarQuantizer(ctx, dim)
sq.train(base)
idx.set_sq8_codes(sq, sq.encode(base))
idx.search_sq8(q, k)
print("sq8 ok", flush=True)

l0, upper, entry = graphs.build_hnsw(base, m=8, seed=1)
idx.set_hnsw_graph(l0, upper, entry, m=8)
oh = o.HnswIndex(base, dim, l0, upper, entry)
finite_rows_match("hnsw", idx.search_hnsw(q, k, 64), lambda v: oh.search(v, k, 64)[:2])
finite_rows_match("hnsw_split_heaps", idx.search_hnsw(q, k, 700), lambda v: oh.search(v, k, 700)[:2])
idx.search_hnsw_pq(q, k, 64)
print("hnsw_pq ok", flush=True)
for mode in (0, 1):
    finite_rows_match(f"brute_{mode}", idx.search_hnsw_brute(q, k, mode), lambda v, mode=mode: oh.brute_search(v, k, mode, None))
g, ventry = graphs.build_vamana(base, r=16, seed=2)
idx.set_vamana_graph(g, ventry)
for kind in (0, 1, 2):
    idx.search_vamana(q, k, kind=kind)
    print(f"vamana_{kind} ok", flush=True)
idx.rerank(q, rng.integers(0, n, (16, 40)).astype(np.uint32), k)
print("rerank ok", flush=True)

# non-finite ROWS (a fresh index: set_vectors computes norms of them)
bad = base.copy()
bad[5, 2] = np.nan
bad[900] = np.inf
bad[2000, 7] = -np.inf
bad[17] = 3e38
idx2 = vg.Index(ctx, n, dim)
idx2.set_vectors(bad)
idx2.search_flat(q, k)
idx2.search_flat(q[:1], k)
print("flat_bad_rows ok", flush=True)
idx2.set_hnsw_graph(l0, upper, entry, m=8)
idx2.search_hnsw(q, k, 64)
for mode in (0, 1):
    idx2.search_hnsw_brute(q, k, mode)
print("graph_bad_rows ok", flush=True)
idx2.set_vamana_graph(g, ventry)
idx2.search_vamana(q, k, kind=0)
print("vamana_bad_rows ok", flush=True)
vg.kmeans_train(ctx, bad, dim, 8, max_iter=3, seed=1)
vg.kmeans_assign(ctx, bad, base[:8].copy(), dim)
pq2 = vg.ProductQuantizer(ctx, dim, 8, 256)
pq2.train(bad, iters=2, seed=1)
pq2.encode(bad)
print("build_side_bad_rows ok", flush=True)
print("ALL OK", flush=True)
