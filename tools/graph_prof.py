"""Graph searches for rocprofv3 passes: HNSW built by vg_hnsw_build on N x 768 i.i.d. normal rows, then
vg_search_hnsw (ef = 128), vg_search_hnsw_pq (ef = 128) and vg_search_vamana (PQ scoring) over 8192 queries, and the
SQ8 scan.  Prints the per-query counters the algorithmic-bytes model multiplies."""
import sys, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
import vecgo_amd as vg, bench

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
D, K, NQ = 768, 10, 8192
ctx = vg.Context(0); dev = torch.device("cuda", 0)
rows = bench.gen_rows(0, N, dev)
idx = vg.Index(ctx, N, D); idx.set_vectors(rows)
idx.build_hnsw(m=32, ef_construction=300, max_batch=8192, growth_div=32)
pq = vg.ProductQuantizer(ctx, D, 96, 256); pq.train(rows[:32768], iters=5, seed=1)
idx.set_pq_codes(pq, pq.encode(rows))
l0, _, entry = idx.get_hnsw_graph(); idx.set_vamana_graph(l0, entry)
q = bench.gen_queries(8, dev).reshape(-1, D)[:NQ].contiguous()
out = {}
for name, fn in (("hnsw_search", lambda s: idx.search_hnsw(q, K, 128, stats="full" if s else False)),
                 ("hnsw_search_pq", lambda s: idx.search_hnsw_pq(q, K, 128, stats="full" if s else False)),
                 ("vamana_search", lambda s: idx.search_vamana(q, K, kind=1, stats=s))):
    _, _, st = fn(True)
    for _ in range(3): fn(False)
    torch.cuda.synchronize()
    out[name] = {"distance_computations_per_launch": float(st[:, 1].sum()), "pops_per_launch": float(st[:, 3].sum())}
    if st.shape[1] > 4:
        out[name]["descent_distance_computations_per_launch"] = float(st[:, 4].sum())
print(json.dumps({"n": N, "nq": NQ, **out}))
