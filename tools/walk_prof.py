"""One graph search per run, for rocprofv3 passes: the graph vg_hnsw_build makes of N x 768 i.i.d. normal rows, then
ONE of  f32 EF | pq EF | vamana_pq | vamana_rabitq  over 8192 queries (3 calls).  Prints the per-query counters the
algorithmic-bytes model multiplies.  argv: N MODE [EF]."""
import sys, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
import vecgo_amd as vg, bench

N = int(sys.argv[1]); mode = sys.argv[2]; ef = int(sys.argv[3]) if len(sys.argv) > 3 else 128
D, K, NQ = 768, 10, 8192
ctx = vg.Context(0); dev = torch.device("cuda", 0)
rows = bench.gen_rows(0, N, dev)
idx = vg.Index(ctx, N, D); idx.set_vectors(rows)
idx.build_hnsw(m=32, ef_construction=300, max_batch=8192, growth_div=32)
if mode != "f32":
    pq = vg.ProductQuantizer(ctx, D, 96, 256); pq.train(rows[:32768], iters=5, seed=1)
    idx.set_pq_codes(pq, pq.encode(rows))
    idx.set_rabitq_codes(vg.RaBitQuantizer(ctx, D).encode(rows))
    l0, _, entry = idx.get_hnsw_graph(); idx.set_vamana_graph(l0, entry)
q = bench.gen_queries(8, dev).reshape(-1, D)[:NQ].contiguous()
fn = {"f32": lambda s: idx.search_hnsw(q, K, ef, stats="full" if s else False),
      "pq": lambda s: idx.search_hnsw_pq(q, ef, ef, stats="full" if s else False),
      "vamana_pq": lambda s: idx.search_vamana(q, K, kind=1, stats=s),
      "vamana_rabitq": lambda s: idx.search_vamana(q, K, kind=2, stats=s)}[mode]
_, _, st = fn(True)
for _ in range(2): fn(False)
torch.cuda.synchronize()
print(json.dumps({"n": N, "nq": NQ, "mode": mode, "ef": ef, "distance_computations": float(st[:, 1].sum()),
                  "pops": float(st[:, 3].sum()),
                  "descent_distance_computations": float(st[:, 4].sum()) if st.shape[1] > 4 else 0.0}))
