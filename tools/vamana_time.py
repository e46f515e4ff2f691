"""vamana_search_kernel time per launch (8192 queries in flight, k = 10) for each node scorer — fp32 rows, PQ codes
(ComputeAsymmetricDistance order), RaBitQ codes — over the layer 0 (R = 64) of a graph built by vg_hnsw_build on
N x 768 i.i.d. normal rows.  argv: [N].  Library: VECGO_HIP_LIB."""
import sys, os
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
idx.set_rabitq_codes(vg.RaBitQuantizer(ctx, D).encode(rows))
l0, _, entry = idx.get_hnsw_graph(); idx.set_vamana_graph(l0, entry)
q = bench.gen_queries(8, dev).reshape(-1, D)[:NQ].contiguous()
st = torch.cuda.current_stream()
for kind, name in ((0, "fp32"), (1, "pq"), (2, "rabitq")):
    ids, _, stats = idx.search_vamana(q, K, kind=kind, stats=True, stream=st)
    torch.cuda.synchronize()
    ctx.profile_read("vamana_search"); ctx.profile_enable(True)
    for _ in range(3): idx.search_vamana(q, K, kind=kind, stream=st)
    torch.cuda.synchronize()
    l, ms = ctx.profile_read("vamana_search"); ctx.profile_enable(False)
    dc = float(stats[:, 1].sum())
    print(f"{os.environ.get('VECGO_HIP_LIB', 'default'):28s} N={N} {name:7s}: {ms / l:7.3f} ms per {NQ} queries, "
          f"{dc / NQ:6.0f} node scores and {float(stats[:, 3].sum()) / NQ:5.1f} pops per query, "
          f"{dc / (ms / l * 1e-3) / 1e9:6.2f} G scores/s, ids checksum {int(ids.to(torch.int64).sum())}")
