import sys, os
from pathlib import Path
sys.path.insert(0, "/root/repo")
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
import vecgo_amd as vg, bench
N=1000000; D,K,NQ=768,10,8192
ctx = vg.Context(0); dev = torch.device("cuda", 0)
rows = bench.gen_rows(0, N, dev)
idx = vg.Index(ctx, N, D); idx.set_vectors(rows)
idx.build_hnsw(m=32, ef_construction=300, max_batch=8192, growth_div=32)
pq = vg.ProductQuantizer(ctx, D, 96, 256); pq.train(rows[:32768], iters=5, seed=1)
idx.set_pq_codes(pq, pq.encode(rows))
q = bench.gen_queries(8, dev).reshape(-1, D)[:NQ].contiguous()
for name, fn in (("f32", idx.search_hnsw), ("pq", idx.search_hnsw_pq)):
    for ef in (128, 512, 2048):
        k = K if name == "f32" else ef
        fn(q, k, ef)
        _, _, st = fn(q, k, ef, stats="full")
        pops = st[:, 3].astype(np.float64)
        lo = (st[:, [0, 1, 2]] & 0xFFFFFFFF).astype(np.float64); hi = (st[:, [0, 1, 2]] >> 32).astype(np.float64)
        P = pops.sum(); npush = hi[:, 2].sum()
        print(f"{name} ef={ef}: pops/query {pops.mean():.1f}; shader-clock cycles per pop: pop {lo[:, 0].sum() / P:.0f} adj+visited {lo[:, 1].sum() / P:.0f} "
              f"score {lo[:, 2].sum() / P:.0f} push loop {st[:, 4].sum() / P:.0f}; pushed nodes per pop {npush / P:.1f}, per pushed node: "
              f"cand push {hi[:, 0].sum() / npush:.0f} res push {hi[:, 1].sum() / npush:.0f} cycles", flush=True)
