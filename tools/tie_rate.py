import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import vecgo_amd as vg, bench
N=1000000; D=768; NQ=2048
ctx = vg.Context(0); dev = torch.device("cuda", 0)
rows = bench.gen_rows(0, N, dev)
idx = vg.Index(ctx, N, D); idx.set_vectors(rows)
idx.build_hnsw(m=32, ef_construction=300, max_batch=8192, growth_div=32)
pq = vg.ProductQuantizer(ctx, D, 96, 256); pq.train(rows[:32768], iters=5, seed=1)
idx.set_pq_codes(pq, pq.encode(rows))
q = bench.gen_queries(8, dev).reshape(-1, D)[:NQ].contiguous()
for name, fn in (("f32", idx.search_hnsw), ("pq", idx.search_hnsw_pq)):
    for ef in (128, 512, 2048):
        ids, sc = fn(q, ef, ef)
        s = sc.cpu().numpy()
        dup_any = np.mean([(np.diff(np.sort(r)) == 0).any() for r in s])
        dup_top11 = np.mean([(np.diff(np.sort(r)[:11]) == 0).any() for r in s])
        print(f"{name} ef={ef}: queries with an equal-distance pair among the final ef results {dup_any:.3f}; among the 11 smallest {dup_top11:.4f}")
