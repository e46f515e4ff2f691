"""HNSW built by vg_hnsw_build on the BASELINE corpus (N x 768 i.i.d. normal, M = 32, EF = 300), then the
recall / cost frontier of vg_search_hnsw over it.  usage: hnsw_build_scale.py [N] [max_batch] [ef_c] [efs...]"""
import sys, time, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
import vecgo_amd as vg, bench

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
MAXB = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
EFC = int(sys.argv[3]) if len(sys.argv) > 3 else 300
EFS = [int(x) for x in sys.argv[4:]] or [128, 256, 512, 1024, 2048]
D, K = 768, 10
ctx = vg.Context(0); dev = torch.device("cuda", 0)
rows = bench.gen_rows(0, N, dev)
idx = vg.Index(ctx, N, D); idx.set_vectors(rows)
ctx.profile_enable(True)
torch.cuda.synchronize(); t0 = time.time()
idx.build_hnsw(m=32, ef_construction=EFC, max_batch=MAXB, growth_div=32)
torch.cuda.synchronize(); bt = time.time() - t0
prof = {k: ctx.profile_read(k) for k in ("hnsw_build_search", "hnsw_build_select", "hnsw_build_link")}
ctx.profile_enable(False)
print(json.dumps({"n": N, "max_batch": MAXB, "ef_construction": EFC, "build_s": bt, "stages": prof}), flush=True)
queries = bench.gen_queries(1, dev).reshape(-1, D)[:1024].contiguous()
gt, _ = idx.search_flat(queries, K)
gt = gt.cpu().numpy().view(np.uint32)
for ef in EFS:
    ids, scs, st = idx.search_hnsw(queries, K, ef, stats=True)
    torch.cuda.synchronize(); t = time.perf_counter()
    idx.search_hnsw(queries, K, ef)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    got = ids.cpu().numpy().view(np.uint32)
    rec = float(np.mean([len(set(got[i]) & set(gt[i])) / K for i in range(1024)]))
    print(json.dumps({"ef": ef, "recall_at_10": rec, "ms_per_1024": dt * 1e3, "qps": 1024 / dt,
                      "dist_comp": float(st[:, 1].mean()), "pops": float(st[:, 3].mean())}), flush=True)
