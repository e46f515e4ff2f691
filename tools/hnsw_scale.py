"""BASELINE config 3 at scale: HNSW layer-0 search, 1M x 768 L2, ef=128, on an exact kNN graph
built on the GPU with this library's own flat search (graph CONSTRUCTION is out of scope; a kNN
graph is the stand-in).  Prints QPS, recall@10 vs exact, distance computations and gathered GB/s."""
import sys, time, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
import vecgo_amd as vg, bench

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
EF = int(sys.argv[2]) if len(sys.argv) > 2 else 128
D, K, DEG = 768, 10, 32
ctx = vg.Context(0); dev = torch.device("cuda", 0)
rows = bench.gen_rows(0, N, dev)
idx = vg.Index(ctx, N, D); idx.set_vectors(rows)
t0 = time.time()
l0 = torch.empty((N, DEG), dtype=torch.int32, device=dev)
sc = torch.empty((4096, DEG), device=dev)
for s in range(0, N, 4096):
    e = min(N, s + 4096)
    idx.search_flat(rows[s:e], DEG, out=(l0[s:e], sc[:e - s]))
torch.cuda.synchronize()
print(f"kNN graph (k={DEG}) built in {time.time()-t0:.1f} s", flush=True)
l0 = l0.cpu().numpy().view(np.uint32)
self_col = l0 == np.arange(N, dtype=np.uint32)[:, None]
l0 = np.where(self_col, np.uint32(0xFFFFFFFF), l0)
# move the removed self entry to the end so lists stay contiguous
order = np.argsort(l0 == 0xFFFFFFFF, axis=1, kind="stable")
l0 = np.take_along_axis(l0, order, axis=1)
idx.set_hnsw_graph(l0, (), entry_point=0, m=16)
queries = bench.gen_queries(8, dev).reshape(-1, D)
gt, _ = idx.search_flat(queries[:1024], K)
gt = gt.cpu().numpy().view(np.uint32)
out = {}
for nq in (1024, 8192):
    q = queries[:nq]
    ids, scs, st = idx.search_hnsw(q, K, EF, stats=True)
    torch.cuda.synchronize(); t = time.perf_counter()
    reps = 3
    for _ in range(reps): idx.search_hnsw(q, K, EF)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / reps
    got = ids.cpu().numpy().view(np.uint32)
    rec = float(np.mean([len(set(got[i]) & set(gt[i])) / K for i in range(1024)]))
    dc = float(st[:, 1].mean()); pops = float(st[:, 3].mean())
    gb = st[:, 1].sum() * D * 4 / dt / 1e9
    out[nq] = dict(qps=nq / dt, ms=dt * 1e3, recall_at_10=rec, dist_comp_per_query=dc, pops_per_query=pops, gathered_GBps=float(gb))
    print(json.dumps({"nq": nq, **out[nq]}), flush=True)
