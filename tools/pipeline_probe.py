"""Dev probe: 1M x 768 pipeline timings and recall vs refine factor."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
import vecgo_amd as vg

N, D, M, K = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 768, 96, 10
ctx = vg.Context(0)
g = torch.Generator(device="cuda"); g.manual_seed(20260130)
base = torch.randn(N, D, device="cuda", generator=g)
g.manual_seed(20260131)
queries = torch.randn(1024, D, device="cuda", generator=g)
st = torch.cuda.current_stream()
def timed(name, fn, reps=1):
    torch.cuda.synchronize(); t = time.time()
    for _ in range(reps): r = fn()
    torch.cuda.synchronize(); dt = (time.time() - t) / reps
    print(f"{name}: {dt*1e3:.2f} ms"); return r
idx = vg.Index(ctx, N, D)
timed("set_vectors", lambda: idx.set_vectors(base))
pq = vg.ProductQuantizer(ctx, D, M, 256)
timed("pq.train 65536x20it", lambda: pq.train(base[:65536], iters=20, seed=1))
codes = timed("pq.encode N", lambda: pq.encode(base))
timed("set_pq_codes", lambda: idx.set_pq_codes(pq, codes))
gt_ids, gt_sc = timed("flat search 1024q k=10 (first)", lambda: idx.search_flat(queries, K, stream=st))
timed("flat search 1024q k=10", lambda: idx.search_flat(queries, K, stream=st), reps=3)
gt = gt_ids.cpu().numpy().view(np.uint32)
for kp in (10, 32, 64, 128, 256, 512, 1024):
    cid, csc = timed(f"adc k'={kp}", lambda: idx.search_pq_adc(queries, kp, stream=st))
    if kp <= 1024:
        kk = min(kp, K)
    rid, rsc = timed(f"  rerank k'={kp}", lambda: idx.rerank(queries, cid, min(K, kp), stream=st))
    r = rid.cpu().numpy().view(np.uint32)
    rec = np.mean([len(set(r[i]) & set(gt[i])) / K for i in range(1024)])
    print(f"  recall@10 with k'={kp}: {rec:.4f}")
