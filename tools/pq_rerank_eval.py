"""Quantized search at the metric's recall bar: PQ (m=96, K=256) ADC scan -> top-R candidates ->
exact fp32 rerank (engine/search.go:914-965) -> top-10, against the exact flat search."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
import vecgo_amd as vg
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
nq, dim, m, k = 1024, 768, 96, 10
ctx = vg.Context(0)
g = torch.Generator(device="cuda"); g.manual_seed(20260130)
base = torch.randn(n, dim, device="cuda", generator=g)
q = torch.randn(nq, dim, device="cuda", generator=g)
pq = vg.ProductQuantizer(ctx, dim, m, 256)
t0 = time.perf_counter(); pq.train(base[:65536], iters=20, seed=1); torch.cuda.synchronize(); t1 = time.perf_counter()
codes = pq.encode(base); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"train 65536x{dim}: {t1-t0:.2f} s   encode {n}: {(t2-t1)*1e3:.1f} ms")
idx = vg.Index(ctx, n, dim); idx.set_vectors(base); idx.set_pq_codes(pq, codes)
gt, _ = idx.search_flat(q, k)
gt = gt.cpu().numpy() if hasattr(gt, "cpu") else np.asarray(gt)
st = torch.cuda.current_stream()
for R in (32, 64, 128, 256, 512, 1024):
    def run():
        cid, _ = idx.search_pq_adc(q, R, stream=st)
        return idx.rerank(q, cid, k, stream=st)
    ids, sc = run(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(3): run()
    e1.record(st); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    ids = ids.cpu().numpy() if hasattr(ids, "cpu") else np.asarray(ids)
    rec = np.mean([len(set(ids[i]) & set(gt[i])) / k for i in range(nq)])
    print(f"R={R:5d}: recall@10 {rec:.4f}   {ms:.2f} ms per {nq} queries = {nq/ms*1e3:.0f} QPS")
