"""Micro-benchmark of the PQ-ADC scan (dev tool, not the judged bench)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
import vecgo_amd as vg

def run(n, nq, k=10, dim=768, m=96, reps=10):
    ctx = run.ctx
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    codes = torch.randint(0, 256, (n, m), dtype=torch.uint8, device="cuda", generator=g)
    rng = np.random.default_rng(0)
    sd = dim // m
    pq = vg.ProductQuantizer(ctx, dim, m, 256)
    pq.set_codebooks(rng.integers(-128, 128, m*256*sd).astype(np.int8),
                     (rng.random(m)*0.02+0.005).astype(np.float32), np.zeros(m, np.float32))
    idx = vg.Index(ctx, n, dim)
    idx.set_pq_codes(pq, codes)
    del codes
    q = torch.randn(nq, dim, device="cuda")
    ids = torch.empty(nq, k, dtype=torch.int32, device="cuda"); sc = torch.empty(nq, k, device="cuda")
    st = torch.cuda.current_stream()
    for _ in range(3): idx.search_pq_adc(q, k, out=(ids, sc), stream=st)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): idx.search_pq_adc(q, k, out=(ids, sc), stream=st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    passes = nq
    print(f"n={n} nq={nq} k={k}: {ms*1e3:.1f} us/call  {nq/ms*1e3:.0f} QPS  "
          f"algorithmic {passes*n*m/ms/1e9*1e3/1e3:.2f} TB/s (per-query passes)")
    idx.close(); pq.close()

run.ctx = vg.Context(0)
for n, nq in [(10_000_000, 1), (10_000_000, 8), (10_000_000, 64), (1_000_000, 1), (1_000_000, 256), (1_000_000, 1024)]:
    run(n, nq)
run(10_000_000, 1, k=100)
run(10_000_000, 1, k=1000)
