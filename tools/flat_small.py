"""Flat exact search latency by query-batch size (BASELINE configs[1]: Q-batch 1...1024)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import vecgo_amd as vg
ctx = vg.Context(0)
n, dim, k = 1_000_000, 768, 10
g = torch.Generator(device="cuda"); g.manual_seed(1)
base = torch.randn(n, dim, device="cuda", generator=g)
idx = vg.Index(ctx, n, dim); idx.set_vectors(base)
st = torch.cuda.current_stream()
for nq in (1, 2, 4, 8, 16, 32, 64, 96, 128, 256, 1024):
    q = torch.randn(nq, dim, device="cuda", generator=g)
    out = (torch.empty(nq, k, dtype=torch.int32, device="cuda"), torch.empty(nq, k, device="cuda"))
    for _ in range(2): idx.search_flat(q, k, out=out, stream=st)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps): idx.search_flat(q, k, out=out, stream=st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"nq={nq:5d}: {ms:8.3f} ms/call  {nq/ms*1e3:9.0f} QPS  rows read {n*dim*4/ms/1e6:7.0f} GB/s-equivalent")
