import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
import vecgo_amd as vg
ctx = vg.Context(0)
g = torch.Generator(device="cuda"); g.manual_seed(1)
base = torch.randn(1_000_000, 768, device="cuda", generator=g)
q = torch.randn(1024, 768, device="cuda", generator=g)
idx = vg.Index(ctx, 1_000_000, 768); idx.set_vectors(base)
ids = torch.empty(1024, 10, dtype=torch.int32, device="cuda"); sc = torch.empty(1024, 10, device="cuda")
for _ in range(3): idx.search_flat(q, 10, out=(ids, sc))
torch.cuda.synchronize()
for rep in range(3):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(5): idx.search_flat(q, 10, out=(ids, sc))
    t1 = time.perf_counter(); e1.record(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"host enqueue {(t1-t0)/5*1e3:.2f} ms/call, gpu {e0.elapsed_time(e1)/5:.2f} ms/call, wall {(t2-t0)/5*1e3:.2f}")
