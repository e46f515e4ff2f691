"""SQ8 exhaustive scan (flat/segment.go:517-604 shape): N x 768 uint8 codes, timing."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
import vecgo_amd as vg
n, nq, k, dim = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), 768
ctx = vg.Context(0)
g = torch.Generator(device="cuda"); g.manual_seed(1)
codes = torch.randint(0, 256, (n, dim), dtype=torch.uint8, device="cuda", generator=g)
sq = vg.ScalarQuantizer(ctx, dim)
sq.set_bounds(np.full(dim, -4.0, np.float32), np.full(dim, 4.0, np.float32))
idx = vg.Index(ctx, n, dim); idx.set_sq8_codes(sq, codes); del codes
q = torch.randn(nq, dim, device="cuda")
ids = torch.empty(nq, k, dtype=torch.int32, device="cuda"); sc = torch.empty(nq, k, device="cuda")
st = torch.cuda.current_stream()
for _ in range(3): idx.search_sq8(q, k, out=(ids, sc), stream=st)
torch.cuda.synchronize()
ctx.profile_read("sq8_scan"); ctx.profile_enable(True)
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
reps = 10
e0.record()
for _ in range(reps): idx.search_sq8(q, k, out=(ids, sc), stream=st)
e1.record(); torch.cuda.synchronize()
launches, kms = ctx.profile_read("sq8_scan")
ms = e0.elapsed_time(e1) / reps
print(f"n={n} nq={nq} k={k}: call {ms*1e3:.1f} us, scan kernel {kms/launches*1e3:.1f} us = "
      f"{n*dim/(kms/launches)/1e6:.1f} GB/s of codes, {nq/ms*1e3:.0f} QPS")
