"""RaBitQ exhaustive scan (BASELINE config 5 shape: N x 768 -> 100 B/row): timing via vg_profile."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
import vecgo_amd as vg
n, nq, k, dim = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), 768
ctx = vg.Context(0)
cb = (dim + 63) // 64 * 8 + 4
g = torch.Generator(device="cuda"); g.manual_seed(1)
codes = torch.randint(0, 256, (n, cb), dtype=torch.uint8, device="cuda", generator=g)
norms = (torch.rand(n, device="cuda", generator=g) * 5 + 25).view(torch.uint8).reshape(n, 4)
codes[:, cb - 4:] = norms
idx = vg.Index(ctx, n, dim); idx.set_rabitq_codes(codes); del codes
q = torch.randn(nq, dim, device="cuda")
ids = torch.empty(nq, k, dtype=torch.int32, device="cuda"); sc = torch.empty(nq, k, device="cuda")
st = torch.cuda.current_stream()
ctx.profile_enable(False)
for _ in range(50): idx.search_rabitq(q, k, out=(ids, sc), stream=st)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
reps = 20
ctx.profile_read("rabitq_scan"); ctx.profile_enable(True)
e0.record()
for _ in range(reps): idx.search_rabitq(q, k, out=(ids, sc), stream=st)
e1.record(); torch.cuda.synchronize()
launches, kms = ctx.profile_read("rabitq_scan")
ms = e0.elapsed_time(e1) / reps
print(f"n={n} nq={nq} k={k}: {ms*1e3:.1f} us/call, scan kernel {kms/launches*1e3:.1f} us = {n*cb/(kms/launches)/1e6:.1f} GB/s of codes, {nq/ms*1e3:.0f} QPS")
