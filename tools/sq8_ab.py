import sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import vecgo_amd as vg
n, dim, k = 4_000_000, 768, 10
ctx = vg.Context(0)
g = torch.Generator(device="cuda"); g.manual_seed(1)
codes = torch.randint(0, 256, (n, dim), dtype=torch.uint8, device="cuda", generator=g)
sq = vg.ScalarQuantizer(ctx, dim); sq.set_bounds(np.full(dim, -4.0, np.float32), np.full(dim, 4.0, np.float32))
idx = vg.Index(ctx, n, dim); idx.set_sq8_codes(sq, codes); del codes
q = torch.randn(1, dim, device="cuda", generator=g)
out = (torch.empty(1, k, dtype=torch.int32, device="cuda"), torch.empty(1, k, device="cuda"))
for rep in range(3):
    for _ in range(100): idx.search_sq8(q, k, out=out)
    torch.cuda.synchronize()
    ctx.profile_read("sq8_scan"); ctx.profile_enable(True)
    for _ in range(50): idx.search_sq8(q, k, out=out)
    torch.cuda.synchronize()
    l, ms = ctx.profile_read("sq8_scan"); ctx.profile_enable(False)
    print(f"sq8 scan kernel {ms / l * 1e3:.1f} us = {n * dim / (ms / l) / 1e9:.2f} TB/s")
