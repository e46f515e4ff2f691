"""RaBitQ one-query scan kernel at 10M x 768 with warm clocks (A/B under VECGO_HIP_LIB)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import vecgo_amd as vg
n, dim, k = 10_000_000, 768, 10
ctx = vg.Context(0)
g = torch.Generator(device="cuda"); g.manual_seed(11)
cb = 100
codes = torch.randint(0, 256, (n, cb), dtype=torch.uint8, device="cuda", generator=g)
codes[:, cb - 4:] = (torch.rand(n, device="cuda", generator=g) * 5 + 25).view(torch.uint8).reshape(n, 4)
idx = vg.Index(ctx, n, dim); idx.set_rabitq_codes(codes); del codes
q = torch.randn(1, dim, device="cuda", generator=g)
out = (torch.empty(1, k, dtype=torch.int32, device="cuda"), torch.empty(1, k, device="cuda"))
for rep in range(3):
    for _ in range(200): idx.search_rabitq(q, k, out=out)
    torch.cuda.synchronize()
    ctx.profile_read("rabitq_scan"); ctx.profile_enable(True)
    for _ in range(50): idx.search_rabitq(q, k, out=out)
    torch.cuda.synchronize()
    l, ms = ctx.profile_read("rabitq_scan"); ctx.profile_enable(False)
    print(f"rabitq scan kernel {ms / l * 1e3:.1f} us = {n * cb / (ms / l) / 1e9:.2f} TB/s, ids checksum {int(out[0].sum())}")
