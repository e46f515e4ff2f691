"""PQ-ADC scan kernel time vs corpus size at one query per pass: slope = streaming rate,
intercept = per-launch fixed cost (LUT staging, ramp, merge of the wave lists)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
import vecgo_amd as vg
nq, k, dim, m = int(sys.argv[1]) if len(sys.argv) > 1 else 1, 10, 768, 96
ctx = vg.Context(0)
rng = np.random.default_rng(0)
pq = vg.ProductQuantizer(ctx, dim, m, 256)
pq.set_codebooks(rng.integers(-128, 128, m*256*8).astype(np.int8), (rng.random(m)*0.02+0.005).astype(np.float32), np.zeros(m, np.float32))
st = torch.cuda.current_stream()
pts = []
for n in (1_000_000, 2_500_000, 5_000_000, 10_000_000, 20_000_000, 40_000_000):
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    codes = torch.randint(0, 256, (n, m), dtype=torch.uint8, device="cuda", generator=g)
    idx = vg.Index(ctx, n, dim); idx.set_pq_codes(pq, codes); del codes
    q = torch.randn(nq, dim, device="cuda")
    ids = torch.empty(nq, k, dtype=torch.int32, device="cuda"); sc = torch.empty(nq, k, device="cuda")
    for _ in range(3): idx.search_pq_adc(q, k, out=(ids, sc), stream=st)
    torch.cuda.synchronize()
    ctx.profile_read("pq_adc_scan"); ctx.profile_enable(True)
    for _ in range(10): idx.search_pq_adc(q, k, out=(ids, sc), stream=st)
    torch.cuda.synchronize()
    l, ms = ctx.profile_read("pq_adc_scan"); ctx.profile_enable(False)
    us = ms / l * 1e3
    pts.append((n * m / 1e6, us))
    print(f"n={n:>9d} nq={nq}: scan kernel {us:7.1f} us  {nq * n * m / us / 1e6:.2f} TB/s")
    idx.close()
x = np.array([p[0] for p in pts]); y = np.array([p[1] for p in pts])
b, a = np.polyfit(x, y, 1)
print(f"fit: {a:.1f} us + bytes / {1 / b / 1e6 * 1e6:.2f} TB/s" if b > 0 else "fit failed")
