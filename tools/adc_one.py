"""PQ-ADC scan kernel time at 10M x 96 B, one query per pass (the BASELINE configs[3] point), for the library
named by VECGO_HIP_LIB (tools/build_variant.sh) — kernel experiments.  argv: [rows [queries [gap_us]]]; gap_us > 0
idles the GPU that long between launches (synchronize + host spin): the kernel's time depends on it (DVFS)."""
import sys, os, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
import vecgo_amd as vg
n, nq, k, dim, m = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000, int(sys.argv[2]) if len(sys.argv) > 2 else 1, 10, 768, 96
gap_us = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
ctx = vg.Context(0)
rng = np.random.default_rng(0)
pq = vg.ProductQuantizer(ctx, dim, m, 256)
pq.set_codebooks(rng.integers(-128, 128, m*256*8).astype(np.int8), (rng.random(m)*0.02+0.005).astype(np.float32), np.zeros(m, np.float32))
st = torch.cuda.current_stream()
g = torch.Generator(device="cuda"); g.manual_seed(1)
codes = torch.randint(0, 256, (n, m), dtype=torch.uint8, device="cuda", generator=g)
idx = vg.Index(ctx, n, dim); idx.set_pq_codes(pq, codes); del codes
q = torch.randn(nq, dim, device="cuda")
ids = torch.empty(nq, k, dtype=torch.int32, device="cuda"); sc = torch.empty(nq, k, device="cuda")
for _ in range(5): idx.search_pq_adc(q, k, out=(ids, sc), stream=st)
torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    ctx.profile_read("pq_adc_scan"); ctx.profile_enable(True)
    for _ in range(20):
        idx.search_pq_adc(q, k, out=(ids, sc), stream=st)
        if gap_us > 0:
            torch.cuda.synchronize()
            t_end = time.perf_counter() + gap_us * 1e-6
            while time.perf_counter() < t_end: pass
    torch.cuda.synchronize()
    l, ms = ctx.profile_read("pq_adc_scan"); ctx.profile_enable(False)
    best = min(best, ms / l * 1e3)
print(f"{os.environ.get('VECGO_HIP_LIB', 'default'):30s} gap {gap_us:7.0f} us n={n} nq={nq}: scan kernel {best:7.1f} us  {nq * n * m / best / 1e6:.2f} TB/s  checksum {int(ids.sum())}")
