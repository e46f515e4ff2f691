"""Int4Quantizer.L2DistanceBatch / L2Distance over N x 768 codes (one query): kernel time and code bytes per second
for both summation orders (precomputed = False: int4L2DistanceBatchAvx512; True: the lookup-table kernel per code).
argv: [N]."""
import sys, os
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
import vecgo_amd as vg, bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
D = 768
ctx = vg.Context(0); dev = torch.device("cuda", 0)
rows = bench.gen_rows(0, min(N, 1_000_000), dev)
iq = vg.Int4Quantizer(ctx, D); iq.train(rows[:65536])
# distinct random codes (bench.py's leg does the same): a corpus that repeats 1M encoded rows sits partly in the
# 256 MB memory-side cache and reads 5-7 % faster than a stream from HBM
g = torch.Generator(device=dev); g.manual_seed(19)
codes = torch.randint(0, 256, (N, D // 2), dtype=torch.uint8, device=dev, generator=g)
q = bench.gen_queries(1, dev)[0][0].contiguous()
out = torch.empty(N, device=dev)
for pre, fn in ((False, iq.l2_distance_batch), (True, iq.l2_distance)):
    for _ in range(150): fn(q, codes, out=out)   # ~50+ ms of launches: an idle GPU needs that long to reach its clocks
    torch.cuda.synchronize()
    ctx.profile_read("int4_scan"); ctx.profile_enable(True)
    for _ in range(50): fn(q, codes, out=out)
    torch.cuda.synchronize()
    l, tot = ctx.profile_read("int4_scan"); ctx.profile_enable(False)
    ms = tot / max(l, 1)
    print(f"{os.environ.get('VECGO_HIP_LIB','default'):24s} int4 l2 batch precomputed={pre}: kernel {ms:8.3f} ms per {N} codes = {N * D / 2 / ms / 1e9:6.2f} TB/s of codes, checksum {float(out.double().sum()):.6e}", flush=True)
