"""The batch forms of the simd / quantizer distance functions over N rows, one query: call time (HIP events, after a
clock warm-up) and bytes of rows per second — squared_l2_batch, dot_batch, hamming_batch, pq_adc_lookup_batch,
ScalarQuantizer.l2_distance_batch, RaBitQuantizer distance.  argv: [N]."""
import sys, os
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
import vecgo_amd as vg, bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
D = 768
ctx = vg.Context(0); dev = torch.device("cuda", 0)
rows = bench.gen_rows(0, N, dev)
q = bench.gen_queries(1, dev)[0][0].contiguous()
g = torch.Generator(device=dev); g.manual_seed(3)

def timed(name, fn, nbytes, reps=20, warm=60):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name:44s}: {ms:8.3f} ms per call = {nbytes / ms / 1e9:6.2f} TB/s of rows", flush=True)

out = torch.empty(N, device=dev)
timed("squared_l2_batch 1M x 768 fp32", lambda: vg.squared_l2_batch(ctx, q, rows, D, out=out), N * D * 4)
timed("dot_batch 1M x 768 fp32", lambda: vg.dot_batch(ctx, q, rows, D, out=out), N * D * 4)
n2 = 10 * N
codes = torch.randint(0, 256, (n2, 96), dtype=torch.uint8, device=dev, generator=g)
a = torch.randint(0, 256, (96,), dtype=torch.uint8, device=dev, generator=g)
oi = torch.empty(n2, dtype=torch.int32, device=dev)
timed("hamming_batch 10M x 96 B", lambda: vg.hamming_batch(ctx, a, codes, out=oi), n2 * 96)
table = torch.randn(96 * 256, device=dev, generator=g)
timed("pq_adc_lookup_batch 10M x 96 B", lambda: vg.pq_adc_lookup_batch(ctx, table, codes, 96), n2 * 96)
sq = vg.ScalarQuantizer(ctx, D); sq.train(rows[:65536])
sc = sq.encode(rows)
timed("ScalarQuantizer.l2_distance_batch 1M x 768 B", lambda: sq.l2_distance_batch(q, sc, out=out), N * D)
