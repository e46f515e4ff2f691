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
# quantizer batch methods (one query against n codes; encode / decode of n rows)
pq = vg.ProductQuantizer(ctx, D, 96, 256); pq.train(rows[:32768], iters=3, seed=1)
pcodes = pq.encode(rows)
timed("ProductQuantizer.asymmetric_distance 1M x 96 B", lambda: pq.asymmetric_distance(q, pcodes, out=out), N * 96)
timed("ProductQuantizer.encode 1M x 768 fp32", lambda: pq.encode(rows, out=pcodes), N * D * 4, reps=3, warm=2)
dec = torch.empty((N, D), device=dev)
timed("ProductQuantizer.decode 1M x 96 B -> fp32", lambda: pq.decode(pcodes, out=dec), N * D * 4)
rq = vg.RaBitQuantizer(ctx, D); rcodes = rq.encode(rows)
timed("RaBitQuantizer.distance 1M x 100 B", lambda: rq.distance(q, rcodes, out=out), N * 100)
timed("RaBitQuantizer.encode 1M x 768 fp32", lambda: rq.encode(rows, out=rcodes), N * D * 4)
timed("ScalarQuantizer.encode 1M x 768 fp32", lambda: sq.encode(rows, out=sc), N * D * 4)
timed("ScalarQuantizer.decode 1M x 768 B -> fp32", lambda: sq.decode(sc, out=dec), N * D * 4)
iq = vg.Int4Quantizer(ctx, D); iq.train(rows[:65536]); icodes = iq.encode(rows)
timed("Int4Quantizer.encode 1M x 768 fp32", lambda: iq.encode(rows, out=icodes), N * D * 4)
timed("Int4Quantizer.decode 1M x 384 B -> fp32", lambda: iq.decode(icodes, out=dec), N * D * 4)
bq = vg.BinaryQuantizer(ctx, D); bq.train(rows[:65536]) if hasattr(bq, "train") else None
bcodes = bq.encode(rows)
oi2 = torch.empty(N, dtype=torch.int32, device=dev)
timed("BinaryQuantizer.compute_hamming_distance 1M x 96 B", lambda: bq.compute_hamming_distance(q, bcodes, out=oi2), N * 96)
timed("BinaryQuantizer.encode 1M x 768 fp32", lambda: bq.encode(rows, out=bcodes), N * D * 4)
bnd = torch.full((1,), 1400.0, device=dev)
timed("squared_l2_bounded_batch 1M x 768 fp32", lambda: vg.squared_l2_bounded_batch(ctx, q, rows, D, bnd), N * D * 4)
# the rest of the boundary's array-shaped entry points
nv = rows.clone()
timed("normalize_l2 1M x 768 fp32 (in place)", lambda: vg.normalize_l2(ctx, nv, D), N * D * 4 * 2)
cent = vg.kmeans_train(ctx, rows[:131072], D, 122, max_iter=3, seed=1)
timed("kmeans_assign 1M x 768 vs 122 centroids", lambda: vg.kmeans_assign(ctx, rows, cent, D), N * D * 4, reps=5, warm=5)
q1024 = bench.gen_queries(1, dev)[0]
tab = pq.build_distance_table(q1024) if hasattr(pq, "build_distance_table") else None
if tab is not None:
    timed("ProductQuantizer.build_distance_table 1024 queries", lambda: pq.build_distance_table(q1024), 1024 * 96 * 256 * 4)
