"""(r05) bench.flat_ivf_probe on its own (1M x 768 in rows / 8192 k-means partitions, 1024 queries, nprobes 1 and 8), plus nprobes 4 /
16 / 32 and the exact kernels alone (VG_PROBE_NO_GEMM) beside the matrix-core nomination.  argv: [N]"""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
import vecgo_amd as vg
import bench
from tests import hooks

n = int(sys.argv[1]) if len(sys.argv) > 1 else bench.N_ROWS
dev = torch.device("cuda", 0)
ctx = vg.Context(0)
rows = bench.gen_rows(0, n, dev)
queries = bench.gen_queries(1, dev)
st = torch.cuda.current_stream()
gt = np.zeros((0, bench.K), np.int64)
print(bench.flat_ivf_probe(vg, ctx, rows, queries, gt, st), flush=True)
parts = n // 8192
cent = vg.kmeans_train(ctx, rows, bench.DIM, parts, max_iter=5, seed=1)
assign = vg.kmeans_assign(ctx, rows, cent, bench.DIM).to(torch.int64)
order = torch.argsort(assign, stable=True)
off = np.concatenate([[0], np.cumsum(torch.bincount(assign, minlength=parts).cpu().numpy())]).astype(np.uint32)
idx = vg.Index(ctx, n, bench.DIM)
idx.set_vectors(rows[order].contiguous())
idx.set_partitions(cent.cpu().numpy(), off)
q = queries.reshape(-1, bench.DIM)[:1024]


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, r


for nprobes in (1, 2, 4, 8, 16, 32):
    ctx.profile_read("flat_probe_gemm")
    ctx.profile_enable(True)
    a, ra = timed(lambda: idx.search_flat_probed(q, 10, nprobes))
    gl, gms = ctx.profile_read("flat_probe_gemm")
    ctx.profile_enable(False)
    if gl:
        print(f"             main grouped GEMM {gms / gl:6.2f} ms of it", flush=True)
    hooks.set_hook("VG_PROBE_NO_GEMM", "1")
    try:
        b, rb = timed(lambda: idx.search_flat_probed(q, 10, nprobes))
    finally:
        hooks.set_hook("VG_PROBE_NO_GEMM", 0)
    same = bool(torch.equal(ra[0], rb[0])) and bool(torch.equal(ra[1].view(torch.int32), rb[1].view(torch.int32)))
    print(f"nprobes {nprobes:3d}: {a:7.2f} ms per 1024 queries ({1024 / a:7.1f} k q/s)   exact kernels alone {b:7.2f} ms   same results: {same}", flush=True)

# 8192 queries x 8 probes: 65536 pairs, one more than a grouped nomination takes — the batch runs in two chunks of queries
q8 = torch.randn((8192, bench.DIM), device=dev, dtype=torch.float32)
a, ra = timed(lambda: idx.search_flat_probed(q8, 10, 8))
hooks.set_hook("VG_PROBE_NO_GEMM", "1")
try:
    b, rb = timed(lambda: idx.search_flat_probed(q8, 10, 8), reps=1)
finally:
    hooks.set_hook("VG_PROBE_NO_GEMM", 0)
same = bool(torch.equal(ra[0], rb[0])) and bool(torch.equal(ra[1].view(torch.int32), rb[1].view(torch.int32)))
print(f"8192 queries, nprobes 8: {a:7.2f} ms ({8192 / a:7.1f} k q/s)   exact kernels alone {b:7.2f} ms   same results: {same}", flush=True)

# the code scans of the same partitioned segment: SQ8 (grouped by partition) and PQ m = 96 (one workgroup per query and share of
# its probe list), beside the unprobed scans of the whole segment
sq = vg.ScalarQuantizer(ctx, bench.DIM); sq.train(rows[:100000])
idx.set_sq8_codes(sq, sq.encode(rows[order].contiguous()))
pq = vg.ProductQuantizer(ctx, bench.DIM, 96, 256); pq.train(rows[:65536], iters=2, seed=1)
idx.set_pq_codes(pq, pq.encode(rows[order].contiguous()))
for name, scan, whole in (("sq8", idx.SCAN_SQ8, idx.search_sq8), ("sq8 + bf16 nomination", idx.SCAN_SQ8, idx.search_sq8),
                          ("pq96", idx.SCAN_PQ, idx.search_pq_adc)):
    idx.enable_sq8_nomination(name.endswith("nomination"))
    w, _ = timed(lambda: whole(q, 10))
    line = [f"{name}: whole segment {w:7.2f} ms per 1024 queries"]
    for nprobes in (1, 8, 32):
        a, _ = timed(lambda: idx.search_flat_probed(q, 10, nprobes, scan=scan))
        line.append(f"nprobes {nprobes}: {a:6.2f}")
    print("   ".join(line), flush=True)
