"""(r05) batches of 1024 queries at k = 10 / 64 / 100 / 160 / 256 on 1M x 768 in rows / 8192 partitions: the partition-probed fp32 and
SQ8 scans (nprobes 8) and the whole-segment SQ8 scan, the matrix-core nomination against the scan kernels alone (VG_PROBE_NO_GEMM /
nomination off), with equality of ids and score bits.  argv: [N]"""
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
q = torch.randn((1024, bench.DIM), device=dev, dtype=torch.float32)
parts = n // 8192
cent = vg.kmeans_train(ctx, rows, bench.DIM, parts, max_iter=5, seed=1)
assign = vg.kmeans_assign(ctx, rows, cent, bench.DIM).to(torch.int64)
order = torch.argsort(assign, stable=True)
off = np.concatenate([[0], np.cumsum(torch.bincount(assign, minlength=parts).cpu().numpy())]).astype(np.uint32)
rows = rows[order].contiguous()
idx = vg.Index(ctx, n, bench.DIM)
idx.set_vectors(rows)
idx.set_partitions(cent.cpu().numpy(), off)
sq = vg.ScalarQuantizer(ctx, bench.DIM); sq.train(rows[:100000])
idx.set_sq8_codes(sq, sq.encode(rows))


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, r


def same(a, b):
    return bool(torch.equal(a[0], b[0])) and bool(torch.equal(a[1].view(torch.int32), b[1].view(torch.int32)))


for k in (10, 64, 100, 160, 256):
    line = [f"k {k:3d}:"]
    if k <= 160:
        a, ra = timed(lambda: idx.search_flat_probed(q, k, 8))
        hooks.set_hook("VG_PROBE_NO_GEMM", "1")
        try:
            b, rb = timed(lambda: idx.search_flat_probed(q, k, 8), reps=1)
        finally:
            hooks.set_hook("VG_PROBE_NO_GEMM", 0)
        line.append(f"fp32 nprobes 8 {a:6.2f} ms (scan kernels {b:6.2f}, same {same(ra, rb)})")
    idx.enable_sq8_nomination(False)
    sw, rsw = timed(lambda: idx.search_sq8(q, k), reps=1)
    sp, rsp = timed(lambda: idx.search_flat_probed(q, k, 8, scan=idx.SCAN_SQ8), reps=1)
    idx.enable_sq8_nomination(True)
    nw, rnw = timed(lambda: idx.search_sq8(q, k))
    np_, rnp = timed(lambda: idx.search_flat_probed(q, k, 8, scan=idx.SCAN_SQ8))
    line.append(f"sq8 whole {nw:6.2f} ms (scan {sw:6.2f}, same {same(rsw, rnw)})")
    line.append(f"sq8 nprobes 8 {np_:6.2f} ms (scan {sp:6.2f}, same {same(rsp, rnp)})")
    print("   ".join(line), flush=True)
