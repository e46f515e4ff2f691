"""vg_search_flat_filtered at 1M x 768 (fp32 / SQ8 / PQ m=96 scans, no partitions): ms per call by batch size and filter
selectivity, next to the unfiltered scan of the same rows.  argv: [N]"""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
import vecgo_amd as vg
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else bench.N_ROWS
dev = torch.device("cuda", 0)
ctx = vg.Context(0)
rows = bench.gen_rows(0, n, dev)
dim = rows.shape[1]
idx = vg.Index(ctx, n, dim)
idx.set_vectors(rows)
sq = vg.ScalarQuantizer(ctx, dim); sq.train(rows[:100000])
idx.set_sq8_codes(sq, sq.encode(rows))
pq = vg.ProductQuantizer(ctx, dim, 96, 256); pq.train(rows[:65536], iters=2, seed=1)
idx.set_pq_codes(pq, pq.encode(rows))
rng = np.random.default_rng(0)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


for scan, name, plain in ((idx.SCAN_F32, "fp32", idx.search_flat), (idx.SCAN_SQ8, "sq8", idx.search_sq8),
                          (idx.SCAN_PQ, "pq96", idx.search_pq_adc)):
    for nq in (1, 16, 256, 1024):
        q = torch.randn((nq, dim), device=dev, dtype=torch.float32)
        line = [f"{name:5s} nq={nq:5d}  unfiltered {timed(lambda: plain(q, 10)):8.3f} ms"]
        for keep in (1.0, 0.5, 0.1, 0.01):
            m = np.packbits(rng.random(n) < keep, bitorder="little")
            ms = timed(lambda: idx.search_flat_filtered(q, 10, m, 0, scan=scan))
            line.append(f"keep {keep:4.2f}: {ms:8.3f}")
        print("  ".join(line), flush=True)
