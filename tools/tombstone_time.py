"""vg_search_hnsw / _hnsw_pq at 1M x 768 (structured corpus) with and without tombstones: the tuned walk against the pass that
reads the bitmap (float comparisons, each lane at its turn).  argv: [N]"""
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
rows = bench.gen_structured(0, n, dev, seed=0)
q = bench.gen_structured(0, 8192, dev, seed=1)
idx = vg.Index(ctx, n, rows.shape[1])
idx.set_vectors(rows)
idx.build_hnsw(m=bench.HNSW_M, ef_construction=bench.HNSW_EFC)
pq = vg.ProductQuantizer(ctx, rows.shape[1], bench.PQ_M, 256)
pq.train(rows[:65536], iters=5, seed=1)
idx.set_pq_codes(pq, pq.encode(rows))


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for ef in (128, 512, 1024):
    idx.set_hnsw_tombstones(None)
    a = timed(lambda: idx.search_hnsw(q, 10, ef))
    ap = timed(lambda: idx.search_hnsw_pq(q, 10, ef))
    dead = np.packbits(np.random.default_rng(1).random(n) < 0.05, bitorder="little")
    idx.set_hnsw_tombstones(dead)
    b = timed(lambda: idx.search_hnsw(q, 10, ef))
    bp = timed(lambda: idx.search_hnsw_pq(q, 10, ef))
    print(f"ef {ef:5d}  fp32 walk {a:7.2f} ms / 8192 q, with 5 % tombstones {b:7.2f} ({b / a:.2f}x)   PQ walk {ap:7.2f}, with tombstones {bp:7.2f} ({bp / ap:.2f}x)", flush=True)
