"""-DVG_WALK_TIMING build (tools/build_variant.sh timing k_graph.hip -DVG_WALK_TIMING): shader-clock cycles per pop by phase of
hnsw_search_sorted_kernel (queries it handed to the heap kernel are left out), next to tools/walk_timing.py for the heaps."""
import sys, os
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
import vecgo_amd as vg, bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
D, K, NQ = 768, 10, 8192
ctx = vg.Context(0); dev = torch.device("cuda", 0)
rows = bench.gen_rows(0, N, dev)
idx = vg.Index(ctx, N, D); idx.set_vectors(rows)
idx.build_hnsw(m=32, ef_construction=300, max_batch=8192, growth_div=32)
pq = vg.ProductQuantizer(ctx, D, 96, 256); pq.train(rows[:32768], iters=5, seed=1)
idx.set_pq_codes(pq, pq.encode(rows))
q = bench.gen_queries(8, dev).reshape(-1, D)[:NQ].contiguous()
for name, fn in (("f32", idx.search_hnsw), ("pq", idx.search_hnsw_pq)):
    for ef in (64, 128, 256):
        k = K if name == "f32" else ef
        fn(q, k, ef)
        _, _, st = fn(q, k, ef, stats="full")
        mine = (st[:, 0] >> 32) == 0          # the heap kernel packs two fields per column
        s = st[mine].astype(np.float64)
        P = s[:, 3].sum()
        print(f"{name} ef={ef}: {mine.sum()} of {NQ} queries on sorted arrays; pops/query {s[:, 3].mean():.1f}; cycles per pop: pop {s[:, 0].sum() / P:.0f} "
              f"adj+visited {s[:, 1].sum() / P:.0f} score {s[:, 2].sum() / P:.0f} merge {s[:, 4].sum() / P:.0f}", flush=True)
