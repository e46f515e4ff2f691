"""vg_search_flat by k at 1M x 768, 1024 queries: ms per call and how many queries the proof sent to the
exhaustive kernel (k <= 48: the 64 best GEMM scores are re-scored and proved; above: every appended row)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch, bench
import vecgo_amd as vg
ctx = vg.Context(0); dev = torch.device("cuda:0")
rows = bench.gen_rows(0, bench.N_ROWS, dev)
q = bench.gen_queries(1, dev).reshape(-1, bench.DIM)[:1024].contiguous()
idx = vg.Index(ctx, bench.N_ROWS, bench.DIM); idx.set_vectors(rows)
for k in (10, 48, 64, 100, 256, 512):
    idx.search_flat(q, k); torch.cuda.synchronize()
    s0 = idx.flat_stats()
    t0 = time.perf_counter(); idx.search_flat(q, k); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    s1 = idx.flat_stats()
    print(f"k={k}: {dt*1e3:.2f} ms per 1024 queries; searched {s1[0]-s0[0]} fell back to exhaustive {s1[1]-s0[1]}")
