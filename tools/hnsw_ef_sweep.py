"""hnsw_search_kernel time per launch over ef (LDS heaps up to 512, HBM scratch beyond) on a graph built by
vg_hnsw_build over N x 768 i.i.d. normal rows; 8192 queries in flight.  argv: [N [ef ...]].  Prints a checksum of
the ids so that variants can be compared (VECGO_HIP_LIB)."""
import sys, os
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
import vecgo_amd as vg, bench

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
efs = [int(a) for a in sys.argv[2:]] or [128, 512, 1024, 2048, 4096]
D, K, NQ = 768, 10, 8192
ctx = vg.Context(0); dev = torch.device("cuda", 0)
rows = bench.gen_rows(0, N, dev)
idx = vg.Index(ctx, N, D); idx.set_vectors(rows)
idx.build_hnsw(m=32, ef_construction=300, max_batch=8192, growth_div=32)
q = bench.gen_queries(8, dev).reshape(-1, D)[:NQ].contiguous()
st = torch.cuda.current_stream()
for ef in efs:
    ids, _, stats = idx.search_hnsw(q, K, ef, stats=True, stream=st)
    torch.cuda.synchronize()
    ctx.profile_read("hnsw_search"); ctx.profile_enable(True)
    reps = 3 if ef <= 512 else 1
    for _ in range(reps): idx.search_hnsw(q, K, ef, stream=st)
    torch.cuda.synchronize()
    l, ms = ctx.profile_read("hnsw_search"); ctx.profile_enable(False)
    dc = float(stats[:, 1].sum())
    print(f"{os.environ.get('VECGO_HIP_LIB', 'default'):28s} N={N} ef={ef:5d}: {ms / l:8.2f} ms per {NQ} queries, "
          f"{dc * D * 4 / (ms / l * 1e-3) / 1e9:7.0f} GB/s gathered, ids checksum {int(ids.to(torch.int64).sum())}")
