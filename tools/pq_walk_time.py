"""PQ-scored graph walks against the fp32 walk on one graph: `hnsw_search_kernel<PQ>` and `hnsw_search_kernel<fp32>`
time per launch over ef, and `vamana_search_kernel` with the PQ node scorer, on the graph vg_hnsw_build makes of
N x 768 i.i.d. normal rows; NQ (env, default 8192) queries per call.  Prints node scores/s, queries/s and an ids checksum so that
library variants (VECGO_HIP_LIB) can be compared.  argv: [N [ef ...]]."""
import sys, os
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
import vecgo_amd as vg, bench

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
efs = [int(a) for a in sys.argv[2:]] or [128, 512, 2048]
D, K, NQ = 768, 10, int(os.environ.get("NQ", 8192))
ctx = vg.Context(0); dev = torch.device("cuda", 0)
rows = bench.gen_rows(0, N, dev)
idx = vg.Index(ctx, N, D); idx.set_vectors(rows)
idx.build_hnsw(m=32, ef_construction=300, max_batch=8192, growth_div=32)
pq = vg.ProductQuantizer(ctx, D, 96, 256); pq.train(rows[:32768], iters=5, seed=1)
idx.set_pq_codes(pq, pq.encode(rows))
l0, _, entry = idx.get_hnsw_graph(); idx.set_vamana_graph(l0, entry)
q = bench.gen_queries((NQ + 1023) // 1024, dev).reshape(-1, D)[:NQ].contiguous()
st = torch.cuda.current_stream()
tag = os.environ.get("VECGO_HIP_LIB", "default")


def timed(label, prof, fn, reps):
    ids, _, stats = fn(True)
    torch.cuda.synchronize()
    ctx.profile_read(prof); ctx.profile_enable(True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps): fn(False)
    e1.record(st)
    torch.cuda.synchronize()
    l, ms = ctx.profile_read(prof); ctx.profile_enable(False)
    dc = float(stats[:, 1].sum()); pops = float(stats[:, 3].sum())
    t = ms / reps * 1e-3     # the call's kernel launches (large ef: several chunks of queries)
    print(f"{tag:24s} N={N} {label:18s}: kernel {ms / reps:8.2f} ms ({l // reps} launches), whole call {e0.elapsed_time(e1) / reps:8.2f} ms per {NQ} queries = {NQ / t / 1e3:8.1f} k queries/s (kernel), "
          f"{dc / NQ:7.0f} scores and {pops / NQ:6.1f} pops per query, {dc / t / 1e9:6.2f} G scores/s, "
          f"{pops / t / 1e6:6.1f} M pops/s, ids checksum {int(ids.to(torch.int64).sum())}", flush=True)


for ef in efs:
    reps = 3 if ef <= 512 else 1
    timed(f"hnsw fp32 ef={ef}", "hnsw_search", lambda s: idx.search_hnsw(q, K, ef, stats=s, stream=st), reps)
    timed(f"hnsw PQ   ef={ef}", "hnsw_search_pq", lambda s: idx.search_hnsw_pq(q, ef, ef, stats=s, stream=st), reps)
timed("vamana PQ k=10", "vamana_search", lambda s: idx.search_vamana(q, K, kind=1, stats=s, stream=st), 3)
