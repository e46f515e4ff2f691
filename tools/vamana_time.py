"""vamana_search_kernel per node scorer (fp32 / PQ / RaBitQ / INT4) on the layer-0 graph vg_hnsw_build makes of
N x 768 i.i.d. normal rows, 8192 queries in flight: kernel time per call, node scores/s and an ids checksum so
that library variants (VECGO_HIP_LIB) can be compared.  argv: [N [k ...]]."""
import sys, os
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
import vecgo_amd as vg, bench

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
ks = [int(a) for a in sys.argv[2:]] or [10]
D, NQ = 768, 8192
ctx = vg.Context(0); dev = torch.device("cuda", 0)
rows = bench.gen_rows(0, N, dev)
idx = vg.Index(ctx, N, D); idx.set_vectors(rows)
idx.build_hnsw(m=32, ef_construction=300, max_batch=8192, growth_div=32)
pq = vg.ProductQuantizer(ctx, D, 96, 256); pq.train(rows[:32768], iters=5, seed=1)
idx.set_pq_codes(pq, pq.encode(rows))
rq = vg.RaBitQuantizer(ctx, D); idx.set_rabitq_codes(rq.encode(rows))
i4 = vg.Int4Quantizer(ctx, D); i4.train(rows[:32768]); idx.set_int4_codes(i4, i4.encode(rows))
l0, _, entry = idx.get_hnsw_graph(); idx.set_vamana_graph(l0, entry)
q = bench.gen_queries(8, dev).reshape(-1, D)[:NQ].contiguous()
st = torch.cuda.current_stream()
tag = os.environ.get("VECGO_HIP_LIB", "default")
for k in ks:
    for kind, name in ((0, "fp32"), (1, "PQ"), (2, "RaBitQ"), (3, "INT4")):
        ids, _, stats = idx.search_vamana(q, k, kind=kind, stats=True, stream=st)
        torch.cuda.synchronize()
        ctx.profile_read("vamana_search"); ctx.profile_enable(True)
        for _ in range(3): idx.search_vamana(q, k, kind=kind, stream=st)
        torch.cuda.synchronize()
        l, ms = ctx.profile_read("vamana_search"); ctx.profile_enable(False)
        dc = float(stats[:, 1].sum()); t = ms / 3 * 1e-3
        print(f"{tag:28s} N={N} vamana {name:6s} k={k:3d}: kernel {ms / 3:7.2f} ms per {NQ} queries, {dc / NQ:6.0f} scores per query, "
              f"{dc / t / 1e9:6.2f} G scores/s, ids checksum {int(ids.to(torch.int64).sum())}", flush=True)
