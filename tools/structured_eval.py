"""Recall / cost frontier of the two graph pipelines on candidate STRUCTURED corpora (bench.gen_structured):
argv: N latent noise [latent noise ...]."""
import sys, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
import vecgo_amd as vg, bench

N = int(sys.argv[1]); pairs = [(int(sys.argv[i]), float(sys.argv[i + 1])) for i in range(2, len(sys.argv), 2)]
ctx = vg.Context(0); dev = torch.device("cuda", 0); st = torch.cuda.current_stream()
for latent, noise in pairs:
    rows = bench.gen_structured(0, N, dev, latent, noise, 0)
    queries = bench.gen_structured(0, 8192, dev, latent, noise, 1).reshape(8, 1024, bench.DIM)
    gi, _ = bench.fp64_topk_local(rows, queries[0], 0, bench.K)
    rep, _, idx, pq = bench.hnsw_pq_frontier(vg, ctx, rows, queries, gi.cpu().numpy(), 11.6, st, False,
                                            efs_f32=(16, 32, 64, 128, 256, 512), efs_pq=(32, 64, 128, 256, 512))
    print(f"latent {latent} noise {noise}: build {rep['graph_build_s']:.1f} s")
    for e in rep["frontier_f32"]: print(f"  f32 ef={e['ef']:5d} recall {e['recall_at_10']:.3f} {e['qps'] / 1e3:9.1f} k q/s  dc {e['distance_computations_per_query']:.0f}")
    for e in rep["frontier_pq_rerank"]: print(f"  pq  ef={e['ef']:5d} recall {e['recall_at_10']:.3f} {e['qps'] / 1e3:9.1f} k q/s  walk {e['walk_ms']:.2f} rerank {e['rerank_ms']:.2f}")
    idx.close(); pq.close(); del rows
