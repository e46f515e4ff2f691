"""bench.build_side_legs on its own (kmeans.TrainKMeans, pq.Train / Encode / BuildDistanceTable, Segment.Rerank,
hnsw.BruteSearch at 1M x 768): the rows the bench line carries for them, printed one per line.  argv: [--no-cpu] [N].
Also the command under the r05 PMC passes of the k-means / PQ train / encode kernels (tools/collect_pmc.sh)."""
import json
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import vecgo_amd as vg
import bench

args = [a for a in sys.argv[1:] if not a.startswith("--")]
cpu = "--no-cpu" not in sys.argv
n = int(args[0]) if args else bench.N_ROWS
dev = torch.device("cuda", 0)
ctx = vg.Context(0)
rows = bench.gen_rows(0, n, dev)
queries = bench.gen_queries(8, dev)
out = bench.build_side_legs(vg, ctx, rows, queries, torch.cuda.current_stream(), cpu, rows.cpu().numpy() if cpu else None)
for k, v in out.items():
    print(k, json.dumps({a: (float(f"{b:.5g}") if isinstance(b, float) else b) for a, b in v.items() if a not in ("workload", "note")}))
