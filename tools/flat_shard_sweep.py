"""What one rank of the row-sharded flat search sees at N = 1, 2, 4, 8: 1024 queries against
1M/N rows through ShardedFlatIndex.search (world 1: local search + merge), GPU ms and host ms per
step, next to the ideal 1/N of the full-corpus step."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench
import vecgo_amd as vg
from vecgo_amd import sharded

ctx = vg.Context(0)
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream()
queries = bench.gen_queries(8, dev)
full = None
worlds = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
for world in worlds:
    n = bench.N_ROWS // world
    rows = bench.gen_rows(0, n, dev)
    index = sharded.ShardedFlatIndex(ctx, rows, bench.DIM, [0, n], metric=0)
    for i in range(3):
        index.search(queries[i % 8], bench.K, stream=stream)
    torch.cuda.synchronize()
    steps = 40
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for i in range(steps):
        index.search(queries[i % 8], bench.K, stream=stream)
    t1 = time.perf_counter(); e1.record(); torch.cuda.synchronize(); t2 = time.perf_counter()
    gpu = e0.elapsed_time(e1) / steps
    full = full or gpu
    print(f"rows/rank={n:8d} (N={world}): gpu {gpu:6.3f} ms/step  host enqueue {(t1 - t0) / steps * 1e3:6.3f} ms/step  "
          f"wall {(t2 - t0) / steps * 1e3:6.3f}  ideal {full / world:6.3f}  -> scaling eff {full / world / ((t2 - t0) / steps * 1e3):.2f}")
    index.index.close(); del rows
