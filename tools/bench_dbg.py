import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
import vecgo_amd as vg, bench
from vecgo_amd import sharded
ctx = vg.Context(0); dev = torch.device("cuda", 0)
rows = bench.gen_rows(0, 1_000_000, dev); queries = bench.gen_queries(8, dev)
index = sharded.ShardedFlatIndex(ctx, rows, 768, [0, 1_000_000])
st = torch.cuda.current_stream()
def run(name, fn, n=5):
    for _ in range(2): fn(0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): fn(i)
    torch.cuda.synchronize(); print(f"{name}: {(time.perf_counter()-t0)/n*1e3:.2f} ms/step", flush=True)
run("direct same batch", lambda i: index.index.search_flat(queries[0], 10, stream=st))
run("direct cycling batches", lambda i: index.index.search_flat(queries[i % 8], 10, stream=st))
run("sharded", lambda i: index.search(queries[i % 8], 10, stream=st))
ctx.profile_enable(True)
run("sharded+profiling", lambda i: index.search(queries[i % 8], 10, stream=st))
print(ctx.profile_read("flat_gemm"))
ctx.profile_enable(False)
run("sharded again", lambda i: index.search(queries[i % 8], 10, stream=st))
