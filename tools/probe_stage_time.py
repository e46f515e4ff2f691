"""The partition-probed fp32 search (bench.flat_ivf_probe's shape: 1M x 768 in 122 k-means partitions, 1024 queries, k = 10) run
`reps` times at one nprobes, for `rocprofv3 --kernel-trace --stats`: every kernel of the call by name, so that the stages of the
matrix-core nomination (sample pass, thresholds, grouped GEMM, pick, exact re-score + proof, fall-back scans) are separated.
    tools/kernel_stats.sh r06_probe8 python3 tools/probe_stage_time.py 8 20     (through gpurun)
The kernels of the set-up (k-means, assignment) carry their own names; divide a search kernel's total by reps for its share."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
import vecgo_amd as vg
import bench

nprobes = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n = bench.N_ROWS
dev = torch.device("cuda", 0)
ctx = vg.Context(0)
rows = bench.gen_rows(0, n, dev)
q = bench.gen_queries(1, dev).reshape(-1, bench.DIM)[:1024].contiguous()
st = torch.cuda.current_stream()
parts = n // 8192
cent = vg.kmeans_train(ctx, rows, bench.DIM, parts, max_iter=5, seed=1)
assign = vg.kmeans_assign(ctx, rows, cent, bench.DIM).to(torch.int64)
order = torch.argsort(assign, stable=True)
off = np.concatenate([[0], np.cumsum(torch.bincount(assign, minlength=parts).cpu().numpy())]).astype(np.uint32)
idx = vg.Index(ctx, n, bench.DIM)
idx.set_vectors(rows[order].contiguous())
idx.set_partitions(cent.cpu().numpy(), off)
for _ in range(2):
    idx.search_flat_probed(q, bench.K, nprobes, stream=st)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(st)
for _ in range(reps):
    idx.search_flat_probed(q, bench.K, nprobes, stream=st)
e1.record(st)
torch.cuda.synchronize()
print(f"nprobes {nprobes}: {e0.elapsed_time(e1) / reps:.3f} ms per call of 1024 queries ({reps + 2} calls in this process)")
