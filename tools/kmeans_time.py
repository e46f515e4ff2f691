"""Time kmeans.TrainKMeans at the flat writer's shape (flat/writer.go:109: k = N/8192 partitions,
10 iterations) on device-resident rows, with the per-kernel split left to rocprofv3."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import vecgo_amd as vg

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dim = 768
k = int(sys.argv[2]) if len(sys.argv) > 2 else max(n // 8192, 2)
ctx = vg.Context(0)
g = torch.Generator(device="cuda"); g.manual_seed(11)
x = torch.randn((n, dim), dtype=torch.float32, device="cuda", generator=g)
for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    c = vg.kmeans_train(ctx, x, dim, k, max_iter=10, seed=3)
    torch.cuda.synchronize()
    print(f"kmeans train n={n} dim={dim} k={k} iters=10: {time.perf_counter() - t0:.3f} s")
t0 = time.perf_counter()
a = vg.kmeans_assign(ctx, x, c, dim)
torch.cuda.synchronize()
print(f"assign partition: {(time.perf_counter() - t0) * 1e3:.1f} ms")
