"""Time PQ training (k-means++ seeding + Lloyd iterations) at BASELINE config 5's shape."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import vecgo_amd as vg

n, dim, m, k, iters = 65536, 768, 96, 256, 20
if len(sys.argv) > 1:
    n = int(sys.argv[1])
rng = np.random.default_rng(3)
x = rng.standard_normal((n, dim), dtype=np.float32)
ctx = vg.Context(0)
pq = vg.ProductQuantizer(ctx, dim, m, k)
for rep in range(3):
    t0 = time.perf_counter()
    pq.train(x, iters=iters, seed=7)
    print(f"train n={n} dim={dim} m={m} k={k} iters={iters}: {time.perf_counter() - t0:.3f} s")

# Encode of device-resident rows (the index-build leg: 1M x 768 -> 1M x 96 code bytes)
import torch
g = torch.Generator(device="cuda"); g.manual_seed(5)
xd = torch.randn((1_000_000, dim), dtype=torch.float32, device="cuda", generator=g)
out = torch.empty((xd.shape[0], m), dtype=torch.uint8, device="cuda")
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pq.encode(xd, out=out)
    torch.cuda.synchronize()
    print(f"encode 1M x {dim}: {(time.perf_counter() - t0) * 1e3:.2f} ms")
