"""Time the index-build legs on 1M x 768 device-resident rows: RaBitQ / SQ8 / INT4 / PQ encode,
SQ8 / INT4 train, PQ decode, LUT build."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
import vecgo_amd as vg

n, dim = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 768
ctx = vg.Context(0)
g = torch.Generator(device="cuda"); g.manual_seed(2)
x = torch.randn((n, dim), dtype=torch.float32, device="cuda", generator=g)


def timed(label, fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(f"{label:28s} {best * 1e3:9.2f} ms   ({n * dim * 4 / best / 1e9:8.1f} GB/s of fp32 rows)")
    return r


rq = vg.RaBitQuantizer(ctx, dim)
timed("rabitq encode", lambda: rq.encode(x))
sq = vg.ScalarQuantizer(ctx, dim)
timed("sq8 train", lambda: sq.train(x))
c8 = timed("sq8 encode", lambda: sq.encode(x))
timed("sq8 decode", lambda: sq.decode(c8))
iq = vg.Int4Quantizer(ctx, dim)
timed("int4 train", lambda: iq.train(x))
c4 = timed("int4 encode", lambda: iq.encode(x))
timed("int4 decode", lambda: iq.decode(c4))
pq = vg.ProductQuantizer(ctx, dim, 96, 256)
pq.train(x[:65536].contiguous(), iters=5, seed=1)
cp = timed("pq encode", lambda: pq.encode(x))
timed("pq decode", lambda: pq.decode(cp))
q = x[:1024].contiguous()
timed("pq build_distance_table x1024", lambda: pq.build_distance_table(q))
