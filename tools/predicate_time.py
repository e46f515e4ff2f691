"""Filtered graph search at 1M x 768 on the structured corpus (bench.gen_structured): vg_search_hnsw_predicate (selectivity <= 0.3)
and vg_search_hnsw_filtered's post-filter walk (> 0.3) against the exact filtered answer (vg_search_flat_filtered): ms per 1024
queries, recall@10, counters; the one-off edge-distance pass.  argv: [N] [ef]"""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
import vecgo_amd as vg
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else bench.N_ROWS
ef = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device("cuda", 0)
ctx = vg.Context(0)
rows = bench.gen_structured(0, n, dev, seed=0)
q = bench.gen_structured(0, 1024, dev, seed=1)
dim = rows.shape[1]
idx = vg.Index(ctx, n, dim)
idx.set_vectors(rows)
t = time.perf_counter()
idx.build_hnsw(m=bench.HNSW_M, ef_construction=bench.HNSW_EFC)
torch.cuda.synchronize()
print(f"graph built in {time.perf_counter() - t:.2f} s", flush=True)
t = time.perf_counter()
idx.set_hnsw_edge_distances(None)
torch.cuda.synchronize()
print(f"edge distances recomputed in {(time.perf_counter() - t) * 1e3:.1f} ms", flush=True)
g = torch.Generator(device=dev)
g.manual_seed(3)


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, r


nb = (n + 7) // 8
for keep in (0.5, 0.3, 0.1, 0.03, 0.01):
    bits_ = torch.rand((n,), device=dev, generator=g) < keep
    pad = torch.zeros(nb * 8, dtype=torch.bool, device=dev)
    pad[:n] = bits_
    packed = (pad.view(nb, 8).to(torch.uint8) << torch.arange(8, device=dev, dtype=torch.uint8)).sum(1).to(torch.uint8)
    ems, (eid, _) = timed(lambda: idx.search_flat_filtered(q, 10, packed, 0))
    if keep > 0.3:
        ms, (ids, sc, st) = timed(lambda: idx.search_hnsw_filtered(q, 10, ef, packed, keep, stats=True))
        name = "post-filter walk"
    else:
        ms, (ids, sc, st) = timed(lambda: idx.search_hnsw_predicate(q, 10, ef, packed, stats=True))
        name = "predicate-aware "
    bms, _ = timed(lambda: idx.search_hnsw_brute(q[:64], 10, 1, packed), reps=1)
    a, b = ids.cpu().numpy().astype(np.int64), eid.cpu().numpy().astype(np.int64)
    rec = np.mean([len(set(a[i][a[i] != 0xFFFFFFFF]) & set(b[i])) / max(1, (b[i] != 0xFFFFFFFF).sum()) for i in range(a.shape[0])])
    print(f"keep {keep:4.2f}  {name} ef={ef}: {ms:8.2f} ms / 1024 q  recall@10 {rec:.3f}  visited {st[:, 0].mean():8.0f}  scored {st[:, 1].mean():8.0f}"
          f"  skipped {st[:, 2].mean():7.0f}  pops {st[:, 3].mean():6.0f} | exact filtered {ems:6.2f} ms | bitmap brute {bms * 16:7.1f} ms", flush=True)
