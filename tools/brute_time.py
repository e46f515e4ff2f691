"""vg_search_hnsw_brute (hnsw.BruteSearch / searchBitmap replayed through the reference's PriorityQueue) at 1M x 768:
ms per call by batch size and mode, the two kernels apart (vg_profile: "hnsw_brute_dist", "hnsw_brute_replay"), next to
vg_search_flat on the same batch.  argv: [N]."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
import vecgo_amd as vg, bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
D, K = 768, 10
ctx = vg.Context(0); dev = torch.device("cuda", 0)
rows = bench.gen_rows(0, N, dev)
idx = vg.Index(ctx, N, D); idx.set_vectors(rows)
qs = bench.gen_queries(1, dev)[0]
mask = np.random.default_rng(1).random(N) < 0.01
for nq in (1, 4, 16, 64, 256):
    q = qs[:nq].contiguous()
    for name, fn in (("scan", lambda: idx.search_hnsw_brute(q, K, 0)), ("bitmap 1%", lambda: idx.search_hnsw_brute(q, K, 1, mask)),
                     ("search_flat", lambda: idx.search_flat(q, K))):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        for k_ in ("hnsw_brute_dist", "hnsw_brute_replay"): ctx.profile_read(k_)
        ctx.profile_enable(True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        reps = 5
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        ctx.profile_enable(False)
        dms = ctx.profile_read("hnsw_brute_dist")[1] / reps; rms = ctx.profile_read("hnsw_brute_replay")[1] / reps
        ms = e0.elapsed_time(e1) / reps
        print(f"nq={nq:4d} {name:12s}: {ms:8.3f} ms per call ({nq / ms:8.1f} k queries/s); dist kernel {dms:7.3f} ms = "
              f"{nq * N * D * 4 / max(dms, 1e-9) / 1e9:7.2f} TB/s of rows scored, replay {rms:6.3f} ms", flush=True)
