"""vg_search_flat of 1024 queries over shards of the 1M x 768 corpus (argv: row counts; default 1M 500k 250k 125k):
the per-rank step of the row-sharded headline at N = 1 / 2 / 4 / 8, before the exchange."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch, vecgo_amd as vg
ctx = vg.Context(0)
g = torch.Generator(device="cuda"); g.manual_seed(1)
q = torch.randn(1024, 768, device="cuda", generator=g)
ids = torch.empty(1024, 10, dtype=torch.int32, device="cuda"); sc = torch.empty(1024, 10, device="cuda")
for n in ([int(a) for a in sys.argv[1:]] or [1_000_000, 500_000, 250_000, 125_000]):
    base = torch.randn(n, 768, device="cuda", generator=g)
    idx = vg.Index(ctx, n, 768); idx.set_vectors(base)
    for _ in range(10): idx.search_flat(q, 10, out=(ids, sc))
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(20): idx.search_flat(q, 10, out=(ids, sc))
    t1 = time.perf_counter(); e1.record(); torch.cuda.synchronize()
    print(f"n={n}: host enqueue {(t1-t0)/20*1e3:.3f} ms/call, gpu {e0.elapsed_time(e1)/20:.3f} ms/call -> {1024/(e0.elapsed_time(e1)/20)*1e3/1e3:.1f} k q/s per GPU-shard", flush=True)
    idx.close(); del base
