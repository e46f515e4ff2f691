"""Batched scans: `vg_search_rabitq` / `vg_search_pq_adc` with 1024 queries over 10M codes (the query-blocked kernels)
against one-query passes over the same codes; checks that a sample of the batch's results equals the one-query
passes' bit for bit.  argv: [rabitq|adc] [N] [NQ]."""
import sys, os
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
import vecgo_amd as vg, bench

what = sys.argv[1] if len(sys.argv) > 1 else "rabitq"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
NQ = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
D, K = 768, 10
ctx = vg.Context(0); dev = torch.device("cuda", 0); st = torch.cuda.current_stream()
q = bench.gen_queries(max(1, NQ // 1024), dev).reshape(-1, D)[:NQ].contiguous()
idx = vg.Index(ctx, N, D)
if what == "rabitq":
    idx.set_rabitq_codes(bench.gen_rabitq_codes(0, N, dev))
    search, prof, row_bytes = idx.search_rabitq, "rabitq_scan_mq", 100
else:
    g = torch.Generator(device=dev); g.manual_seed(7)
    codes = torch.randint(0, 256, (N, 96), dtype=torch.uint8, device=dev, generator=g)
    rng = np.random.default_rng(0)
    pq = vg.ProductQuantizer(ctx, D, 96, 256)
    pq.set_codebooks(rng.integers(-128, 128, 96 * 256 * 8).astype(np.int8),
                     (rng.random(96) * 0.02 + 0.005).astype(np.float32), np.zeros(96, np.float32))
    idx.set_pq_codes(pq, codes)
    search, prof, row_bytes = idx.search_pq_adc, "pq_adc_scan", 96
ids, sc = search(q, K, stream=st)
torch.cuda.synchronize()
ok = True
for i in list(range(0, NQ, max(1, NQ // 16)))[:16]:
    i1, s1 = search(q[i:i + 1], K, stream=st)
    ok &= bool(torch.equal(i1[0], ids[i]) and torch.equal(s1[0].view(torch.int32), sc[i].view(torch.int32)))
ctx.profile_read(prof); ctx.profile_enable(True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 3
e0.record(st)
for _ in range(reps): search(q, K, stream=st)
e1.record(st); torch.cuda.synchronize()
l, ms = ctx.profile_read(prof); ctx.profile_enable(False)
call = e0.elapsed_time(e1) / reps
print(f"{os.environ.get('VECGO_HIP_LIB', 'default'):24s} {what} N={N} NQ={NQ}: call {call:8.2f} ms = {NQ / call:7.1f} k queries/s; "
      f"scan kernel {ms / reps:8.2f} ms ({l // reps} launches) = {N * NQ / (ms / reps * 1e-3) / 1e9:7.1f} G row scores/s, "
      f"{N * row_bytes * NQ / (ms / reps * 1e-3) / 1e12:6.1f} TB/s of code bytes scored; sample equals one-query passes: {ok}")
