"""What the NaN replay costs at the bench's size: vg_search_flat, 1024 queries x 1M x 768, with 0 / 1 / 64 / 1024 queries holding a NaN
(one workgroup walks all rows per query at risk: include/vecgo_hip.h "NaN scores")."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import vecgo_amd as vg, bench
ctx = vg.Context(0); dev = torch.device("cuda", 0)
rows = bench.gen_rows(0, 1_000_000, dev)
q = bench.gen_queries(2, dev)[0].clone()
idx = vg.Index(ctx, 1_000_000, 768); idx.set_vectors(rows)
st = torch.cuda.current_stream()
for bad in (0, 1, 64, 1024):
    qq = q.clone()
    qq[:bad, 5] = float("nan")
    for _ in range(2): idx.search_flat(qq, 10, stream=st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(3): idx.search_flat(qq, 10, stream=st)
    e1.record(st); torch.cuda.synchronize()
    print(f"{bad:5d} of 1024 queries hold a NaN: {e0.elapsed_time(e1) / 3:9.2f} ms per call", flush=True)
