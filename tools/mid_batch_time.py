"""vg_search_flat (fp32, 1M x 768) for 65 .. 128 queries per call: the 128-query tile against the tile of three / four 32-row
blocks (hook VG_FLAT_NO_SMALL_TILE) — ms per call, same results."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import vecgo_amd as vg, bench
from tests import hooks
ctx = vg.Context(0); dev = torch.device("cuda", 0)
rows = bench.gen_rows(0, 1_000_000, dev)
q = bench.gen_queries(2, dev)[0]
idx = vg.Index(ctx, 1_000_000, 768); idx.set_vectors(rows)
st = torch.cuda.current_stream()
for nq in (48, 64, 65, 80, 96, 100, 128):
    qs = q[:nq].contiguous()
    out = {}
    for wide in (1, 0, 1, 0):
        hooks.set_hook("VG_FLAT_NO_SMALL_TILE", wide)
        for _ in range(3): r = idx.search_flat(qs, 10, stream=st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(10): r = idx.search_flat(qs, 10, stream=st)
        e1.record(st); torch.cuda.synchronize()
        out[wide] = (e0.elapsed_time(e1) / 10, r)
    same = bool(torch.equal(out[0][1][0], out[1][1][0]) and torch.equal(out[0][1][1].view(torch.int32), out[1][1][1].view(torch.int32)))
    print(f"{nq:4d} queries: 128-query tile {out[1][0]:7.3f} ms   32-row blocks {out[0][0]:7.3f} ms   same: {same}", flush=True)
print("with the bf16 filter:")
idx.enable_bf16_filter(True)
for nq in (64, 65, 80, 96, 100, 128):
    qs = q[:nq].contiguous()
    out = {}
    for wide in (1, 0, 1, 0):
        hooks.set_hook("VG_FLAT_NO_SMALL_TILE", wide)
        for _ in range(3): r = idx.search_flat(qs, 10, stream=st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(10): r = idx.search_flat(qs, 10, stream=st)
        e1.record(st); torch.cuda.synchronize()
        out[wide] = (e0.elapsed_time(e1) / 10, r)
    same = bool(torch.equal(out[0][1][0], out[1][1][0]) and torch.equal(out[0][1][1].view(torch.int32), out[1][1][1].view(torch.int32)))
    print(f"{nq:4d} queries: 128-query tile {out[1][0]:7.3f} ms   32-row blocks {out[0][0]:7.3f} ms   same: {same}", flush=True)
hooks.set_hook("VG_FLAT_NO_SMALL_TILE", 0)
