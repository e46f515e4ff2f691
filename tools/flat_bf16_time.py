"""vg_search_flat, 1024 queries x 1M x 768, with and without the bfloat16 filter (vg_index_enable_bf16_filter): ms per
call, the nomination GEMM's time per launch, equality of the results and the number of proof fall-backs."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import vecgo_amd as vg, bench
ctx = vg.Context(0); dev = torch.device("cuda", 0)
args = [a for a in sys.argv[1:] if a not in ("no_big_tile", "early_b")]
if "early_b" in sys.argv[1:]:          # (row-tile fills behind the first matrix group: DESIGN.md section 9 item 3)
    from tests import hooks
    hooks.set_hook("VG_FLAT_BIG_EARLY_B", 1)
if "no_big_tile" in sys.argv[1:]:      # (the 128 x 128 tile above 128 queries too: what the persistent 256 x 256 tile replaced)
    from tests import hooks
    hooks.set_hook("VG_FLAT_NO_BIG_TILE", 1)
n = int(args[0]) if args else 1_000_000
rows = bench.gen_rows(0, n, dev)
q = bench.gen_queries(2, dev)
idx = vg.Index(ctx, n, 768); idx.set_vectors(rows)
st = torch.cuda.current_stream()
res = {}
for name, on in (("fp32", False), ("bf16 filter", True), ("fp32 again", False)):
    idx.enable_bf16_filter(on)
    for _ in range(3): idx.search_flat(q[0], 10, stream=st)
    torch.cuda.synchronize()
    s0 = idx.flat_stats()
    ctx.profile_read("flat_gemm"); ctx.profile_enable(True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for i in range(10): ids, sc = idx.search_flat(q[i % 2], 10, stream=st)
    e1.record(st); torch.cuda.synchronize()
    l, ms = ctx.profile_read("flat_gemm"); ctx.profile_enable(False)
    s1 = idx.flat_stats()
    res[name] = (idx.search_flat(q[0], 10, stream=st))
    print(f"{name:12s}: {e0.elapsed_time(e1) / 10:7.3f} ms per 1024 queries = {1024 / (e0.elapsed_time(e1) / 10) * 1e3:9.0f} queries/s; "
          f"nomination GEMM {ms / l:6.3f} ms per launch; proof fall-backs {s1[1] - s0[1]} of {s1[0] - s0[0]} queries")
for nqs in (32, 64):
    qs = q[0][:nqs].contiguous()
    for on in (False, True):
        idx.enable_bf16_filter(on)
        for _ in range(20): idx.search_flat(qs, 10, stream=st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for i in range(20): idx.search_flat(qs, 10, stream=st)
        e1.record(st); torch.cuda.synchronize()
        print(f"{nqs} queries, {'bf16 filter' if on else 'fp32':12s}: {e0.elapsed_time(e1) / 20:7.3f} ms per call = {nqs / (e0.elapsed_time(e1) / 20) * 1e3:9.0f} queries/s")
idx.enable_bf16_filter(False)
a, b = res["fp32"], res["bf16 filter"]
print("ids equal:", bool(torch.equal(a[0], b[0])), " scores bit-equal:", bool(torch.equal(a[1].view(torch.int32), b[1].view(torch.int32))))
