"""vg_search_pq_adc, nq queries x 1M x 768 (m = 96, K = 256), with and without the bfloat16 nomination (vg_index_enable_pq_nomination):
ms per call, equality of the results, and the smallest batch the nomination should take.
    python tools/pq_nominate_time.py [rows]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import vecgo_amd as vg, bench
ctx = vg.Context(0); dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
rows = bench.gen_rows(0, n, dev)
q = bench.gen_queries(2, dev)[0]
st = torch.cuda.current_stream()
pq = vg.ProductQuantizer(ctx, 768, 96, 256)
pq.train(rows[:65536], iters=5)
codes = pq.encode(rows)
from tests import hooks
hooks.set_hook("VG_PQ_NOM_ALWAYS", 1)                  # every batch size takes the nomination here: where the crossover is
idx = vg.Index(ctx, n, 768)
idx.set_pq_codes(pq, codes)


def timed(nq, k, reps):
    qs = q[:nq].contiguous()
    for _ in range(2): r = idx.search_pq_adc(qs, k, stream=st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps): r = idx.search_pq_adc(qs, k, stream=st)
    e1.record(st); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, r


for k in (10, 100):
    for nq in (16, 32, 48, 64, 128, 256, 1024):
        idx.enable_pq_nomination(False)
        ms0, r0 = timed(nq, k, 3)
        idx.enable_pq_nomination(True)
        ctx.profile_read("sq8_nominate_gemm"); ctx.profile_enable(True)
        ms1, r1 = timed(nq, k, 5)
        l, gms = ctx.profile_read("sq8_nominate_gemm"); ctx.profile_enable(False)
        same = bool(torch.equal(r0[0], r1[0]) and torch.equal(r0[1].view(torch.int32), r1[1].view(torch.int32)))
        print(f"k {k:3d}  {nq:5d} queries: scan {ms0:8.3f} ms, nominated {ms1:8.3f} ms (GEMM {gms / max(l, 1):6.3f} ms per launch), bits equal: {same}", flush=True)
