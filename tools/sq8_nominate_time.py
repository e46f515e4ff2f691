"""(r05) vg_search_sq8 batches at 1M x 768 with and without vg_index_enable_sq8_nomination: ms per call, equality of ids and score bits,
how many queries' proofs fail — on the bench's random-normal corpus and on the structured one.  argv: [N]"""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import vecgo_amd as vg
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else bench.N_ROWS
dev = torch.device("cuda", 0)
ctx = vg.Context(0)


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, r


for name, rows, qs in (("random-normal", bench.gen_rows(0, n, dev), bench.gen_queries(1, dev).reshape(-1, bench.DIM)),
                       ("structured", bench.gen_structured(0, n, dev, seed=0), bench.gen_structured(0, 1024, dev, seed=1))):
    idx = vg.Index(ctx, n, bench.DIM)
    sq = vg.ScalarQuantizer(ctx, bench.DIM); sq.train(rows[:200000])
    idx.set_sq8_codes(sq, sq.encode(rows))
    for nq in (16, 64, 256, 1024):
        q = qs[:nq].contiguous()
        idx.enable_sq8_nomination(False)
        a, ra = timed(lambda: idx.search_sq8(q, 10), reps=2)
        t0 = time.perf_counter()
        idx.enable_sq8_nomination(True)
        torch.cuda.synchronize()
        build = (time.perf_counter() - t0) * 1e3
        ctx.profile_read("sq8_nominate_gemm")
        ctx.profile_enable(True)
        b, rb = timed(lambda: idx.search_sq8(q, 10))
        gl, gms = ctx.profile_read("sq8_nominate_gemm")
        ctx.profile_enable(False)
        same = bool(torch.equal(ra[0], rb[0])) and bool(torch.equal(ra[1].view(torch.int32), rb[1].view(torch.int32)))
        print(f"{name:14s} nq {nq:5d}: scan {a:8.2f} ms   nominated {b:7.2f} ms (GEMM {gms / max(gl, 1):5.2f}; image built in {build:6.1f} ms)   same: {same}", flush=True)
    idx.close()
    del rows
