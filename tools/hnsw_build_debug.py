import sys, time
from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch, vecgo_amd as vg, bench
N = int(sys.argv[1]); dbg = int(sys.argv[2])
ctx = vg.Context(0); dev = torch.device("cuda", 0)
rows = bench.gen_rows(0, N, dev)
idx = vg.Index(ctx, N, 768); idx.set_vectors(rows)
if dbg:
    from tests import hooks; hooks.set_hook("VG_BUILD_DEBUG", 1)
ctx.profile_enable(True)
torch.cuda.synchronize(); t0 = time.time()
idx.build_hnsw(m=32, ef_construction=300, max_batch=8192, growth_div=32)
torch.cuda.synchronize(); print("build_s", time.time() - t0, {k: ctx.profile_read(k) for k in ("hnsw_build_search", "hnsw_build_select", "hnsw_build_link")})
l0, up, e = idx.get_hnsw_graph()
import numpy as np, zlib
print("graph crc", zlib.crc32(l0.tobytes()), e)
