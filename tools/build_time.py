"""vg_hnsw_build of 1M x 768 (the bench's settings), twice, and the CRC of the layer-0 table: the same CRC before and
after a change of the build kernels = the same graph (VECGO_HIP_LIB selects a library variant)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch, vecgo_amd as vg, bench, zlib, numpy as np
ctx = vg.Context(0); dev = torch.device("cuda", 0)
rows = bench.gen_rows(0, 1_000_000, dev)
idx = vg.Index(ctx, 1_000_000, 768); idx.set_vectors(rows)
for r in range(2):
    torch.cuda.synchronize(); t = time.perf_counter()
    idx.build_hnsw(m=32, ef_construction=300, max_batch=8192, growth_div=32)
    torch.cuda.synchronize(); print("build s", time.perf_counter() - t)
l0, up, entry = idx.get_hnsw_graph()
print("l0 crc", zlib.crc32(np.ascontiguousarray(l0).tobytes()), "entry", entry)
