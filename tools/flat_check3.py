import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch, time
import vecgo_amd as vg
ctx = vg.Context(0)
g = torch.Generator(device="cuda"); g.manual_seed(1)
base = torch.randn(1_000_000, 768, device="cuda", generator=g)
q = torch.randn(1024, 768, device="cuda", generator=g)
idx = vg.Index(ctx, 1_000_000, 768); idx.set_vectors(base)
for nq in (536, 537, 600, 1024, 1024):
    torch.cuda.synchronize(); t = time.time()
    ids, sc = idx.search_flat(q[:nq], 10)
    torch.cuda.synchronize(); print(nq, f"{(time.time()-t)*1e3:.1f} ms", flush=True)
