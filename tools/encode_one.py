"""pq_nominate_kernel<true> alone: Encode of N x 768 (m = 96) a few times, kernel time from vg_profile; for library
variants under VECGO_HIP_LIB (tools/build_variant.sh nom1 k_pq_train.hip -DVG_NOM_PROBE=1: no scan of the matrix
results; =2: no matrix instructions).  argv: [N]."""
import sys, os
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import vecgo_amd as vg
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
ctx = vg.Context(0)
g = torch.Generator(device="cuda"); g.manual_seed(2)
x = torch.randn((n, 768), dtype=torch.float32, device="cuda", generator=g)
pq = vg.ProductQuantizer(ctx, 768, 96, 256)
pq.train(x[:65536].contiguous(), iters=20, seed=1)
out = torch.empty((n, 96), dtype=torch.uint8, device="cuda")
for _ in range(2): pq.encode(x, out=out)
torch.cuda.synchronize()
ctx.profile_read("pq_encode"); ctx.profile_enable(True)
for _ in range(5): pq.encode(x, out=out)
torch.cuda.synchronize()
l, t = ctx.profile_read("pq_encode")
print(f"{os.environ.get('VECGO_HIP_LIB', 'default'):32s} encode {n} x 768: {t / l:.3f} ms per call, checksum {int(out.to(torch.int64).sum())}")
