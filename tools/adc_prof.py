import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
import vecgo_amd as vg
n, nq, k, dim, m = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), 768, 96
ctx = vg.Context(0)
g = torch.Generator(device="cuda"); g.manual_seed(1)
codes = torch.randint(0, 256, (n, m), dtype=torch.uint8, device="cuda", generator=g)
rng = np.random.default_rng(0)
pq = vg.ProductQuantizer(ctx, dim, m, 256)
pq.set_codebooks(rng.integers(-128, 128, m*256*8).astype(np.int8), (rng.random(m)*0.02+0.005).astype(np.float32), np.zeros(m, np.float32))
idx = vg.Index(ctx, n, dim); idx.set_pq_codes(pq, codes); del codes
q = torch.randn(nq, dim, device="cuda")
ids = torch.empty(nq, k, dtype=torch.int32, device="cuda"); sc = torch.empty(nq, k, device="cuda")
st = torch.cuda.current_stream()
for _ in range(5): idx.search_pq_adc(q, k, out=(ids, sc), stream=st)
torch.cuda.synchronize()
