import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch, time
import vecgo_amd as vg
ctx = vg.Context(0)
D, K = 768, 10
for N, nq in [(100_000, 16), (300_000, 16), (1_000_000, 16), (1_000_000, 600), (1_000_000, 1024)]:
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    base = torch.randn(N, D, device="cuda", generator=g)
    q = torch.randn(nq, D, device="cuda", generator=g)
    idx = vg.Index(ctx, N, D); idx.set_vectors(base)
    torch.cuda.synchronize(); t = time.time()
    ids, sc = idx.search_flat(q, K, stream=torch.cuda.current_stream())
    torch.cuda.synchronize(); dt = time.time() - t
    d = (q.double()**2).sum(1, keepdim=True) + (base.double()**2).sum(1)[None, :] - 2 * q.double() @ base.double().T if N <= 300_000 else None
    if d is None:
        d = torch.cat([(q.double()**2).sum(1, keepdim=True) + (base[s:s+100000].double()**2).sum(1)[None, :] - 2 * q.double() @ base[s:s+100000].double().T for s in range(0, N, 100000)], 1)
    gt = torch.topk(d, K, dim=1, largest=False).indices.cpu().numpy()
    got = ids.cpu().numpy().view(np.uint32).astype(np.int64)
    rec = np.mean([len(set(got[i]) & set(gt[i])) / K for i in range(nq)])
    print(f"N={N} nq={nq}: {dt*1e3:.1f} ms recall={rec:.3f}", got[0][:5], gt[0][:5], flush=True)
    idx.close(); del base
