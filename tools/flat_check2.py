import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch, time
import vecgo_amd as vg
import bench
from vecgo_amd import sharded
ctx = vg.Context(0)
dev = torch.device("cuda", 0)
rows = bench.gen_rows(0, 1_000_000, dev)
queries = bench.gen_queries(2, dev)
print("rows stats", rows.mean().item(), rows.std().item(), torch.isnan(rows).any().item(), rows[999_999, :3])
idx = vg.Index(ctx, 1_000_000, 768); idx.set_vectors(rows)
q = queries[0][:64]
gi, gs = bench.fp64_topk_local(rows, q, 0, 10)
gt = gi.cpu().numpy()
for name, fn in [("direct", lambda: idx.search_flat(q, 10)),
                 ("sharded", lambda: sharded.ShardedFlatIndex(ctx, rows, 768, [0, 1_000_000]).search(q, 10))]:
    torch.cuda.synchronize(); t = time.time()
    ids, sc = fn()
    torch.cuda.synchronize(); dt = time.time() - t
    got = ids.cpu().numpy().view(np.uint32).astype(np.int64)
    rec = np.mean([len(set(got[i]) & set(gt[i])) / 10 for i in range(64)])
    print(name, f"{dt*1e3:.1f} ms recall={rec:.3f}", got[0][:4], gt[0][:4], sc[0][:3].cpu().numpy(), gs[0][:3].cpu().numpy())
