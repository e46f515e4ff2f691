"""Partition-probed flat search (flat/segment.go:727-749) at the flat writer's shape: 1M x 768 rows in
rows/8192 = 122 k-means partitions, 1024 queries, by nprobes and scan type; recall@10 against the
exhaustive search of the same rows."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
import bench
import vecgo_amd as vg

N, DIM, K = bench.N_ROWS, bench.DIM, bench.K
dev = torch.device("cuda:0")
ctx = vg.Context(0)
rows = bench.gen_rows(0, N, dev)
parts = N // 8192
cent = vg.kmeans_train(ctx, rows, DIM, parts, max_iter=10, seed=1)
assign = vg.kmeans_assign(ctx, rows, cent, DIM)
order = torch.argsort(assign.to(torch.int64), stable=True)
rows = rows[order].contiguous()
counts = torch.bincount(assign.to(torch.int64), minlength=parts).cpu().numpy()
off = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint32)
idx = vg.Index(ctx, N, DIM)
idx.set_vectors(rows)
idx.set_partitions(cent.cpu().numpy(), off)
sq = vg.ScalarQuantizer(ctx, DIM); sq.train(rows)
idx.set_sq8_codes(sq, sq.encode(rows))
pq = vg.ProductQuantizer(ctx, DIM, 96, 256); pq.train(rows[:65536].contiguous(), iters=10, seed=1)
idx.set_pq_codes(pq, pq.encode(rows))
q = bench.gen_queries(1, dev).reshape(-1, DIM)[:1024].contiguous()
gt, _ = idx.search_flat(q, K)
gt = gt.cpu().numpy()
print(f"partitions {parts}: rows per partition min {counts.min()} mean {counts.mean():.0f} max {counts.max()}")
for scan, name in ((idx.SCAN_F32, "fp32"), (idx.SCAN_SQ8, "sq8"), (idx.SCAN_PQ, "pq")):
    for nprobes in (1, 4, 16):
        ids, _ = idx.search_flat_probed(q, K, nprobes, scan=scan)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            idx.search_flat_probed(q, K, nprobes, scan=scan)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        got = ids.cpu().numpy()
        rec = float(np.mean([len(set(got[i]) & set(gt[i])) / K for i in range(q.shape[0])]))
        print(f"{name:5s} nprobes={nprobes:3d}: {dt * 1e3:8.2f} ms / 1024 queries = {1024 / dt / 1e3:8.1f} kQPS   recall@10 {rec:.3f}")
