"""How navigable are bench-style graphs on the BASELINE corpus (1M x 768 i.i.d. normal)?
Compares the exact 31-NN graph bench.py searches with (a) the same candidates plus random long
edges pruned by robustPrune, (b) that graph with reverse edges added, at ef = 128 / 256 / 512."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
import bench
import vecgo_amd as vg

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
DIM, K = bench.DIM, bench.K
dev = torch.device("cuda:0")
ctx = vg.Context(0)
rows = bench.gen_rows(0, N, dev)
q = bench.gen_queries(1, dev).reshape(-1, DIM)[:1024].contiguous()
idx = vg.Index(ctx, N, DIM)
idx.set_vectors(rows)
gt, _ = idx.search_flat(q, K)
gt = gt.cpu().numpy().view(np.uint32)

deg = 32
l0 = torch.empty((N, deg), dtype=torch.int32, device=dev)
sc = torch.empty((4096, deg), device=dev)
t0 = time.time()
for s in range(0, N, 4096):
    e = min(N, s + 4096)
    idx.search_flat(rows[s:e], deg, out=(l0[s:e], sc[:e - s]))
torch.cuda.synchronize()
print(f"exact {deg}-NN lists: {time.time() - t0:.1f} s")
INV = np.uint32(0xFFFFFFFF)
knn = l0.cpu().numpy().view(np.uint32)
knn = np.where(knn == np.arange(N, dtype=np.uint32)[:, None], INV, knn)
knn = np.take_along_axis(knn, np.argsort(knn == INV, axis=1, kind="stable"), axis=1)


def evaluate(label, graph, entry=0):
    idx.set_hnsw_graph(graph, (), entry_point=entry, m=graph.shape[1] // 2)
    for ef in (128, 256, 512):
        ids, _, st = idx.search_hnsw(q, K, ef, stats=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            idx.search_hnsw(q, K, ef)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        got = ids.cpu().numpy().view(np.uint32)
        rec = float(np.mean([len(set(got[i]) & set(gt[i])) / K for i in range(q.shape[0])]))
        print(f"{label:34s} ef={ef:4d} recall@10={rec:.3f}  dist/query={float(st[:, 1].sum()) / q.shape[0]:8.0f}  "
              f"{q.shape[0] / dt / 1e3:8.1f} kQPS")


def add_reverse(graph, cap):
    n, d = graph.shape
    src = np.repeat(np.arange(n, dtype=np.uint32), d)
    dst = graph.reshape(-1)
    ok = dst != INV
    src, dst = src[ok], dst[ok]
    out = np.full((n, cap), INV, np.uint32)
    out[:, :d] = graph
    fill = (graph != INV).sum(1).astype(np.int64)
    order = np.argsort(dst, kind="stable")
    rs, rd = src[order], dst[order]          # reverse edge rd -> rs
    start = np.searchsorted(rd, np.arange(n, dtype=np.uint32))
    rank = np.arange(rd.size) - start[rd]
    pos = fill[rd] + rank
    keep = pos < cap
    out[rd[keep], pos[keep]] = rs[keep]
    # drop duplicates inside a list (keep first)
    srt = np.sort(out, axis=1)
    dup_rows = np.where((srt[:, 1:] == srt[:, :-1]) & (srt[:, 1:] != INV))[0]
    for r in np.unique(dup_rows):
        _, first = np.unique(out[r], return_index=True)
        row = np.full(cap, INV, np.uint32)
        vals = out[r][np.sort(first)]
        vals = vals[vals != INV]
        row[:vals.size] = vals
        out[r] = row
    return out


evaluate("exact 31-NN (bench.py)", knn)
rng = np.random.default_rng(5)
rand = rng.integers(0, N, size=(N, 32), dtype=np.uint32)
cands = np.concatenate([knn, rand], axis=1)
t0 = time.time()
pruned = np.empty((N, 32), np.uint32)
for s in range(0, N, 65536):
    e = min(N, s + 65536)
    kept, _ = idx.robust_prune(np.arange(s, e, dtype=np.uint32), cands[s:e], 32, alpha=1.2)
    pruned[s:e] = kept
print(f"robustPrune of 64 candidates x {N}: {time.time() - t0:.1f} s; mean degree {(pruned != INV).mean() * 32:.1f}")
evaluate("31-NN + 32 random, robustPrune", pruned)
both = add_reverse(pruned, 64)
print(f"with reverse edges: mean degree {(both != INV).sum(1).mean():.1f}")
evaluate("... + reverse edges (cap 64)", both)
knn_rev = add_reverse(knn, 64)
evaluate("exact 31-NN + reverse (cap 64)", knn_rev)
