"""Does building the HNSW graph in batches cost recall?  vg_hnsw_build inserts nodes in id order in batches of
clamp(inserted / growth_div, 1, max_batch): every node of a batch searches the graph as it stood when the batch began
(what ApplyBatchInsert's goroutines are to one another, hnsw.go:639-684); max_batch = 1 is the reference's sequential
hnsw.Insert loop (hnsw.go:579).  This tool builds the same rows (bench generator, i.i.d. normal x 768, M = 32,
EF = 300) both ways and prints recall@10 vs ef for each: SMALL rows sequentially (max_batch = 1; minutes on a GPU — a
build is one dependent chain of inserts) against the bench's setting (8192 / 32), and LARGE rows with small batches
(max_batch = 64) against the bench's setting.  argv: [SMALL [LARGE]]; writes one JSON object."""
import sys, time, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
import vecgo_amd as vg, bench

SMALL = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000
LARGE = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
D, K, EFS = 768, 10, (64, 128, 256, 512, 1024)
ctx = vg.Context(0); dev = torch.device("cuda", 0)
queries = bench.gen_queries(1, dev).reshape(-1, D)[:1024].contiguous()


def frontier(n, max_batch, growth_div):
    rows = bench.gen_rows(0, n, dev)
    idx = vg.Index(ctx, n, D); idx.set_vectors(rows)
    torch.cuda.synchronize(); t0 = time.time()
    idx.build_hnsw(m=32, ef_construction=300, max_batch=max_batch, growth_div=growth_div)
    torch.cuda.synchronize(); secs = time.time() - t0
    gt = idx.search_flat(queries, K)[0].cpu().numpy().view(np.uint32)
    l0, _, _ = idx.get_hnsw_graph()
    out = {"rows": n, "max_batch": max_batch, "growth_div": growth_div, "build_s": secs,
           "mean_layer0_degree": float((l0 != 0xFFFFFFFF).sum(1).mean()), "recall_at_10": {}}
    for ef in EFS:
        got = idx.search_hnsw(queries, K, ef)[0].cpu().numpy().view(np.uint32)
        out["recall_at_10"][str(ef)] = float(np.mean([len(set(got[i]) & set(gt[i])) / K for i in range(1024)]))
    idx.close()
    return out


res = {"workload": "bench generator rows x 768 (i.i.d. normal), HNSW M=32 EF=300, recall@10 of vg_search_hnsw over 1024 queries "
                   "against the exact search", "small": [], "large": []}
for mb, gd in ((1, 1), (8192, 32)):
    res["small"].append(frontier(SMALL, mb, gd)); print(json.dumps(res["small"][-1]), flush=True)
for mb, gd in ((64, 32), (8192, 32)) if LARGE > 0 else ():
    res["large"].append(frontier(LARGE, mb, gd)); print(json.dumps(res["large"][-1]), flush=True)
print("RESULT " + json.dumps(res))
