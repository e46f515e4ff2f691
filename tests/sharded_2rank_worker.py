"""Worker of tests/test_gpu_sharded_2rank.py (run under torch.distributed.run, 2 ranks, gloo, BOTH on GPU 0): a
row-sharded search with the real pieces — each rank's shard resident in a vecgo_amd Index, the HIP local scorers
(flat exact / PQ-ADC / RaBitQ), the packed [2][nq][k] block all-gathered, vg_merge_topk_packed — must equal the
search of ONE index over the whole corpus.  Prints 'OK <checks>' on rank 0."""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
import torch.distributed as dist

import vecgo_amd as vg
from vecgo_amd import sharded


def main():
    dist.init_process_group("gloo")
    world, rank = dist.get_world_size(), dist.get_rank()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    ctx = vg.Context(0)
    n, dim, nq, k = 30011, 128, 37, 10          # ragged: the shards differ in size
    g = torch.Generator(device=dev); g.manual_seed(1234)
    rows = torch.randn((n, dim), generator=g, device=dev)
    rows[n // 3] = rows[5]; rows[n - 7] = rows[5]                  # duplicates across the shard boundary: ties by row id
    q = torch.randn((nq, dim), generator=g, device=dev); q[2] = rows[5]
    bounds = sharded.partition(n, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    checks = 0
    # 1. exact flat search
    whole = vg.Index(ctx, n, dim); whole.set_vectors(rows)
    want_i, want_s = whole.search_flat(q, k)
    sh = sharded.ShardedFlatIndex(ctx, rows[lo:hi].contiguous(), dim, bounds, metric=0)
    got_i, got_s = sh.search(q, k)
    assert torch.equal(got_i, want_i) and torch.equal(got_s.view(torch.int32), want_s.view(torch.int32)), "flat"
    checks += 1
    # 2. PQ-ADC and RaBitQ scans through sharded_search with the HIP scorers and the packed merge
    pq = vg.ProductQuantizer(ctx, dim, 16, 256); pq.train(rows[:4096], iters=4, seed=3)
    codes = pq.encode(rows)
    whole.set_pq_codes(pq, codes)
    rcodes = vg.RaBitQuantizer(ctx, dim).encode(rows)
    whole.set_rabitq_codes(rcodes)
    local = vg.Index(ctx, hi - lo, dim)
    local.set_pq_codes(pq, codes[lo:hi].contiguous()); local.set_rabitq_codes(rcodes[lo:hi].contiguous())

    def merge(packed, kk, off):
        lists, _, nq_, _ = packed.shape
        return vg.merge_topk_packed(ctx, packed, lists, nq_, kk, metric=0, id_offsets=off)
    for name, fw, fl in (("adc", whole.search_pq_adc, local.search_pq_adc), ("rabitq", whole.search_rabitq, local.search_rabitq)):
        wi, ws = fw(q, k)
        gi, gs = sharded.sharded_search(lambda qq, kk, out: fl(qq, kk, out=out), merge, q, k, bounds, None, metric=0)
        assert torch.equal(gi, wi) and torch.equal(gs.view(torch.int32), ws.view(torch.int32)), name
        checks += 1
    # 3. the metric's pipeline over query-sharded replicas (sharded.ReplicatedGraphIndex): every rank holds the whole index
    # (graph + PQ codes + fp32 rows) and answers its slice of the batch; the gathered result = the single-process one, in order
    whole.build_hnsw(m=8, ef_construction=100, max_batch=64, growth_div=32)
    for ef in (32, 64):
        cand, _ = whole.search_hnsw_pq(q, ef, ef)
        wi, ws = whole.rerank(q, cand, k)
        gi, gs = sharded.ReplicatedGraphIndex(whole).search(q, k, ef)
        assert torch.equal(gi.view(torch.int32), wi.view(torch.int32)) and torch.equal(gs.view(torch.int32), ws.view(torch.int32)), "replicas"
        fi, fs = whole.search_hnsw(q, k, ef)
        gi, gs = sharded.ReplicatedGraphIndex(whole).search_f32(q, k, ef)
        assert torch.equal(gi.view(torch.int32), fi.view(torch.int32)) and torch.equal(gs.view(torch.int32), fs.view(torch.int32)), "replicas f32"
        checks += 2
    dist.barrier()
    if rank == 0:
        print(f"OK {checks}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
