"""N>1 path on CPU: two gloo ranks, row-sharded corpus, one all-gather of per-shard top-k,
merge with the reference tie-break.  The local scorer and the merge are injected (oracle /
numpy) because no HIP device exists here; what is under test is vecgo_amd.sharded's
partitioning, collective layout and id offsets."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]


def _np_merge(packed, k, off):
    """the packed image sharded_search hands to vg_merge_topk_packed: [world, 2, nq, k] int32"""
    ids, scores = packed[:, 0], packed[:, 1].contiguous().view(torch.float32)
    ids = ids.numpy().view(np.uint32).astype(np.int64) + off.numpy().astype(np.int64)[:, None, None]
    sc = scores.numpy()
    world, nq, kk = sc.shape
    out_i = np.zeros((nq, k), np.uint32); out_s = np.zeros((nq, k), np.float32)
    for q in range(nq):
        cand = [(sc[w, q, j], ids[w, q, j]) for w in range(world) for j in range(kk)
                if ids[w, q, j] - off.numpy()[w] != 0xFFFFFFFF]
        cand.sort()
        for j, (s, i) in enumerate(cand[:k]):
            out_i[q, j] = i; out_s[q, j] = s
    return torch.from_numpy(out_i.view(np.int32)), torch.from_numpy(out_s)


def _worker(rank, world, port, n, dim, nq, k, ret):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as o
    from vecgo_amd import sharded
    rng = np.random.default_rng(123)
    base = rng.standard_normal((n, dim)).astype(np.float32)
    queries = rng.standard_normal((nq, dim)).astype(np.float32)
    bounds = sharded.partition(n, world)
    lo, hi = bounds[rank], bounds[rank + 1]

    def local(q, kk, out):
        ids = np.full((nq, kk), 0xFFFFFFFF, np.uint32); sc = np.full((nq, kk), np.inf, np.float32)
        for i in range(nq):
            a, b = o.flat_search_f32(base[lo:hi], dim, q[i], kk)
            ids[i, :a.size] = a; sc[i, :b.size] = b
        if rank == 0:   # one rank fills the caller's block in place, the other returns fresh tensors
            out[0].copy_(torch.from_numpy(ids.view(np.int32))); out[1].copy_(torch.from_numpy(sc))
            return out
        return torch.from_numpy(ids.view(np.int32)), torch.from_numpy(sc)

    ids, sc = sharded.sharded_search(local, _np_merge, queries, k, bounds)
    ok = True
    for i in range(nq):
        eid, esc = o.flat_search_f32(base, dim, queries[i], k)
        ok &= np.array_equal(ids[i].numpy().view(np.uint32), eid)
        ok &= np.array_equal(sc[i].numpy().view(np.uint32), esc.view(np.uint32))
    ret[rank] = bool(ok)
    dist.destroy_process_group()


def test_partition_covers_rows():
    from vecgo_amd import sharded
    for n in (0, 1, 7, 1000, 1_000_003):
        for w in (1, 2, 3, 8):
            b = sharded.partition(n, w)
            assert b[0] == 0 and b[-1] == n and all(b[i] <= b[i + 1] for i in range(w))


@pytest.mark.parametrize("n", [501, 64])
def test_two_rank_sharded_search_equals_single_segment(n):
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, n, 32, 5, 10, ret), nprocs=world, join=True)
    assert ret[0] and ret[1]


class _OraclePQ:
    """Stands in for vecgo_amd.ProductQuantizer on a box without a GPU: trains with the CPU
    oracle (sub-quantizers are independent and their random streams keyed by (seed, sub), so
    training everything and handing out a range is what vg_pq_train_subset computes)."""

    def __init__(self, dim, m, k):
        from oracle import oracle as o
        self._o = o.ProductQuantizer(dim, m, k)
        self.num_subvectors, self.num_centroids, self.subvector_dim = m, k, dim // m
        self.result = None

    def train_subset(self, vectors, lo, cnt, iters=20, seed=1, stream=None):
        self._o.train(np.asarray(vectors), iters=iters, seed=seed)
        self._range = (lo, cnt)

    def codebooks_range(self, lo, cnt):
        per = self.num_centroids * self.subvector_dim
        return (self._o.codebooks[lo * per:(lo + cnt) * per].copy(), self._o.scales[lo:lo + cnt].copy(),
                self._o.offsets[lo:lo + cnt].copy())

    def set_codebooks(self, cb, sc, of):
        self.result = (np.asarray(cb).copy(), np.asarray(sc).copy(), np.asarray(of).copy())


def _pq_worker(rank, world, port, ret):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as o
    from vecgo_amd import sharded
    dim, m, k = 24, 6, 16
    x = np.random.default_rng(5).standard_normal((300, dim)).astype(np.float32)
    pq = _OraclePQ(dim, m, k)
    # poison the ranges this rank does not own: the result must come from the owners
    orig = pq.codebooks_range
    sharded.train_pq_sharded(pq, torch.from_numpy(x), iters=5, seed=9)
    full = o.ProductQuantizer(dim, m, k)
    full.train(x, iters=5, seed=9)
    cb, sc, of = pq.result
    ret[rank] = bool(np.array_equal(cb, full.codebooks) and np.array_equal(sc.view(np.uint32), full.scales.view(np.uint32))
                     and np.array_equal(of.view(np.uint32), full.offsets.view(np.uint32)))
    dist.destroy_process_group()


def test_two_rank_sharded_pq_training_equals_single_process():
    world = 2
    port = 31500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_pq_worker, args=(world, port, ret), nprocs=world, join=True)
    assert ret[0] and ret[1]


def _sq_worker(rank, world, port, ret):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as o
    from vecgo_amd import sharded
    dim = 20
    x = np.random.default_rng(8).standard_normal((301, dim)).astype(np.float32)
    x[:, 3] = 5.0                                   # a constant dimension: Train's own handling (max = min + 1e-6), not SetBounds'
    x[:150, 7] = 100.0                              # constant on rank 0's rows only
    bounds = sharded.partition(x.shape[0], world)
    ok = True
    for rows in (x[bounds[rank]:bounds[rank + 1]], torch.from_numpy(x[bounds[rank]:bounds[rank + 1]]),
                 x[:0] if rank == 1 else x):        # numpy shards, torch shards, an empty shard on rank 1
        sq = o.ScalarQuantizer(dim)
        sharded.train_sq8_sharded(sq, rows)
        full = o.ScalarQuantizer(dim); full.train(x)
        for a, b in ((sq.mins, full.mins), (sq.maxs, full.maxs), (sq.scales, full.scales), (sq.inv_scales, full.inv_scales)):
            ok &= bool(np.array_equal(a.view(np.uint32), b.view(np.uint32)))
    ret[rank] = ok
    dist.destroy_process_group()


def test_two_rank_sharded_sq8_training_equals_single_process():
    world = 2
    port = 33500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_sq_worker, args=(world, port, ret), nprocs=world, join=True)
    assert ret[0] and ret[1]


def _replica_worker(rank, world, port, nq, ret):
    """Query-sharded replicas (sharded.replicated_search): every rank holds the whole corpus and answers its slice of the batch;
    the local search is the oracle's HNSW-style stand-in (exact flat search: what is under test is the split, the padding of the
    ragged last slice, the single all-gather and the order of the concatenation)."""
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as o
    from vecgo_amd import sharded
    rng = np.random.default_rng(77)
    dim, n, k = 24, 400, 7
    base = rng.standard_normal((n, dim)).astype(np.float32)
    queries = rng.standard_normal((nq, dim)).astype(np.float32)
    calls = []

    def local(q, kk):
        calls.append(len(q))
        ids = np.zeros((len(q), kk), np.uint32); sc = np.zeros((len(q), kk), np.float32)
        for i in range(len(q)):
            ids[i], sc[i] = o.flat_search_f32(base, dim, q[i], kk)
        return (torch.from_numpy(ids.view(np.int32)), torch.from_numpy(sc)) if rank == 0 else (ids, sc)   # tensors or arrays

    ids, sc = sharded.replicated_search(local, torch.from_numpy(queries) if rank == 0 else queries, k)
    ok = ids.shape == (nq, k) and sc.shape == (nq, k)
    qb = sharded.partition(nq, world)
    ok &= calls == ([qb[rank + 1] - qb[rank]] if qb[rank + 1] > qb[rank] else [])     # one local call, on this rank's slice only
    for i in range(nq):
        eid, esc = o.flat_search_f32(base, dim, queries[i], k)
        ok &= np.array_equal(ids[i].numpy().view(np.uint32), eid) and np.array_equal(sc[i].numpy().view(np.uint32), esc.view(np.uint32))
    ret[rank] = bool(ok)
    dist.destroy_process_group()


@pytest.mark.parametrize("nq", [9, 1, 2])      # ragged slices; fewer queries than ranks (an empty slice)
def test_two_rank_query_sharded_replicas_equal_single_process(nq):
    world = 2
    port = 35500 + (os.getpid() % 2000) + nq
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_replica_worker, args=(world, port, nq, ret), nprocs=world, join=True)
    assert ret[0] and ret[1]


def test_sq8_sharded_train_skips_nan_and_refuses_empty():
    """train_sq8_sharded at world 1: a NaN is skipped as the reference's `val < min` / `val > max` loop skips it
    (quantizer.go:156-163), and a corpus with no rows is an error as in Train (quantizer.go:128-130)."""
    from oracle import oracle as o
    from vecgo_amd import sharded
    x = np.random.default_rng(4).standard_normal((50, 6)).astype(np.float32)
    y = x.copy(); y[7, 2] = np.nan
    clean = np.delete(x, 7, axis=0) if False else x.copy()
    clean[7, 2] = x[:, 2].min()          # replacing the NaN by a value inside the range leaves min / max as if it were skipped
    for rows in (y, torch.from_numpy(y)):
        a, b = o.ScalarQuantizer(6), o.ScalarQuantizer(6)
        sharded.train_sq8_sharded(a, rows)
        b.train(clean)
        assert np.array_equal(a.mins.view(np.uint32), b.mins.view(np.uint32)) and np.array_equal(a.maxs.view(np.uint32), b.maxs.view(np.uint32))
    with pytest.raises(ValueError):
        sharded.train_sq8_sharded(o.ScalarQuantizer(6), x[:0])
