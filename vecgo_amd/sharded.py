"""Row-sharded search across the GPUs of one node (one process per GPU).

The reference is single-process; its engine fans a query out over segments and merges the
per-segment candidate lists into one bounded heap (internal/engine/search.go:835-908).  Here
the "segments" are contiguous row shards, one per rank: every rank scores the same query batch
against its shard, the per-shard top-k (k ids + k scores per query) is exchanged with ONE
all-gather (RCCL over xGMI; nq*k*8 bytes per rank, ids and score bits in one buffer), and every rank merges with the reference's
tie-break (score, then RowID — searcher/candidate_queue.go:12-23).  No other collective.

Graph search does not shard by rows (a traversal needs the whole graph: SURVEY.md section 8e, "replicas only"): there every rank
holds a REPLICA and takes a contiguous slice of the query batch (replicated_search, ReplicatedGraphIndex) — the reference's
analogue is one goroutine per query over one shared index (engine/search.go:835-908 fans out per segment, benchmark_test's
concurrent searches per query).  One all-gather of the per-slice results, concatenated in query order; nothing to merge.
"""
from __future__ import annotations

from typing import Callable, Sequence

import numpy as np
import torch
import torch.distributed as dist


def partition(n: int, world: int) -> list[int]:
    """Shard bounds: rank r owns rows [bounds[r], bounds[r+1])."""
    return [n * r // world for r in range(world + 1)]


_OFFSETS = {}


def _shard_offsets(bounds: Sequence[int], world: int, device):
    """First global row of every shard as a device tensor, built once per (bounds, device): creating
    it per search is a synchronous host-to-device copy that keeps the host from running ahead."""
    key = (tuple(bounds[:world]), str(device))
    off = _OFFSETS.get(key)
    if off is None:
        off = torch.tensor(list(bounds[:world]), dtype=torch.int32, device=device)
        _OFFSETS[key] = off
    return off


def sharded_search(local_search: Callable, merge: Callable, queries, k: int, bounds: Sequence[int],
                   group=None, comm=None, metric=0, stream=None):
    """local_search(queries, k, out=(ids, scores)) -> (ids[nq,k] int32 bit-pattern of uint32, LOCAL row ids;
    scores[nq,k] f32) on this rank's shard.  merge(packed[world,2,nq,k], k, offsets[world]) -> (ids[nq,k],
    scores[nq,k]) global.  ONE collective per search: ids and the scores' bit patterns of a rank are one
    [2, nq, k] int32 block, written in place by the local search; the gathered [world, 2, nq, k] image goes to
    the merge kernel as it lies (vg_merge_topk_packed).  comm = a vecgo_amd.Comm: the exchange runs through the
    C ABI (direct ncclAllGather); otherwise torch.distributed (RCCL on GPUs, gloo in the CPU tests)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    nq = queries.shape[0]
    if comm is not None:
        ids, scores = local_search(queries, k, None)
        return comm.all_gather_topk(ids, scores, k, metric=metric, id_offsets=_shard_offsets(bounds, world, ids.device),
                                    stream=stream)
    dev = queries.device if isinstance(queries, torch.Tensor) else "cpu"
    mine = torch.empty((2, nq, k), dtype=torch.int32, device=dev)
    out = local_search(queries, k, (mine[0], mine[1].view(torch.float32)))
    if out[0].data_ptr() != mine.data_ptr():   # a local search that does not take `out`
        mine[0].copy_(out[0].view(torch.int32))
        mine[1].copy_(out[1].view(torch.int32))
    if world == 1:
        return merge(mine.view(1, 2, nq, k), k, _shard_offsets(bounds, 1, dev))
    gathered = torch.empty((world * 2 * nq, k), dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(gathered, mine.view(2 * nq, k), group=group)
    return merge(gathered.view(world, 2, nq, k), k, _shard_offsets(bounds, world, dev))


def make_comm(ctx, group=None, report: dict = None):
    """A vecgo_amd.Comm spanning the process group: rank 0's id travels over torch.distributed's store.  Returns
    None (callers fall back to torch.distributed collectives) when the world is 1, the backend is not RCCL, or any
    rank cannot load RCCL / join — and says WHY in `report` (the bench line carries it: a run must not degrade to
    another collective silently).
    Order matters: vg_comm_create is a collective (ncclCommInitRank), so everything that can fail on ONE rank —
    loading RCCL, making the id — is done first and agreed on with an all-reduce; only then does every rank enter
    the collective together (a rank that failed before it left its peers blocked inside it)."""
    report = report if report is not None else {}
    report.update({"collective": "torch.distributed", "fell_back": True, "reason": None})
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        report["reason"] = "world size 1"
        return None
    if dist.get_backend(group) != "nccl":
        report["reason"] = f"torch.distributed backend is {dist.get_backend(group)}, not nccl (RCCL)"
        return None
    from . import api
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = torch.device("cuda", torch.cuda.current_device())
    why = ""
    try:
        api.Comm.probe()
        ok = 1
    except Exception as e:
        ok, why = 0, f"rank {rank}: {e}"
    box = [None]
    if rank == 0 and ok:
        try:
            box[0] = api.Comm.unique_id()
        except Exception as e:
            ok, why = 0, f"rank 0: {e}"
    flag = torch.tensor([ok], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    if int(flag.item()) == 0:          # some rank cannot load RCCL through the C ABI: nobody enters the collective
        whys = [None] * world
        dist.all_gather_object(whys, why, group=group)
        report["reason"] = "; ".join(w for w in whys if w) or "a rank could not load RCCL"
        return None
    dist.broadcast_object_list(box, src=0, group=group)
    comm = api.Comm(ctx, world, rank, box[0])     # collective; a failure here is a failure of the job
    info = comm.describe()
    devs = [None] * world
    dist.all_gather_object(devs, {"rank": rank, "hip_device": ctx.device, **info}, group=group)
    ranks_ok = all(d["rccl_ranks"] == world for d in devs)
    report.update({"collective": "vg_comm (ncclAllGather through the C ABI)", "fell_back": not ranks_ok,
                   "reason": None if ranks_ok else "ncclCommCount disagrees with the world size", "rccl_ranks": info["rccl_ranks"],
                   "rccl_path": info["rccl_path"], "reused_mapped_rccl": info["reused_mapped_rccl"], "per_rank": devs})
    if not ranks_ok:
        comm.close()
        return None
    return comm


class ShardedFlatIndex:
    """Exact brute force over a row-sharded corpus (BASELINE config 2 at N GPUs)."""

    def __init__(self, ctx, local_rows, dim: int, bounds: Sequence[int], metric=0, group=None, comm=None,
                 bf16_filter: bool = False):
        from . import api
        self._api = api
        self.ctx, self.dim, self.bounds, self.metric, self.group, self.comm = ctx, dim, list(bounds), metric, group, comm
        n_local = local_rows.shape[0]
        self.index = api.Index(ctx, n_local, dim, api.Metric(metric))
        self.index.set_vectors(local_rows)
        if bf16_filter:   # per-shard results stay exact, so the merged result does too
            self.index.enable_bf16_filter(True)

    def search(self, queries, k: int, stream=None):
        def local(q, kk, out):
            return self.index.search_flat(q, kk, out=out, stream=stream)

        def merge(packed, kk, off):
            lists, _, nq, _ = packed.shape
            return self._api.merge_topk_packed(self.ctx, packed, lists, nq, kk, metric=self.metric, id_offsets=off,
                                               stream=stream)
        return sharded_search(local, merge, queries, k, self.bounds, self.group, comm=self.comm, metric=self.metric,
                              stream=stream)

    def search_filtered(self, queries, k: int, local_mask, stream=None):
        """flat.Segment.Search with a row filter over the sharded corpus: every rank passes the filter bits of ITS rows
        (bool[n_local] / packed bits, one for the batch or one per query); the fan-in is the unfiltered one."""
        def local(q, kk, out):
            return self.index.search_flat_filtered(q, kk, local_mask, 0, out=out, stream=stream)

        def merge(packed, kk, off):
            lists, _, nq, _ = packed.shape
            return self._api.merge_topk_packed(self.ctx, packed, lists, nq, kk, metric=self.metric, id_offsets=off,
                                               stream=stream)
        return sharded_search(local, merge, queries, k, self.bounds, self.group, comm=self.comm, metric=self.metric,
                              stream=stream)


def replicated_search(local_search: Callable, queries, k: int, group=None):
    """Query-sharded search over replicas: rank r answers queries[qb[r]:qb[r+1]] (qb = partition(nq, world)) with
    local_search(q_slice, k) -> (ids[nq_r, k] int32 bit-pattern of uint32 GLOBAL row ids, scores[nq_r, k] f32) on its replica,
    and ONE all-gather of the [2, nq_pad, k] int32 blocks (ids + score bits; nq_pad = the largest slice, so that every rank
    contributes the same count) puts the whole batch on every rank, in query order.  No merge: a query is answered by exactly
    one rank, with the ids and score bits the single-process search returns."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    nq = queries.shape[0]
    qb = partition(nq, world)
    lo, hi = qb[rank], qb[rank + 1]
    pad = max(qb[r + 1] - qb[r] for r in range(world))
    q_local = queries[lo:hi]
    is_t = isinstance(queries, torch.Tensor)
    dev = queries.device if is_t else "cpu"
    mine = torch.zeros((2, pad, k), dtype=torch.int32, device=dev)
    if hi > lo:
        ids, scores = local_search(q_local, k)
        ids = ids if isinstance(ids, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(ids).view(np.int32))
        scores = scores if isinstance(scores, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(scores, np.float32))
        mine[0, :hi - lo].copy_(ids.view(torch.int32).reshape(hi - lo, k))
        mine[1, :hi - lo].copy_(scores.reshape(hi - lo, k).view(torch.int32))
    if world == 1:
        return mine[0, :nq], mine[1, :nq].view(torch.float32)
    gathered = torch.empty((world, 2, pad, k), dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(gathered.view(world * 2 * pad, k), mine.view(2 * pad, k), group=group)
    ids = torch.cat([gathered[r, 0, :qb[r + 1] - qb[r]] for r in range(world)])
    scores = torch.cat([gathered[r, 1, :qb[r + 1] - qb[r]] for r in range(world)]).view(torch.float32)
    return ids, scores


class ReplicatedGraphIndex:
    """The metric's pipeline — HNSW walk on PQ codes (ef candidates) + exact rerank (engine/search.go:914-965) — at N GPUs:
    every rank holds the whole index (graph + PQ codes + fp32 rows: 3.3 GB at 1M x 768, of 288 GB) and answers a contiguous
    slice of the query batch.  `index`: a vecgo_amd.Index with vectors, PQ codes and an HNSW graph attached."""

    def __init__(self, index, group=None):
        self.index, self.group = index, group

    def search(self, queries, k: int, ef: int, stream=None):
        def local(q, kk):
            cand, _ = self.index.search_hnsw_pq(q, ef, ef, stream=stream)
            return self.index.rerank(q, cand, kk, stream=stream)
        return replicated_search(local, queries, k, self.group)

    def search_f32(self, queries, k: int, ef: int, stream=None):
        """the fp32 walk (hnsw.KNNSearch) over the replicas, for the same split"""
        return replicated_search(lambda q, kk: self.index.search_hnsw(q, kk, ef, stream=stream), queries, k, self.group)


def train_pq_sharded(pq, vectors, iters: int = 20, seed: int = 1, group=None, device=None, stream=None, comm=None):
    """PQ training partitioned by sub-quantizer (BASELINE configs[4]; pq.go:83-138 runs the m
    k-means problems independently).  Every rank holds the same training sample, trains
    m/world sub-quantizers, then ONE all-gather of codebooks + scales + offsets (m*K*sd + 8m
    bytes per rank) and SetCodebooks.  The random stream is keyed by (seed, sub-quantizer), so
    the result equals single-GPU training bit for bit."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    m = pq.num_subvectors
    per = pq.num_centroids * pq.subvector_dim
    bounds = partition(m, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    pq.train_subset(vectors, lo, hi - lo, iters=iters, seed=seed, stream=stream)
    if world == 1:
        cb, sc, of = pq.codebooks_range(0, m)
        pq.set_codebooks(cb, sc, of)
        return
    cb_l, sc_l, of_l = pq.codebooks_range(lo, hi - lo)
    if device is not None:
        dev = device
    elif isinstance(vectors, torch.Tensor):
        dev = vectors.device
    elif dist.get_backend(group) == "nccl":   # host training sample on a GPU rank: RCCL needs device buffers
        dev = torch.device("cuda", torch.cuda.current_device())
    else:
        dev = "cpu"
    # ONE buffer per rank: [codebooks m*per int8 | scales 4m bytes | offsets 4m bytes], own range filled in
    import numpy as np
    nb = m * per + 8 * m
    host = np.zeros(nb, np.uint8)
    host[lo * per:hi * per] = cb_l.view(np.uint8)
    host[m * per + 4 * lo:m * per + 4 * hi] = sc_l.astype(np.float32).view(np.uint8)
    host[m * per + 4 * m + 4 * lo:m * per + 4 * m + 4 * hi] = of_l.astype(np.float32).view(np.uint8)
    mine = torch.from_numpy(host).to(dev)
    gathered = torch.empty((world * nb,), dtype=torch.uint8, device=dev)
    if comm is not None:
        comm.all_gather_bytes(mine, gathered, stream=stream)
    else:
        dist.all_gather_into_tensor(gathered, mine, group=group)
    g = gathered.view(world, nb).cpu().numpy()
    cb = np.empty(m * per, np.int8)
    sc = np.empty(m, np.float32)
    of = np.empty(m, np.float32)
    for r in range(world):
        a_, b_ = bounds[r], bounds[r + 1]
        cb[a_ * per:b_ * per] = g[r, a_ * per:b_ * per].view(np.int8)
        sc[a_:b_] = g[r, m * per + 4 * a_:m * per + 4 * b_].view(np.float32)
        of[a_:b_] = g[r, m * per + 4 * m + 4 * a_:m * per + 4 * m + 4 * b_].view(np.float32)
    pq.set_codebooks(cb, sc, of)


def train_sq8_sharded(sq, local_rows, group=None):
    """ScalarQuantizer.Train (quantizer.go:127-180) over a row-sharded corpus.  Train depends on the rows only through the
    per-dimension minimum and maximum, and those are associative: every rank reduces ITS rows, one all-reduce (MIN, MAX) makes
    them global, and Train itself runs on the two-row matrix [mins; maxs] — so the degenerate-dimension handling is Train's own
    (not SetBounds', which differs for max == min) and the quantizer equals the single-process one bit for bit.
    sq: a vecgo_amd.ScalarQuantizer (or anything with .train(rows)); local_rows: [n_local, dim] torch tensor or numpy array,
    n_local may be 0."""
    big = float(np.finfo(np.float32).max)
    # the reference's loop (`val < min` / `val > max`, quantizer.go:156-163) never takes a NaN: it is skipped, not propagated
    # (torch.amin / np.min would poison the dimension's bounds on every rank) — NaN -> +big for the minimum, -big for the maximum
    if isinstance(local_rows, torch.Tensor):
        dim, count = local_rows.shape[1], local_rows.shape[0]
        if count:
            x = local_rows.float()
            nan = torch.isnan(x)
            lo = torch.where(nan, torch.full_like(x, big), x).amin(dim=0)
            hi = torch.where(nan, torch.full_like(x, -big), x).amax(dim=0)
        else:
            lo = torch.full((dim,), big, dtype=torch.float32, device=local_rows.device)
            hi = -lo
    else:
        x = np.asarray(local_rows, np.float32)
        dim, count = x.shape[1], x.shape[0]
        nan = np.isnan(x)
        lo = torch.from_numpy(np.where(nan, np.float32(big), x).min(axis=0) if count else np.full(dim, big, np.float32))
        hi = torch.from_numpy(np.where(nan, np.float32(-big), x).max(axis=0) if count else np.full(dim, -big, np.float32))
    total = torch.tensor([count], dtype=torch.int64, device=lo.device)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
        dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group)
    if int(total.item()) == 0:
        raise ValueError("no vectors provided")   # quantizer.go:128-130, as single-process Train on an empty corpus
    sq.train(torch.stack([lo, hi]).cpu().numpy())


class ShardedSQ8Index:
    """flat.Segment.Search, SQ8 branch, over a row-sharded corpus: one quantizer for all shards (train_sq8_sharded), every rank
    scans the codes of its rows, the fan-in is the flat index's.  nomination: vg_index_enable_sq8_nomination on every shard
    (per-shard results stay exact, so the merged result does)."""

    def __init__(self, ctx, sq, local_codes, n_local: int, dim: int, bounds: Sequence[int], metric=0, group=None, comm=None,
                 nomination: bool = False):
        from . import api
        self._api = api
        self.ctx, self.dim, self.bounds, self.metric, self.group, self.comm = ctx, dim, list(bounds), metric, group, comm
        self.index = api.Index(ctx, n_local, dim, api.Metric(metric))
        self.index.set_sq8_codes(sq, local_codes)
        if nomination:
            self.index.enable_sq8_nomination(True)

    def _run(self, local, queries, k, stream):
        def merge(packed, kk, off):
            lists, _, nq, _ = packed.shape
            return self._api.merge_topk_packed(self.ctx, packed, lists, nq, kk, metric=self.metric, id_offsets=off, stream=stream)
        return sharded_search(local, merge, queries, k, self.bounds, self.group, comm=self.comm, metric=self.metric, stream=stream)

    def search(self, queries, k: int, stream=None):
        return self._run(lambda q, kk, out: self.index.search_sq8(q, kk, out=out, stream=stream), queries, k, stream)

    def search_filtered(self, queries, k: int, local_mask, stream=None):
        """with a row filter: every rank passes the filter bits of ITS rows (ShardedFlatIndex.search_filtered)"""
        return self._run(lambda q, kk, out: self.index.search_flat_filtered(q, kk, local_mask, 0, scan=self.index.SCAN_SQ8, out=out,
                                                                           stream=stream), queries, k, stream)


class ShardedRaBitQIndex:
    """Exhaustive RaBitQ scan over a row-sharded corpus (BASELINE configs[4]: 10M x 768 split 8 ways)."""

    def __init__(self, ctx, local_codes, n_local: int, dim: int, bounds: Sequence[int], group=None, comm=None):
        from . import api
        self._api = api
        self.ctx, self.dim, self.bounds, self.group, self.comm = ctx, dim, list(bounds), group, comm
        self.index = api.Index(ctx, n_local, dim, api.Metric(0))
        self.index.set_rabitq_codes(local_codes)

    def search(self, queries, k: int, stream=None):
        def local(q, kk, out):
            return self.index.search_rabitq(q, kk, out=out, stream=stream)

        def merge(packed, kk, off):
            lists, _, nq, _ = packed.shape
            return self._api.merge_topk_packed(self.ctx, packed, lists, nq, kk, metric=0, id_offsets=off, stream=stream)
        return sharded_search(local, merge, queries, k, self.bounds, self.group, comm=self.comm, metric=0, stream=stream)
