"""Row-sharded search across the GPUs of one node (one process per GPU).

The reference is single-process; its engine fans a query out over segments and merges the
per-segment candidate lists into one bounded heap (internal/engine/search.go:835-908).  Here
the "segments" are contiguous row shards, one per rank: every rank scores the same query batch
against its shard, the per-shard top-k (k ids + k scores per query) is exchanged with ONE
all-gather (RCCL over xGMI; nq*k*8 bytes per rank, ids and score bits in one buffer), and every rank merges with the reference's
tie-break (score, then RowID — searcher/candidate_queue.go:12-23).  No other collective.
"""
from __future__ import annotations

from typing import Callable, Sequence

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
                   group=None):
    """local_search(queries, k) -> (ids[nq,k] int32 bit-pattern of uint32, LOCAL row ids;
    scores[nq,k] f32) on this rank's shard.  merge(ids[world,nq,k], scores[world,nq,k], k,
    offsets[world]) -> (ids[nq,k], scores[nq,k]) global.  Returns the merged global result."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    ids, scores = local_search(queries, k)
    if world == 1:
        return merge(ids.unsqueeze(0), scores.unsqueeze(0), k, _shard_offsets(bounds, 1, ids.device))
    nq = ids.shape[0]
    # ONE collective per search: ids and the scores' bit patterns travel in the same int32 buffer
    # ([2, nq, k] per rank); gathered layout = concatenation along dim 0 (accepted by RCCL and gloo)
    mine = torch.stack((ids.contiguous().view(torch.int32), scores.contiguous().view(torch.int32)))
    gathered = torch.empty((world * 2 * nq, k), dtype=torch.int32, device=ids.device)
    dist.all_gather_into_tensor(gathered, mine.view(2 * nq, k), group=group)
    gathered = gathered.view(world, 2, nq, k)
    all_ids = gathered[:, 0].contiguous().view(ids.dtype)
    all_scores = gathered[:, 1].contiguous().view(torch.float32)
    return merge(all_ids, all_scores, k, _shard_offsets(bounds, world, ids.device))


class ShardedFlatIndex:
    """Exact brute force over a row-sharded corpus (BASELINE config 2 at N GPUs)."""

    def __init__(self, ctx, local_rows, dim: int, bounds: Sequence[int], metric=0, group=None):
        from . import api
        self._api = api
        self.ctx, self.dim, self.bounds, self.metric, self.group = ctx, dim, list(bounds), metric, group
        n_local = local_rows.shape[0]
        self.index = api.Index(ctx, n_local, dim, api.Metric(metric))
        self.index.set_vectors(local_rows)

    def search(self, queries, k: int, stream=None):
        def local(q, kk):
            return self.index.search_flat(q, kk, stream=stream)

        def merge(ids, scores, kk, off):
            return self._api.merge_topk(self.ctx, ids, scores, kk, metric=self.metric, id_offsets=off,
                                        stream=stream)
        return sharded_search(local, merge, queries, k, self.bounds, self.group)


def assemble_codebooks(gathered_cb, gathered_scales, gathered_offsets, bounds: Sequence[int], per_sub: int):
    """gathered_*[r] = full-size arrays of rank r in which only the sub-quantizers
    [bounds[r], bounds[r+1]) are meaningful; returns the arrays with every range taken from its
    owner.  gathered_cb: [world, m*per_sub] int8; gathered_scales/offsets: [world, m] float32."""
    cb = gathered_cb[0].clone()
    sc = gathered_scales[0].clone()
    of = gathered_offsets[0].clone()
    for r in range(1, len(bounds) - 1):
        lo, hi = bounds[r], bounds[r + 1]
        cb[lo * per_sub:hi * per_sub] = gathered_cb[r][lo * per_sub:hi * per_sub]
        sc[lo:hi] = gathered_scales[r][lo:hi]
        of[lo:hi] = gathered_offsets[r][lo:hi]
    return cb, sc, of


def train_pq_sharded(pq, vectors, iters: int = 20, seed: int = 1, group=None, device=None, stream=None):
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
    dev = device if device is not None else (vectors.device if isinstance(vectors, torch.Tensor) else "cpu")
    cb = torch.zeros(m * per, dtype=torch.int8, device=dev)
    sc = torch.zeros(m, dtype=torch.float32, device=dev)
    of = torch.zeros(m, dtype=torch.float32, device=dev)
    cb[lo * per:hi * per] = torch.from_numpy(cb_l).to(dev)
    sc[lo:hi] = torch.from_numpy(sc_l).to(dev)
    of[lo:hi] = torch.from_numpy(of_l).to(dev)
    g_cb = torch.empty((world * m * per,), dtype=torch.int8, device=dev)
    g_sc = torch.empty((world * m,), dtype=torch.float32, device=dev)
    g_of = torch.empty((world * m,), dtype=torch.float32, device=dev)
    dist.all_gather_into_tensor(g_cb, cb, group=group)
    dist.all_gather_into_tensor(g_sc, sc, group=group)
    dist.all_gather_into_tensor(g_of, of, group=group)
    cb, sc, of = assemble_codebooks(g_cb.view(world, -1), g_sc.view(world, -1), g_of.view(world, -1), bounds, per)
    pq.set_codebooks(cb.cpu().numpy(), sc.cpu().numpy(), of.cpu().numpy())


class ShardedRaBitQIndex:
    """Exhaustive RaBitQ scan over a row-sharded corpus (BASELINE configs[4]: 10M x 768 split 8 ways)."""

    def __init__(self, ctx, local_codes, n_local: int, dim: int, bounds: Sequence[int], group=None):
        from . import api
        self._api = api
        self.ctx, self.dim, self.bounds, self.group = ctx, dim, list(bounds), group
        self.index = api.Index(ctx, n_local, dim, api.Metric(0))
        self.index.set_rabitq_codes(local_codes)

    def search(self, queries, k: int, stream=None):
        def local(q, kk):
            return self.index.search_rabitq(q, kk, stream=stream)

        def merge(ids, scores, kk, off):
            return self._api.merge_topk(self.ctx, ids, scores, kk, metric=0, id_offsets=off, stream=stream)
        return sharded_search(local, merge, queries, k, self.bounds, self.group)
