"""Row-sharded search across the GPUs of one node (one process per GPU).

The reference is single-process; its engine fans a query out over segments and merges the
per-segment candidate lists into one bounded heap (internal/engine/search.go:835-908).  Here
the "segments" are contiguous row shards, one per rank: every rank scores the same query batch
against its shard, the per-shard top-k (k ids + k scores per query) is exchanged with ONE
all-gather (RCCL over xGMI; nq*k*8 bytes per rank), and every rank merges with the reference's
tie-break (score, then RowID — searcher/candidate_queue.go:12-23).  No other collective.
"""
from __future__ import annotations

from typing import Callable, Sequence

import torch
import torch.distributed as dist


def partition(n: int, world: int) -> list[int]:
    """Shard bounds: rank r owns rows [bounds[r], bounds[r+1])."""
    return [n * r // world for r in range(world + 1)]


def sharded_search(local_search: Callable, merge: Callable, queries, k: int, bounds: Sequence[int],
                   group=None):
    """local_search(queries, k) -> (ids[nq,k] int32 bit-pattern of uint32, LOCAL row ids;
    scores[nq,k] f32) on this rank's shard.  merge(ids[world,nq,k], scores[world,nq,k], k,
    offsets[world]) -> (ids[nq,k], scores[nq,k]) global.  Returns the merged global result."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    ids, scores = local_search(queries, k)
    if world == 1:
        off = torch.tensor([bounds[0]], dtype=torch.int32, device=ids.device)
        return merge(ids.unsqueeze(0), scores.unsqueeze(0), k, off)
    nq = ids.shape[0]
    # gathered layout = concatenation along dim 0 (accepted by both RCCL and gloo): [world*nq, k]
    all_ids = torch.empty((world * nq, k), dtype=ids.dtype, device=ids.device)
    all_scores = torch.empty((world * nq, k), dtype=scores.dtype, device=scores.device)
    dist.all_gather_into_tensor(all_ids, ids.contiguous(), group=group)
    dist.all_gather_into_tensor(all_scores, scores.contiguous(), group=group)
    off = torch.tensor(list(bounds[:-1]), dtype=torch.int32, device=ids.device)
    return merge(all_ids.view(world, nq, k), all_scores.view(world, nq, k), k, off)


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
