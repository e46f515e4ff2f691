"""The exchange step behind the C ABI (vg_comm.hip) as far as one GPU can exercise it: RCCL is loaded and
joined with world = 1 (the all-gather is then a copy), and the packed merge is checked with several lists
against the dense merge and the oracle's CandidateHeap order."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vg():
    import vecgo_amd
    return vecgo_amd


@pytest.fixture(scope="module")
def ctx(vg):
    return vg.Context(0)


def test_packed_merge_equals_dense_merge(vg, ctx):
    import torch
    from vecgo_amd import api
    rng = np.random.default_rng(1)
    lists, nq, k = 5, 37, 10
    ids = rng.integers(0, 1000, (lists, nq, k)).astype(np.uint32)
    sc = np.sort(rng.integers(0, 50, (lists, nq, k)).astype(np.float32), axis=2)  # ties across lists
    ids[2, :, 7:] = 0xFFFFFFFF                                                    # short lists
    off = (np.arange(lists) * 1000).astype(np.uint32)
    dense = api.merge_topk(ctx, ids, sc, k, metric=0, id_offsets=off)
    packed = np.stack([ids.view(np.int32), sc.view(np.int32)], axis=1)            # [lists, 2, nq, k]
    dev = torch.device("cuda", 0)
    p = torch.from_numpy(packed).to(dev)
    o = torch.from_numpy(off.view(np.int32)).to(dev)
    got = api.merge_topk_packed(ctx, p, lists, nq, k, metric=0, id_offsets=o)
    assert np.array_equal(got[0].cpu().numpy().view(np.uint32), dense[0])
    assert np.array_equal(got[1].cpu().numpy().view(np.uint32), dense[1].view(np.uint32))
    # and both are the k smallest (score, global id) pairs
    for q in range(nq):
        cand = sorted((float(sc[l, q, j]), int(ids[l, q, j]) + int(off[l])) for l in range(lists) for j in range(k)
                      if ids[l, q, j] != 0xFFFFFFFF)[:k]
        assert [c[1] for c in cand] == dense[0][q].tolist()


def test_comm_world_of_one(vg, ctx):
    import torch
    from vecgo_amd import api
    import torch.distributed  # noqa: F401  (torch maps its own librccl when RCCL is first used; probe() must reuse a mapped one)
    path = api.Comm.probe()
    assert "rccl" in path
    uid = api.Comm.unique_id()
    assert len(uid) == 128
    comm = api.Comm(ctx, 1, 0, uid)
    d = comm.describe()      # what RCCL itself says about the communicator
    assert d["rccl_ranks"] == 1 and d["rccl_rank"] == 0 and d["rccl_device"] == ctx.device and d["rccl_path"] == path
    # one RCCL per process: if a librccl was mapped before ours was resolved, ours is that file
    mapped = {line.split()[-1] for line in open("/proc/self/maps") if "librccl" in line}
    assert len(mapped) == 1, mapped
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(2)
    nq, k = 19, 10
    sc = np.sort(rng.standard_normal((nq, k)).astype(np.float32), axis=1)
    ids = rng.integers(0, 500, (nq, k)).astype(np.uint32)
    out = comm.all_gather_topk(torch.from_numpy(ids.view(np.int32)).to(dev), torch.from_numpy(sc).to(dev), k,
                               id_offsets=torch.tensor([7], dtype=torch.int32, device=dev))
    want = api.merge_topk(ctx, ids[None], sc[None], k, metric=0, id_offsets=np.array([7], np.uint32))
    assert np.array_equal(out[0].cpu().numpy().view(np.uint32), want[0])
    assert np.array_equal(out[1].cpu().numpy().view(np.uint32), want[1].view(np.uint32))
    a = torch.arange(1000, dtype=torch.uint8, device=dev)
    b = torch.zeros(1000, dtype=torch.uint8, device=dev)
    comm.all_gather_bytes(a, b)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    comm.close()


def test_sharded_flat_index_on_one_rank(vg, ctx):
    """world = 1 through vecgo_amd.sharded: the packed block written in place by the local search, merged."""
    import torch
    from vecgo_amd import sharded
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(3)
    rows = torch.randn((5000, 64), device=dev, generator=g)
    q = torch.randn((33, 64), device=dev, generator=g)
    idx = sharded.ShardedFlatIndex(ctx, rows, 64, [0, 5000])
    ids, sc = idx.search(q, 10)
    want = idx.index.search_flat(q, 10)
    assert torch.equal(ids, want[0]) and torch.equal(sc, want[1])


def test_sharded_flat_index_with_a_filter(vg, ctx):
    """ShardedFlatIndex.search_filtered on one rank = the index's own filtered search (every rank filters its own rows)"""
    import torch
    from vecgo_amd import sharded
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(4)
    rows = torch.randn((6000, 64), device=dev, generator=g)
    q = torch.randn((20, 64), device=dev, generator=g)
    mask = (torch.rand((20, 6000), device=dev, generator=g) < 0.3).cpu().numpy()
    idx = sharded.ShardedFlatIndex(ctx, rows, 64, [0, 6000])
    ids, sc = idx.search_filtered(q, 10, mask)
    want = idx.index.search_flat_filtered(q, 10, mask, 0)
    assert torch.equal(ids, want[0]) and torch.equal(sc, want[1])
    assert mask[np.arange(20)[:, None], ids.cpu().numpy().astype(np.int64)].all()


def test_sharded_sq8_index_on_one_rank(vg, ctx):
    """train_sq8_sharded = ScalarQuantizer.Train bit for bit; ShardedSQ8Index (nomination on) = the index's own searches"""
    import torch
    from vecgo_amd import sharded
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(5)
    rows = torch.randn((7000, 64), device=dev, generator=g)
    rows[:, 9] = 2.5
    q = torch.randn((24, 64), device=dev, generator=g)
    sq = vg.ScalarQuantizer(ctx, 64)
    sharded.train_sq8_sharded(sq, rows)
    full = vg.ScalarQuantizer(ctx, 64); full.train(rows)
    for a, b in zip(sq.params(), full.params()):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    codes = sq.encode(rows)
    idx = sharded.ShardedSQ8Index(ctx, sq, codes, 7000, 64, [0, 7000], nomination=True)
    mask = (torch.rand((24, 7000), device=dev, generator=g) < 0.3).cpu().numpy()
    for k in (10, 100):
        ids, sc = idx.search(q, k)
        want = idx.index.search_sq8(q, k)
        assert torch.equal(ids, want[0]) and torch.equal(sc, want[1])
        ids, sc = idx.search_filtered(q, k, mask)
        want = idx.index.search_flat_filtered(q, k, mask, 0, scan=idx.index.SCAN_SQ8)
        assert torch.equal(ids, want[0]) and torch.equal(sc, want[1])
