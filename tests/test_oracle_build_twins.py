"""The build-side CPU twins of bench.py (oracle/vg_cpu_bench.c: vgo_bench_build_run) return what the oracle's
one-at-a-time functions return — the timed loops are the checked loops."""
import numpy as np
import pytest

from oracle import oracle as o


@pytest.fixture(scope="module")
def data():
    rng = np.random.default_rng(77)
    n, dim, m = 700, 64, 8
    rows = rng.standard_normal((n, dim)).astype(np.float32)
    pq = o.ProductQuantizer(dim, m, 256)
    pq.train(rows, iters=3, seed=5)
    return rows, pq


@pytest.mark.parametrize("metric", [o.METRIC_L2, o.METRIC_DOT])
@pytest.mark.parametrize("ref_kernels", [False, True])
def test_km_assign_twin(data, metric, ref_kernels):
    rows, _ = data
    cent = rows[::23][:17].copy()
    o.use_reference_kernels(ref_kernels)
    try:
        r = o.bench_build_run(o.BUILD_KM_ASSIGN, rows, 3, 0.0, metric=metric, centroids=cent, want_out=True)
        # budget 0: one unit per thread; a second run with a budget covers every row
        r = o.bench_build_run(o.BUILD_KM_ASSIGN, rows, 3, 0.2, metric=metric, centroids=cent, want_out=True)
    finally:
        o.use_reference_kernels(False)
    assert r["units"] >= rows.shape[0]
    want = np.array([o.assign_partition(v, cent, rows.shape[1], metric) for v in rows], np.int32)
    assert np.array_equal(r["out"]["assign"], want)


def test_pq_encode_and_lut_twin(data):
    rows, pq = data
    r = o.bench_build_run(o.BUILD_PQ_ENCODE, rows, 4, 0.2, pq=pq, want_out=True)
    assert r["units"] >= rows.shape[0]
    assert np.array_equal(r["out"]["codes"], pq.encode_batch(rows))
    r = o.bench_build_run(o.BUILD_PQ_LUT, rows[:32], 4, 0.05, pq=pq)
    assert r["units"] >= 4 and r["rate"] > 0


def test_rerank_twin(data):
    rows, _ = data
    rng = np.random.default_rng(3)
    q = rng.standard_normal((9, rows.shape[1])).astype(np.float32)
    cand = rng.integers(0, rows.shape[0], (9, 40)).astype(np.uint32)
    r = o.bench_build_run(o.BUILD_RERANK, q, 3, 0.1, base=rows, cand=cand, topk=5, want_out=True)
    for i in range(9):
        sc = o.rerank_f32(rows, rows.shape[1], q[i], cand[i])
        order = sorted(range(40), key=lambda c: (sc[c], cand[i][c]))[:5]
        assert list(r["out"]["ids"][i]) == [int(cand[i][c]) for c in order]
        assert np.array_equal(r["out"]["scores"][i].view(np.uint32), np.asarray([sc[c] for c in order], np.float32).view(np.uint32))


def test_pq_train_subspace_twin(data):
    rows, _ = data
    pq = o.ProductQuantizer(rows.shape[1], 8, 16)
    pq.train(rows, iters=4, seed=9)
    r = o.bench_build_run(o.BUILD_PQ_TRAIN_SUB, rows, 4, 0.0, pq_m=8, pq_k=16, iters=4, seed=9, want_out=True)
    got = r["out"]["cent"].reshape(-1)
    want = pq.centroids_f32
    # budget 0 = one unit per thread: the first 4 sub-quantizers
    per = 16 * (rows.shape[1] // 8)
    assert np.array_equal(got[:4 * per].view(np.uint32), want[:4 * per].view(np.uint32))
