"""Randomised differential run of round 5's matrix-core paths against the oracle: kmeans.AssignPartition / TrainKMeans
(fp32 and bfloat16-split passes), ProductQuantizer.Encode and Train (bf16 nomination + reference-order fix), and
vg_search_hnsw_filtered.  Shapes, scales, cluster structure, duplicates and near-duplicates are drawn at random; every
mismatch is printed with its configuration; exit code 1 if any.  `python tools/fuzz_build_side.py [seconds] [seed]`"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import vecgo_amd as vg
from oracle import oracle as o
from tests import graphs
from tests.hooks import set_hook

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ctx = vg.Context(0)
bits = lambda x: np.asarray(x, np.float32).view(np.uint32)
fails = 0
runs = {"km_assign": 0, "km_train": 0, "pq_encode": 0, "pq_train": 0, "hnsw_filtered": 0}


def rows(n, dim):
    """random rows with random structure: i.i.d., clustered, low-rank, integer grid (ties), one scale or many"""
    kind = rng.integers(0, 5)
    if kind == 0:
        x = rng.standard_normal((n, dim))
    elif kind == 1:
        c = rng.standard_normal((int(rng.integers(2, 40)), dim)) * rng.choice([0.5, 3.0, 30.0])
        x = c[rng.integers(0, c.shape[0], n)] + rng.standard_normal((n, dim)) * rng.choice([1e-3, 0.1, 1.0])
    elif kind == 2:
        r = int(rng.integers(1, max(2, dim // 4)))
        x = rng.standard_normal((n, r)) @ rng.standard_normal((r, dim)) + 0.01 * rng.standard_normal((n, dim))
    elif kind == 3:
        x = rng.integers(-3, 4, (n, dim)).astype(np.float64)
    else:
        x = rng.standard_normal((n, dim)) * np.exp(rng.uniform(-3, 3, (1, dim)))
    x = x * float(rng.choice([1e-6, 1e-2, 1.0, 1.0, 1.0, 100.0, 1e6]))
    return np.ascontiguousarray(x, np.float32)


def report(tag, cfg, what):
    global fails
    fails += 1
    print(f"MISMATCH {tag} {cfg}: {what}", flush=True)


t_end = time.time() + budget
while time.time() < t_end:
    which = rng.choice(list(runs))
    runs[which] += 1
    try:
        if which == "km_assign":
            dim = int(rng.choice([32, 36, 64, 96, 100, 128, 256, 384, 768, 1024]))
            n = int(rng.choice([4096, 4100, 5000, 9000, 20000]))
            k = int(rng.choice([2, 3, 16, 37, 122, 128, 129, 300]))
            metric = int(rng.choice([0, 1, 2]))
            x = rows(n, dim)
            mode = rng.integers(0, 4)
            if mode == 0:
                c = x[rng.choice(n, k, replace=False)].copy()
            elif mode == 1:
                c = rows(k, dim)
            elif mode == 2:   # means of random subsets (what Lloyd leaves), some duplicated
                c = np.stack([x[rng.choice(n, 50)].mean(0) for _ in range(k)]).astype(np.float32)
                c[rng.integers(0, k)] = c[rng.integers(0, k)]
            else:             # pairs a few ulps apart
                c = x[rng.choice(n, k, replace=False)].copy()
                for _ in range(max(1, k // 4)):
                    a, b = rng.integers(0, k, 2)
                    c[b] = c[a]
                    j = rng.integers(0, dim, 3)
                    c[b, j] = (c[a, j].view(np.int32) + np.int32(rng.integers(1, 9))).astype(np.int32).view(np.float32)
            bf16 = bool(rng.integers(0, 2))
            cfg = dict(n=n, dim=dim, k=k, metric=metric, mode=int(mode), bf16=bf16)
            set_hook("VG_KM_BF16", bf16)
            try:
                got = np.asarray(vg.kmeans_assign(ctx, x, c, dim, metric))
            finally:
                set_hook("VG_KM_BF16", 0)
            want = o.assign_partition_batch(x, c, metric)
            bad = np.nonzero(got != want)[0]
            if bad.size:
                report(which, cfg, f"{bad.size} rows, first {bad[:4]} got {got[bad[:4]]} want {want[bad[:4]]}")
        elif which == "km_train":
            dim = int(rng.choice([32, 64, 100, 128, 256]))
            n = int(rng.choice([4096, 6000, 12000]))
            k = int(rng.choice([2, 9, 40, 130]))
            metric = int(rng.choice([0, 1, 2]))
            iters = int(rng.choice([1, 2, 3, 5]))
            x = rows(n, dim)
            s = int(rng.integers(1, 1000))
            cfg = dict(n=n, dim=dim, k=k, metric=metric, iters=iters, seed=s)
            got = vg.kmeans_train(ctx, x, dim, k, metric, iters, seed=s)
            want = o.kmeans_train(x, dim, k, metric, iters, seed=s)
            if not np.array_equal(bits(np.asarray(got).reshape(-1)), bits(want)):
                report(which, cfg, "centroid bits differ")
        elif which == "pq_encode":
            m = int(rng.choice([1, 2, 8, 12, 32, 96]))
            dim = 8 * m
            n = int(rng.choice([1, 31, 32, 33, 1000, 4096, 5000, 20000]))
            x = rows(n, dim)
            opq = o.ProductQuantizer(dim, m, 256)
            if rng.random() < 0.5 and n >= 300:
                opq.train(x[:3000], iters=int(rng.integers(1, 4)), seed=int(rng.integers(1, 99)))
            else:   # arbitrary codebooks: duplicates and near-duplicates of centroids included
                cb = rng.integers(-128, 128, m * 256 * 8).astype(np.int8)
                if rng.random() < 0.5:
                    cb.reshape(m, 256, 8)[:, 7] = cb.reshape(m, 256, 8)[:, 200]
                amp = float(np.abs(x).max()) + 1e-30
                opq.set_codebooks(cb, (rng.random(m) * amp / 100 + amp / 1e4).astype(np.float32),
                                  ((rng.random(m) * 2 - 1) * amp / 10).astype(np.float32))
            pq = vg.ProductQuantizer(ctx, dim, m, 256)
            pq.set_codebooks(opq.codebooks, opq.scales, opq.offsets)
            cfg = dict(n=n, m=m)
            got = np.asarray(pq.encode(x))
            want = opq.encode_batch(x)
            if not np.array_equal(got, want):
                bad = np.argwhere(got != want)
                report(which, cfg, f"{bad.shape[0]} codes, first {bad[:3].tolist()}")
            pq.close()
        elif which == "pq_train":
            m = int(rng.choice([1, 4, 12]))
            dim = 8 * m
            n = int(rng.choice([300, 1000, 4096, 6000]))
            iters = int(rng.choice([1, 3, 6]))
            s = int(rng.integers(1, 1000))
            x = rows(n, dim)
            cfg = dict(n=n, m=m, iters=iters, seed=s)
            opq = o.ProductQuantizer(dim, m, 256)
            opq.train(x, iters=iters, seed=s)
            pq = vg.ProductQuantizer(ctx, dim, m, 256)
            pq.train(x, iters=iters, seed=s)
            cb, sc, of = pq.codebooks()
            if not (np.array_equal(np.asarray(cb).reshape(-1), opq.codebooks.reshape(-1)) and np.array_equal(bits(sc), bits(opq.scales))):
                report(which, cfg, "codebooks differ")
            pq.close()
        else:
            dim = int(rng.choice([8, 16, 33, 64, 128]))
            n = int(rng.choice([200, 1000, 2500]))
            metric = int(rng.choice([0, 1, 2]))
            base = rows(n, dim)
            if metric:
                base = base / np.maximum(np.linalg.norm(base, axis=1, keepdims=True), 1e-30)
                base = np.ascontiguousarray(base, np.float32)
            mdeg = int(rng.choice([4, 8, 16]))
            l0, upper, entry = graphs.build_hnsw(base, m=mdeg, seed=int(rng.integers(0, 99)))
            oidx = o.HnswIndex(base, dim, l0, upper, entry, metric=metric)
            idx = vg.Index(ctx, n, dim, vg.Metric(metric))
            idx.set_vectors(base)
            idx.set_hnsw_graph(l0, upper, entry, m=mdeg)
            nq = int(rng.choice([1, 3, 9]))
            q = rows(nq, dim)
            k = int(rng.choice([1, 5, 10, 64, 100]))
            ef = int(rng.choice([1, 10, 50, 128, 300, 600]))
            sel = float(rng.choice([0.31, 0.5, 0.8, 1.0]))
            per_query = bool(rng.integers(0, 2))
            mask = rng.random((nq, n) if per_query else n) < sel
            cfg = dict(n=n, dim=dim, metric=metric, k=k, ef=ef, sel=sel, per_query=per_query)
            ids, sc, st = idx.search_hnsw_filtered(q, k, ef, mask, sel, stats=True)
            for qi in range(nq):
                eid, esc, est = oidx.search_filtered(q[qi], k, ef, mask[qi] if per_query else mask, sel)
                r = eid.size
                if not (np.array_equal(ids[qi, :r], eid) and np.array_equal(bits(sc[qi, :r]), bits(esc)) and
                        tuple(int(v) for v in st[qi]) == (est.nodes_visited, est.distance_computations,
                                                          est.distance_short_circuits, est.pops)):
                    report(which, cfg, f"query {qi}")
                    break
            idx.close()
    except vg.VecgoHipError as e:
        report(which, "-", f"raised {e}")
print(f"fuzz_build_side: {sum(runs.values())} configurations {runs}, {fails} mismatches, seed {seed}")
sys.exit(1 if fails else 0)
