"""Randomised differential run of the filtered searches against the oracle: vg_search_flat_filtered (fp32 / PQ / SQ8 scans,
partitions or whole segment, few queries or a matrix-core batch, pages beyond 64 results) vg_search_vamana_filtered
(fp32 / PQ / RaBitQ / INT4 node scorers) and vg_search_hnsw_predicate (tombstones, host-held edge distances).  Row structure (ties included), filter selectivity (down to nothing passing), shapes and
k are drawn at random; every mismatch is printed with its configuration; exit code 1 if any.
`python tools/fuzz_filtered.py [seconds] [seed] [family ...]` (families: flat_f32 flat_pq flat_sq8 vamana hnsw_predicate)"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import vecgo_amd as vg
from oracle import oracle as o
from tests import graphs

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ctx = vg.Context(0)
bits = lambda x: np.asarray(x, np.float32).view(np.uint32)
fails = 0
runs = {"flat_f32": 0, "flat_pq": 0, "flat_sq8": 0, "vamana": 0, "hnsw_predicate": 0}
only = [a for a in sys.argv[3:] if a in runs]


def rows(n, dim):
    kind = rng.integers(0, 5)
    if kind == 0:
        x = rng.standard_normal((n, dim))
    elif kind == 1:
        c = rng.standard_normal((int(rng.integers(2, 40)), dim)) * rng.choice([0.5, 3.0, 30.0])
        x = c[rng.integers(0, c.shape[0], n)] + rng.standard_normal((n, dim)) * rng.choice([1e-3, 0.1, 1.0])
    elif kind == 2:
        x = rng.integers(-2, 3, (n, dim)).astype(np.float64)          # integer grid: many equal scores
    elif kind == 3:
        x = np.repeat(rng.standard_normal(((n + 7) // 8, dim)), 8, axis=0)[:n]   # every row eight times
    else:
        x = rng.standard_normal((n, dim)) * np.exp(rng.uniform(-3, 3, (1, dim)))
    return np.ascontiguousarray(x * float(rng.choice([1e-3, 1.0, 1.0, 1.0, 100.0])), np.float32)


def make_mask(nq, n):
    keep = float(rng.choice([0.0, 0.0005, 0.01, 0.1, 0.5, 0.9, 1.0]))
    per_query = bool(rng.integers(0, 2))
    m = rng.random((nq, n) if per_query else n) < keep
    return m, keep, per_query


def report(tag, cfg, what):
    global fails
    fails += 1
    print(f"MISMATCH {tag} {cfg}: {what}", flush=True)


def partition(x, dim, parts, metric):
    cent = x[rng.choice(x.shape[0], parts, replace=False)] + 0.01
    a = o.assign_partition_batch(x, np.ascontiguousarray(cent, np.float32), metric)
    order = np.argsort(a, kind="stable")
    return np.ascontiguousarray(x[order]), np.ascontiguousarray(cent, np.float32), np.searchsorted(a[order], np.arange(parts + 1)).astype(np.uint32)


t_end = time.time() + budget
while time.time() < t_end:
    which = rng.choice(only or list(runs))
    runs[which] += 1
    try:
        if which.startswith("flat"):
            metric = int(rng.choice([0, 0, 1, 2]))
            if which == "flat_pq":
                m = int(rng.choice([2, 8, 12, 96]))
                dim = 8 * m
            else:
                dim = int(rng.choice([8, 17, 32, 64, 100, 128, 768]))
            n = int(rng.choice([300, 4097, 9000, 20000])) if dim < 768 else int(rng.choice([300, 5000]))
            x = rows(n, dim)
            parts = int(rng.choice([0, 0, 3, 11]))
            cent = off = None
            if parts:
                x, cent, off = partition(x, dim, parts, metric)
            idx = vg.Index(ctx, n, dim, vg.Metric(metric))
            kw = {}
            keep_alive = []
            if which == "flat_f32":
                idx.set_vectors(x)
                scan = idx.SCAN_F32
            elif which == "flat_pq":
                opq = o.ProductQuantizer(dim, m, 256)
                opq.train(x[:2000], iters=2, seed=int(rng.integers(1, 99)))
                codes = opq.encode_batch(x)
                pq = vg.ProductQuantizer(ctx, dim, m, 256)
                pq.set_codebooks(opq.codebooks, opq.scales, opq.offsets)
                idx.set_pq_codes(pq, codes)
                keep_alive.append(pq)
                kw = dict(pq=opq, codes=codes)
                scan = idx.SCAN_PQ
            else:
                ref = o.ScalarQuantizer(dim); ref.train(x)
                sq = vg.ScalarQuantizer(ctx, dim); sq.train(x)
                codes = np.asarray(sq.encode(x))
                idx.set_sq8_codes(sq, codes)
                nominate = bool(rng.integers(0, 2))      # the bf16 nomination (dim % 64 == 0, 5 queries up) or the scans
                idx.enable_sq8_nomination(nominate)
                keep_alive.append(sq)
                kw = dict(sq=ref, codes=codes)
                scan = idx.SCAN_SQ8
            if parts:
                idx.set_partitions(cent, off)
            seg = o.FlatSegment(x, dim, metric=metric, centroids=cent, part_offsets=off, **kw)
            nq = int(rng.choice([1, 3, 5, 9, 40, 130]))
            q = rows(nq, dim)
            if rng.random() < 0.3:
                q[0] = x[rng.integers(0, n)]
            k = int(rng.choice([1, 10, 48, 49, 64, 65, 100, 160, 200, 256]))
            nprobes = int(rng.choice([0, 1, 2, max(parts, 1)]))
            mask, keep, per_query = make_mask(nq, n)
            cfg = dict(n=n, dim=dim, metric=metric, parts=parts, nq=nq, k=k, nprobes=nprobes, keep=keep, per_query=per_query,
                       nominate=(which == "flat_sq8" and nominate))
            ids, sc = idx.search_flat_filtered(q, k, mask, nprobes, scan=scan)
            for qi in sorted(set([0, nq // 2, nq - 1])):
                mi = mask[qi] if per_query else mask
                eid, esc = seg.search(q[qi], k, nprobes, mask=mi)
                r = eid.size
                if not (np.array_equal(ids[qi, :r], eid) and np.array_equal(bits(sc[qi, :r]), bits(esc)) and np.all(ids[qi, r:] == 0xFFFFFFFF)):
                    report(which, cfg, f"query {qi}: got {ids[qi, :4]} want {eid[:4]}")
                    break
            idx.close()
        elif which == "hnsw_predicate":
            dim = int(rng.choice([8, 16, 33, 64, 128]))
            n = int(rng.choice([200, 1000, 2500]))
            metric = int(rng.choice([0, 1, 2]))
            base = rows(n, dim)
            if metric:
                base = np.ascontiguousarray(base / np.maximum(np.linalg.norm(base, axis=1, keepdims=True), 1e-30), np.float32)
            mdeg = int(rng.choice([4, 8, 16]))
            l0, upper, entry = graphs.build_hnsw(base, m=mdeg, seed=int(rng.integers(0, 99)))
            oidx = o.HnswIndex(base, dim, l0, upper, entry, metric=metric)
            idx = vg.Index(ctx, n, dim, vg.Metric(metric))
            idx.set_vectors(base)
            idx.set_hnsw_graph(l0, upper, entry, m=mdeg)
            l0_dist = None
            if rng.random() < 0.4:      # host-held edge distances that are not the recomputed ones (zeros included)
                l0_dist = (rng.random(l0.shape) * float(rng.choice([0.1, 10.0]))).astype(np.float32)
                l0_dist[rng.random(l0.shape) < 0.3] = 0.0
                idx.set_hnsw_edge_distances(l0_dist)
            nq = int(rng.choice([1, 3, 9]))
            q = rows(nq, dim)
            k = int(rng.choice([1, 5, 10, 64, 100]))
            ef = int(rng.choice([1, 10, 50, 128, 300, 1000]))
            mask, keep, per_query = make_mask(nq, n)
            dead = (rng.random(n) < float(rng.choice([0.05, 0.5]))) if rng.random() < 0.4 else None
            cfg = dict(n=n, dim=dim, metric=metric, m=mdeg, k=k, ef=ef, keep=keep, per_query=per_query, dead=dead is not None,
                       own_edges=l0_dist is not None)
            ids, sc, st = idx.search_hnsw_predicate(q, k, ef, mask, deleted=dead, stats=True)
            for qi in range(nq):
                eid, esc, est = oidx.search_predicate(q[qi], k, ef, mask[qi] if per_query else mask, deleted=dead, l0_dist=l0_dist)
                r = eid.size
                if not (np.array_equal(ids[qi, :r], eid) and np.array_equal(bits(sc[qi, :r]), bits(esc)) and
                        tuple(int(v) for v in st[qi]) == (est.nodes_visited, est.distance_computations,
                                                          est.distance_short_circuits, est.pops)):
                    report(which, cfg, f"query {qi}")
                    break
            idx.close()
        else:
            kind = int(rng.choice([0, 1, 2, 3]))
            dim = int(rng.choice([16, 32, 64, 96])) if kind != 3 else int(rng.choice([32, 64, 96]))
            n = int(rng.choice([90, 600, 2000]))
            metric = int(rng.choice([0, 0, 2])) if kind == 0 else 0
            base = rows(n, dim)
            r = int(rng.choice([8, 16, 32]))
            g, entry = graphs.build_vamana(base, r=r, seed=int(rng.integers(0, 99)))
            idx = vg.Index(ctx, n, dim, vg.Metric(metric))
            idx.set_vamana_graph(g, entry)
            keep_alive = []
            if kind == 0:
                ov = o.VamanaIndex(g, entry, dim, o.VAMANA_F32, metric=metric, base=base)
                idx.set_vectors(base)
            elif kind == 1:
                m = dim // 8
                opq = o.ProductQuantizer(dim, m, 256)
                opq.train(base, iters=2, seed=3)
                codes = opq.encode_batch(base)
                ov = o.VamanaIndex(g, entry, dim, o.VAMANA_PQ, pq=opq, codes=codes)
                pq = vg.ProductQuantizer(ctx, dim, m, 256)
                pq.set_codebooks(opq.codebooks, opq.scales, opq.offsets)
                idx.set_pq_codes(pq, codes)
                keep_alive.append(pq)
            elif kind == 2:
                codes = o.rabitq_encode_batch(base, dim)
                ov = o.VamanaIndex(g, entry, dim, o.VAMANA_RABITQ, codes=codes)
                idx.set_rabitq_codes(codes)
            else:
                oi = o.Int4Quantizer(dim); oi.train(base)
                iq = vg.Int4Quantizer(ctx, dim); iq.train(base)
                codes = oi.encode_batch(base)
                ov = o.VamanaIndex(g, entry, dim, kind=o.VAMANA_INT4, codes=codes, int4_table=oi.table)
                idx.set_int4_codes(iq, codes)
                keep_alive.append(iq)
            nq = int(rng.choice([1, 4, 9]))
            q = rows(nq, dim)
            k = int(rng.choice([1, 10, 64, 100]))
            mask, keep, per_query = make_mask(nq, n)
            cfg = dict(kind=kind, n=n, dim=dim, r=r, metric=metric, nq=nq, k=k, keep=keep, per_query=per_query)
            ids, sc, st = idx.search_vamana_filtered(q, k, mask, kind=kind, stats=True)
            for qi in range(nq):
                eid, esc, est = ov.search(q[qi], k, mask=mask[qi] if per_query else mask)
                rr = eid.size
                if not (np.array_equal(ids[qi, :rr], eid) and np.array_equal(bits(sc[qi, :rr]), bits(esc)) and
                        (int(st[qi][0]), int(st[qi][1]), int(st[qi][3])) == (est.nodes_visited, est.distance_computations, est.pops)):
                    report(which, cfg, f"query {qi}")
                    break
            idx.close()
    except vg.VecgoHipError as e:
        report(which, "-", f"raised {e}")
print(f"fuzz_filtered: {sum(runs.values())} configurations {runs}, {fails} mismatches, seed {seed}")
sys.exit(1 if fails else 0)
