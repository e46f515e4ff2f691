"""Randomised differential test of the NaN-score behaviour (include/vecgo_hip.h "NaN scores"): random shapes, with NaN / +-Inf / huge
values thrown into queries, rows, quantizer parameters and stored norms, every exhaustive search entry point and the beam search
against the oracle (ids equal, score bits equal with NaN == NaN, the beam search's counters equal).  Prints every mismatch with the
configuration that produced it; exit code 1 if any.
    python tools/fuzz_nonfinite.py [seconds] [seed]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import vecgo_amd as vg
from oracle import oracle as o
from tests import graphs, hooks

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ctx = vg.Context(0)
fails = runs = 0
by_tag = {}
BAD = np.array([np.nan, np.inf, -np.inf, 3e38, -3e38, 1e25, 0.0], np.float32)


def same(a, b):
    a = np.asarray(a, np.float32); b = np.asarray(b, np.float32)
    return a.shape == b.shape and bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))))


LAST = None   # the inputs of the comparison under way: saved next to a mismatch (gpurun_out/fuzz_nonfinite_<n>.npz)


def compare(tag, cfg, ids, sc, exp, stats=None):
    global fails
    by_tag[tag] = by_tag.get(tag, 0) + 1
    for i, e in enumerate(exp):
        eid, esc = e[0], e[1]
        r = eid.size
        ok = np.array_equal(ids[i, :r], eid) and same(sc[i, :r], esc) and np.all(ids[i, r:] == 0xFFFFFFFF)
        if ok and stats is not None:
            est = e[2]
            ok = (int(stats[i][0]), int(stats[i][1]), int(stats[i][3])) == (est.nodes_visited, est.distance_computations, est.pops)
        if not ok:
            fails += 1
            print(f"MISMATCH {tag} {cfg} query {i}: got {ids[i][:12]} {sc[i][:12]} want {eid[:12]} {esc[:12]}", flush=True)
            if LAST is not None and fails <= 20:
                Path("gpurun_out").mkdir(exist_ok=True)
                np.savez(f"gpurun_out/fuzz_nonfinite_{fails}.npz", tag=tag, cfg=str(cfg), query=i, ids=ids, sc=sc, want_ids=eid, want_sc=esc,
                         **{k_: v for k_, v in LAST.items() if v is not None})
            return


def poison(a, frac):
    """a fraction of the entries of `a` (float32, any shape) replaced by values from BAD"""
    a = a.copy()
    flat = a.reshape(-1)
    cnt = max(1, int(frac * flat.size)) if frac > 0 else 0
    if cnt:
        flat[rng.integers(0, flat.size, cnt)] = BAD[rng.integers(0, BAD.size, cnt)]
    return a


hooks.set_hook("VG_PQ_NOM_ALWAYS", 1)
t_end = time.time() + budget
while time.time() < t_end:
    runs += 1
    dim = int(rng.choice([4, 8, 16, 17, 32, 48, 64, 100, 128, 256]))
    n = int(rng.choice([5, 64, 65, 257, 1000, 3000, 5000]))
    nq = int(rng.choice([1, 3, 9, 33, 140]))
    k = int(rng.choice([1, 2, 10, 33, 64, 65, 200]))
    metric = int(rng.choice([0, 1, 2]))
    x = rng.standard_normal((n, dim)).astype(np.float32)
    if n > 3:
        x[n // 2] = x[0]
    q = (x[rng.integers(0, n, nq)] + 0.1 * rng.standard_normal((nq, dim))).astype(np.float32)
    what = rng.choice(["queries", "data", "both", "first_rows"])
    if what in ("queries", "both"):
        q = poison(q, float(rng.choice([0.0, 1.0 / q.size, 0.02, 0.5])))
        if rng.random() < 0.3:
            q[rng.integers(0, nq)] = BAD[rng.integers(0, BAD.size)]
    xr = x
    if what in ("data", "both"):
        xr = poison(x, float(rng.choice([1.0 / x.size, 0.001, 0.05])))
    elif what == "first_rows":
        xr = x.copy()
        xr[: min(n, k)] = poison(xr[: min(n, k)], 0.1)
    cfg = dict(n=n, dim=dim, nq=nq, k=k, metric=metric, what=str(what), run=runs, seed=seed)
    which = int(rng.integers(0, 8))
    LAST = dict(x=x, xr=xr, q=q)
    if len(sys.argv) > 3:                                # a log of what is about to run: the last line names a crash
        with open(sys.argv[3], "a") as f:
            f.write(f"{cfg} which {which}\n")
        np.savez(sys.argv[3] + ".npz", x=x, xr=xr, q=q, cfg=str(cfg), which=which)
    try:
        if which == 0:                                   # flat fp32 (+ bf16 filter)
            idx = vg.Index(ctx, n, dim, vg.Metric(metric)); idx.set_vectors(xr)
            if rng.random() < 0.5:
                idx.enable_bf16_filter(True)
            ids, sc = idx.search_flat(q, min(k, 512))
            compare("flat", cfg, ids, sc, [o.flat_search_f32(xr, dim, q[i], min(k, 512), metric) for i in range(nq)])
        elif which == 1 and dim % 8 == 0:                # PQ table scan (+ nomination)
            m = dim // int(rng.choice([4, 8]))
            sd = dim // m
            opq = o.ProductQuantizer(dim, m, 256)
            scales = (rng.random(m) * 0.02 + 0.005).astype(np.float32)
            offsets = ((rng.random(m) * 2 - 1) * 0.1).astype(np.float32)
            if what in ("data", "both") and rng.random() < 0.5:
                scales = poison(scales, 1.0 / m) if rng.random() < 0.5 else scales
                offsets = poison(offsets, 1.0 / m)
            opq.set_codebooks(rng.integers(-128, 128, m * 256 * sd).astype(np.int8), scales, offsets)
            codes = rng.integers(0, 256, (n, m)).astype(np.uint8)
            pq = vg.ProductQuantizer(ctx, dim, m, 256); pq.set_codebooks(opq.codebooks, opq.scales, opq.offsets)
            idx = vg.Index(ctx, n, dim, vg.Metric.L2); idx.set_pq_codes(pq, codes)
            if rng.random() < 0.5:
                idx.enable_pq_nomination(True)
            ids, sc = idx.search_pq_adc(q, k)
            compare("pq_adc", cfg, ids, sc, [o.flat_search_pq(opq, codes, q[i], k) for i in range(nq)])
        elif which == 2 and metric != 1:                 # SQ8 (L2 / Dot)
            sq = vg.ScalarQuantizer(ctx, dim)
            mins, maxs = x.min(0), x.max(0) + 1e-3
            if what in ("data", "both"):
                mins = poison(mins, 1.0 / dim)
            sq.set_bounds(mins, maxs)
            ref = o.ScalarQuantizer(dim)
            for dst, src in zip((ref.mins, ref.maxs, ref.scales, ref.inv_scales), sq.params()):
                dst[:] = src
            ref.trained = True
            codes = rng.integers(0, 256, (n, dim)).astype(np.uint8)
            idx = vg.Index(ctx, n, dim, vg.Metric(metric)); idx.set_sq8_codes(sq, codes)
            if rng.random() < 0.5:
                idx.enable_sq8_nomination(True)
            kk = min(k, 512)
            ids, sc = idx.search_sq8(q, kk)
            seg = o.FlatSegment(x, dim, metric=metric, sq=ref, codes=codes)
            compare("sq8", cfg, ids, sc, [seg.search(q[i], kk) for i in range(nq)])
        elif which == 3:                                 # RaBitQ
            codes = o.rabitq_encode_batch(x, dim)
            cb = codes.shape[1]
            if what in ("data", "both", "first_rows"):
                rows = rng.integers(0, min(n, k) if what == "first_rows" else n, max(1, n // 200))
                for r_ in rows:
                    codes[r_, cb - 4:] = np.frombuffer(np.float32(BAD[rng.integers(0, BAD.size)]).tobytes(), np.uint8)
            idx = vg.Index(ctx, n, dim, vg.Metric.L2); idx.set_rabitq_codes(codes)
            kk = min(k, 512)
            ids, sc = idx.search_rabitq(q, kk)
            compare("rabitq", cfg, ids, sc, [o.flat_search_rabitq(codes, dim, q[i], kk) for i in range(nq)])
        elif which == 4:                                 # hnsw brute, both loops
            mode = int(rng.integers(0, 2))
            idx = vg.Index(ctx, n, dim, vg.Metric(metric)); idx.set_vectors(xr)
            oidx = o.HnswIndex(xr, dim, np.full((n, 2), 0xFFFFFFFF, np.uint32), metric=metric)
            mk = rng.choice(["none", "one", "each"])
            mask = None if mk == "none" else (rng.random(n) < 0.5 if mk == "one" else rng.random((nq, n)) < 0.4)
            ids, sc = idx.search_hnsw_brute(q, k, mode, mask)
            compare("brute", dict(cfg, mode=mode, mask=str(mk)), ids, sc,
                    [oidx.brute_search(q[i], k, mode, None if mask is None else (mask if mask.ndim == 1 else mask[i])) for i in range(nq)])
        elif which == 5 and n >= 64:                     # Vamana beam, fp32 rows
            mt = 0 if metric == 1 else metric
            g, entry = graphs.build_vamana(x, r=int(rng.choice([8, 16, 32])), seed=runs)
            idx = vg.Index(ctx, n, dim, vg.Metric(mt)); idx.set_vectors(xr); idx.set_vamana_graph(g, entry)
            ov = o.VamanaIndex(g, entry, dim, o.VAMANA_F32, metric=mt, base=xr)
            kk = min(k, 512)
            use_mask = rng.random() < 0.5
            mask = rng.random((nq, n)) < 0.4
            if use_mask:
                ids, sc, st = idx.search_vamana_filtered(q, kk, mask, kind=0, stats=True)
            else:
                ids, sc, st = idx.search_vamana(q, kk, kind=0, stats=True)
            compare("vamana_f32", dict(cfg, metric=mt, mask=use_mask), ids, sc,
                    [ov.search(q[i], kk, mask=mask[i] if use_mask else None) for i in range(nq)], stats=st)
        elif which == 6 and n >= 64 and dim % 8 == 0:    # Vamana beam, PQ / RaBitQ node scorers
            g, entry = graphs.build_vamana(x, r=16, seed=runs)
            idx = vg.Index(ctx, n, dim); idx.set_vamana_graph(g, entry)
            kk = min(k, 512)
            if rng.random() < 0.5:
                m = dim // 8
                opq = o.ProductQuantizer(dim, m, 256)
                offsets = ((rng.random(m) * 2 - 1) * 0.1).astype(np.float32)
                if what in ("data", "both"):
                    offsets = poison(offsets, 1.0 / m)
                opq.set_codebooks(rng.integers(-128, 128, m * 256 * 8).astype(np.int8), (rng.random(m) * 0.02 + 0.005).astype(np.float32), offsets)
                codes = rng.integers(0, 256, (n, m)).astype(np.uint8)
                pq = vg.ProductQuantizer(ctx, dim, m, 256); pq.set_codebooks(opq.codebooks, opq.scales, opq.offsets)
                idx.set_pq_codes(pq, codes)
                ov = o.VamanaIndex(g, entry, dim, o.VAMANA_PQ, pq=opq, codes=codes)
                kind, tag = 1, "vamana_pq"
            else:
                codes = o.rabitq_encode_batch(x, dim)
                if what in ("data", "both"):
                    cb = codes.shape[1]
                    for r_ in rng.integers(0, n, max(1, n // 100)):
                        codes[r_, cb - 4:] = np.frombuffer(np.float32(BAD[rng.integers(0, BAD.size)]).tobytes(), np.uint8)
                idx.set_rabitq_codes(codes)
                ov = o.VamanaIndex(g, entry, dim, o.VAMANA_RABITQ, codes=codes)
                kind, tag = 2, "vamana_rabitq"
            ids, sc, st = idx.search_vamana(q, kk, kind=kind, stats=True)
            compare(tag, cfg, ids, sc, [ov.search(q[i], kk) for i in range(nq)], stats=st)
        elif which == 7 and n >= 64:                     # partition-probed / filtered flat scans (fp32, SQ8)
            parts = int(rng.integers(2, 7))
            cuts = np.sort(rng.integers(0, n + 1, parts - 1))
            off = np.concatenate([[0], cuts, [n]]).astype(np.uint32)
            cent = rng.standard_normal((parts, dim)).astype(np.float32)
            nprobes = int(rng.integers(1, parts + 1))
            kk = min(k, 512)
            idx = vg.Index(ctx, n, dim, vg.Metric(metric))
            use_mask = str(rng.choice(["none", "one", "each"]))
            mask = None if use_mask == "none" else (rng.random(n) < 0.5 if use_mask == "one" else rng.random((nq, n)) < 0.4)
            if rng.random() < 0.5 or metric == 1:
                idx.set_vectors(xr)
                seg = o.FlatSegment(xr, dim, metric=metric, centroids=cent, part_offsets=off)
                code, tag = idx.SCAN_F32, "probed_f32"
            else:
                sq = vg.ScalarQuantizer(ctx, dim)
                mins, maxs = x.min(0), x.max(0) + 1e-3
                if what in ("data", "both"):
                    mins = poison(mins, 1.0 / dim)
                sq.set_bounds(mins, maxs)
                ref = o.ScalarQuantizer(dim)
                for dst, src in zip((ref.mins, ref.maxs, ref.scales, ref.inv_scales), sq.params()):
                    dst[:] = src
                ref.trained = True
                codes = rng.integers(0, 256, (n, dim)).astype(np.uint8)
                idx.set_sq8_codes(sq, codes)
                seg = o.FlatSegment(x, dim, metric=metric, sq=ref, codes=codes, centroids=cent, part_offsets=off)
                code, tag = idx.SCAN_SQ8, "probed_sq8"
            idx.set_partitions(cent, off)
            if mask is None:
                ids, sc = idx.search_flat_probed(q, kk, nprobes, scan=code)
            else:
                ids, sc = idx.search_flat_filtered(q, kk, mask, nprobes, scan=code)
            compare(tag, dict(cfg, parts=parts, nprobes=nprobes, mask=use_mask, off=off.tolist()), ids, sc,
                    [seg.search(q[i], kk, nprobes, mask=None if mask is None else (mask if mask.ndim == 1 else mask[i])) for i in range(nq)])
    except vg.VecgoHipError as e:
        if e.status != -5:                               # (an unsupported shape is not a finding)
            fails += 1
            print(f"ERROR {cfg} which {which}: {e}", flush=True)
hooks.set_hook("VG_PQ_NOM_ALWAYS", 0)
print(f"{runs} configurations, {fails} mismatches; compared by entry point: {by_tag}")
sys.exit(1 if fails else 0)
