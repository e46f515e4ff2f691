"""Randomised differential test of the search entry points against the oracle: odd dims, tiny and
ragged n, k around n, every metric.  Prints every mismatch with the configuration that produced it;
exit code 1 if any.  `python tools/fuzz_parity.py [seconds] [seed] [structured]` (structured: clustered /
integer-grid / repeated rows and queries that are rows — equal scores everywhere)"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import vecgo_amd as vg
from oracle import oracle as o
from tests import graphs, hooks

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
STRUCTURED = len(sys.argv) > 3 and sys.argv[3] == "structured"
rng = np.random.default_rng(seed)
ctx = vg.Context(0)
bits = lambda x: np.asarray(x, np.float32).view(np.uint32)
fails = 0
runs = 0


def compare(tag, cfg, ids, sc, exp):
    global fails
    for i, (eid, esc) in enumerate(exp):
        r = eid.size
        ok = np.array_equal(ids[i, :r], eid) and np.array_equal(bits(sc[i, :r]), bits(esc)) and \
            np.all(ids[i, r:] == 0xFFFFFFFF)
        if not ok:
            fails += 1
            print(f"MISMATCH {tag} {cfg} query {i}: got {ids[i]} {sc[i]} want {eid} {esc}", flush=True)
            return


t_end = time.time() + budget
while time.time() < t_end:
    runs += 1
    dim = int(rng.choice([1, 3, 4, 7, 8, 15, 16, 17, 31, 32, 48, 63, 64, 65, 96, 100, 127, 128, 130, 256, 300, 768, 1024, 1100]))
    n = int(rng.choice([1, 2, 5, 15, 16, 17, 63, 64, 65, 100, 255, 256, 257, 1000, 3000]))
    nq = int(rng.choice([1, 2, 3, 5, 9, 33, 70]))
    k = int(rng.choice([1, 2, 5, 10, 31, 32, 33, 64, 65, 200]))
    metric = int(rng.choice([0, 1, 2]))
    structured = STRUCTURED and rng.random() < 0.6         # argv[3] == "structured": rows with MANY equal scores
    if structured:
        kind = int(rng.integers(0, 3))
        if kind == 0:                                      # a few clusters, tight or loose
            c = rng.standard_normal((int(rng.integers(2, 40)), dim)) * rng.choice([0.5, 3.0, 30.0])
            x = c[rng.integers(0, c.shape[0], n)] + rng.standard_normal((n, dim)) * rng.choice([0.0, 1e-3, 0.1])
        elif kind == 1:
            x = rng.integers(-2, 3, (n, dim)).astype(np.float64)                     # integer grid
        else:
            x = np.repeat(rng.standard_normal(((n + 7) // 8, dim)), 8, axis=0)[:n]   # every row eight times
        x = np.ascontiguousarray(x * float(rng.choice([1e-3, 1.0, 1.0, 100.0])), np.float32)
    else:
        x = rng.standard_normal((n, dim)).astype(np.float32)
    if n > 3 and rng.random() < 0.5:
        x[n // 2] = x[0]                                  # duplicates: ties broken by row id
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    if structured and rng.random() < 0.5:
        q = x[rng.integers(0, n, nq)].copy()               # queries that ARE rows
    cfg = dict(n=n, dim=dim, nq=nq, k=k, metric=metric)
    idx = vg.Index(ctx, n, dim, vg.Metric(metric))
    idx.set_vectors(x)
    which = rng.integers(0, 12)
    try:
        if which == 0:
            tag = "flat"
            if rng.random() < 0.6:                     # the bfloat16 nomination filter (takes effect above 4 queries; any dim)
                nq = int(rng.choice([5, 33, 64, 65, 130, 200]))
                q = rng.standard_normal((nq, dim)).astype(np.float32)
                if rng.random() < 0.3:                 # rows bf16 cannot tell apart: the proof must fail over to the scan
                    x = (x[0] + 1e-4 * rng.standard_normal((n, dim))).astype(np.float32)
                    q = (x[0] + 1e-4 * rng.standard_normal((nq, dim))).astype(np.float32)
                    idx.set_vectors(x)
                cfg = dict(cfg, nq=nq)
                idx.enable_bf16_filter(True)
                tag = "flat_bf16_filter"
            ids, sc = idx.search_flat(q, k)
            compare(tag, cfg, ids, sc, [o.flat_search_f32(x, dim, q[i], k, metric) for i in range(nq)])
        elif which == 1:
            parts = int(rng.integers(2, 9))
            cuts = np.sort(rng.integers(0, n + 1, parts - 1))
            off = np.concatenate([[0], cuts, [n]]).astype(np.uint32)
            cent = rng.standard_normal((parts, dim)).astype(np.float32)
            if structured:                             # centroids that are rows: equal centroid distances
                cent = x[rng.integers(0, n, parts)].copy()
            nprobes = int(rng.integers(0, parts + 2))
            idx.set_partitions(cent, off)
            kk = k
            kind = int(rng.integers(0, 3))
            if kind == 1 and metric != 1:        # SQ8 codes: L2Distance / DotProduct by metric
                sq = vg.ScalarQuantizer(ctx, dim); sq.train(x)
                ref = o.ScalarQuantizer(dim); ref.train(x)
                codes = sq.encode(x)
                idx.set_sq8_codes(sq, codes)
                ids, sc = idx.search_flat_probed(q, kk, nprobes, scan=idx.SCAN_SQ8)
                seg = o.FlatSegment(x, dim, metric=metric, sq=ref, codes=codes, centroids=cent, part_offsets=off)
                tag = "probed_sq8"
            elif kind == 2 and metric == 0 and n >= 256 and [d for d in (1, 2, 4, 8, 16, 20) if dim % d == 0 and dim // d <= 152]:
                m = int(rng.choice([d for d in (1, 2, 4, 8, 16, 20) if dim % d == 0 and dim // d <= 152]))
                pq = vg.ProductQuantizer(ctx, dim, m, 256)
                pq.train(x, iters=2, seed=3)
                codes = pq.encode(x)
                cb, scales, offsets = pq.codebooks()
                opq = o.ProductQuantizer(dim, m, 256); opq.set_codebooks(cb, scales, offsets)
                idx.set_pq_codes(pq, codes)
                ids, sc = idx.search_flat_probed(q, kk, nprobes, scan=idx.SCAN_PQ)
                seg = o.FlatSegment(x, dim, pq=opq, codes=codes, centroids=cent, part_offsets=off)
                tag = "probed_pq"
            else:
                ids, sc = idx.search_flat_probed(q, kk, nprobes, scan=idx.SCAN_F32)
                seg = o.FlatSegment(x, dim, metric=metric, centroids=cent, part_offsets=off)
                tag = "probed"
            compare(tag, dict(cfg, parts=parts, nprobes=nprobes, off=off.tolist()), ids, sc,
                    [seg.search(q[i], kk, nprobes) for i in range(nq)])
        elif which == 2 and metric != 1:
            sq = vg.ScalarQuantizer(ctx, dim); sq.train(x)
            ref = o.ScalarQuantizer(dim); ref.train(x)
            codes = sq.encode(x)
            idx.set_sq8_codes(sq, codes)
            kk = k
            tag = "sq8"
            if rng.random() < 0.5:                     # the bfloat16 nomination (5 queries up; more than 128: the persistent tile)
                nq = int(rng.choice([5, 33, 64, 130, 200]))
                q = rng.standard_normal((nq, dim)).astype(np.float32)
                if structured:
                    q = x[rng.integers(0, n, nq)].copy()
                cfg = dict(cfg, nq=nq)
                idx.enable_sq8_nomination(True)
                tag = "sq8_nominated"
            ids, sc = idx.search_sq8(q, kk)
            seg = o.FlatSegment(x, dim, metric=metric, sq=ref, codes=codes)
            compare(tag, cfg, ids, sc, [seg.search(q[i], kk) for i in range(nq)])
        elif which == 3 and metric == 0:
            codes = vg.RaBitQuantizer(ctx, dim).encode(x)
            idx.set_rabitq_codes(codes)
            kk = k
            ids, sc = idx.search_rabitq(q, kk)
            compare("rabitq", cfg, ids, sc, [o.flat_search_rabitq(codes, dim, q[i], kk) for i in range(nq)])
        elif which == 4 and metric == 0 and n >= 256:
            ms = [d for d in (1, 2, 4, 8, 16, 20, 96) if dim % d == 0 and dim // d <= 152]  # K*subdim*4 <= 152 KiB of LDS
            if not ms:
                idx.close()
                continue
            m = int(rng.choice(ms))
            pq = vg.ProductQuantizer(ctx, dim, m, 256)
            pq.train(x, iters=2, seed=int(rng.integers(1, 100)))
            codes = pq.encode(x)
            cb, scales, offsets = pq.codebooks()
            opq = o.ProductQuantizer(dim, m, 256); opq.set_codebooks(cb, scales, offsets)
            idx.set_pq_codes(pq, codes)
            tag = "pq_adc"
            if rng.random() < 0.5:                     # the bfloat16 nomination over the decoded rows (test hook: any batch size)
                nq = int(rng.choice([5, 33, 64, 130, 200]))
                q = rng.standard_normal((nq, dim)).astype(np.float32)
                if structured:
                    q = x[rng.integers(0, n, nq)].copy()
                cfg = dict(cfg, nq=nq)
                hooks.set_hook("VG_PQ_NOM_ALWAYS", 1)
                idx.enable_pq_nomination(True)
                tag = "pq_adc_nominated"
            ids, sc = idx.search_pq_adc(q, k)
            hooks.set_hook("VG_PQ_NOM_ALWAYS", 0)
            compare(tag, dict(cfg, m=m), ids, sc, [o.flat_search_pq(opq, codes, q[i], k) for i in range(nq)])
        elif which == 5 and n >= 16:
            gm = int(rng.choice([4, 8, 16]))
            l0, upper, entry = graphs.build_hnsw(x, m=gm, seed=int(rng.integers(0, 1000)))
            ef = int(rng.choice([1, 8, 33, 100, 300, 513, 600, 1500, 5000]))   # > 512: heaps split between LDS and HBM scratch
            kk = min(k, 64)
            oidx = o.HnswIndex(x, dim, l0, upper, entry, metric=metric)
            idx.set_hnsw_graph(l0, upper, entry, m=gm)
            ids, sc, st = idx.search_hnsw(q, kk, ef, stats=True)
            exp = [oidx.search(q[i], kk, ef) for i in range(nq)]
            compare("hnsw", dict(cfg, m=gm, ef=ef), ids, sc, [(e[0], e[1]) for e in exp])
            for i in range(nq):
                est = exp[i][2]
                want = (est.nodes_visited, est.distance_computations, est.distance_short_circuits, est.pops)
                if tuple(int(v) for v in st[i]) != want:
                    fails += 1
                    print(f"STATS MISMATCH hnsw {cfg} m={gm} ef={ef} query {i}: {st[i]} want {want}", flush=True)
                    break
        elif which == 6 and n >= 16 and metric != 1:
            r = int(rng.choice([8, 16, 32]))
            g, entry = graphs.build_vamana(x, r=r, seed=int(rng.integers(0, 1000)))
            kk = k
            oidx = o.VamanaIndex(g, entry, dim, kind=0, metric=metric, base=x)
            idx.set_vamana_graph(g, entry)
            ids, sc, st = idx.search_vamana(q, kk, kind=0, stats=True)
            compare("vamana", dict(cfg, r=r), ids, sc, [oidx.search(q[i], kk)[:2] for i in range(nq)])
            if metric == 0:
                # the same beam scored from INT4 codes (dim % 32 == 0: terms evaluated in place, else the table) ...
                oiq = o.Int4Quantizer(dim); oiq.train(x)
                icodes = oiq.encode_batch(x)
                iq = vg.Int4Quantizer(ctx, dim); iq.train(x)
                idx.set_int4_codes(iq, icodes)
                ids, sc = idx.search_vamana(q, kk, kind=3)
                ov = o.VamanaIndex(g, entry, dim, kind=o.VAMANA_INT4, codes=icodes, int4_table=oiq.table)
                compare("vamana_int4", dict(cfg, r=r), ids, sc, [ov.search(q[i], kk)[:2] for i in range(nq)])
                # ... and from RaBitQ codes
                rcodes = o.rabitq_encode_batch(x, dim)
                idx.set_rabitq_codes(rcodes)
                ids, sc = idx.search_vamana(q, kk, kind=2)
                ov = o.VamanaIndex(g, entry, dim, kind=o.VAMANA_RABITQ, codes=rcodes)
                compare("vamana_rabitq", dict(cfg, r=r), ids, sc, [ov.search(q[i], kk)[:2] for i in range(nq)])
                iq.close()
        elif which == 7 and 16 <= n <= 1000:
            # hnsw.Insert loop on the GPU (batched) = the letter-by-letter CPU build, ties included; then a search over it
            gm = int(rng.choice([4, 8, 16]))
            efc = int(rng.choice([8, 40, 100]))
            mb = int(rng.choice([1, 7, 64]))
            gd = int(rng.choice([1, 4, 32]))
            idx.build_hnsw(m=gm, ef_construction=efc, max_batch=mb, growth_div=gd)
            l0, upper, entry = idx.get_hnsw_graph()
            ol0, oupper, oentry = o.hnsw_build(x, dim, metric=metric, m=gm, ef=efc, max_batch=mb, growth_div=gd)
            same = entry == oentry and np.array_equal(l0, ol0) and len(upper) == len(oupper) and \
                all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(upper, oupper))
            if not same:
                fails += 1
                print(f"MISMATCH hnsw_build {cfg} m={gm} ef={efc} max_batch={mb} growth_div={gd}", flush=True)
        elif which == 8 and metric == 0 and n >= 256 and dim % 8 == 0 and dim // 8 <= 152:
            # graph walk scored from PQ codes (ComputeAsymmetricDistance order), incl. the split-heap range
            gm = int(rng.choice([4, 8, 16]))
            l0, upper, entry = graphs.build_hnsw(x, m=gm, seed=int(rng.integers(0, 1000)))
            # sub-dimension 8 (m = dim / 8, odd m included): terms computed from the codebook (packed pairs, LDS-staged
            # query constants); m = 8: the per-query table form
            m = dim // 8 if rng.random() < 0.6 else 8
            pq = vg.ProductQuantizer(ctx, dim, m, 256)
            pq.train(x, iters=2, seed=5)
            codes = pq.encode(x)
            cb, scales, offsets = pq.codebooks()
            opq = o.ProductQuantizer(dim, m, 256); opq.set_codebooks(cb, scales, offsets)
            idx.set_hnsw_graph(l0, upper, entry, m=gm)
            idx.set_pq_codes(pq, codes)
            ef = int(rng.choice([10, 64, 300, 700, 2000]))
            kk = min(k, 64)
            ids, sc = idx.search_hnsw_pq(q, kk, ef)
            oidx = o.HnswIndex(x, dim, l0, upper, entry, m=gm, pq=opq, codes=codes)
            compare("hnsw_pq", dict(cfg, m=gm, pq_m=m, ef=ef), ids, sc, [oidx.search(q[i], kk, ef)[:2] for i in range(nq)])
            # the Vamana beam over layer 0 with the same PQ node scorer
            idx.set_vamana_graph(l0, entry)
            ids, sc = idx.search_vamana(q, kk, kind=1)
            ov = o.VamanaIndex(l0, entry, dim, o.VAMANA_PQ, pq=opq, codes=codes)
            compare("vamana_pq", dict(cfg, m=gm, pq_m=m), ids, sc, [ov.search(q[i], kk)[:2] for i in range(nq)])
        elif which == 9:
            # hnsw.BruteSearch / searchBitmap replayed through the reference's PriorityQueue: tie-heavy integer grids
            # half of the time (where the two disciplines return different ids), masks shared / per query / none
            if rng.random() < 0.5:
                grid = int(rng.choice([2, 3, 5, 30]))
                x = rng.integers(0, grid, (n, dim)).astype(np.float32)
                q = rng.integers(0, grid, (nq, dim)).astype(np.float32)
                if metric == 1:
                    x[np.abs(x).sum(1) == 0, 0] = 1; q[np.abs(q).sum(1) == 0, 0] = 1
                    x /= np.linalg.norm(x, axis=1, keepdims=True); q /= np.linalg.norm(q, axis=1, keepdims=True)
                idx.set_vectors(x)
            mode = int(rng.integers(0, 2))
            mk = int(rng.integers(0, 3))
            mask = None if mk == 0 else (rng.random(n) < rng.choice([0.02, 0.5, 0.95]) if mk == 1 else rng.random((nq, n)) < 0.4)
            kk = min(k, 1024)
            oidx = o.HnswIndex(x, dim, np.full((n, 1), 0xFFFFFFFF, np.uint32), metric=metric)
            ids, sc = idx.search_hnsw_brute(q, kk, mode, mask)
            exp = [oidx.brute_search(q[i], kk, mode, None if mask is None else (mask if mask.ndim == 1 else mask[i])) for i in range(nq)]
            for i, (eid, esc) in enumerate(exp):
                r = eid.size
                if not (np.array_equal(ids[i, :r], eid) and np.array_equal(bits(sc[i, :r]), bits(esc)) and np.all(ids[i, r:] == 0xFFFFFFFF)):
                    fails += 1
                    print(f"MISMATCH hnsw_brute {cfg} mode={mode} mask={mk} query {i}: got {ids[i]} want {eid}", flush=True)
                    break
        elif which == 10:
            # searcher.PriorityQueue scripts on the device heap (vg_debug_heap_replay) vs the oracle's heap: flags, popped
            # items and the final heap array, float and unsigned-key sifts
            from tests import heap_kats
            for is_max, script in heap_kats.random_scripts(int(rng.integers(0, 1 << 30)), n_scripts=6):
                ops = o.heap_script_array(script)
                want = o.prioq_replay(is_max, script)
                for uk in (False, True):
                    out, nodes, dists = vg.heap_replay(ctx, is_max, ops, unsigned_keys=uk, cap=4096)
                    if not heap_kats.same((out, (nodes, dists)), want):
                        fails += 1
                        print(f"MISMATCH heap script is_max={is_max} uk={uk} ops={len(script)}", flush=True)
        elif which == 11:
            # the L0 batch kernels (internal/simd: SquaredL2Batch / DotBatch / SquaredL2Bounded incl. the partial sum of
            # its early exit / PqAdcLookup / Hamming) against the oracle's lane-order restatement, every dim
            got = vg.squared_l2_batch(ctx, q[0], x, dim)
            if not np.array_equal(bits(got), bits(o.l2_batch(q[0], x.reshape(-1), dim))):
                fails += 1; print(f"MISMATCH l2_batch {cfg}", flush=True)
            got = vg.dot_batch(ctx, q[0], x, dim)
            if not np.array_equal(bits(got), bits(o.dot_batch(q[0], x.reshape(-1), dim))):
                fails += 1; print(f"MISMATCH dot_batch {cfg}", flush=True)
            full = o.l2_batch(q[0], x.reshape(-1), dim)
            bounds = (full * rng.choice([0.3, 0.9, 1.0, 1.5], n)).astype(np.float32) if rng.random() < 0.7 else np.float32(np.median(full))
            d, e = vg.squared_l2_bounded_batch(ctx, q[0], x, dim, bounds)
            for i in range(n):
                want, exc = o.l2_bounded(q[0], x[i], float(bounds[i] if np.ndim(bounds) else bounds))
                if bits(d[i]) != bits(want) or bool(e[i]) != bool(exc):
                    fails += 1; print(f"MISMATCH l2_bounded {cfg} row {i}: {d[i]} {e[i]} want {want} {exc}", flush=True)
                    break
            m = int(rng.choice([1, 2, 8, 15, 16, 17, 32, 96, 100]))
            table = rng.standard_normal(m * 256).astype(np.float32)
            codes = rng.integers(0, 256, (n, m), dtype=np.uint8)
            got = vg.pq_adc_lookup_batch(ctx, table, codes, m)
            want = np.array([o.adc(table, codes[i], m) for i in range(n)], np.float32)
            if not np.array_equal(bits(got), bits(want)):
                fails += 1; print(f"MISMATCH adc_lookup_batch {cfg} m={m}", flush=True)
            nb = int(rng.choice([1, 8, 12, 16, 96, 100]))
            a = rng.integers(0, 256, nb, dtype=np.uint8); cb = rng.integers(0, 256, (n, nb), dtype=np.uint8)
            got = vg.hamming_batch(ctx, a, cb)
            if not np.array_equal(np.asarray(got), np.array([o.hamming(a, cb[i]) for i in range(n)])):
                fails += 1; print(f"MISMATCH hamming_batch {cfg} bytes={nb}", flush=True)
    except vg.VecgoHipError as e:
        fails += 1
        print(f"ERROR {cfg} which={which}: {e}", flush=True)
    idx.close()
print(f"{runs} configurations, {fails} failures")
sys.exit(1 if fails else 0)
