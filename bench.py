#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on MI355X: queries/s at recall@10 >= 0.95 on 1M x 768 fp32 (i.i.d. normal).

Which pipeline answers at that recall is MEASURED here, not assumed (N = 1):
  * a real HNSW graph is built on the corpus on the GPU (vg_hnsw_build: hnsw.Insert with the reference's
    defaults M = 32 / M0 = 64 / EF = 300, levels of ApplyInsert) and PQ (m = 96, 256 centroids) is trained and
    the corpus encoded on the GPU;
  * `hnsw_pq.frontier_f32`: hnsw.KNNSearch (fp32 node scoring) swept over ef;
  * `hnsw_pq.frontier_pq_rerank`: the graph walked on PQ codes (ComputeAsymmetricDistance), the ef results
    re-scored exactly (vg_rerank = engine/search.go:914-965), swept over ef;
  * configs[1]: exact brute force (fp32 MFMA GEMM nomination + exact re-score + proof).
The operating point = the fastest of these with recall@10 >= 0.95 against an fp64 ground truth; exactly K steps
of it (1024 queries per step) are timed as `value`.  On this corpus (no low-dimensional structure) the graph
paths cross the exact path's cost long before they reach the recall bar — the whole frontier is in the line
so that the claim is witnessed, next to the same searches on the host's cores (`*.cpu`: C threads, one query
per thread, the reference's compiled AVX-512 kernels, same graph — identical answers).

N > 1: the corpus is sharded by rows over the ranks (strong scaling) on the exact path, ONE all-gather of
per-shard top-k (RCCL over xGMI), merge with the reference tie-break (vecgo_amd/sharded.py); `rabitq_sharded`
and `pq_train_sharded` are BASELINE configs[4].  A single graph does not shard (replicas only).

One JSON line on rank 0; extra objects: `roofline` (dominant kernel of the timed region, HIP events from inside
the library), `adc_scan` (configs[3]: 10M x 96 B codes, HBM roofline), `rabitq_scan` (configs[4] on one GPU),
`sq8_scan`, `int4_scan`, `flat_small_batch`, `hnsw_layer0` (configs[2]), `flat_ivf_probe`, `cpu_baseline`.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

N_ROWS, DIM, K, Q_BATCH = 1_000_000, 768, 10, 1024
SCAN_ROWS = 10_000_000         # configs[3] / configs[4]: rows of the PQ / RaBitQ code matrices
# smoke-test overrides (tests/test_gpu_bench_2rank.py): a line produced with them carries "reduced_sizes": true and is
# not a measurement of BASELINE's configs
N_ROWS = int(os.environ.get("VECGO_BENCH_ROWS", N_ROWS))
SCAN_ROWS = int(os.environ.get("VECGO_BENCH_SCAN_ROWS", SCAN_ROWS))
REDUCED = (N_ROWS, SCAN_ROWS) != (1_000_000, 10_000_000)
SEED_BASE, SEED_QUERY = 20260130, 20260131
BLOCK = 65536  # rows per generation block: data is identical for every world size
PEAK_MFMA_F32_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32-input MFMA dense peak
PEAK_MFMA_BF16_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16 MFMA dense peak (16x the fp32 form)
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec peak
RQ_MQ_INSTR_PER_ROW = 81.7     # vector instructions per 64 (row, query) pairs in rabitq_scan_mq_kernel<6>: SQ_INSTS_VALU of a
                               # 10M x 1024 call / (10M * 1024 / 64), profiles/r03_traffic.json rabitq_scan_mq


def gen_rows(lo: int, hi: int, device) -> torch.Tensor:
    """Rows [lo, hi) of the synthetic corpus: i.i.d. standard normal (testutil.go:128-145
    GaussianVectors analogue), generated per 65536-row block with its own seed."""
    out = torch.empty((hi - lo, DIM), dtype=torch.float32, device=device)
    b = lo // BLOCK
    while b * BLOCK < hi:
        g = torch.Generator(device=device)
        g.manual_seed(SEED_BASE * 4096 + b)  # disjoint from SEED_QUERY's stream
        blk = torch.randn((BLOCK, DIM), generator=g, device=device, dtype=torch.float32)
        s, e = max(lo, b * BLOCK), min(hi, (b + 1) * BLOCK)
        out[s - lo:e - lo] = blk[s - b * BLOCK:e - b * BLOCK]
        b += 1
    return out


STRUCT_LATENT, STRUCT_NOISE = 16, 0.1


def gen_structured(lo: int, hi: int, device, latent: int = STRUCT_LATENT, noise: float = STRUCT_NOISE, seed: int = 0) -> torch.Tensor:
    """Rows [lo, hi) of the STRUCTURED extra corpus: x = z A + noise * e with z ~ N(0, I_latent), A a fixed latent x 768
    matrix of N(0, 1/latent) entries (unit variance per dimension), e ~ N(0, I_768): data on a `latent`-dimensional
    subspace plus isotropic noise — neighbourhoods a graph and an 8-dim sub-quantizer can exploit, unlike the
    i.i.d. normal corpus of the headline.  Per 65536-row block with its own seed (identical for every world size);
    seed 0 = corpus, 1 = queries."""
    g = torch.Generator(device=device)
    g.manual_seed(SEED_BASE * 16384 + 5)
    a = torch.randn((latent, DIM), generator=g, device=device, dtype=torch.float32) / latent ** 0.5
    out = torch.empty((hi - lo, DIM), dtype=torch.float32, device=device)
    b = lo // BLOCK
    while b * BLOCK < hi:
        g.manual_seed(SEED_BASE * 32768 + 1000003 * seed + b)
        z = torch.randn((BLOCK, latent), generator=g, device=device, dtype=torch.float32)
        blk = z @ a + noise * torch.randn((BLOCK, DIM), generator=g, device=device, dtype=torch.float32)
        s, e = max(lo, b * BLOCK), min(hi, (b + 1) * BLOCK)
        out[s - lo:e - lo] = blk[s - b * BLOCK:e - b * BLOCK]
        b += 1
    return out


def gen_queries(n_batches: int, device) -> torch.Tensor:
    g = torch.Generator(device=device)
    g.manual_seed(SEED_QUERY)
    return torch.randn((n_batches, Q_BATCH, DIM), generator=g, device=device, dtype=torch.float32)


def fp64_topk_local(rows: torch.Tensor, q: torch.Tensor, lo: int, k: int):
    """Independent ground truth (checker only, outside the timed region): fp64 distances."""
    qd = q.double()
    best_s = torch.full((q.shape[0], k), float("inf"), dtype=torch.float64, device=q.device)
    best_i = torch.full((q.shape[0], k), -1, dtype=torch.int64, device=q.device)
    qn = (qd * qd).sum(1, keepdim=True)
    for s in range(0, rows.shape[0], 131072):
        blk = rows[s:s + 131072].double()
        d = qn + (blk * blk).sum(1)[None, :] - 2.0 * (qd @ blk.T)
        cs = torch.cat([best_s, d], 1)
        ci = torch.cat([best_i, torch.arange(s, s + blk.shape[0], device=q.device)[None, :].expand(q.shape[0], -1) + lo], 1)
        top = torch.topk(cs, k, dim=1, largest=False)
        best_s, best_i = top.values, torch.gather(ci, 1, top.indices)
    return best_i, best_s


def effective_cpus() -> int:
    """CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota (a container can
    see 256 logical CPUs and be scheduled on a fraction of them — threads beyond the quota only add
    time-slicing)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, int(quota / period + 0.5)))
        except (OSError, ValueError):
            pass
    return max(1, n)


def host_info():
    model = "?"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"cpu": model, "logical_cpus": os.cpu_count() or 1, "usable_cpus": effective_cpus()}


def cpu_leg(kind, queries_host, k, budget_s, **kw):
    """One CPU twin (oracle/vg_cpu_bench.c): C threads, one query per thread (the reference's concurrency model:
    one goroutine per query), nothing but C in the timed region; the reference's own compiled AVX-512 kernels
    when oracle/_ref is present ("reference"), the scalar restatement otherwise ("port")."""
    from oracle import oracle as o
    threads = min(effective_cpus(), queries_host.shape[0])
    is_ref = o.use_reference_kernels(True)
    try:
        r = o.bench_run(kind, queries_host, k, threads, budget_s, **kw)
    finally:
        o.use_reference_kernels(False)
    r.update(cores=threads, kind="reference" if is_ref else "port")
    return r


def cpu_baseline(rows_host: np.ndarray, queries_host: np.ndarray, k: int, budget_s: float = 12.0):
    """configs[1] on the host: flat.Segment.Search's fp32 scan (flat/segment.go:691-721: squaredL2Avx512 per row
    + CandidateHeap), corpus pages interleaved over the NUMA nodes."""
    from oracle import oracle as o
    ic = o.InterleavedCopy(rows_host)
    try:
        r = cpu_leg(o.BENCH_FLAT, queries_host, k, budget_s, base=ic.array)
        numa = ic.numa_nodes
    finally:
        ic.close()
    n = rows_host.shape[0]
    gbs = r["queries"] * n * DIM * 4 / r["seconds"] / 1e9
    return {"value": r["qps"], "unit": "queries/s", "cores": r["cores"], "kind": r["kind"],
            "scan_gbs": gbs, "numa_interleave_nodes": numa, **host_info(),
            "sample": f"{r['queries']} queries x {n} rows x {DIM} fp32, exact L2 top-{k}, {r['cores']} C threads "
                      f"(1 query/thread), {r['seconds']:.1f} s; squaredL2Avx512 per row + CandidateHeap "
                      f"(flat/segment.go:691-721), {gbs:.0f} GB/s of rows"}


def adc_scan_roofline(vg, ctx, stream, device, with_cpu=False):
    """BASELINE configs[3]: PQ (m=96, K=256) ADC scan over 10M codes, one query per pass:
    algorithmic bytes = N*m per launch."""
    n, m = SCAN_ROWS, 96
    g = torch.Generator(device=device)
    g.manual_seed(7)
    codes = torch.randint(0, 256, (n, m), dtype=torch.uint8, device=device, generator=g)
    rng = np.random.default_rng(0)
    pq = vg.ProductQuantizer(ctx, DIM, m, 256)
    pq.set_codebooks(rng.integers(-128, 128, m * 256 * (DIM // m)).astype(np.int8),
                     (rng.random(m) * 0.02 + 0.005).astype(np.float32), np.zeros(m, np.float32))
    idx = vg.Index(ctx, n, DIM)
    idx.set_pq_codes(pq, codes)
    q = torch.randn((1, DIM), device=device)
    out = (torch.empty((1, K), dtype=torch.int32, device=device), torch.empty((1, K), device=device))
    for _ in range(200):   # ~50 ms: an idle GPU needs more than a handful of launches to reach its clocks
        idx.search_pq_adc(q, K, out=out, stream=stream)
    torch.cuda.synchronize()
    ctx.profile_read("pq_adc_scan")
    ctx.profile_enable(True)
    reps = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        idx.search_pq_adc(q, K, out=out, stream=stream)
    e1.record(stream)
    torch.cuda.synchronize()
    launches, ms = ctx.profile_read("pq_adc_scan")
    ctx.profile_enable(False)
    kern_ms = ms / max(launches, 1)
    achieved = n * m / (kern_ms * 1e-3) / 1e9
    res = {"workload": "pq_adc_scan_10Mx768_m96_K256_k10_nq1", "bound": "hbm", "achieved": achieved,
           "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": achieved / PEAK_HBM_GBS, "traffic": measured_traffic("pq_adc_scan"),
           "kernel": "pq_adc_scan_kernel<6,true,true>", "kernel_ms": kern_ms,
           "bytes_per_launch": n * m, "search_call_ms": e0.elapsed_time(e1) / reps,
           "qps_single_query_passes": 1e3 / (e0.elapsed_time(e1) / reps)}
    res["batch"] = scan_batch(ctx, idx.search_pq_adc, "pq_adc_scan", n, m, stream, device,
                              bound="lds", peak=PEAK_LDS_LOOKUPS_G, unit="G table lookups/s", per_row=m,
                              note="one fp32 table (96 KiB) fills a workgroup's LDS, so a workgroup scores ONE query; the batch "
                                   "shares each slice of codes through its XCD's L2 (block order) and runs at the LDS gather "
                                   "rate: peak = 15 lookups/clk/CU (tools/ubench/lds_pattern.hip, rotated image) x 256 CUs x 2.4 GHz")
    cpu = None
    if with_cpu:  # flat.Segment.Search's PQ branch on the host: BuildDistanceTable + pqAdcLookupAvx512 per row
        from oracle import oracle as o
        opq = o.ProductQuantizer(DIM, m, 256)
        opq.set_codebooks(*[np.asarray(x) for x in pq.codebooks()])
        ic = o.InterleavedCopy(codes.cpu().numpy())
        try:
            qh = np.random.default_rng(3).standard_normal((effective_cpus(), DIM)).astype(np.float32)
            r = cpu_leg(o.BENCH_ADC, qh, K, 5.0, pq=opq, codes=ic.array, n=n, want_ids=True)
            gi, _ = idx.search_pq_adc(torch.from_numpy(qh[:4]).to(device), K)
            gi = gi.cpu().numpy().view(np.uint32)
            filled = [i for i in range(4) if r["dist_comps"][i] >= 0]
            cpu = {"qps": r["qps"], "cores": r["cores"], "kind": r["kind"], "queries": r["queries"], "seconds": r["seconds"],
                   "scan_gbs": r["queries"] * n * m / r["seconds"] / 1e9, "numa_interleave_nodes": ic.numa_nodes,
                   "ids_equal_gpu": bool(filled) and all(np.array_equal(r["ids"][i], gi[i]) for i in filled)}
        finally:
            ic.close()
    del codes
    if cpu:
        res["cpu"] = cpu
        res["batch"]["cpu_qps"] = cpu["qps"]   # the reference scans one query at a time whatever the batch
        res["batch"]["gpu_over_cpu"] = res["batch"]["qps"] / cpu["qps"]
    idx.close()
    pq.close()
    return res


PEAK_LDS_LOOKUPS_G = 15 * 256 * 2.4        # G ds_read_b32 gathers/s: 15 per clock and CU measured for the rotated table image
PEAK_VALU_GINSTR = 256 * 4 * 2.4 / 4       # G wave-instructions/s: 4 SIMDs per CU, one 64-lane instruction per 4 clocks


def scan_batch(ctx, search, prof, n, row_bytes, stream, device, bound, peak, unit, per_row, note, nq=1024):
    """A batch of `nq` queries through one search call (the query-blocked scan kernels), 10M rows: queries/s, the
    kernel's rate against the roofline that bounds it, and a check that a sample of the batch equals one-query passes
    (which tests/test_gpu_fullsize.py pins to whole-query oracle replays)."""
    q = gen_queries(1, device)[0][:nq].contiguous()
    ids, sc = search(q, K, stream=stream)
    torch.cuda.synchronize()
    same = True
    for i in (0, nq // 2 + 1, nq - 1):
        i1, s1 = search(q[i:i + 1], K, stream=stream)
        same &= bool(torch.equal(i1[0], ids[i]) and torch.equal(s1[0].view(torch.int32), sc[i].view(torch.int32)))
    ctx.profile_read(prof)
    ctx.profile_enable(True)
    reps = 3
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        search(q, K, stream=stream)
    e1.record(stream)
    torch.cuda.synchronize()
    launches, ms = ctx.profile_read(prof)
    ctx.profile_enable(False)
    call_ms = e0.elapsed_time(e1) / reps
    kern_ms = ms / reps
    achieved = n * nq * per_row / (kern_ms * 1e-3) / 1e9
    return {"workload": f"{nq} queries x {n} rows, k={K}, one call", "qps": nq / (call_ms * 1e-3), "call_ms": call_ms,
            "kernel_ms": kern_ms, "launches_per_call": launches // reps, "row_scores_per_s": n * nq / (kern_ms * 1e-3),
            "code_bytes_scored_tbs": n * nq * row_bytes / (kern_ms * 1e-3) / 1e12,
            "roofline": {"bound": bound, "achieved": achieved, "peak": peak, "unit": unit, "frac": achieved / peak, "note": note},
            "sample_equals_one_query_passes": same}


def flat_bf16_filter(vg, ctx, idx, queries, gt_ids, steps, stream):
    """The exact path again with vg_index_enable_bf16_filter: the two nomination GEMMs run as bfloat16 MFMA over a bf16
    copy of the rows; the nominated rows are re-scored from the fp32 rows and the proof (widened by the rounding of the
    copies) or the exhaustive fallback decides, so ids and fp32 scores are the reference's, bit for bit — checked here
    against the unfiltered path on every query of a batch.  Reported beside the headline, which stays on the fp32 MFMA
    GEMM BASELINE's configs[1] names."""
    nb = queries.shape[0]
    ref_ids, ref_sc = idx.search_flat(queries[0], K, stream=stream)
    idx.enable_bf16_filter(True, stream=stream)
    try:
        for i in range(3):
            idx.search_flat(queries[i % nb], K, stream=stream)
        torch.cuda.synchronize()
        s0 = idx.flat_stats()
        ctx.profile_read("flat_gemm")
        ctx.profile_enable(True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for i in range(steps):
            idx.search_flat(queries[i % nb], K, stream=stream)
        e1.record(stream)
        torch.cuda.synchronize()
        launches, ms = ctx.profile_read("flat_gemm")
        ctx.profile_enable(False)
        s1 = idx.flat_stats()
        ids, sc = idx.search_flat(queries[0], K, stream=stream)
        torch.cuda.synchronize()
        small = {}
        for nqs in (32, 64):   # the HBM-bound tiles: half the bytes per row
            qs = queries[1][:nqs].contiguous()
            for _ in range(20):
                idx.search_flat(qs, K, stream=stream)
            torch.cuda.synchronize()
            s0e, s1e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s0e.record(stream)
            for _ in range(10):
                idx.search_flat(qs, K, stream=stream)
            s1e.record(stream)
            torch.cuda.synchronize()
            small[f"q{nqs}_call_ms"] = s0e.elapsed_time(s1e) / 10
    finally:
        idx.enable_bf16_filter(False, stream=stream)
    step_ms = e0.elapsed_time(e1) / steps
    kern_ms = ms / max(launches, 1)
    flops = 2.0 * Q_BATCH * N_ROWS * DIM
    got = ids.cpu().numpy().view(np.uint32).astype(np.int64)
    return {"workload": "flat_exact_l2_1Mx768_top10, nomination by bfloat16 MFMA GEMM over a bf16 copy of the rows, exact "
                        "fp32 re-score + proof (opt-in: vg_index_enable_bf16_filter)",
            "qps": Q_BATCH / (step_ms * 1e-3), "ms_per_step": step_ms, "steps": steps,
            "recall_at_10": recall_at_k(got[:gt_ids.shape[0]], gt_ids),
            "ids_equal_fp32_path": bool(torch.equal(ids, ref_ids)),
            "scores_bit_equal_fp32_path": bool(torch.equal(sc.view(torch.int32), ref_sc.view(torch.int32))),
            "proof_fallbacks": int(s1[1] - s0[1]), "queries": int(s1[0] - s0[0]),
            "extra_hbm_bytes": N_ROWS * DIM * 2, "small_batches": small,
            "roofline": {"bound": "mfma", "kernel": "flat_gemm_bf16_big_kernel<false,3> (v_mfma_f32_32x32x16_bf16)",
                         "kernel_ms": kern_ms, "launches": launches, "flops_per_launch": flops,
                         "achieved": flops / (kern_ms * 1e-3) / 1e12, "peak": PEAK_MFMA_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": flops / (kern_ms * 1e-3) / 1e12 / PEAK_MFMA_BF16_TFLOPS,
                         "note": "persistent 256 x 256 tile, one workgroup per CU; the matrix-only loop of this tile runs at 0.78 ms, the K loop with its LDS traffic and barrier at ~1.0, the threshold epilogue adds the rest (profiles/r06_gemm_bf16_probe.txt)"}}


def flat_small_batch(vg, ctx, idx, queries, stream):
    """BASELINE configs[1] below the MFMA regime (HBM-bound for Q < ~40): a batch
    of 1 (exact scan) or 32 (32-query GEMM tile) = one pass over the 1M x 768 fp32 rows, N*d*4
    algorithmic bytes."""
    res = {"bound": "hbm", "peak": PEAK_HBM_GBS, "unit": "GB/s", "bytes_per_pass": N_ROWS * DIM * 4}
    for nq in (1, 32):
        label, kname = ("flat_scan", "flat_scan_mq_kernel<false>") if nq <= 4 else \
            ("flat_gemm", "flat_gemm_dma32_kernel<false,2>")
        q = queries[:nq].contiguous()
        out = (torch.empty((nq, K), dtype=torch.int32, device=q.device), torch.empty((nq, K), device=q.device))
        for _ in range(60):
            idx.search_flat(q, K, out=out, stream=stream)
        torch.cuda.synchronize()
        ctx.profile_read(label)
        ctx.profile_enable(True)
        reps = 10
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            idx.search_flat(q, K, out=out, stream=stream)
        e1.record(stream)
        torch.cuda.synchronize()
        launches, ms = ctx.profile_read(label)
        ctx.profile_enable(False)
        kern_ms = ms / max(launches, 1)
        res[f"q{nq}"] = {"call_ms": e0.elapsed_time(e1) / reps, "kernel": kname,
                         "kernel_ms": kern_ms, "achieved": N_ROWS * DIM * 4 / (kern_ms * 1e-3) / 1e9,
                         "frac": N_ROWS * DIM * 4 / (kern_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                         "qps": nq * 1e3 / (e0.elapsed_time(e1) / reps)}
    return res


def rabitq_scan_roofline(vg, ctx, stream, device, with_cpu=False):
    """BASELINE configs[4], one GPU's view: exhaustive RaBitQ scan, 10M x 768 -> 100 B per row
    (96 B of sign bits + f32 norm): algorithmic bytes = N*100 per launch."""
    n = SCAN_ROWS
    cb = (DIM + 63) // 64 * 8 + 4
    g = torch.Generator(device=device)
    g.manual_seed(11)
    codes = torch.randint(0, 256, (n, cb), dtype=torch.uint8, device=device, generator=g)
    codes[:, cb - 4:] = (torch.rand(n, device=device, generator=g) * 5 + 25).view(torch.uint8).reshape(n, 4)
    idx = vg.Index(ctx, n, DIM)
    idx.set_rabitq_codes(codes)
    q = torch.randn((1, DIM), device=device)
    out = (torch.empty((1, K), dtype=torch.int32, device=device), torch.empty((1, K), device=device))
    for _ in range(200):
        idx.search_rabitq(q, K, out=out, stream=stream)
    torch.cuda.synchronize()
    ctx.profile_read("rabitq_scan")
    ctx.profile_enable(True)
    reps = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        idx.search_rabitq(q, K, out=out, stream=stream)
    e1.record(stream)
    torch.cuda.synchronize()
    launches, ms = ctx.profile_read("rabitq_scan")
    ctx.profile_enable(False)
    kern_ms = ms / max(launches, 1)
    achieved = n * cb / (kern_ms * 1e-3) / 1e9
    res = {"workload": "rabitq_scan_10Mx768_100B_k10_nq1", "bound": "hbm", "achieved": achieved,
           "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": achieved / PEAK_HBM_GBS,
           "traffic": measured_traffic("rabitq_scan"), "kernel": "rabitq_scan_kernel", "kernel_ms": kern_ms,
           "bytes_per_launch": n * cb, "search_call_ms": e0.elapsed_time(e1) / reps,
           "qps_single_query_passes": 1e3 / (e0.elapsed_time(e1) / reps)}
    res["batch"] = scan_batch(ctx, idx.search_rabitq, "rabitq_scan_mq", n, cb, stream, device,
                              bound="valu", peak=PEAK_VALU_GINSTR, unit="G wave-instructions/s", per_row=RQ_MQ_INSTR_PER_ROW / 64.0,
                              note="rabitq_scan_mq_kernel: 16 queries per workgroup pass, a row's sign bits loaded once and kept in "
                                   "registers; per (row, query) 2 vector instructions per 32 dimensions (xor + popcount-accumulate) "
                                   "+ the formula with its IEEE division (rabitq.go:170-175) + the top-k pre-test = "
                                   f"{RQ_MQ_INSTR_PER_ROW} instructions per 64 (row, query) pairs (ISA count); peak = 4 SIMDs x 256 CUs "
                                   "x 2.4 GHz / 4 clocks per 64-lane instruction")
    cpu = None
    if with_cpu:  # rq.Distance per row (rabitq.go:119-176: query norm + sign-pack recomputed per call, hammingAvx512)
        from oracle import oracle as o
        # rq.Distance costs ~0.1 us per row and core (it re-derives the query's norm and sign bits on every call), a
        # whole 10M-row query ~1 s per core: the timed sample is the first 1M rows, the rate is per row
        n_cpu = 1_000_000
        sub = codes[:n_cpu].contiguous()
        sidx = vg.Index(ctx, n_cpu, DIM)
        sidx.set_rabitq_codes(sub)
        ic = o.InterleavedCopy(sub.cpu().numpy())
        try:
            qh = np.random.default_rng(4).standard_normal((effective_cpus(), DIM)).astype(np.float32)
            r = cpu_leg(o.BENCH_RABITQ, qh, K, 5.0, codes=ic.array, n=n_cpu, dim=DIM, want_ids=True)
            gi, _ = sidx.search_rabitq(torch.from_numpy(qh[:4]).to(device), K)
            gi = gi.cpu().numpy().view(np.uint32)
            filled = [i for i in range(4) if r["dist_comps"][i] >= 0]
            rows_s = r["queries"] * n_cpu / r["seconds"]
            cpu = {"rows_per_s": rows_s, "qps_at_10M_rows": rows_s / n, "cores": r["cores"], "kind": r["kind"],
                   "sample": f"{r['queries']} queries x {n_cpu} rows (first 1M rows of the corpus), {r['seconds']:.1f} s",
                   "scan_gbs": rows_s * cb / 1e9, "numa_interleave_nodes": ic.numa_nodes,
                   "ids_equal_gpu_on_the_sample": bool(filled) and all(np.array_equal(r["ids"][i], gi[i]) for i in filled)}
        finally:
            ic.close()
            sidx.close()
    del codes
    if cpu:
        res["cpu"] = cpu
        res["batch"]["cpu_qps"] = cpu["qps_at_10M_rows"]   # the reference scans one query at a time whatever the batch
        res["batch"]["gpu_over_cpu"] = res["batch"]["qps"] / cpu["qps_at_10M_rows"]
    idx.close()
    return res


def sq8_scan_roofline(vg, ctx, stream, device):
    """exhaustive SQ8 scan (flat/segment.go:517-604), 4M x 768 one-byte codes
    = 3.07 GB per pass (the byte count of the fp32 1M x 768 corpus)."""
    n = 4_000_000
    g = torch.Generator(device=device)
    g.manual_seed(17)
    codes = torch.randint(0, 256, (n, DIM), dtype=torch.uint8, device=device, generator=g)
    sq = vg.ScalarQuantizer(ctx, DIM)
    sq.set_bounds(np.full(DIM, -4.0, np.float32), np.full(DIM, 4.0, np.float32))
    idx = vg.Index(ctx, n, DIM)
    idx.set_sq8_codes(sq, codes)
    del codes
    q = torch.randn((1, DIM), device=device)
    out = (torch.empty((1, K), dtype=torch.int32, device=device), torch.empty((1, K), device=device))
    for _ in range(40):
        idx.search_sq8(q, K, out=out, stream=stream)
    torch.cuda.synchronize()
    ctx.profile_read("sq8_scan")
    ctx.profile_enable(True)
    reps = 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        idx.search_sq8(q, K, out=out, stream=stream)
    e1.record(stream)
    torch.cuda.synchronize()
    launches, ms = ctx.profile_read("sq8_scan")
    ctx.profile_enable(False)
    kern_ms = ms / max(launches, 1)
    achieved = n * DIM / (kern_ms * 1e-3) / 1e9
    res = {"workload": "sq8_scan_4Mx768_k10_nq1", "bound": "hbm", "achieved": achieved, "peak": PEAK_HBM_GBS,
           "unit": "GB/s", "frac": achieved / PEAK_HBM_GBS, "traffic": measured_traffic("sq8_scan"), "kernel": "sq8_scan_kernel",
           "kernel_ms": kern_ms, "bytes_per_launch": n * DIM, "search_call_ms": e0.elapsed_time(e1) / reps}
    idx.close()
    sq.close()
    return res


def int4_scan_roofline(vg, ctx, stream, device):
    """Int4Quantizer.L2DistanceBatch / L2Distance of one query over 4M x 768 INT4 codes
    (int4.go:133-164; 384 B per code = 1.54 GB per pass), both summation orders."""
    n = 4_000_000
    g = torch.Generator(device=device)
    g.manual_seed(19)
    train = torch.randn((65536, DIM), generator=g, device=device)
    iq = vg.Int4Quantizer(ctx, DIM)
    iq.train(train)
    codes = torch.randint(0, 256, (n, DIM // 2), dtype=torch.uint8, device=device, generator=g)
    q = torch.randn((DIM,), device=device, generator=g)
    out = torch.empty((n,), device=device)
    res = {"workload": "int4_l2_distance_4Mx768_nq1", "bound": "hbm", "peak": PEAK_HBM_GBS, "unit": "GB/s",
           "bytes_per_launch": n * DIM // 2, "kernel": "int4_scan_tab_kernel",
           "note": "bound by INSTRUCTION ISSUE: 2.0 vector + 1.06 LDS instructions per (64 rows x 1 dimension), one issue slot per SIMD "
                   "per 4 clocks whatever the unit — 91-93 % of all slots carry an instruction (profiles/r04_pmc_i4_issue.csv, r04_traffic.json int4_scan.issue: "
                   "SQ_ACTIVE_INST_ANY / (SIMDs x cycles / 4)); a per-lane lookup per dimension is the floor of this formulation: DESIGN.md section 4"}
    for name, fn in (("batch_order", iq.l2_distance_batch), ("lookup_table_order", iq.l2_distance)):
        for _ in range(100):
            fn(q, codes, out=out, stream=stream)
        torch.cuda.synchronize()
        ctx.profile_read("int4_scan")
        ctx.profile_enable(True)
        for _ in range(20):
            fn(q, codes, out=out, stream=stream)
        torch.cuda.synchronize()
        launches, ms = ctx.profile_read("int4_scan")
        ctx.profile_enable(False)
        kern_ms = ms / max(launches, 1)
        achieved = n * DIM / 2 / (kern_ms * 1e-3) / 1e9
        res[name] = {"kernel_ms": kern_ms, "achieved": achieved, "frac": achieved / PEAK_HBM_GBS}
    iq.close()
    return res


def recall_at_k(got: np.ndarray, gt: np.ndarray) -> float:
    n = min(got.shape[0], gt.shape[0])
    return float(np.mean([len(set(got[i].tolist()) & set(gt[i].tolist())) / K for i in range(n)]))


HNSW_M, HNSW_EFC, PQ_M = 32, 300, 96          # hnsw.go:34-37 defaults; BASELINE configs[3] PQ shape
EFS_F32 = (128, 256, 512, 1024, 2048, 4096)
EFS_PQ = (128, 512, 2048, 8192)
NQ_FLIGHT = 8192                               # graph searches are latency chains: many queries in flight


def hnsw_pq_frontier(vg, ctx, rows, queries, gt_ids, exact_ms_per_1024, stream, with_cpu, efs_f32=None, efs_pq=None,
                     cpu_efs=(128, 512, 2048)):
    """The metric's own configuration (BASELINE.json: recall@10 >= 0.95 on 1M x 768 HNSW+PQ), measured:
    real graph (vg_hnsw_build), PQ m = 96 trained + encoded on the GPU, then the (ef, recall, cost) frontier
    of (a) hnsw.KNNSearch on fp32 rows and (b) graph walk on PQ codes -> exact rerank of the ef results.
    Returns (report, index)."""
    n = rows.shape[0]
    idx = vg.Index(ctx, n, DIM)
    idx.set_vectors(rows)
    ctx.profile_enable(True)
    for kname in ("hnsw_build_search", "hnsw_build_select", "hnsw_build_link"):
        ctx.profile_read(kname)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    idx.build_hnsw(m=HNSW_M, ef_construction=HNSW_EFC, max_batch=8192, growth_div=32, stream=stream)
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    stages = {kname: ctx.profile_read(kname)[1] for kname in ("hnsw_build_search", "hnsw_build_select", "hnsw_build_link")}
    ctx.profile_enable(False)
    t0 = time.perf_counter()
    pq = vg.ProductQuantizer(ctx, DIM, PQ_M, 256)
    pq.train(rows[:65536], iters=20, seed=1, stream=stream)
    codes = pq.encode(rows, stream=stream)
    idx.set_pq_codes(pq, codes, stream=stream)
    torch.cuda.synchronize()
    pq_s = time.perf_counter() - t0
    q = queries.reshape(-1, DIM)[:NQ_FLIGHT].contiguous()
    nrec = gt_ids.shape[0]

    def timed(fn, reps=2):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            fn()
        e1.record(stream)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    f32 = []
    for ef in (efs_f32 or EFS_F32):
        ids, _, st = idx.search_hnsw(q, K, ef, stats="full", stream=stream)
        ctx.profile_read("hnsw_search")
        ctx.profile_enable(True)
        ms = timed(lambda: idx.search_hnsw(q, K, ef, stream=stream))
        ctx.profile_enable(False)
        kern_ms = ctx.profile_read("hnsw_search")[1] / 3     # timed() makes 3 calls; the kernel launches of one call
        dc = float(st[:, 1].mean())
        f32.append({"ef": ef, "recall_at_10": recall_at_k(ids.cpu().numpy().view(np.uint32)[:nrec], gt_ids),
                    "qps": q.shape[0] / (ms * 1e-3), "ms_per_1024": ms * 1024 / q.shape[0], "kernel_ms": kern_ms,
                    "distance_computations_per_query": dc, "pops_per_query": float(st[:, 3].mean()),
                    "descent_distance_computations_per_query": float(st[:, 4].mean()),
                    # rows scored = layer-0 DistanceComputations + the descent's (uncounted by the reference's stats)
                    "gathered_gbs": ((float(st[:, 1].sum()) + float(st[:, 4].sum())) * DIM * 4
                                     + float(st[:, 3].sum()) * 2 * HNSW_M * 4) / (kern_ms * 1e-3) / 1e9})
    pqr = []
    for ef in (efs_pq or EFS_PQ):
        cand, _, st = idx.search_hnsw_pq(q, ef, ef, stats=True, stream=stream)
        ids, _ = idx.rerank(q, cand, K, stream=stream)
        ms_walk = timed(lambda: idx.search_hnsw_pq(q, ef, ef, stream=stream))
        ms_rr = timed(lambda: idx.rerank(q, cand, K, stream=stream))
        ms = ms_walk + ms_rr
        pqr.append({"ef": ef, "rerank_candidates": ef, "recall_at_10": recall_at_k(ids.cpu().numpy().view(np.uint32)[:nrec], gt_ids),
                    "qps": q.shape[0] / (ms * 1e-3), "ms_per_1024": ms * 1024 / q.shape[0],
                    "walk_ms": ms_walk, "rerank_ms": ms_rr, "pq_scores_per_query": float(st[:, 1].mean()),
                    "pq_scores_per_s": float(st[:, 1].sum()) / (ms_walk * 1e-3)})
    rep = {"workload": f"1M x 768 L2, HNSW M={HNSW_M} M0={2 * HNSW_M} EF={HNSW_EFC} built by vg_hnsw_build "
                       f"(batches <= 8192), PQ m={PQ_M} K=256 trained on 65536 rows (20 iterations), k={K}, "
                       f"{q.shape[0]} queries in flight, recall over {nrec} queries vs fp64 brute force",
           "graph_build_s": build_s, "graph_build_stage_ms": stages, "pq_train_encode_s": pq_s,
           "frontier_f32": f32, "frontier_pq_rerank": pqr,
           "exact_path": {"recall_at_10": 1.0, "ms_per_1024": exact_ms_per_1024, "qps": 1024 / (exact_ms_per_1024 * 1e-3)}}
    ok = [("hnsw_f32", e) for e in f32 if e["recall_at_10"] >= 0.95] + \
         [("hnsw_pq_rerank", e) for e in pqr if e["recall_at_10"] >= 0.95]
    best = max(ok, key=lambda t: t[1]["qps"]) if ok else None
    if best and best[1]["qps"] > rep["exact_path"]["qps"]:
        rep["operating_point"] = {"path": best[0], **best[1]}
    else:
        rep["operating_point"] = {"path": "flat_exact", **rep["exact_path"]}
    best_f32 = max(f32, key=lambda e: e["recall_at_10"])
    best_pq = max(pqr, key=lambda e: e["recall_at_10"])
    rep["conclusion"] = (
        f"highest graph recall reached inside the sweep: fp32 walk {best_f32['recall_at_10']:.3f} at ef={best_f32['ef']} "
        f"({best_f32['ms_per_1024']:.1f} ms per 1024 queries), PQ walk + rerank {best_pq['recall_at_10']:.3f} at ef={best_pq['ef']} "
        f"({best_pq['ms_per_1024']:.1f} ms); the exact path answers 1024 queries in {exact_ms_per_1024:.1f} ms at recall 1.0")
    if with_cpu:
        rep["cpu"] = hnsw_cpu_twin(idx, rows, q, f32, cpu_efs)
        try:
            # ... including the ef of the pipeline's best recall, the entry the line's "metric pipeline" row quotes
            best_ef = max(pqr, key=lambda e: e["recall_at_10"])["ef"]
            rep["cpu_pq_rerank"] = hnsw_pq_cpu_twin(idx, pq, codes, rows, q, pqr,
                                                    tuple(cpu_efs) + (() if best_ef in cpu_efs else (best_ef,)))
        except Exception as e:   # an extra leg: named in the record, not fatal
            rep["cpu_pq_rerank"] = {"error": f"{type(e).__name__}: {e}"}
    # configs[2] as a bandwidth statement: ef = 128 on the real graph
    e128 = f32[0]
    alg128 = e128["gathered_gbs"] * 1e9 * e128["kernel_ms"] * 1e-3
    tr128 = measured_traffic("hnsw_search", alg128)
    rep128 = {"workload": f"hnsw_ef128_1Mx768_k10 on the built graph (M0 = {2 * HNSW_M}), {q.shape[0]} queries in flight",
              "bound": "hbm", "achieved": e128["gathered_gbs"], "peak": PEAK_HBM_GBS, "unit": "GB/s",
              # `achieved` counts every row the walk scored, some of them served by L2 / the memory-side cache (hub rows, the
              # shared descent path): it is a GATHER rate set against the HBM peak, not an HBM fraction
              "gather_rate_over_hbm_peak": e128["gathered_gbs"] / PEAK_HBM_GBS,
              # the HBM-side fraction: fabric bytes (FETCH_SIZE x 2 + WRITE_SIZE, PMC) per second over the peak
              "frac": (tr128 / (e128["kernel_ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS) if tr128 else None,
              "traffic": tr128,
              "kernel": "hnsw_search_kernel<0, *>",
              "kernel_ms": e128["kernel_ms"], "bytes_per_launch": e128["gathered_gbs"] * 1e9 * e128["kernel_ms"] * 1e-3,
              "recall_at_10": e128["recall_at_10"],
              "distance_computations_per_query": e128["distance_computations_per_query"],
              "note": "gather rate of the candidate-batch distance kernel; not a QPS claim (recall is far below the bar). "
                      "`frac` is FABRIC bytes (FETCH_SIZE x 2 + WRITE_SIZE) over the HBM peak: FETCH_SIZE counts reads served by the "
                      "memory-side Infinity Cache as well as HBM reads, so part of it is MALL — MI355X_MICROARCH.md measures "
                      "5.7-5.8 TB/s for a pure-HBM gather of whole rows, below this kernel's fabric rate"}
    return rep, rep128, idx, pq


def structured_corpus(vg, ctx, stream, device, with_cpu):
    """The metric's NAMED pipeline at the NAMED bar, on data that has neighbourhoods: 1M x 768 rows on a 16-dimensional
    subspace + isotropic noise (gen_structured; the headline's i.i.d. normal corpus has none, §5 of DESIGN.md).  Same
    build, same PQ shape, same sweep as `hnsw_pq`; reports where HNSW fp32 and HNSW-on-PQ + exact rerank cross
    recall@10 = 0.95, their queries/s, the exact path on the same rows, and the CPU twin on the same graph.  An
    extra: `value` stays on the random-normal corpus BASELINE names."""
    rows = gen_structured(0, N_ROWS, device, seed=0)
    queries = gen_structured(0, 8 * Q_BATCH, device, seed=1).reshape(8, Q_BATCH, DIM)
    flat = vg.Index(ctx, N_ROWS, DIM)
    flat.set_vectors(rows)
    for _ in range(2):
        flat.search_flat(queries[1], K, stream=stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for i in range(3):
        flat.search_flat(queries[i], K, stream=stream)
    e1.record(stream)
    torch.cuda.synchronize()
    exact_ms = e0.elapsed_time(e1) / 3
    st0 = flat.flat_stats()
    flat.close()
    gi, _ = fp64_topk_local(rows, queries[0], 0, K)
    rep, _, idx, pq = hnsw_pq_frontier(vg, ctx, rows, queries, gi.cpu().numpy(), exact_ms, stream, with_cpu,
                                       efs_f32=(64, 128, 160, 192, 224, 256, 384), efs_pq=(64, 128, 160, 192, 224, 256, 384),
                                       cpu_efs=(128, 192, 256))
    idx.close()
    pq.close()
    rep["workload"] = (f"STRUCTURED extra corpus: 1M x 768 = z A + {STRUCT_NOISE} e, z ~ N(0, I_{STRUCT_LATENT}) (bench.gen_structured, "
                       "seeded), queries drawn the same way; ") + rep["workload"]

    def first(front):
        ok = [e for e in front if e["recall_at_10"] >= 0.95]
        return max(ok, key=lambda e: e["qps"]) if ok else None
    a, b = first(rep["frontier_f32"]), first(rep["frontier_pq_rerank"])
    rep["at_recall_0_95"] = {"hnsw_f32": a and {k_: a[k_] for k_ in ("ef", "recall_at_10", "qps")},
                             "hnsw_pq_rerank": b and {k_: b[k_] for k_ in ("ef", "recall_at_10", "qps")},
                             "exact_qps": rep["exact_path"]["qps"], "exact_fallback_queries": int(st0[1])}
    if with_cpu and "cpu" in rep:
        cpu_best = max((c for c in rep["cpu"]["sweep"] if c["recall_at_10"] >= 0.95), key=lambda c: c["qps"], default=None)
        if cpu_best and b:
            rep["at_recall_0_95"]["cpu_hnsw_f32"] = {k_: cpu_best[k_] for k_ in ("ef", "recall_at_10", "qps", "cores", "ids_equal_gpu")}
            rep["at_recall_0_95"]["gpu_hnsw_pq_rerank_over_cpu_hnsw"] = b["qps"] / cpu_best["qps"]
            if a:
                rep["at_recall_0_95"]["gpu_hnsw_f32_over_cpu_hnsw"] = a["qps"] / cpu_best["qps"]
        # the same pipeline on both sides: HNSW on PQ codes + exact rerank, each at ITS fastest ef that meets the bar
        cpu_pq = max((c for c in rep.get("cpu_pq_rerank", {}).get("sweep", []) if c["recall_at_10"] >= 0.95),
                     key=lambda c: c["qps"], default=None)
        if cpu_pq and b:
            rep["at_recall_0_95"]["cpu_hnsw_pq_rerank"] = {k_: cpu_pq[k_] for k_ in ("ef", "recall_at_10", "qps", "cores", "ids_equal_gpu")}
            rep["at_recall_0_95"]["gpu_hnsw_pq_rerank_over_cpu_hnsw_pq_rerank"] = b["qps"] / cpu_pq["qps"]
    return rep


def hnsw_cpu_twin(idx, rows, q, f32, efs=(128, 512, 2048)):
    """The same searches on the host: the graph the GPU built, hnsw.KNNSearch restated in C (oracle), the
    reference's compiled AVX-512 distance kernels, one query per thread.  Same graph + same algorithm = same
    answers: the first queries' ids are compared with the GPU's."""
    from oracle import oracle as o
    l0, upper, entry = idx.get_hnsw_graph()
    ic = o.InterleavedCopy(rows.cpu().numpy())
    try:
        h = o.HnswIndex(ic.array, DIM, l0, upper, entry, m=HNSW_M)
        qh = q.cpu().numpy()
        out = []
        for ef in efs:
            r = cpu_leg(o.BENCH_HNSW, qh, K, 4.0, hnsw=h, ef=ef, want_ids=True)
            gids, _ = idx.search_hnsw(q[:256], K, ef)
            gids = gids.cpu().numpy().view(np.uint32)
            filled = np.nonzero(r["dist_comps"][:256] >= 0)[0]
            same = bool(filled.size) and all(np.array_equal(r["ids"][i], gids[i]) for i in filled)
            gpu = next(e for e in f32 if e["ef"] == ef)
            out.append({"ef": ef, "qps": r["qps"], "cores": r["cores"], "kind": r["kind"], "queries": r["queries"],
                        "seconds": r["seconds"], "ids_equal_gpu": same, "compared_queries": int(filled.size),
                        "recall_at_10": gpu["recall_at_10"], "gpu_qps": gpu["qps"], "gpu_over_cpu": gpu["qps"] / r["qps"]})
        return {"sweep": out, **host_info()}
    finally:
        ic.close()


def hnsw_pq_cpu_twin(idx, pq, codes, rows, q, pqr, efs=(128, 512, 2048)):
    """The metric's NAMED pipeline on the host: the same graph walked on the same PQ codes (oracle searchLayer with
    distFunc = ComputeAsymmetricDistance, pq.go:234-260) for ef candidates, Segment.Rerank's exact distances
    (the reference's compiled squaredL2Avx512) and the best k by (Score, RowID) — engine/search.go:914-965 — one
    query per C thread.  Same graph + same codes + same algorithm = same answers: ids compared with the GPU's."""
    from oracle import oracle as o
    l0, upper, entry = idx.get_hnsw_graph()
    opq = o.ProductQuantizer(DIM, PQ_M, 256)
    opq.set_codebooks(*[np.asarray(x) for x in pq.codebooks()])
    hcodes = codes.cpu().numpy() if hasattr(codes, "cpu") else np.asarray(codes)
    ic = o.InterleavedCopy(rows.cpu().numpy())
    try:
        h = o.HnswIndex(ic.array, DIM, l0, upper, entry, m=HNSW_M, pq=opq, codes=hcodes)
        qh = q.cpu().numpy()
        out = []
        for ef in efs:
            gpu = next((e for e in pqr if e["ef"] == ef), None)
            if gpu is None:
                continue
            r = cpu_leg(o.BENCH_HNSW_PQ_RERANK, qh, K, 4.0, hnsw=h, ef=ef, want_ids=True)
            cand, _ = idx.search_hnsw_pq(q[:256], ef, ef)
            gids, _ = idx.rerank(q[:256], cand, K)
            gids = gids.cpu().numpy().view(np.uint32)
            filled = np.nonzero(r["dist_comps"][:256] >= 0)[0]
            same = bool(filled.size) and all(np.array_equal(r["ids"][i], gids[i]) for i in filled)
            out.append({"ef": ef, "qps": r["qps"], "cores": r["cores"], "kind": r["kind"], "queries": r["queries"],
                        "seconds": r["seconds"], "ids_equal_gpu": same, "compared_queries": int(filled.size),
                        "recall_at_10": gpu["recall_at_10"], "gpu_qps": gpu["qps"], "gpu_over_cpu": gpu["qps"] / r["qps"]})
        return {"pipeline": "hnsw walk on PQ codes (ef candidates) + exact rerank, top-10", "sweep": out}
    finally:
        ic.close()


PQ_SCORE_VALU_SLOTS = 24 * 213  # vector issue slots per PQ node score (m = 96): see vamana_pq
NOMINATE_VALU_PER_BLOCK = 428  # vector instructions pq_nominate_bf16_kernel executes per (32 rows, sub-quantizer) block of a wave:
                               # SQ_INSTS_VALU / blocks (profiles/r06_pmc_encode_inst.csv; tools/isa_count.py on the hot loop: 404 + the flush)
PEAK_VALU_LANEOPS = 78.6e12    # fp32 vector lane-operations per second: 256 CUs x 4 SIMD-32 x 2.4 GHz (MI355X_MICROARCH.md:
                               # "4 SIMD-32 units", v_fma_f32 wave64 = 2 cycles); an FMA counts as ONE lane-op here, so the
                               # same rate is the guide's 157.3 TFLOP/s vector peak


def _cpu_build(kind, units, budget_s, **kw):
    """One build-side CPU twin (oracle/vg_cpu_bench.c vgo_bench_build_run): one unit per C thread, the reference's
    compiled AVX-512 kernels where the reference calls internal/simd."""
    from oracle import oracle as o
    threads = effective_cpus()
    is_ref = o.use_reference_kernels(True)
    try:
        r = o.bench_build_run(kind, units, threads, budget_s, want_out=True, **kw)
    finally:
        o.use_reference_kernels(False)
    r.update(cores=threads, kind="reference" if is_ref else "port")
    return r


def build_side_legs(vg, ctx, rows, queries, stream, with_cpu, rows_host=None):
    """The functions north_star names that the search legs do not time: kmeans.TrainKMeans / AssignPartition
    (kmeans.go:16-138,142-196), ProductQuantizer.Train (pq.go:68-143: seeding and Lloyd apart), Encode (pq.go:147-176),
    BuildDistanceTable (pq.go:468-491), Segment.Rerank (flat/segment.go:754-780, engine/search.go:914-965) and
    hnsw.BruteSearch (hnsw.go:2021-2101) — each at the bench's 1M x 768 shape, kernel time from HIP events inside the
    library (vg_profile), priced against the bound that applies, next to its CPU twin (oracle/vg_cpu_bench.c, one unit
    per thread) with the GPU's results compared with the twin's."""
    from oracle import oracle as o
    n = rows.shape[0]
    out = {}

    def prof(keys, fn, reps=1):
        """fn() `reps` times under vg_profile; -> ({key: (launches, total_ms)}, wall_ms per rep)"""
        fn()
        torch.cuda.synchronize()
        for key in keys:
            ctx.profile_read(key)
        ctx.profile_enable(True)
        t0 = time.perf_counter()
        for _ in range(reps):
            r = fn()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e3 / reps
        ctx.profile_enable(False)
        return {key: ctx.profile_read(key) for key in keys}, wall, r

    def valu_row(workload, kernel, kernel_ms, lane_ops, **extra):
        ach = lane_ops / (kernel_ms * 1e-3) / 1e12
        return {"workload": workload, "kernel": kernel, "kernel_ms": kernel_ms, "bound": "valu", "achieved": ach,
                "peak": PEAK_VALU_LANEOPS / 1e12, "unit": "T lane-ops/s", "frac": ach * 1e12 / PEAK_VALU_LANEOPS, **extra}

    # ---- kmeans.TrainKMeans at the flat writer's shape (flat/writer.go:109: k = N / 8192 partitions, 10 iterations)
    k_parts = max(n // 8192, 2)
    pr, wall, cent = prof(("km_assign", "km_update"), lambda: vg.kmeans_train(ctx, rows, DIM, k_parts, max_iter=10, seed=3, stream=stream))
    la, ta = pr["km_assign"]
    lu, tu = pr["km_update"]
    assign_ms = ta / max(la, 1)
    # One assignment pass = km_gemm_kernel (a training run of >= 3 iterations: bfloat16 splits [hi|lo] x [hi|lo] on
    # v_mfma_f32_32x32x16_bf16, three products per 64-element chunk) + the decision + the reference-order kernels over the points
    # the matrix scores cannot decide.  `bound` / `frac` = what limits THIS pass: every point read once, n dim 4 algorithmic bytes
    # over the HBM peak (the [hi|lo] image read is the same 4 bytes per element).  `over_reference_valu` = the pass's n k dim
    # element-steps priced as the REFERENCE writes them — (sub, fma) = 2 lane-operations on the vector ALU — over the vector peak:
    # above 1 = faster than any kernel that computes the distances that way (r04's ran at 0.40); it is not a fraction of anything
    # this kernel executes.
    km_ops = 2.0 * n * k_parts * DIM
    km_gbs = n * DIM * 4.0 / (assign_ms * 1e-3) / 1e9
    row = {"workload": f"kmeans.TrainKMeans {n} x {DIM}, k = {k_parts}, 10 iterations: one assignment pass (kmeans.go:54-99)",
           "kernel": "km_gemm_kernel<bf16 splits> + km_decide / km_pairs / km_assign_regs<LIST>", "kernel_ms": assign_ms,
           "bound": "hbm", "achieved": km_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": km_gbs / PEAK_HBM_GBS,
           "over_reference_valu": km_ops / (assign_ms * 1e-3) / PEAK_VALU_LANEOPS,
           "mfma_f32_equiv": km_ops / (assign_ms * 1e-3) / 1e12 / PEAK_MFMA_F32_TFLOPS,
           "short": f"kmeans_{n}x{DIM}_k{k_parts}_10it", "train_ms": wall, "assign_launches": la,
           "update_ms_per_iter": tu / max(lu, 1), "lane_ops_per_pass": km_ops}
    ga = vg.kmeans_assign(ctx, rows, cent, DIM, stream=stream)
    if with_cpu and rows_host is not None:
        r = _cpu_build(o.BUILD_KM_ASSIGN, rows_host, 4.0, centroids=cent.cpu().numpy())
        ca = r["out"]["assign"]
        done = ca >= 0
        row.update(cpu_ms=1e3 * n / r["rate"], cpu_rows_per_s=r["rate"], cpu_cores=r["cores"], cpu_kind=r["kind"],
                   bits_equal=bool(done.any()) and bool(np.array_equal(ga.cpu().numpy()[done], ca[done])),
                   compared=int(done.sum()))
    out["kmeans_assign"] = row

    # ---- ProductQuantizer.Train on 65536 rows (m = 96, K = 256, 20 iterations): seeding and Lloyd apart
    ntrain = min(65536, n)
    pq = vg.ProductQuantizer(ctx, DIM, PQ_M, 256)
    train_rows = rows[:ntrain].contiguous()
    pr, wall, _ = prof(("pq_kmeanspp", "pq_assign", "pq_update"), lambda: pq.train(train_rows, iters=20, seed=1, stream=stream))
    lpp, tpp = pr["pq_kmeanspp"]
    las, tas = pr["pq_assign"]
    lup, tup = pr["pq_update"]
    sd = DIM // PQ_M
    # seeding (pq.go:281-338): per sub-quantizer and new centroid one pass over the sub-quantizer's slab (n x sd floats) and
    # minDistSq (read + write): 96 slabs of 2 MB do not stay in the 4 MB L2s, so the passes stream from HBM / the memory-side
    # cache.  (Through r04 the bound was the reference's chain of n dependent additions per centroid: 60 ms; the running sum
    # is blocked since r05, one definition shared with the oracle — oracle/vg_oracle.c.)
    seed_ms = tpp / max(lpp, 1)
    seed_bytes = PQ_M * 255.0 * ntrain * (sd * 4 + 8)
    out["pq_train_seeding"] = {
        "workload": f"ProductQuantizer.Train {ntrain} x {DIM}, m = {PQ_M}, K = 256: k-means++ seeding (pq.go:281-338)",
        "kernel": "pq_kmeanspp_kernel", "kernel_ms": seed_ms, "bound": "hbm",
        "achieved": seed_bytes / (seed_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
        "frac": seed_bytes / (seed_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, "bytes": seed_bytes, "train_ms": wall,
        "short": f"pq_train_{ntrain}x{DIM}_m{PQ_M}_K256_20it"}
    lloyd_ms = tas / max(las, 1)

    def mfma_row(workload, kernel, kernel_ms, pairs, valu_ops_per_elem, **extra):
        """pq_nominate_bf16_kernel: `pairs` (row, sub-quantizer) pairs x 256 centroids x 8 dimensions.  `bound` / `frac` = what
        limits the kernel: vector-ALU issue.  Its loop executes NOMINATE_VALU_PER_BLOCK vector instructions per (32 rows,
        sub-quantizer) block of a wave (tools/isa_count.py on the kernel's hot loop; 2.25 of them per score keep the smallest /
        second smallest key: v_and_or + med3 / min3 over four keys), and none of them is the fp32 add / fma that gfx950 issues at
        twice the rate: every one costs a SIMD 4 cycles (tools/ubench/valu_rate.hip; the SQ counts them so: SQ_ACTIVE_INST_VALU =
        SQ_INSTS_VALU quad-cycles, profiles/r06_pmc_encode_valu.csv) — peak = 256 CUs x 4 SIMDs x 2.4 GHz / 4 wave-instructions/s.
        The arithmetic itself is on the bf16 matrix cores (2 instructions per 1024 scores, ~25 % busy).  `over_reference_valu` =
        the reference's arithmetic (valu_ops_per_elem lane-operations per element) over the vector peak: above 1 = faster than any
        kernel computing the distances as the reference writes them; not a fraction of anything this kernel executes."""
        ops = valu_ops_per_elem * pairs * 256 * sd
        blocks = pairs / 32.0
        ginstr = NOMINATE_VALU_PER_BLOCK * blocks / (kernel_ms * 1e-3) / 1e9
        return {"workload": workload, "kernel": kernel, "kernel_ms": kernel_ms, "bound": "valu_issue",
                "achieved": ginstr, "peak": PEAK_VALU_GINSTR, "unit": "G wave-instructions/s", "frac": ginstr / PEAK_VALU_GINSTR,
                "valu_instructions_per_block": NOMINATE_VALU_PER_BLOCK,
                "over_reference_valu": ops / (kernel_ms * 1e-3) / PEAK_VALU_LANEOPS,
                "mfma_f32_equiv": 2.0 * pairs * 256 * sd / (kernel_ms * 1e-3) / 1e12 / PEAK_MFMA_F32_TFLOPS, **extra}

    out["pq_train_lloyd"] = mfma_row(
        f"ProductQuantizer.Train {ntrain} x {DIM}, m = {PQ_M}, K = 256: one Lloyd assignment pass (pq.go:353-386)",
        "pq_nominate_bf16_kernel<false> + pq_fix_kernel", lloyd_ms, float(ntrain) * PQ_M, 2.0, iterations=las,
        update_ms_per_iter=tup / max(lup, 1), train_ms=wall, short=f"pq_train_{ntrain}x{DIM}_m{PQ_M}_K256_20it")
    if with_cpu and rows_host is not None:
        r = _cpu_build(o.BUILD_PQ_TRAIN_SUB, rows_host[:ntrain], 0.0, pq_m=PQ_M, pq_k=256, iters=20, seed=1)
        nsub = min(r["units"], PQ_M)
        cb, sc, of = [np.asarray(x) for x in pq.codebooks()]
        same = True
        for sub in range(nsub):
            qc, qs, qo = o.pq_quantize_centroids(r["out"]["cent"][sub])
            same = same and np.array_equal(qc.reshape(-1), cb.reshape(PQ_M, -1)[sub]) and \
                np.float32(qs).view(np.uint32) == np.float32(sc[sub]).view(np.uint32)
        cpu_ms = 1e3 * PQ_M / r["rate"]
        for key in ("pq_train_seeding", "pq_train_lloyd"):
            out[key].update(cpu_train_ms=cpu_ms, cpu_cores=r["cores"], cpu_kind=r["kind"], bits_equal=bool(same),
                            compared=f"{nsub} sub-quantizers' codebooks")

    # ---- Encode 1M x 768 (pq.go:147-176)
    pr, wall, codes = prof(("pq_encode",), lambda: pq.encode(rows, stream=stream), reps=3)
    le, te = pr["pq_encode"]
    enc_ms = te / max(le, 1)
    # FindNearestCentroidInt8 (kernels.go:376-396): per (row, sub-quantizer, centroid, dimension) d = q - v, dd = d * d,
    # sum = sum + dd — three separately rounded operations (Go on amd64 does not fuse), the dequantisation hoisted
    row = mfma_row(f"ProductQuantizer.Encode {n} x {DIM} -> {PQ_M} B (pq.go:147-176)", "pq_nominate_bf16_kernel<true> + pq_fix_kernel",
                   enc_ms, float(n) * PQ_M, 3.0, rows_per_s=n / (enc_ms * 1e-3), short=f"pq_encode_{n}x{DIM}_m{PQ_M}")
    if with_cpu and rows_host is not None:
        opq = o.ProductQuantizer(DIM, PQ_M, 256)
        opq.set_codebooks(*[np.asarray(x) for x in pq.codebooks()])
        r = _cpu_build(o.BUILD_PQ_ENCODE, rows_host[:200_000], 6.0, pq=opq)
        cov = min(r["units"], 200_000)
        gc = codes[:200_000].cpu().numpy()
        # strided units: thread t covers rows t, t + T, ...; after `units` completions every row below units - T is done
        lim = max(cov - r["cores"], 0)
        row.update(cpu_ms=1e3 * n / r["rate"], cpu_rows_per_s=r["rate"], cpu_cores=r["cores"], cpu_kind="port (generic Go on amd64)",
                   bits_equal=bool(lim) and bool(np.array_equal(gc[:lim], r["out"]["codes"][:lim])), compared=int(lim))
    out["pq_encode"] = row

    # ---- BuildDistanceTable x 1024 queries (pq.go:468-491)
    q1024 = queries.reshape(-1, DIM)[:1024].contiguous()
    tables = torch.empty((1024, PQ_M * 256), dtype=torch.float32, device=rows.device)
    pr, wall, _ = prof(("pq_build_table",), lambda: pq.build_distance_table(q1024, out=tables, stream=stream), reps=10)
    lt, tt = pr["pq_build_table"]
    lut_ms = tt / max(lt, 1)
    lut_bytes = 1024.0 * PQ_M * 256 * 4
    row = {"workload": f"ProductQuantizer.BuildDistanceTable x 1024 queries (pq.go:468-491): {lut_bytes / 1e6:.0f} MB of tables written",
           "kernel": "pq_build_table_rows8_kernel", "kernel_ms": lut_ms, "bound": "hbm", "achieved": lut_bytes / (lut_ms * 1e-3) / 1e9,
           "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": lut_bytes / (lut_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
           "valu_frac": 5.0 * 1024 * PQ_M * 256 * sd / (lut_ms * 1e-3) / PEAK_VALU_LANEOPS, "tables_per_s": 1024 / (lut_ms * 1e-3),
           "short": f"pq_lut_x1024_m{PQ_M}_K256"}
    if with_cpu:
        r = _cpu_build(o.BUILD_PQ_LUT, q1024.cpu().numpy(), 2.0, pq=opq)
        same = all(np.array_equal(opq.build_table(q1024[i].cpu().numpy()).view(np.uint32), tables[i].cpu().numpy().view(np.uint32))
                   for i in (0, 511, 1023))
        row.update(cpu_ms=1e3 * 1024 / r["rate"], cpu_tables_per_s=r["rate"], cpu_cores=r["cores"],
                   cpu_kind="port (generic Go on amd64)", bits_equal=same)
    out["pq_build_table"] = row
    del tables

    # ---- Segment.Rerank: 8192 queries x 512 candidates each (the PQ walk's ef = 512 result lists), top-10
    idx = vg.Index(ctx, n, DIM)
    idx.set_vectors(rows)
    qf = queries.reshape(-1, DIM)[:NQ_FLIGHT].contiguous()
    nc = 512
    g = torch.Generator(device=rows.device)
    g.manual_seed(99)
    cand = torch.randint(0, n, (qf.shape[0], nc), device=rows.device, generator=g, dtype=torch.int64).to(torch.int32)
    pr, wall, rr = prof(("rerank",), lambda: idx.rerank(qf, cand, K, stream=stream), reps=3)
    lr, tr = pr["rerank"]
    rr_ms = tr / max(lr, 1)
    rr_bytes = float(qf.shape[0]) * nc * DIM * 4
    row = {"workload": f"Segment.Rerank + top-{K}: {qf.shape[0]} queries x {nc} candidates x {DIM} fp32 (flat/segment.go:754-780, engine/search.go:914-965)",
           "kernel": "rerank_kernel", "kernel_ms": rr_ms, "bound": "hbm", "achieved": rr_bytes / (rr_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS,
           "unit": "GB/s", "frac": rr_bytes / (rr_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, "queries_per_s": qf.shape[0] / (rr_ms * 1e-3),
           "note": "gathered row bytes (uniform random candidates: no row reuse) over the HBM peak",
           "short": f"rerank_{qf.shape[0]}q_x{nc}cand_{DIM}d_top{K}"}
    if with_cpu and rows_host is not None:
        r = _cpu_build(o.BUILD_RERANK, qf.cpu().numpy(), 4.0, base=rows_host, cand=cand.cpu().numpy().view(np.uint32), topk=K)
        cov = max(min(r["units"], qf.shape[0]) - r["cores"], 0)
        gi = rr[0].cpu().numpy().view(np.uint32)
        gs = rr[1].cpu().numpy().view(np.uint32)
        row.update(cpu_qps=r["rate"], cpu_cores=r["cores"], cpu_kind=r["kind"], compared=int(cov),
                   bits_equal=bool(cov) and bool(np.array_equal(gi[:cov], r["out"]["ids"][:cov])) and
                   bool(np.array_equal(gs[:cov], r["out"]["scores"][:cov].view(np.uint32))))
    out["rerank"] = row

    # ---- hnsw.BruteSearch + scanSegment (hnsw.go:2021-2101): one query (rows streamed once) and 256 queries
    brute = {}
    for nq in (1, 256):
        qb = qf[:nq].contiguous()
        pr, wall, br = prof(("hnsw_brute_dist", "hnsw_brute_replay"), lambda: idx.search_hnsw_brute(qb, K, 0, stream=stream), reps=3)
        ld, td = pr["hnsw_brute_dist"]
        lp, tp = pr["hnsw_brute_replay"]
        d_ms = td / 3
        if nq == 1:
            ach = n * DIM * 4.0 / (d_ms * 1e-3) / 1e9
            # one unmasked query: vg_search_flat's exact scan for k + 1 results (flat_scan_mq_kernel + merge), turned into the
            # reference's answer by brute_from_flat_kernel when no two of the k + 1 distances tie (else: distance pass + replay)
            e = {"kernel": "flat_scan_mq_kernel (k + 1) + topk_merge + brute_from_flat_kernel", "kernel_ms": d_ms, "bound": "hbm", "achieved": ach,
                 "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBS}
        else:
            # two or more unmasked queries: vg_search_flat's MFMA nomination + exact re-score + proof for k + 1 results,
            # turned into the reference's answer when no two of the k + 1 distances are equal (ties: the replay)
            tf = 2.0 * nq * n * DIM / (d_ms * 1e-3) / 1e12
            e = {"kernel": "flat_gemm_dma_kernel (k + 1) + brute_from_flat_kernel", "kernel_ms": d_ms, "bound": "mfma", "achieved": tf,
                 "peak": PEAK_MFMA_F32_TFLOPS, "unit": "TFLOP/s", "frac": tf / PEAK_MFMA_F32_TFLOPS,
                 "valu_equiv": 2.0 * nq * n * DIM / (d_ms * 1e-3) / PEAK_VALU_LANEOPS}
        e.update(workload=f"hnsw.BruteSearch {nq} quer{'y' if nq == 1 else 'ies'} x {n} x {DIM}, top-{K}, heap replayed (hnsw.go:2021-2101)",
                 replay_ms=tp / 3, call_ms=wall, queries_per_s=nq / (wall * 1e-3), short=f"hnsw_brute_{nq}q_{n}x{DIM}_top{K}")
        brute[nq] = (e, br)
    if with_cpu and rows_host is not None:
        h = o.HnswIndex(rows_host, DIM, np.zeros((n, 1), np.uint32), (), 0, m=1)
        r = _cpu_build(o.BUILD_BRUTE, qf[:64].cpu().numpy(), 4.0, hnsw=h, mode=0, topk=K)
        cov = max(min(r["units"], 64) - r["cores"], 0)
        gi = brute[256][1][0][:64].cpu().numpy().view(np.uint32)
        gs = brute[256][1][1][:64].cpu().numpy().view(np.uint32)
        for nq in (1, 256):
            brute[nq][0].update(cpu_qps=r["rate"], cpu_cores=r["cores"], cpu_kind=r["kind"], compared=int(cov),
                                bits_equal=bool(cov) and bool(np.array_equal(gi[:cov], r["out"]["ids"][:cov])) and
                                bool(np.array_equal(gs[:cov], r["out"]["scores"][:cov].view(np.uint32))))
    out["brute_q1"] = brute[1][0]
    out["brute_q256"] = brute[256][0]

    # ---- flat.Segment.Search with `filter` set (flat/segment.go:631-635): 1024 queries, each its own filter keeping 1/8 of
    # the rows (the AND of three random bytes); one query with one such filter (rows read = the rows it keeps)
    nqf = min(1024, qf.shape[0])
    g = torch.Generator(device=rows.device)
    g.manual_seed(5)
    nb = (n + 7) // 8
    fm = torch.randint(0, 256, (nqf, nb), device=rows.device, generator=g, dtype=torch.uint8)
    for _ in range(2):
        fm &= torch.randint(0, 256, (nqf, nb), device=rows.device, generator=g, dtype=torch.uint8)
    qb = qf[:nqf].contiguous()
    pr, wall, fr = prof(("flat_gemm",), lambda: idx.search_flat_filtered(qb, K, fm, 0, stream=stream), reps=3)
    lg, tg = pr["flat_gemm"]
    g_ms = tg / max(lg, 1)
    tf = 2.0 * nqf * n * DIM / (g_ms * 1e-3) / 1e12
    row = {"workload": f"flat.Segment.Search with a row filter per query (1/8 of the rows pass): {nqf} queries x {n} x {DIM}, top-{K} "
                       "(flat/segment.go:631-635)",
           "kernel": "flat_gemm_dma_kernel<false,2> + row filter", "kernel_ms": g_ms, "bound": "mfma",
           "achieved": tf, "peak": PEAK_MFMA_F32_TFLOPS, "unit": "TFLOP/s", "frac": tf / PEAK_MFMA_F32_TFLOPS, "call_ms": wall,
           "queries_per_s": nqf / (wall * 1e-3), "short": f"flat_filtered_{nqf}q_{n}x{DIM}_keep0.125_top{K}"}
    one = fm[0].contiguous()
    pr1, wall1, _ = prof(("flat_probe",), lambda: idx.search_flat_filtered(qb[:1], K, one, 0, stream=stream), reps=5)
    l1, t1 = pr1["flat_probe"]
    kept = int(np.unpackbits(one.cpu().numpy(), bitorder="little")[:n].sum())
    p_ms = t1 / max(l1, 1)
    row.update(one_query_call_ms=wall1, one_query_kernel_ms=p_ms, one_query_rows_kept=kept,
               one_query_gbs=kept * DIM * 4.0 / (p_ms * 1e-3) / 1e9)
    if with_cpu and rows_host is not None:
        seg = o.FlatSegment(rows_host, DIM)
        same, t0 = True, time.perf_counter()
        checked = 0
        for i in (0, 1, nqf // 2, nqf - 1):
            mi = np.unpackbits(fm[i].cpu().numpy(), bitorder="little")[:n].astype(bool)
            eid, esc = seg.search(qb[i].cpu().numpy(), K, mask=mi)
            same = same and np.array_equal(fr[0][i].cpu().numpy().view(np.uint32)[:eid.size], eid) and \
                np.array_equal(fr[1][i].cpu().numpy().view(np.uint32)[:eid.size], esc.view(np.uint32))
            checked += 1
        row.update(cpu_qps=checked / (time.perf_counter() - t0), cpu_cores=1, cpu_kind="port", compared=checked, bits_equal=bool(same))
    out["flat_filtered"] = row
    del fm

    # ---- flat.Segment.Search, SQ8 branch, 1024 queries: the multi-query scan (every code decoded once per 4 queries, vector-ALU
    # bound) and the opt-in bf16 nomination (vg_index_enable_sq8_nomination) + exact re-score from the codes + proof
    sq = vg.ScalarQuantizer(ctx, DIM)
    sq.train(rows[:200000])
    idx.set_sq8_codes(sq, sq.encode(rows))
    pr, wall_scan, r_scan = prof(("sq8_scan",), lambda: idx.search_sq8(qb, K, stream=stream), reps=1)
    idx.enable_sq8_nomination(True)
    pr, wall_nom, r_nom = prof(("sq8_nominate_gemm",), lambda: idx.search_sq8(qb, K, stream=stream), reps=3)
    lg, tg = pr["sq8_nominate_gemm"]
    g_ms = tg / max(lg, 1)
    tf = 2.0 * nqf * n * DIM / (g_ms * 1e-3) / 1e12
    out["sq8_batch"] = {"workload": f"flat.Segment.Search SQ8 branch, {nqf} queries x {n} x {DIM}, top-{K} (flat/segment.go:517-604): bf16 nomination "
                                    "over the dequantised rows + L2Distance of the 64 nominated rows from the codes + proof",
                        "kernel": "flat_gemm_bf16_big_kernel<false,3> + sq8_verify_kernel", "kernel_ms": g_ms, "bound": "mfma",
                        "achieved": tf, "peak": PEAK_MFMA_BF16_TFLOPS, "unit": "TFLOP/s", "frac": tf / PEAK_MFMA_BF16_TFLOPS, "call_ms": wall_nom,
                        "queries_per_s": nqf / (wall_nom * 1e-3), "scan_call_ms": wall_scan, "scan_queries_per_s": nqf / (wall_scan * 1e-3),
                        "bits_equal": bool(torch.equal(r_scan[0], r_nom[0]) and torch.equal(r_scan[1].view(torch.int32), r_nom[1].view(torch.int32))),
                        "compared": nqf, "short": f"sq8_batch_{nqf}q_{n}x{DIM}_top{K}"}
    idx.enable_sq8_nomination(False)

    # ---- flat.Segment.Search, PQ branch, 1024 queries: one table scan per query (LDS gather rate) and the opt-in bf16 nomination
    # over the DECODED rows (vg_index_enable_pq_nomination) + table sums of the nominated rows from the codes + proof
    idx.set_pq_codes(pq, codes, stream=stream)
    pr, wall_scan, r_scan = prof(("pq_adc_scan",), lambda: idx.search_pq_adc(qb, K, stream=stream), reps=1)
    idx.enable_pq_nomination(True)
    pr, wall_nom, r_nom = prof(("sq8_nominate_gemm",), lambda: idx.search_pq_adc(qb, K, stream=stream), reps=3)
    lg, tg = pr["sq8_nominate_gemm"]
    g_ms = tg / max(lg, 1)
    tf = 2.0 * nqf * n * DIM / (g_ms * 1e-3) / 1e12
    out["pq_batch"] = {"workload": f"flat.Segment.Search PQ branch, {nqf} queries x {n} x m{PQ_M}, top-{K} (flat/segment.go:476-483,678-689): bf16 "
                                   "nomination over the decoded rows + BuildDistanceTable / pqAdcLookup sums of the 64 nominated rows + proof",
                       "kernel": "flat_gemm_bf16_big_kernel<false,3> + pq_verify_kernel", "kernel_ms": g_ms, "bound": "mfma",
                       "achieved": tf, "peak": PEAK_MFMA_BF16_TFLOPS, "unit": "TFLOP/s", "frac": tf / PEAK_MFMA_BF16_TFLOPS, "call_ms": wall_nom,
                       "queries_per_s": nqf / (wall_nom * 1e-3), "scan_call_ms": wall_scan, "scan_queries_per_s": nqf / (wall_scan * 1e-3),
                       "extra_hbm_bytes": n * ((DIM + 63) // 64 * 64) * 2,
                       "bits_equal": bool(torch.equal(r_scan[0], r_nom[0]) and torch.equal(r_scan[1].view(torch.int32), r_nom[1].view(torch.int32))),
                       "compared": nqf, "short": f"pq_batch_{nqf}q_{n}x{DIM}_m{PQ_M}_top{K}"}
    idx.close()
    pq.close()
    return out


def vamana_pq(vg, ctx, idx, rows, q, gt_ids, stream):
    """BASELINE configs[3], graph half: Vamana beam search (diskann/segment.go:503-706) with PQ node scoring
    (ComputeAsymmetricDistance order) over the built graph's layer 0 as the adjacency (R = 64).  A node-scoring
    rate, not a QPS claim: the beam stops at k results (segment.go:655-668), recall before rerank is low."""
    l0, _, entry = idx.get_hnsw_graph()
    idx.set_vamana_graph(l0, entry)
    ids, _, st = idx.search_vamana(q, K, kind=1, stats=True, stream=stream)
    torch.cuda.synchronize()
    ctx.profile_read("vamana_search")
    ctx.profile_enable(True)
    for _ in range(3):
        idx.search_vamana(q, K, kind=1, stream=stream)
    torch.cuda.synchronize()
    launches, ms = ctx.profile_read("vamana_search")
    ctx.profile_enable(False)
    kern_ms = ms / 3     # the kernel launches of ONE call (r02 divided by launches: a call was 5 launches of 1650 queries)
    dc = float(st[:, 1].sum())
    gathered = dc * PQ_M + float(st[:, 3].sum()) * l0.shape[1] * 4
    # the same beam with the reference's other node scorers (diskann/segment.go:503-565): fp32 rows, RaBitQ codes,
    # INT4 codes — one kernel instance each
    others = {}
    idx.set_rabitq_codes(vg.RaBitQuantizer(ctx, DIM).encode(rows))
    iq = vg.Int4Quantizer(ctx, DIM); iq.train(rows[:65536])
    idx.set_int4_codes(iq, iq.encode(rows))
    for kind, name, row_bytes in ((0, "fp32", DIM * 4), (2, "rabitq", DIM // 8 + 4), (3, "int4", DIM // 2)):
        _, _, st_k = idx.search_vamana(q, K, kind=kind, stats=True, stream=stream)
        torch.cuda.synchronize()
        ctx.profile_read("vamana_search")
        ctx.profile_enable(True)
        for _ in range(3):
            idx.search_vamana(q, K, kind=kind, stream=stream)
        torch.cuda.synchronize()
        _, ms_k = ctx.profile_read("vamana_search")
        ctx.profile_enable(False)
        dck = float(st_k[:, 1].sum())
        others[name] = {"kernel_ms": ms_k / 3, "node_scores_per_s": dck / (ms_k / 3 * 1e-3), "node_scores_per_query": dck / q.shape[0],
                        "gathered_gbs": dck * row_bytes / (ms_k / 3 * 1e-3) / 1e9}
    return {"workload": f"vamana_pq_1Mx768_m96_K256_k10 over the built graph's layer 0 (R = {l0.shape[1]}), {q.shape[0]} queries in flight",
            "kernel": "vamana_search_kernel<4, false>", "kernel_ms": kern_ms, "launches_per_call": launches // 3, "node_scores_per_s": dc / (kern_ms * 1e-3),
            "lut_lookups_per_s": dc * PQ_M / (kern_ms * 1e-3),
            "recall_at_10_before_rerank": recall_at_k(ids.cpu().numpy().view(np.uint32)[:gt_ids.shape[0]], gt_ids),
            "node_scores_per_query": dc / q.shape[0], "gathered_gbs": gathered / (kern_ms * 1e-3) / 1e9,
            "other_node_scorers": others,
            # the node scorer's own bound: vector-ALU issue.  pq_term8_quad (vg_hnsw_layer.hpp:178-210) is 213 issue slots per four
            # (node, sub-quantizer) terms on one lane's node (tools/isa_count.py on tools/ubench/pq_slice.hip's scoring loop: 57
            # single + 78 packed-fp32 instructions, a packed one = 2 slots) = 24 x 213 = 5112 slots per node score at m = 96, one node
            # per lane: 64 scores per wave-instruction slot at the fp32 rate (2 cycles per slot and SIMD)
            "bound": "valu_issue", "achieved": dc * PQ_SCORE_VALU_SLOTS / (kern_ms * 1e-3) / 1e12, "peak": PEAK_VALU_LANEOPS / 1e12,
            "unit": "T lane-ops/s", "frac": dc * PQ_SCORE_VALU_SLOTS / (kern_ms * 1e-3) / PEAK_VALU_LANEOPS,
            "valu_slots_per_node_score": PQ_SCORE_VALU_SLOTS, "gathered_frac_of_hbm": gathered / (kern_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
            "bound_note": "vector-ALU issue of the node terms computed from the shared int8 codebook (no per-query table) next to the "
                          "heaps' serial work on the same ALU (scoring = 36 % of a wave's pop, r04 timing build): DESIGN.md"}


def flat_ivf_probe(vg, ctx, rows, queries, gt_ids, stream):
    """flat.Segment.Search over an IVF-partitioned segment (flat/segment.go:727-749): the corpus in
    rows/8192 k-means partitions as compaction writes it (engine/compaction.go:137-141), trained and
    assigned on the GPU (untimed), 1024 queries, nprobes = 1 (the reference's default) and 8.  The
    recall printed is what probing reaches on i.i.d. normal data, not a kernel property."""
    n = rows.shape[0]
    parts = n // 8192
    t0 = time.perf_counter()
    cent = vg.kmeans_train(ctx, rows, DIM, parts, max_iter=10, seed=1)
    assign = vg.kmeans_assign(ctx, rows, cent, DIM).to(torch.int64)
    order = torch.argsort(assign, stable=True)
    grouped = rows[order].contiguous()
    off = np.concatenate([[0], np.cumsum(torch.bincount(assign, minlength=parts).cpu().numpy())]).astype(np.uint32)
    torch.cuda.synchronize()
    prep_s = time.perf_counter() - t0
    idx = vg.Index(ctx, n, DIM)
    idx.set_vectors(grouped)
    idx.set_partitions(cent.cpu().numpy(), off)
    q = queries.reshape(-1, DIM)[:Q_BATCH]
    order_h = order.cpu().numpy()
    res = {"workload": f"flat_ivf_probe_1Mx768_{parts}_partitions_k10, {Q_BATCH} queries per call", "kmeans_assign_sort_s": prep_s}
    for nprobes in (1, 8):
        ids, _ = idx.search_flat_probed(q, K, nprobes, scan=idx.SCAN_F32, stream=stream)
        torch.cuda.synchronize()
        ctx.profile_read("flat_probe")
        ctx.profile_enable(True)
        reps = 5
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            idx.search_flat_probed(q, K, nprobes, scan=idx.SCAN_F32, stream=stream)
        e1.record(stream)
        torch.cuda.synchronize()
        launches, ms = ctx.profile_read("flat_probe")
        ctx.profile_enable(False)
        call_ms = e0.elapsed_time(e1) / reps
        got = order_h[ids.cpu().numpy().view(np.uint32)[:gt_ids.shape[0]].astype(np.int64)]  # back to corpus row ids
        rec = float(np.mean([len(set(got[i]) & set(gt_ids[i])) / K for i in range(gt_ids.shape[0])]))
        res[f"nprobes_{nprobes}"] = {"call_ms": call_ms, "qps": Q_BATCH / (call_ms * 1e-3),
                                     "scan_kernel_ms": ms / max(launches, 1), "recall_at_10": rec}
    idx.close()
    return res


def gen_rabitq_codes(lo: int, hi: int, device) -> torch.Tensor:
    """Rows [lo, hi) of a synthetic RaBitQ code matrix (96 B of sign bits + f32 norm per row), generated per
    65536-row block with its own seed: identical for every world size."""
    cb = (DIM + 63) // 64 * 8 + 4
    out = torch.empty((hi - lo, cb), dtype=torch.uint8, device=device)
    b = lo // BLOCK
    while b * BLOCK < hi:
        g = torch.Generator(device=device)
        g.manual_seed(SEED_BASE * 8192 + 77 + b)
        blk = torch.randint(0, 256, (BLOCK, cb), dtype=torch.uint8, device=device, generator=g)
        blk[:, cb - 4:] = (torch.rand(BLOCK, device=device, generator=g) * 5 + 25).view(torch.uint8).reshape(BLOCK, 4)
        s_, e_ = max(lo, b * BLOCK), min(hi, (b + 1) * BLOCK)
        out[s_ - lo:e_ - lo] = blk[s_ - b * BLOCK:e_ - b * BLOCK]
        b += 1
    return out


def multi_gpu_legs(vg, ctx, sharded, world, rank, device, stream, comm):
    """BASELINE configs[4] on `world` GPUs: the RaBitQ scan over a row-sharded corpus (strong: 10M x 768 codes
    split `world` ways; weak: 10M per GPU) with ONE all-gather of per-shard top-k, and PQ k-means trained by
    sub-quantizer ranges with ONE all-gather of codebooks.  Every rank takes part; rank 0 reports (max over ranks)."""
    nq = Q_BATCH
    g = torch.Generator(device=device)
    g.manual_seed(SEED_QUERY + 5)
    q = torch.randn((nq, DIM), generator=g, device=device)

    def wall_max(seconds):
        t = torch.tensor([seconds], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def sync():
        dist.barrier()
        torch.cuda.synchronize()

    res = {}
    for mode, total in (("strong", SCAN_ROWS), ("weak", SCAN_ROWS * world)):
        bounds = sharded.partition(total, world)
        lo, hi = bounds[rank], bounds[rank + 1]
        idx = sharded.ShardedRaBitQIndex(ctx, gen_rabitq_codes(lo, hi, device), hi - lo, DIM, bounds, comm=comm)
        for _ in range(2):
            idx.search(q, K, stream=stream)
        sync()
        ctx.profile_read("rabitq_scan_mq")      # nq >= 2: the query-blocked kernel
        ctx.profile_read("comm_all_gather")
        ctx.profile_enable(True)
        steps = 10
        t0 = time.perf_counter()
        for _ in range(steps):
            idx.search(q, K, stream=stream)
        sync()
        dt = wall_max(time.perf_counter() - t0)
        ctx.profile_enable(False)
        launches, scan_ms = ctx.profile_read("rabitq_scan_mq")
        cl, coll_ms = ctx.profile_read("comm_all_gather")
        res[mode] = {"rows_total": total, "rows_per_gpu": hi - lo, "queries_per_step": nq, "ms_per_step": dt / steps * 1e3,
                     "qps": steps * nq / dt, "rank0_scan_ms_per_step": scan_ms / steps,
                     "rank0_all_gather_ms_per_step": (coll_ms / steps) if cl else None,
                     "code_bytes_scanned_per_s_all_gpus": steps * nq * total * 100 / dt}
        idx.index.close()
        del idx
    out = {"rabitq_sharded": {"workload": f"RaBitQ {SCAN_ROWS} x 768 (100 B/row) exhaustive scan, {world} row shards, k={K}",
                              "collective": "vg_comm (ncclAllGather through the C ABI)" if comm is not None else "torch.distributed all_gather_into_tensor",
                              **res}}
    # PQ training by sub-quantizer ranges
    gx = torch.Generator(device=device)
    gx.manual_seed(SEED_BASE + 2)
    x = torch.randn((65536, DIM), generator=gx, device=device)
    pq = vg.ProductQuantizer(ctx, DIM, PQ_M, 256)
    sync()
    t0 = time.perf_counter()
    sharded.train_pq_sharded(pq, x, iters=20, seed=1, stream=stream, comm=comm)
    sync()
    out["pq_train_sharded"] = {"workload": f"ProductQuantizer.Train 65536 x 768, m={PQ_M}, K=256, 20 iterations, "
                                           f"{PQ_M // world} sub-quantizers per GPU + one all-gather of codebooks",
                               "wall_s": wall_max(time.perf_counter() - t0)}
    cb, _, _ = pq.codebooks()
    chk = torch.tensor([int(np.asarray(cb, np.int64).sum())], dtype=torch.int64, device=device)
    lo_, hi_ = chk.clone(), chk.clone()
    dist.all_reduce(lo_, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi_, op=dist.ReduceOp.MAX)
    out["pq_train_sharded"]["codebooks_identical_on_all_ranks"] = bool(lo_.item() == hi_.item())
    pq.close()
    try:
        out["hnsw_pq_replicas"] = hnsw_pq_replicas(vg, ctx, sharded, world, rank, device, stream, wall_max, sync)
    except Exception as e:  # an extra leg: named in the record, not fatal
        out["hnsw_pq_replicas"] = {"error": f"{type(e).__name__}: {e}"}
    return out


REPLICA_ROWS = 200_000 if N_ROWS >= 1_000_000 else 50_000   # rows of the replicated graph (every rank builds the same one: the
                                                             # build is deterministic; fewer in the reduced-size test runs)


def hnsw_pq_replicas(vg, ctx, sharded, world, rank, device, stream, wall_max, sync):
    """The metric's pipeline at N GPUs (graph search = replicas, queries sharded): every rank holds the
    whole index — HNSW graph + PQ codes + fp32 rows of the structured corpus (the one on which the pipeline reaches the recall
    bar; REPLICA_ROWS rows so that `world` builds on one box stay short) — and answers a contiguous slice of the batch
    (sharded.ReplicatedGraphIndex: walk on PQ codes, ef candidates, exact rerank), ONE all-gather of [2][nq/world][k] per search.
    Reports whole-batch queries/s (max over ranks), recall@10 and equality with the single-GPU search of the same batch."""
    n, nq, ef = REPLICA_ROWS, NQ_FLIGHT, 192
    rows = gen_structured(0, n, device, seed=0)
    q = gen_structured(0, nq, device, seed=1)
    idx = vg.Index(ctx, n, DIM)
    idx.set_vectors(rows)
    idx.build_hnsw(m=HNSW_M, ef_construction=HNSW_EFC, max_batch=8192, growth_div=32, stream=stream)
    pq = vg.ProductQuantizer(ctx, DIM, PQ_M, 256)
    pq.train(rows[:65536], iters=20, seed=1, stream=stream)
    idx.set_pq_codes(pq, pq.encode(rows, stream=stream), stream=stream)
    rep = sharded.ReplicatedGraphIndex(idx)
    cand, _ = idx.search_hnsw_pq(q, ef, ef, stream=stream)            # the single-GPU answer to the whole batch
    want_i, want_s = idx.rerank(q, cand, K, stream=stream)
    for _ in range(2):
        got_i, got_s = rep.search(q, K, ef, stream=stream)
    sync()
    steps = 5
    t0 = time.perf_counter()
    for _ in range(steps):
        rep.search(q, K, ef, stream=stream)
    sync()
    dt = wall_max(time.perf_counter() - t0)
    t0 = time.perf_counter()
    for _ in range(steps):
        c_, _ = idx.search_hnsw_pq(q, ef, ef, stream=stream)
        idx.rerank(q, c_, K, stream=stream)
    sync()
    dt1 = wall_max(time.perf_counter() - t0)
    gi, _ = fp64_topk_local(rows, q[:1000], 0, K)
    same = bool(torch.equal(got_i.view(torch.int32), want_i.view(torch.int32)) and
                torch.equal(got_s.view(torch.int32), want_s.view(torch.int32)))
    res = {"workload": f"hnsw walk on PQ codes (ef = {ef}) + exact rerank, top-{K}: structured corpus {n} x {DIM}, {nq} queries per "
                       f"search split over {world} replicas, one all-gather",
           "qps": steps * nq / dt, "ms_per_search": dt / steps * 1e3, "one_replica_whole_batch_qps": steps * nq / dt1,
           "recall_at_10": recall_at_k(got_i.cpu().numpy().view(np.uint32)[:1000], gi.cpu().numpy()),
           "ids_equal_single_gpu": same, "replicas": world,
           "note": "ranks sharing ONE GPU (tests, --backend gloo on a 1-GPU box) split the GPU, not the work: the scaling figure needs "
                   "one GPU per rank"}
    idx.close()
    pq.close()
    return res


def traffic_file():
    """the newest committed PMC summary: profiles/rNN_traffic.json of the highest round"""
    files = sorted((ROOT / "profiles").glob("r[0-9][0-9]_traffic.json"))
    return files[-1] if files else None


def measured_traffic(key: str, algorithmic_bytes: float = None):
    """HBM bytes per launch from the committed PMC passes (profiles/rNN_traffic.json, newest round: rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE of that round's binary at this bench's shapes by tools/collect_pmc.sh, gfx950 correction
    applied).  PMC counters cannot be read from inside this process: the value is a committed measurement, not one
    taken in this run (`roofline.traffic_source` names the file).  For the graph searches the measured traffic per
    algorithmic byte (same 1M-row graph, same 8192 queries) is applied to this launch's algorithmic bytes."""
    try:
        t = json.loads(traffic_file().read_text())[key]
        if algorithmic_bytes is not None:
            return float(t["traffic_per_algorithmic_byte"]) * float(algorithmic_bytes)
        return float(t["traffic_bytes"])
    except Exception:
        return None



def live_traffic(kernel_sub: str, script: str, timeout_s: float = 150.0, script_args=()):
    """HBM-side bytes per launch of `kernel_sub`, MEASURED IN THIS RUN: two child processes (never an exec of this one)
    `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 tools/<script>` — counters in their own passes with
    --kernel-trace only, started from /tmp, as MI355X_MICROARCH.md's HBM section prescribes — at the headline's shape;
    FETCH_SIZE / WRITE_SIZE are KiB and FETCH_SIZE reports half of wide reads on gfx950 (read side x 2).  Returns
    (bytes, detail) or (None, reason): no rocprofv3, this process itself runs under a profiler, a pass failed or timed
    out — the caller then falls back to the committed pass of the same kernel (profiles/rNN_traffic.json)."""
    import csv
    import shutil
    import signal
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    if any(k.startswith(("ROCP", "ROCPROF")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "this process runs under a profiler"
    got = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        with tempfile.TemporaryDirectory(dir="/tmp", prefix="vg_pmc_") as d:
            # the interpreter that runs THIS process (the one that has torch and vecgo_amd), directly after `--`
            cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--",
                   sys.executable, str(ROOT / "tools" / script), *[str(a) for a in script_args]]
            try:
                with open(Path(d) / "child_stderr.txt", "wb") as errf:
                    pr = subprocess.Popen(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL,
                                          stderr=errf, start_new_session=True)
                    try:
                        rc = pr.wait(timeout=timeout_s)
                    except subprocess.TimeoutExpired:
                        os.killpg(pr.pid, signal.SIGKILL)     # the group this call started, nothing else
                        pr.wait()
                        return None, f"{counter} pass timed out after {timeout_s:.0f} s"
                if rc != 0:   # the end of what the child said, so that a broken pass can be diagnosed from the line's source
                    tail = (Path(d) / "child_stderr.txt").read_bytes()[-300:].decode("utf-8", "replace").strip().replace("\n", " | ")
                    return None, f"{counter} pass exited with {rc}: {tail}"
                files = list(Path(d).rglob("*counter_collection.csv"))
                if not files:
                    return None, f"{counter} pass wrote no counter_collection.csv"
                vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(files[0]))
                        if kernel_sub in r["Kernel_Name"] and r["Counter_Name"] == counter]
                if not vals:
                    return None, f"no {counter} rows for {kernel_sub}"
                got[counter] = (sum(vals) / len(vals), len(vals))
            except OSError as e:
                return None, f"{type(e).__name__}: {e}"
    f, w = got["FETCH_SIZE"][0], got["WRITE_SIZE"][0]
    return f * 1024 * 2 + w * 1024, {"fetch_size_kib": f, "write_size_kib": w, "dispatches": got["FETCH_SIZE"][1],
                                     "command": f"rocprofv3 --pmc <counter> --kernel-trace -- python3 tools/{script}" +
                                                "".join(f" {a}" for a in script_args)}


BASELINE_METRIC = "QPS at recall@10≥0.95, 1M×768 HNSW+PQ; PQ-ADC HBM GB/s vs peak"   # BASELINE.json "metric", verbatim
FULL_RECORD = "bench_full.json"
LINE_LIMIT = 9000      # bytes: the driver could not parse r03's 22.7 KB line; r02's 10.8 KB one it could


def _r(x, sig=5):
    """Numbers of the compact line carry 5 significant digits; everything else passes through."""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    return float(f"{x:.{sig}g}")


def _pick(d, *keys):
    return {k: _r(d[k]) for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def compact_line(full: dict) -> dict:
    """The ONE stdout line: the contract's headline keys + `roofline` + `cpu_baseline` + one flat summary object per
    BASELINE config / §8(f) scan ({workload, kernel, kernel_ms, achieved, peak, unit, frac, traffic, cpu_qps,
    ids_equal}).  Frontiers, sweeps, notes and build tables stay in the full record (bench_full.json, copied to
    profiles/): a pure function of that record, so a committed record can be re-compacted by the CPU tests."""
    out = {k: _r(full[k]) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                    "scaling", "vs_baseline", "dtype", "data") if k in full}
    out["config"] = full.get("config", {})
    out.update(_pick(full, "recall_at_10", "recall_queries"))
    rf = full.get("roofline") or {}
    out["roofline"] = {**_pick(rf, "bound", "achieved", "peak", "unit", "frac"), "traffic": _r(rf.get("traffic")),
                       **_pick(rf, "kernel", "kernel_ms", "launches")}  # (traffic_source: the full record)
    cb = full.get("cpu_baseline") or {}
    if "value" in cb:
        out["cpu_baseline"] = _pick(cb, "value", "unit", "cores", "kind", "sample", "logical_cpus", "usable_cpus")
        if isinstance(out["cpu_baseline"].get("sample"), str):  # the whole sentence (and the CPU's name) stay in the full record
            out["cpu_baseline"]["sample"] = out["cpu_baseline"]["sample"][:110]
        out["gpu_over_cpu"] = _r(full["value"] / cb["value"])
    elif cb:
        out["cpu_baseline"] = cb                      # {"error": ...}
    out["exchange"] = full.get("exchange")
    for k in ("scaling_curve", "reduced_sizes", "multi_gpu_legs_error"):
        if k in full:
            out[k] = full[k]

    cfgs = []

    def row(name, src, **extra):
        if not isinstance(src, dict):
            return
        if "error" in src:
            cfgs.append({"config": name, "error": str(src["error"])[:160]})
            return
        r = {"config": name, **_pick(src, "workload", "kernel", "kernel_ms", "bound", "achieved", "peak", "unit", "frac", "traffic"),
             **{k: _r(v) for k, v in extra.items() if v is not None}}
        for k in ("workload", "kernel"):   # (the whole sentences stay in the full record: the line has a size budget)
            if isinstance(r.get(k), str) and len(r[k]) > 64:
                r[k] = r[k][:61] + "..."
        cfgs.append(r)

    # configs[1]: the headline kernel (same numbers as `roofline`, in the per-config shape)
    row("configs[1] flat exact fp32 MFMA GEMM", {**rf, "workload": "flat_exact_l2_1Mx768_top10_nq1024"},
        cpu_qps=cb.get("value"), qps=full.get("value"), traffic_live=("traffic_detail" in rf) or None)
    fs = full.get("flat_small_batch")
    if isinstance(fs, dict) and "q1" in fs:
        for nq in ("q1", "q32"):
            row(f"configs[1] {nq} (HBM-bound batch)", {**fs[nq], "bound": fs.get("bound", "hbm"), "peak": fs["peak"], "unit": fs["unit"],
                                                       "workload": f"flat_exact_l2_1Mx768_top10_n{nq}"}, qps=fs[nq].get("qps"))
    else:
        row("configs[1] small batches", fs)
    h0 = full.get("hnsw_layer0")
    if isinstance(h0, dict):
        f32cpu = ((full.get("hnsw_pq") or {}).get("cpu") or {}).get("sweep") or [{}]
        row("configs[2] hnsw ef=128 layer-0 walk", h0, gather_rate_over_hbm_peak=h0.get("gather_rate_over_hbm_peak"),
            recall_at_10=h0.get("recall_at_10"), cpu_qps=f32cpu[0].get("qps"), gpu_qps=f32cpu[0].get("gpu_qps"),
            ids_equal=f32cpu[0].get("ids_equal_gpu"))
    a = full.get("adc_scan")
    if isinstance(a, dict):
        c = a.get("cpu") or {}
        row("configs[3] pq adc scan", a, qps=a.get("qps_single_query_passes"), cpu_qps=c.get("qps"), ids_equal=c.get("ids_equal_gpu"),
            batch1024_qps=(a.get("batch") or {}).get("qps"), traffic_live=("traffic_detail" in a) or None)
    v = full.get("vamana_pq")
    if isinstance(v, dict):
        row("configs[3] vamana beam, pq node scoring", v, node_scores_per_s=v.get("node_scores_per_s"),
            gathered_gbs=v.get("gathered_gbs"))
    rq = full.get("rabitq_scan")
    if isinstance(rq, dict):
        c = rq.get("cpu") or {}
        row("configs[4] rabitq scan (1 GPU)", rq, qps=rq.get("qps_single_query_passes"), cpu_qps=c.get("qps_at_10M_rows"),
            ids_equal=c.get("ids_equal_gpu_on_the_sample"), batch1024_qps=(rq.get("batch") or {}).get("qps"))
    rs = full.get("rabitq_sharded")
    if isinstance(rs, dict):
        for mode in ("strong", "weak"):
            if mode in rs:
                cfgs.append({"config": f"configs[4] rabitq sharded ({mode})", "collective": rs.get("collective"),
                             **_pick(rs[mode], "rows_total", "rows_per_gpu", "ms_per_step", "qps", "rank0_scan_ms_per_step",
                                     "rank0_all_gather_ms_per_step")})
    hr = full.get("hnsw_pq_replicas")
    if isinstance(hr, dict):
        cfgs.append({"config": "metric pipeline over query-sharded replicas (structured corpus)",
                     **({"error": str(hr["error"])[:160]} if "error" in hr else
                        _pick(hr, "replicas", "qps", "one_replica_whole_batch_qps", "recall_at_10", "ids_equal_single_gpu"))})
    pt = full.get("pq_train_sharded")
    if isinstance(pt, dict):
        cfgs.append({"config": "configs[4] pq kmeans train, sharded by sub-quantizer",
                     **_pick(pt, "wall_s", "codebooks_identical_on_all_ranks")})
    fp = full.get("flat_ivf_probe")
    if isinstance(fp, dict) and "nprobes_8" in fp:  # the partition-probed scan: grouped GEMM nomination + proof per (query, probe)
        p8 = fp["nprobes_8"]
        tf = 2.0 * Q_BATCH * 8 * (N_ROWS / max(N_ROWS // 8192, 1)) * DIM / (p8["scan_kernel_ms"] * 1e-3) / 1e12
        cfgs.append({"config": "flat ivf probe, nprobes 8", "kernel": "flat_gemm_dma_grouped_kernel + verify", "kernel_ms": _r(p8["scan_kernel_ms"]),
                     "bound": "mfma", "achieved": _r(tf), "peak": PEAK_MFMA_F32_TFLOPS, "unit": "TFLOP/s", "frac": _r(tf / PEAK_MFMA_F32_TFLOPS),
                     "qps": _r(p8["qps"]), "nprobes_1_qps": _r(fp["nprobes_1"]["qps"])})
    row("f3 sq8 scan", full.get("sq8_scan"))
    i4 = full.get("int4_scan")
    if isinstance(i4, dict) and "lookup_table_order" in i4:
        row("f3 int4 scan (lookup_table_order)", {**i4, **i4["lookup_table_order"]})  # (batch_order: in the full record)
    else:
        row("f3 int4 scan", i4)
    # the north_star functions the search legs do not time (build_side_legs): bound, fraction, CPU twin, equality
    bs = full.get("build_side")
    if isinstance(bs, dict) and "error" in bs:
        cfgs.append({"config": "build-side legs", "error": str(bs["error"])[:160]})
    elif isinstance(bs, dict):
        for key, name in (("kmeans_assign", "a16 kmeans.TrainKMeans: assignment pass"),
                          ("pq_train_seeding", "a11 pq.Train: k-means++ seeding"), ("pq_train_lloyd", "a11 pq.Train: Lloyd assignment pass"),
                          ("pq_encode", "a12 pq.Encode"), ("pq_build_table", "a13 pq.BuildDistanceTable"),
                          ("rerank", "f1 Segment.Rerank"), ("brute_q1", "a17 hnsw.BruteSearch, 1 query"),
                          ("brute_q256", "a17 hnsw.BruteSearch, 256 queries"),
                          ("flat_filtered", "flat search, a filter per query, 1024 queries"),
                          ("sq8_batch", "f3 sq8 batch, 1024 queries"), ("pq_batch", "pq adc batch, 1024 queries")):
            e = bs.get(key)
            if isinstance(e, dict):
                row(name, {**e, "workload": e.get("short", e.get("workload"))},
                    **{k: e.get(k) for k in ("over_reference_valu", "train_ms", "cpu_ms", "cpu_train_ms", "cpu_qps", "bits_equal")})

    # the metric's NAMED pipeline (HNSW on PQ codes + exact rerank, recall@10 >= 0.95): what it reaches on BASELINE's corpus
    # (the best recall inside the sweep — below the bar) and on the structured extra corpus (the fastest entry at the bar)
    hp = full.get("hnsw_pq")
    if isinstance(hp, dict) and "frontier_f32" in hp:
        bf = max(hp["frontier_f32"], key=lambda e: e["recall_at_10"])
        bp = max(hp["frontier_pq_rerank"], key=lambda e: e["recall_at_10"])
        ct = next((c for c in (hp.get("cpu_pq_rerank") or {}).get("sweep", []) if c.get("ef") == bp["ef"]), {})
        cfgs.append({"config": "metric pipeline: hnsw_pq_rerank, random-normal (BASELINE corpus)",
                     "reaches_recall_bar": bool(bp["recall_at_10"] >= 0.95), "best_recall_at_10": _r(bp["recall_at_10"]),
                     **_pick(bp, "ef", "qps"), "cpu_qps": _r(ct.get("qps")), "ids_equal": ct.get("ids_equal_gpu"),
                     "hnsw_f32_best_recall_at_10": _r(bf["recall_at_10"]), "hnsw_f32_best_ef": bf["ef"], "hnsw_f32_best_qps": _r(bf["qps"]),
                     "exact_qps": _r(hp["exact_path"]["qps"]),
                     "operating_point": hp["operating_point"]["path"]})
    sc = full.get("structured_corpus")
    if isinstance(sc, dict):
        name = "metric pipeline: hnsw_pq_rerank, structured (extra corpus, not a BASELINE config)"
        if "error" in sc:
            cfgs.append({"config": name, "error": str(sc["error"])[:160]})
        else:
            at = sc.get("at_recall_0_95") or {}
            b, cp = at.get("hnsw_pq_rerank") or {}, at.get("cpu_hnsw_pq_rerank") or {}
            cfgs.append({"config": name, "reaches_recall_bar": bool(b), **_pick(b, "ef", "recall_at_10", "qps"),
                         "cpu_qps": _r(cp.get("qps")), "cpu_ef": cp.get("ef"), "ids_equal": cp.get("ids_equal_gpu"),
                         **{"hnsw_f32_" + k_: v_ for k_, v_ in _pick(at.get("hnsw_f32") or {}, "ef", "recall_at_10", "qps").items()},
                         "cpu_hnsw_f32_qps": _r((at.get("cpu_hnsw_f32") or {}).get("qps")), "exact_qps": _r(at.get("exact_qps"))})
    out["configs"] = cfgs
    out["full_record"] = FULL_RECORD
    return out


def emit(full: dict):
    """Full record -> bench_full.json (+ gpurun_out/ when present); compact line -> stdout, ONE line."""
    text = json.dumps(full)
    for d in (ROOT, ROOT / "gpurun_out"):
        try:
            if d.is_dir():
                (d / FULL_RECORD).write_text(text + "\n")
        except OSError:
            pass
    print(f"bench.py: full record ({len(text)} bytes) -> {FULL_RECORD}", file=sys.stderr, flush=True)
    line = json.dumps(compact_line(full), ensure_ascii=True, separators=(",", ":"))
    assert "\n" not in line
    if _REAL_STDOUT_FD is not None:  # fd 1 itself points at stderr by now (guard_stdout)
        sys.stdout.flush()
        os.write(_REAL_STDOUT_FD, (line + "\n").encode("ascii"))
    else:
        print(line, flush=True)


_REAL_STDOUT_FD = None


def guard_stdout():
    """The contract is ONE JSON line on stdout.  Libraries write there too — Gloo's "[Gloo] Rank 1 is connected to 7 peer
    ranks" (seen with --backend gloo), RCCL's version banner, a stray print in a dependency — and every rank of a
    torch.distributed.run job shares the launcher's stdout.  So a rank process keeps a private copy of the real stdout
    for the line and points descriptor 1 (C stdio, C++ iostreams, Python's sys.stdout) at stderr for everything else."""
    global _REAL_STDOUT_FD
    if _REAL_STDOUT_FD is None:
        sys.stdout.flush()
        _REAL_STDOUT_FD = os.dup(1)
        os.dup2(2, 1)


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as a FRESH child (torch.distributed.run, one
    process per GPU) and relay its exit code — rank 0 of the child prints the line on the inherited stdout.  Called
    before this process has touched the GPU (torch is imported, no device call made: device_count() does not
    initialise HIP on this image); never an exec, never a restart of a process that owns a device."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if args.backend == "nccl" and have < args.gpus:
        print(f"bench.py: --gpus {args.gpus} but only {have} HIP device(s) visible; RCCL needs one device per rank "
              f"(--backend gloo lets ranks share a device for a smoke test)", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    print("bench.py: no WORLD_SIZE in the environment, launching " + " ".join(cmd), file=sys.stderr, flush=True)
    return subprocess.run(cmd).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-adc", action="store_true")
    ap.add_argument("--no-hnsw", action="store_true")
    ap.add_argument("--no-build-side", action="store_true", help="skip the k-means / PQ train / encode / LUT / rerank / brute legs")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not run the two rocprofv3 --pmc child passes that measure roofline.traffic in this run")
    ap.add_argument("--torch-collective", action="store_true", help="N > 1: exchange through torch.distributed instead of vg_comm")
    ap.add_argument("--bf16-filter", action="store_true",
                    help="time the exact path WITH vg_index_enable_bf16_filter as the headline (default: the fp32 MFMA GEMM "
                         "BASELINE's configs[1] names; the filtered path is reported as `flat_exact_bf16_filter` either way)")
    ap.add_argument("--backend", default="nccl",
                    help="torch.distributed backend; 'gloo' lets several ranks share one GPU to smoke-test the N>1 path")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    guard_stdout()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a line for a world it did not run")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    if args.backend != "nccl":
        local_rank %= torch.cuda.device_count()  # debugging only: ranks may share a device
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.backend)

    import vecgo_amd as vg
    from vecgo_amd import sharded

    ctx = vg.Context(local_rank)
    stream = torch.cuda.current_stream()
    bounds = sharded.partition(N_ROWS, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    rows = gen_rows(lo, hi, device)
    # the exchange step through the C ABI (vg_comm: direct ncclAllGather) when RCCL can be joined that way;
    # torch.distributed's all-gather otherwise
    comm_report = {"collective": "none (1 GPU)" if world == 1 else "torch.distributed", "fell_back": False,
                   "reason": "--torch-collective" if (world > 1 and args.torch_collective) else None}
    comm = sharded.make_comm(ctx, report=comm_report) if (world > 1 and not args.torch_collective) else None
    index = sharded.ShardedFlatIndex(ctx, rows, DIM, bounds, metric=0, comm=comm)
    if args.bf16_filter:
        index.index.enable_bf16_filter(True)
    if comm is not None:   # untimed cross-check of the two exchange paths on one batch
        a_ids, a_sc = index.search(gen_queries(1, device)[0][:64], K, stream=stream)
        index.comm = None
        b_ids, b_sc = index.search(gen_queries(1, device)[0][:64], K, stream=stream)
        same = torch.tensor([int(torch.equal(a_ids, b_ids) and torch.equal(a_sc, b_sc))], dtype=torch.int32, device=device)
        dist.all_reduce(same, op=dist.ReduceOp.MIN)
        if int(same.item()) == 1:
            index.comm = comm
        else:
            comm = None
            comm_report.update({"collective": "torch.distributed", "fell_back": True,
                                "reason": "vg_comm and torch.distributed disagreed on a cross-check batch"})
    n_batches = 8
    queries = gen_queries(n_batches, device)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def flat_step(i):
        return index.search(queries[i % n_batches], K, stream=stream)[0]

    # ---- which pipeline reaches recall@10 >= 0.95 fastest?  (N = 1: measured; N > 1: the exact path, sharded)
    hp = hnsw128 = hidx = hpq = None
    gt1024 = None
    op = {"path": "flat_exact"}
    if world == 1 and not args.no_hnsw:
        for i in range(3):
            flat_step(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for i in range(3):
            flat_step(i)
        e1.record(stream)
        torch.cuda.synchronize()
        exact_ms = e0.elapsed_time(e1) / 3
        gi, _ = fp64_topk_local(rows, queries[0], lo, K)   # independent ground truth: fp64 brute force
        gt1024 = gi.cpu().numpy()
        hp, hnsw128, hidx, hpq = hnsw_pq_frontier(vg, ctx, rows, queries, gt1024, exact_ms, stream,
                                                  with_cpu=not args.no_cpu_baseline)
        op = hp["operating_point"]

    if op["path"] == "hnsw_f32":
        def step(i):
            return hidx.search_hnsw(queries[i % n_batches], K, op["ef"], stream=stream)[0]
    elif op["path"] == "hnsw_pq_rerank":
        def step(i):
            cand, _ = hidx.search_hnsw_pq(queries[i % n_batches], op["ef"], op["ef"], stream=stream)
            return hidx.rerank(queries[i % n_batches], cand, K, stream=stream)[0]
    else:
        step = flat_step
    prof_key = "flat_gemm" if op["path"] == "flat_exact" else ("hnsw_search" if op["path"] == "hnsw_f32" else "hnsw_search_pq")

    for i in range(args.warmup):
        step(i)
    barrier()
    ctx.profile_read(prof_key)
    ctx.profile_enable(True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        res = step(i)
    barrier()
    dt = time.perf_counter() - t0
    ctx.profile_enable(False)
    launches, kern_ms_total = ctx.profile_read(prof_key)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    qps = args.steps * Q_BATCH / dt

    # ---- recall@10 of the timed pipeline against the fp64 ground truth (checker, untimed) --------
    nrec = Q_BATCH if gt1024 is not None else 64
    qrec = queries[0][:nrec]
    got_ids = step(0)[:nrec] if nrec == Q_BATCH else index.search(qrec, K, stream=stream)[0]
    if gt1024 is not None:
        gt = gt1024
    else:
        gi, gs = fp64_topk_local(rows, qrec, lo, K)
        if world > 1:
            gl_i = [torch.empty_like(gi) for _ in range(world)]
            gl_s = [torch.empty_like(gs) for _ in range(world)]
            dist.all_gather(gl_i, gi)
            dist.all_gather(gl_s, gs)
            ci, cs = torch.cat(gl_i, 1), torch.cat(gl_s, 1)
            top = torch.topk(cs, K, dim=1, largest=False)
            gi = torch.gather(ci, 1, top.indices)
        gt = gi.cpu().numpy()
    got = got_ids.cpu().numpy().view(np.uint32).astype(np.int64)
    recall = recall_at_k(got, gt)

    extra = {}
    if world > 1:   # BASELINE configs[4]: every rank takes part
        try:
            extra = multi_gpu_legs(vg, ctx, sharded, world, rank, device, stream, comm)
        except Exception as e:   # an extra must not cost the run its headline line (the same error on every rank
            extra = {"multi_gpu_legs_error": f"{type(e).__name__}: {e}"}   # leaves the ranks in step)

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    rows_local = hi - lo
    kern_avg_ms = kern_ms_total / max(launches, 1)
    if op["path"] == "flat_exact":
        # a step's query batch runs as ceil(Q / chunk) GEMM launches (the score matrix of one launch
        # is capped at 2 GiB): algorithmic flops per launch = 2 * (queries in the launch) * rows * dim,
        # averaged over the launches of the timed region
        flops_per_launch = 2.0 * Q_BATCH * rows_local * DIM * args.steps / max(launches, 1)
        achieved_tf = flops_per_launch / (kern_avg_ms * 1e-3) / 1e12 if launches else 0.0
        peak_tf = PEAK_MFMA_BF16_TFLOPS if args.bf16_filter else PEAK_MFMA_F32_TFLOPS
        roofline = {"bound": "mfma", "achieved": achieved_tf, "peak": peak_tf,
                    "unit": "TFLOP/s", "frac": achieved_tf / peak_tf,
                    "traffic": measured_traffic("flat_gemm") if (world == 1 and not args.bf16_filter) else None,
                    "kernel": "flat_gemm_bf16_big_kernel<false,3>" if args.bf16_filter else "flat_gemm_dma_kernel<false,2>",
                    "kernel_ms": kern_avg_ms, "launches": launches, "flops_per_launch": flops_per_launch,
                    "traffic_source": (traffic_file().name + " (committed rocprofv3 --pmc pass of the same kernel and shape; "
                                       "not re-measured in this run)") if traffic_file() else None}
        workload = "flat_exact_l2_1Mx768_top10 (BASELINE configs[1]): MFMA GEMM nomination + exact re-score + proof" + \
            (" [nomination in bfloat16: --bf16-filter]" if args.bf16_filter else "")
    else:
        per_q = op["distance_computations_per_query"] * DIM * 4 if op["path"] == "hnsw_f32" else op["pq_scores_per_query"] * PQ_M
        bytes_per_launch = per_q * Q_BATCH
        ach = bytes_per_launch / (kern_avg_ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBS,
                    "traffic": None, "kernel": "hnsw_search_kernel", "kernel_ms": kern_avg_ms, "launches": launches,
                    "bytes_per_launch": bytes_per_launch}
        workload = f"{op['path']} ef={op['ef']} over the built HNSW graph, 1M x 768, top-10"
    out = {
        "metric": BASELINE_METRIC,
        "value": qps, "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None,
        "dtype": "f32" if not args.bf16_filter else "f32 results; nomination GEMM in bf16 (--bf16-filter)", "data": "synthetic",
        "config": {"workload": workload, "operating_point": op["path"],
                   # BASELINE's metric names HNSW+PQ; WHICH pipeline is timed is measured, not assumed
                   "operating_point_note": "fastest of {HNSW fp32, HNSW on PQ codes + exact rerank, exact fp32 MFMA GEMM (configs[1])} "
                                           "with recall@10 >= 0.95 vs fp64 ground truth on this corpus, from the frontier measured "
                                           "in this run (bench_full.json: hnsw_pq)" if world == 1 and not args.no_hnsw else
                                           "exact fp32 path (configs[1]); no frontier measured in this mode",
                   "rows": N_ROWS,
                   "dim": DIM, "k": K, "queries_per_step": Q_BATCH,
                   "parallelism": (f"row-shard x{world}, one all-gather of per-shard top-k per step ("
                                   + ("vg_comm: ncclAllGather through the C ABI" if comm is not None else "torch.distributed") + ")")
                   if world > 1 else "1 GPU"},
        "recall_at_10": recall, "recall_queries": int(min(got.shape[0], gt.shape[0])),
        "roofline": roofline,
        # which exchange ran, witnessed by RCCL itself (ncclCommCount), and why if it is not the C-ABI one
        "exchange": {**comm_report, "world_size": world, "torch_backend": args.backend if world > 1 else None},
    }
    if world > 1:
        out["scaling_curve"] = "not measured here: one line per N; the driver computes efficiency from its own N = 1, 2, 4, 8 runs"
    if REDUCED:
        out["reduced_sizes"] = True
        out["config"]["rows"] = N_ROWS
        out["config"]["workload"] += f" [REDUCED smoke-test sizes: {N_ROWS} rows, scans {SCAN_ROWS} rows — not BASELINE's config]"
    out.update(extra)

    def leg(name, fn):   # the extras run after the headline was timed: a failing one is named in the line, not fatal
        try:
            out[name] = fn()
        except Exception as e:
            out[name] = {"error": f"{type(e).__name__}: {e}"}

    if hp is not None:
        # does building in batches cost recall?  (tools/build_check.py on a GPU box: sequential hnsw.Insert, max_batch = 1,
        # against this bench's 8192 / 32 setting on the same rows; committed tables)
        try:
            hp["build_check"] = {"source": "tools/build_check.py, profiles/r03_build_check_*.json",
                                 **{f.stem.replace("r03_build_check_", "rows_"): json.loads(f.read_text())
                                    for f in sorted((ROOT / "profiles").glob("r03_build_check_*.json"))}}
        except Exception as e:
            hp["build_check"] = {"error": str(e)}
        out["hnsw_pq"] = hp
        out["hnsw_layer0"] = hnsw128
        leg("vamana_pq", lambda: vamana_pq(vg, ctx, hidx, rows, queries.reshape(-1, DIM)[:NQ_FLIGHT].contiguous(), gt1024, stream))
        hidx.close()
        hpq.close()
    if world == 1:
        if args.bf16_filter:
            index.index.enable_bf16_filter(False)
        leg("flat_exact_bf16_filter", lambda: flat_bf16_filter(vg, ctx, index.index, queries, gt, args.steps, stream))
        if "qps" in out["flat_exact_bf16_filter"]:
            out["flat_exact_bf16_filter"]["over_fp32_headline"] = out["flat_exact_bf16_filter"]["qps"] / qps
        leg("flat_small_batch", lambda: flat_small_batch(vg, ctx, index.index, queries[2], stream))
    if world == 1 and not args.no_hnsw:
        leg("flat_ivf_probe", lambda: flat_ivf_probe(vg, ctx, rows, queries, gt[:64], stream))
        leg("structured_corpus", lambda: structured_corpus(vg, ctx, stream, device, with_cpu=not args.no_cpu_baseline))
    cpu_on = world == 1 and not args.no_cpu_baseline
    rows_host = rows.cpu().numpy() if cpu_on else None
    if world == 1 and not args.no_build_side:
        leg("build_side", lambda: build_side_legs(vg, ctx, rows, queries, stream, cpu_on, rows_host))
    if world == 1 and not args.no_adc:
        del index
        leg("adc_scan", lambda: adc_scan_roofline(vg, ctx, stream, device, with_cpu=cpu_on))
        leg("rabitq_scan", lambda: rabitq_scan_roofline(vg, ctx, stream, device, with_cpu=cpu_on))
        leg("sq8_scan", lambda: sq8_scan_roofline(vg, ctx, stream, device))
        leg("int4_scan", lambda: int4_scan_roofline(vg, ctx, stream, device))
    if cpu_on:
        leg("cpu_baseline", lambda: cpu_baseline(rows_host, queries[1].cpu().numpy(), K))
        if "value" in out["cpu_baseline"]:
            out["gpu_over_cpu_at_recall_bar"] = qps / out["cpu_baseline"]["value"]
    # roofline.traffic measured in THIS run (after everything else: a profiler pass that went wrong cannot touch a timing)
    if world == 1 and op["path"] == "flat_exact" and not args.bf16_filter and not args.no_live_traffic and not REDUCED:
        tb, detail = live_traffic("flat_gemm_dma_kernel<false, 2, 0, false>", "flat_time.py")
        if tb is not None:
            out["roofline"]["traffic_committed"] = out["roofline"]["traffic"]
            out["roofline"]["traffic"] = tb
            out["roofline"]["traffic_source"] = "measured in this run: " + detail["command"] + " (two child passes, FETCH_SIZE x 2 + WRITE_SIZE, KiB)"
            out["roofline"]["traffic_detail"] = detail
        else:
            out["roofline"]["traffic_source"] = (out["roofline"].get("traffic_source") or "none") + f"; live pass skipped: {detail}"
        # ... and for the kernel the metric's second half names (PQ-ADC scan, configs[3]), same procedure
        if tb is not None and isinstance(out.get("adc_scan"), dict) and "kernel_ms" in out["adc_scan"]:
            ab, adetail = live_traffic("pq_adc_scan_kernel<6, true, true>", "adc_prof.py", script_args=(SCAN_ROWS, 1, K))
            if ab is not None:
                out["adc_scan"]["traffic_committed"] = out["adc_scan"].get("traffic")
                out["adc_scan"]["traffic"] = ab
                out["adc_scan"]["traffic_source"] = "measured in this run: " + adetail["command"]
                out["adc_scan"]["traffic_detail"] = adetail
            else:
                out["adc_scan"]["traffic_source"] = f"committed pass ({traffic_file().name if traffic_file() else None}); live pass skipped: {adetail}"
    emit(out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
