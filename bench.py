#!/usr/bin/env python3
"""bench.py — the hot path's headline benchmark on MI355X.

Workload (BASELINE.json configs[1], the configuration the metric's QPS is quoted on that fits
one GPU): exact brute-force L2 top-10 over 1M x 768 fp32 rows for a batch of 1024 queries per
step — fp32 MFMA GEMM candidate generation, exact re-score in the reference's AVX-512 summation
order, proof-or-fallback (vecgo_amd/csrc/k_flat.hip).  Recall@10 = 1.0 by construction and is
re-measured here against an independent fp64 ground truth.

N > 1: the 1M-row corpus is sharded by rows over the ranks (strong scaling), every rank scores
the same query batch against its shard, ONE all-gather of per-shard top-k (RCCL over xGMI),
merge with the reference tie-break (vecgo_amd/sharded.py).

One JSON line on rank 0; extra objects: `roofline` (dominant kernel of the timed region, HIP
events from inside the library), `adc_scan` (PQ-ADC scan, BASELINE configs[3]: 10M x 96 B codes,
HBM roofline), `rabitq_scan` (RaBitQ scan, configs[4] shape on one GPU: 10M x 100 B), `sq8_scan`,
`flat_small_batch` (configs[1] below the MFMA regime),
`hnsw_layer0` (configs[2]), `flat_ivf_probe` (the partition-probed flat search the reference runs on
compacted segments), `cpu_baseline` (the CPU oracle = port of the reference's AVX-512 path, timed on
this host's cores on a bounded sample).
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
SEED_BASE, SEED_QUERY = 20260130, 20260131
BLOCK = 65536  # rows per generation block: data is identical for every world size
PEAK_MFMA_F32_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32-input MFMA dense peak
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec peak


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


def cpu_baseline(rows_host: np.ndarray, queries_host: np.ndarray, k: int, budget_s: float = 15.0):
    """The reference's CPU path on this host, one query per thread (the reference's concurrency model:
    one goroutine per query), for about `budget_s` seconds of wall time.
    kind "reference": the distances come from the reference's own AVX-512 kernel
    (internal/simd/src/batch_avx512.c squaredL2BatchAvx512, compiled in place into oracle/_ref — the
    per-row arithmetic of the squaredL2Avx512 calls flat/segment.go:691-701 makes) in chunks of 8192
    rows, the top-k from a partial sort of each chunk.  kind "port": the CPU restatement
    (oracle/vg_oracle.c, pinned bit-for-bit to those kernels) when oracle/_ref is absent or the host
    has no AVX-512."""
    from oracle import oracle as o
    cores = os.cpu_count() or 1
    nthreads = min(cores, queries_host.shape[0])
    ref = o.Ref()
    n = rows_host.shape[0]
    done = [0] * nthreads
    deadline = [0.0]
    chunk = 8192

    def one_query_ref(q):
        best_d = np.full(k, np.inf, np.float32)
        best_i = np.full(k, -1, np.int64)
        out = np.empty(chunk, np.float32)
        for s0 in range(0, n, chunk):
            m = min(chunk, n - s0)
            ref.lib.squaredL2BatchAvx512(q.ctypes.data, rows_host[s0:s0 + m].ctypes.data, DIM, m, out.ctypes.data)
            d = out[:m]
            if m > k:
                part = np.argpartition(d, k)[:k]
            else:
                part = np.arange(m)
            cd = np.concatenate([best_d, d[part]])
            ci = np.concatenate([best_i, part + s0])
            sel = np.argsort(cd, kind="stable")[:k]
            best_d, best_i = cd[sel], ci[sel]
        return best_i

    def work(t):
        i = t
        while True:
            q = queries_host[i % queries_host.shape[0]]
            if ref.ok:
                one_query_ref(q)
            else:
                o.flat_search_f32(rows_host, DIM, q, k)
            done[t] += 1
            i += nthreads
            if time.time() >= deadline[0]:
                return

    th = [threading.Thread(target=work, args=(t,)) for t in range(nthreads)]
    t0 = time.time()
    deadline[0] = t0 + budget_s
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.time() - t0
    nq = sum(done)
    kind = "reference" if ref.ok else "port"
    how = ("squaredL2BatchAvx512 of the reference (oracle/_ref) over 8192-row chunks + partial sort" if ref.ok
           else "oracle/vg_oracle.c restatement")
    return {"value": nq / dt, "unit": "queries/s", "cores": nthreads, "kind": kind,
            "sample": f"{nq} queries x {n} rows x {DIM} fp32, exact L2 top-{k}, {nthreads} threads "
                      f"(1 query/thread), {dt:.1f} s; {how}"}


def adc_scan_roofline(vg, ctx, stream, device):
    """BASELINE configs[3]: PQ (m=96, K=256) ADC scan over 10M codes, one query per pass:
    algorithmic bytes = N*m per launch."""
    n, m = 10_000_000, 96
    g = torch.Generator(device=device)
    g.manual_seed(7)
    codes = torch.randint(0, 256, (n, m), dtype=torch.uint8, device=device, generator=g)
    rng = np.random.default_rng(0)
    pq = vg.ProductQuantizer(ctx, DIM, m, 256)
    pq.set_codebooks(rng.integers(-128, 128, m * 256 * (DIM // m)).astype(np.int8),
                     (rng.random(m) * 0.02 + 0.005).astype(np.float32), np.zeros(m, np.float32))
    idx = vg.Index(ctx, n, DIM)
    idx.set_pq_codes(pq, codes)
    del codes
    q = torch.randn((1, DIM), device=device)
    out = (torch.empty((1, K), dtype=torch.int32, device=device), torch.empty((1, K), device=device))
    for _ in range(5):
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
    idx.close()
    pq.close()
    return res


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
        for _ in range(3):
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


def rabitq_scan_roofline(vg, ctx, stream, device):
    """BASELINE configs[4], one GPU's view: exhaustive RaBitQ scan, 10M x 768 -> 100 B per row
    (96 B of sign bits + f32 norm): algorithmic bytes = N*100 per launch."""
    n = 10_000_000
    cb = (DIM + 63) // 64 * 8 + 4
    g = torch.Generator(device=device)
    g.manual_seed(11)
    codes = torch.randint(0, 256, (n, cb), dtype=torch.uint8, device=device, generator=g)
    codes[:, cb - 4:] = (torch.rand(n, device=device, generator=g) * 5 + 25).view(torch.uint8).reshape(n, 4)
    idx = vg.Index(ctx, n, DIM)
    idx.set_rabitq_codes(codes)
    del codes
    q = torch.randn((1, DIM), device=device)
    out = (torch.empty((1, K), dtype=torch.int32, device=device), torch.empty((1, K), device=device))
    for _ in range(5):
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
    for _ in range(3):
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
           "unit": "GB/s", "frac": achieved / PEAK_HBM_GBS, "traffic": None, "kernel": "sq8_scan_kernel",
           "kernel_ms": kern_ms, "bytes_per_launch": n * DIM, "search_call_ms": e0.elapsed_time(e1) / reps}
    idx.close()
    sq.close()
    return res


def hnsw_layer0(vg, ctx, rows, queries, gt_ids, stream):
    """BASELINE configs[2]: HNSW ef=128 layer-0 search, 1M x 768 L2.  Graph CONSTRUCTION is out of
    scope (and non-deterministic in the reference), so the graph is the exact 31-NN graph of the
    corpus, built here with this library's own flat search (untimed).  On i.i.d. normal data in
    768 dimensions such a graph is barely navigable: the recall printed next to the QPS is what
    the reference's algorithm reaches on it, not a kernel property."""
    n, deg, ef = rows.shape[0], 32, 128
    idx = vg.Index(ctx, n, DIM)
    idx.set_vectors(rows)
    l0 = torch.empty((n, deg), dtype=torch.int32, device=rows.device)
    sc = torch.empty((4096, deg), device=rows.device)
    t0 = time.time()
    for s in range(0, n, 4096):
        e = min(n, s + 4096)
        idx.search_flat(rows[s:e], deg, out=(l0[s:e], sc[:e - s]), stream=stream)
    torch.cuda.synchronize()
    build_s = time.time() - t0
    l0 = l0.cpu().numpy().view(np.uint32)
    l0 = np.where(l0 == np.arange(n, dtype=np.uint32)[:, None], np.uint32(0xFFFFFFFF), l0)
    l0 = np.take_along_axis(l0, np.argsort(l0 == 0xFFFFFFFF, axis=1, kind="stable"), axis=1)
    idx.set_hnsw_graph(l0, (), entry_point=0, m=16)
    q = queries.reshape(-1, DIM)[:8192]
    ids, _, st = idx.search_hnsw(q, K, ef, stats=True, stream=stream)
    torch.cuda.synchronize()
    ctx.profile_read("hnsw_search")
    ctx.profile_enable(True)
    reps = 5
    for _ in range(reps):
        idx.search_hnsw(q, K, ef, stream=stream)
    torch.cuda.synchronize()
    launches, ms = ctx.profile_read("hnsw_search")
    ctx.profile_enable(False)
    kern_ms = ms / max(launches, 1)
    got = ids.cpu().numpy().view(np.uint32)[:gt_ids.shape[0]]
    rec = float(np.mean([len(set(got[i]) & set(gt_ids[i])) / K for i in range(gt_ids.shape[0])]))
    dc = float(st[:, 1].sum())
    gathered = dc * DIM * 4 + float(st[:, 3].sum()) * deg * 4
    vam = vamana_pq(vg, ctx, idx, rows, l0, q, gt_ids, stream)
    idx.close()
    return {"vamana_pq": vam,
            "workload": "hnsw_layer0_1Mx768_ef128_k10 on the exact 31-NN graph, 8192 queries in flight",
            "bound": "hbm", "achieved": gathered / (kern_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": gathered / (kern_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, "traffic": None,
            "kernel": "hnsw_search_kernel", "kernel_ms": kern_ms, "qps": q.shape[0] / (kern_ms * 1e-3),
            "recall_at_10": rec, "distance_computations_per_query": dc / q.shape[0],
            "bytes_per_launch": gathered, "graph_build_s": build_s}


def vamana_pq(vg, ctx, idx, rows, graph, q, gt_ids, stream):
    """BASELINE configs[3], graph half: Vamana beam search (diskann/segment.go:503-706) over the
    same 31-NN graph with PQ (m=96, K=256) node scoring = ComputeAsymmetricDistance per visited
    node (no LUT, :536-557).  PQ trained on the GPU on a 32768-row sample."""
    pq = vg.ProductQuantizer(ctx, DIM, 96, 256)
    t0 = time.perf_counter()
    pq.train(rows[:32768], iters=10, seed=1)
    codes = pq.encode(rows)
    torch.cuda.synchronize()
    prep_s = time.perf_counter() - t0
    idx.set_pq_codes(pq, codes)
    idx.set_vamana_graph(graph, 0)
    ids, _, st = idx.search_vamana(q, K, kind=1, stats=True, stream=stream)
    torch.cuda.synchronize()
    ctx.profile_read("vamana_search")
    ctx.profile_enable(True)
    reps = 3
    for _ in range(reps):
        idx.search_vamana(q, K, kind=1, stream=stream)
    torch.cuda.synchronize()
    launches, ms = ctx.profile_read("vamana_search")
    ctx.profile_enable(False)
    kern_ms = ms / max(launches, 1)
    got = ids.cpu().numpy().view(np.uint32)[:gt_ids.shape[0]]
    rec = float(np.mean([len(set(got[i]) & set(gt_ids[i])) / K for i in range(gt_ids.shape[0])]))
    dc = float(st[:, 1].sum())
    gathered = dc * 96 + float(st[:, 3].sum()) * graph.shape[1] * 4
    return {"workload": "vamana_pq_1Mx768_m96_K256_k10 on the exact 31-NN graph, 8192 queries in flight",
            "kernel": "vamana_search_kernel", "kernel_ms": kern_ms, "qps": q.shape[0] / (kern_ms * 1e-3),
            "recall_at_10_before_rerank": rec, "distance_computations_per_query": dc / q.shape[0],
            "bytes_per_launch": gathered, "gathered_gbs": gathered / (kern_ms * 1e-3) / 1e9,
            "pq_train_encode_s": prep_s}


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


def measured_traffic(key: str):
    """HBM bytes per launch from the committed PMC passes (profiles/r01_traffic.json: rocprofv3
    --pmc FETCH_SIZE / WRITE_SIZE, gfx950 correction applied).  PMC counters cannot be read from
    inside this process; the file names the exact commands."""
    try:
        t = json.loads((ROOT / "profiles" / "r01_traffic.json").read_text())[key]
        return float(t["traffic_bytes"])
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-adc", action="store_true")
    ap.add_argument("--no-hnsw", action="store_true")
    ap.add_argument("--backend", default="nccl",
                    help="torch.distributed backend; 'gloo' lets several ranks share one GPU to smoke-test the N>1 path")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
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
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    import vecgo_amd as vg
    from vecgo_amd import sharded

    ctx = vg.Context(local_rank)
    stream = torch.cuda.current_stream()
    bounds = sharded.partition(N_ROWS, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    rows = gen_rows(lo, hi, device)
    index = sharded.ShardedFlatIndex(ctx, rows, DIM, bounds, metric=0)
    n_batches = 8
    queries = gen_queries(n_batches, device)

    def step(i):
        return index.search(queries[i % n_batches], K, stream=stream)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    barrier()
    ctx.profile_read("flat_gemm")
    ctx.profile_enable(True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        res = step(i)
    barrier()
    dt = time.perf_counter() - t0
    ctx.profile_enable(False)
    launches, gemm_ms = ctx.profile_read("flat_gemm")
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    qps = args.steps * Q_BATCH / dt

    # ---- recall@10 against an independent fp64 ground truth (checker, untimed) ----------------
    nrec = 64
    qrec = queries[0][:nrec]
    got_ids, _ = index.search(qrec, K, stream=stream)
    gi, gs = fp64_topk_local(rows, qrec, lo, K)
    if world > 1:
        gl_i = [torch.empty_like(gi) for _ in range(world)]
        gl_s = [torch.empty_like(gs) for _ in range(world)]
        dist.all_gather(gl_i, gi)
        dist.all_gather(gl_s, gs)
        ci, cs = torch.cat(gl_i, 1), torch.cat(gl_s, 1)
        top = torch.topk(cs, K, dim=1, largest=False)
        gi = torch.gather(ci, 1, top.indices)
    got = got_ids.cpu().numpy().view(np.uint32).astype(np.int64)
    gt = gi.cpu().numpy()
    recall = float(np.mean([len(set(got[i]) & set(gt[i])) / K for i in range(nrec)]))

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    rows_local = hi - lo
    # a step's query batch runs as ceil(Q / chunk) GEMM launches (the score matrix of one launch
    # is capped at 2 GiB): algorithmic flops per launch = 2 * (queries in the launch) * rows * dim,
    # averaged over the launches of the timed region
    flops_per_launch = 2.0 * Q_BATCH * rows_local * DIM * args.steps / max(launches, 1)
    gemm_avg_ms = gemm_ms / max(launches, 1)
    achieved_tf = flops_per_launch / (gemm_avg_ms * 1e-3) / 1e12 if launches else 0.0
    out = {
        "metric": "QPS at recall@10>=0.95, 1M x 768 (exact brute force, fp32 MFMA GEMM + exact re-score)",
        "value": qps, "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "flat_exact_l2_1Mx768_top10 (BASELINE configs[1])", "rows": N_ROWS,
                   "dim": DIM, "k": K, "queries_per_step": Q_BATCH,
                   "parallelism": f"row-shard x{world}, all-gather of per-shard top-k" if world > 1 else "1 GPU"},
        "recall_at_10": recall,
        "roofline": {"bound": "mfma", "achieved": achieved_tf, "peak": PEAK_MFMA_F32_TFLOPS,
                     "unit": "TFLOP/s", "frac": achieved_tf / PEAK_MFMA_F32_TFLOPS,
                     "traffic": measured_traffic("flat_gemm") if world == 1 else None,
                     "kernel": "flat_gemm_dma_kernel<false,2>", "kernel_ms": gemm_avg_ms,
                     "launches": launches, "flops_per_launch": flops_per_launch},
    }
    if world == 1:
        out["flat_small_batch"] = flat_small_batch(vg, ctx, index.index, queries[2], stream)
    if world == 1 and not args.no_hnsw:
        out["hnsw_layer0"] = hnsw_layer0(vg, ctx, rows, queries, gt[:nrec], stream)
    if world == 1 and not args.no_hnsw:
        out["flat_ivf_probe"] = flat_ivf_probe(vg, ctx, rows, queries, gt[:nrec], stream)
    if world == 1 and not args.no_adc:
        del index
        out["adc_scan"] = adc_scan_roofline(vg, ctx, stream, device)
        out["rabitq_scan"] = rabitq_scan_roofline(vg, ctx, stream, device)
        out["sq8_scan"] = sq8_scan_roofline(vg, ctx, stream, device)
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(rows.cpu().numpy(), queries[1].cpu().numpy(), K)
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
