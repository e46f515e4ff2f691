#!/usr/bin/env python3
"""the current-state table, one row per kernel of the shipped binary, written from the bench's full
JSON (profiles/<round>_bench_full.json) so the table cannot drift from the measured run.

    python tools/design_table.py [--round r06] [--write]

Without --write the table goes to stdout; with it the text between the two markers in DESIGN.md is replaced.
"""
import argparse
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BEGIN, END = "<!-- current-state table: begin (tools/design_table.py) -->", "<!-- current-state table: end -->"

# (path in the bench JSON, row label, shape, the PMC / stats files that back the row)
ROWS = [
    ("roofline", "flat fp32 nomination GEMM (headline)", "1024 q × 1M × 768 fp32, top-10", "pmc_gemm_{fetch,write,mfma}.csv, bench_kernel_stats.csv"),
    ("flat_exact_bf16_filter.roofline", "flat bf16 nomination GEMM (256 × 256 persistent tile)", "1024 q × 1M × 768 bf16", "pmc_gemm16_big_*.csv, gemm_bf16_probe.txt"),
    ("build_side.sq8_batch", "SQ8 batch: bf16 nomination + verify from codes", "1024 q × 1M × 768 u8", "pmc_sq8nom_{fetch,mfma}.csv"),
    ("build_side.pq_batch", "PQ batch: bf16 nomination over decoded rows + ADC re-score", "1024 q × 1M × m96", "bench_kernel_stats.csv"),
    ("flat_small_batch.q1", "flat scan, 1 query", "1 q × 1M × 768 fp32", "bench_kernel_stats.csv"),
    ("flat_small_batch.q32", "flat 32-query GEMM", "32 q × 1M × 768 fp32", "bench_kernel_stats.csv"),
    ("build_side.flat_filtered", "flat GEMM + per-query row filter", "1024 q × 1M × 768, 1/8 kept", "bench_kernel_stats.csv"),
    ("build_side.brute_q1", "hnsw.BruteSearch, 1 query", "1 q × 1M × 768", "bench_kernel_stats.csv"),
    ("build_side.brute_q256", "hnsw.BruteSearch, 256 queries", "256 q × 1M × 768", "bench_kernel_stats.csv"),
    ("hnsw_layer0", "HNSW layer-0 walk, fp32 rows", "8192 q in flight, ef 128, 1M × 768", "pmc_walk_f32_128_*.csv"),
    ("vamana_pq", "Vamana beam, PQ node scoring", "8192 q in flight, 1M × m96", "pmc_walk_vamana_pq_*.csv"),
    ("adc_scan", "PQ ADC scan, 1 query", "10M × 96 B", "pmc_adc_{fetch,write,lds}.csv"),
    ("adc_scan.batch.roofline", "PQ ADC scan, batch (LDS gather rate)", "64 q × 10M × 96 B", "pmc_adcmq_fetch.csv"),
    ("rabitq_scan", "RaBitQ scan, 1 query", "10M × 100 B", "pmc_rq_{fetch,write}.csv"),
    ("rabitq_scan.batch.roofline", "RaBitQ scan, batch", "1024 q × 10M × 100 B", "pmc_rqmq_{fetch,valu}.csv"),
    ("sq8_scan", "SQ8 scan, 1 query", "4M × 768 B", "pmc_sq8_fetch.csv"),
    ("int4_scan.lookup_table_order", "INT4 distance scan", "4M × 384 B", "pmc_i4_*.csv"),
    ("build_side.rerank", "Rerank + top-10", "8192 q × 512 cand × 768", "bench_kernel_stats.csv"),
    ("build_side.kmeans_assign", "k-means assignment pass", "1M × 768 × k 122", "pmc_kmeans_*.csv"),
    ("build_side.pq_train_seeding", "PQ k-means++ seeding", "65536 × 768, m96 K256", "pmc_pqtrain_*.csv"),
    ("build_side.pq_train_lloyd", "PQ Lloyd assignment pass", "65536 × 768, m96 K256", "pmc_pqtrain_*.csv"),
    ("build_side.pq_encode", "ProductQuantizer.Encode", "1M × 768 → 96 B", "pmc_encode_*.csv"),
    ("build_side.pq_build_table", "BuildDistanceTable", "1024 tables, m96 K256", "bench_kernel_stats.csv"),
]


def get(d, path):
    for p in path.split("."):
        if not isinstance(d, dict) or p not in d:
            return None
        d = d[p]
    return d


def inherit(d, path, key):
    """A field of the row, or of the nearest enclosing object (bound / peak / unit are often stated once per family)."""
    parts = path.split(".")
    while parts:
        o = get(d, ".".join(parts))
        if isinstance(o, dict) and key in o:
            return o[key]
        parts.pop()
    return None


def table(rnd):
    with open(os.path.join(ROOT, "profiles", f"{rnd}_bench_full.json")) as f:
        d = json.load(f)
    out = ["| kernel(s) | what | shape | ms / launch | bound | achieved / peak | fraction | profile (`profiles/%s_…`) |" % rnd,
           "|---|---|---|---:|---|---|---:|---|"]
    for path, label, shape, files in ROWS:
        o = get(d, path)
        if not isinstance(o, dict) or o.get("frac") is None:
            continue
        kern = inherit(d, path, "kernel") or ""
        ms = o.get("kernel_ms")
        ach, peak, unit = o.get("achieved"), inherit(d, path, "peak"), inherit(d, path, "unit")
        out.append("| `%s` | %s | %s | %s | %s | %s | %.2f | %s |" % (
            kern.replace("|", "/"), label, shape, ("%.3f" % ms) if ms is not None else "—", inherit(d, path, "bound"),
            ("%.4g / %.4g %s" % (ach, peak, unit)) if ach is not None and peak is not None else "—", o["frac"], files))
    return "\n".join(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", default="r06")
    ap.add_argument("--write", action="store_true")
    a = ap.parse_args()
    t = table(a.round)
    if not a.write:
        print(t)
        return
    path = os.path.join(ROOT, "DESIGN.md")
    with open(path) as f:
        s = f.read()
    pat = re.compile(re.escape(BEGIN) + r".*?" + re.escape(END), re.S)
    if not pat.search(s):
        raise SystemExit("DESIGN.md has no table markers")
    with open(path, "w") as f:
        f.write(pat.sub(lambda _: BEGIN + "\n" + t + "\n" + END, s))


if __name__ == "__main__":
    main()
