"""gpurun_out/pmc/*.csv (tools/collect_pmc.sh) -> rNN_traffic.json: HBM-side bytes per launch, corrected as
MI355X_MICROARCH.md prescribes (FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reports half of the bytes of
wide reads, so the read side is doubled), next to the launch's algorithmic bytes, plus the
vector-ALU / MFMA counters of the kernels that are not HBM-bound.  argv: directory of the csv files."""
import csv, glob, json, os, re, sys

d = sys.argv[1]
ROUND = os.environ.get("ROUND", "r06")
if not glob.glob(os.path.join(d, "*.csv")):
    sys.exit(f"make_traffic_json: no csv under {d}: every pmc pass failed (see the .log files)")


def rows(name):
    p = os.path.join(d, name + ".csv")
    if not os.path.exists(p):
        return []
    out = []   # kernel names carry commas (template arguments): split from the right
    for line in list(open(p))[1:]:
        parts = line.rstrip("\n").rsplit(",", 3)
        if len(parts) == 4:
            out.append({"kernel": parts[0].strip('"'), "counter": parts[1], "dispatches": parts[2], "mean_per_dispatch": parts[3]})
    return out


_main = {}


def val(name, kernel_sub, counter):
    """mean per dispatch of `counter` for the kernel matching `kernel_sub`.  Several instantiations can match (the HNSW walk has a
    second, near-empty pass for redone queries since r05): the one that did the work — the largest FETCH_SIZE / GRBM_GUI_ACTIVE /
    first counter of the pass — is chosen once per (pass, substring) and used for every counter of that pass."""
    rs = [r for r in rows(name) if kernel_sub in r["kernel"]]
    if not rs:
        return None
    key = (name, kernel_sub)
    if key not in _main:
        weight = {}
        for r in rs:
            if r["counter"] in ("FETCH_SIZE", "WRITE_SIZE", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES"):
                weight[r["kernel"]] = max(weight.get(r["kernel"], 0.0), float(r["mean_per_dispatch"]))
        _main[key] = max(weight, key=weight.get) if weight else rs[0]["kernel"]
    for r in rs:
        if r["kernel"] == _main[key] and r["counter"] == counter:
            return float(r["mean_per_dispatch"])
    return None


def counts(name):
    """the JSON line tools/walk_prof.py printed under the profiler"""
    p = os.path.join(d, name + ".log")
    if not os.path.exists(p):
        return None
    for line in open(p):
        if line.startswith("{") and '"distance_computations"' in line:
            return json.loads(line)
    return None


def traffic(fetch_name, write_name, kernel_sub, algorithmic, extra=None):
    f, w = val(fetch_name, kernel_sub, "FETCH_SIZE"), val(write_name, kernel_sub, "WRITE_SIZE") if write_name else None
    if f is None:
        return None
    tb = f * 1024 * 2 + (w or 0) * 1024
    out = {"kernel": kernel_sub, "fetch_size_kib": f, "write_size_kib": w, "traffic_bytes": tb}
    if algorithmic:
        out["algorithmic_bytes"] = algorithmic
        out["traffic_per_algorithmic_byte"] = tb / algorithmic
    if extra:
        out.update(extra)
    return out


out = {"_comment": "HBM-side traffic per launch from rocprofv3 --pmc passes of THIS round's binary at the bench's own shapes "
                   "(tools/collect_pmc.sh; counters in their own runs, --kernel-trace only), corrected as MI355X_MICROARCH.md "
                   "section HBM prescribes (KiB; read side doubled on gfx950); per-kernel means in profiles/" + ROUND + "_pmc_*.csv"}
out["pq_adc_scan"] = traffic("adc_fetch", "adc_write", "pq_adc_scan_kernel", 10_000_000 * 96, {
    "workload": "10M x 96 B, 1 query",
    "lds": {c: val("adc_lds", "pq_adc_scan_kernel", c) for c in ("SQ_INSTS_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_LDS")}})
if out["pq_adc_scan"] and out["pq_adc_scan"]["lds"]["SQ_LDS_IDX_ACTIVE"]:
    l = out["pq_adc_scan"]["lds"]
    l["bank_conflict_share_of_lds_active_cycles"] = l["SQ_LDS_BANK_CONFLICT"] / l["SQ_LDS_IDX_ACTIVE"]
out["rabitq_scan"] = traffic("rq_fetch", "rq_write", "rabitq_scan_kernel", 10_000_000 * 100, {"workload": "10M x 100 B, 1 query"})
mq = traffic("rqmq_fetch", None, "rabitq_scan_mq_kernel", None, {"workload": "10M x 100 B, 1024 queries in one call"})
if mq:
    valu, gui = val("rqmq_valu", "rabitq_scan_mq_kernel", "SQ_INSTS_VALU"), val("rqmq_valu", "rabitq_scan_mq_kernel", "GRBM_GUI_ACTIVE")
    mq.update({"SQ_INSTS_VALU": valu, "GRBM_GUI_ACTIVE_sum_over_8_xcds": gui,
               "valu_instructions_per_64_row_query_pairs": valu / (10_000_000 * 1024 / 64) if valu else None,
               "valu_busy_fraction": valu * 4 / 1024 / (gui / 8) if valu and gui else None,
               "note": "traffic_bytes = fabric reads of the whole call: the 1 GB of codes once per XCD-resident query block, "
                       "the rest of the 64 query blocks' passes served by L2"})
out["rabitq_scan_mq"] = mq
out["pq_adc_scan_batch64"] = traffic("adcmq_fetch", None, "pq_adc_scan_kernel", None, {"workload": "10M x 96 B, 64 queries in one call (one pass per query, slices shared through L2)"})
out["sq8_scan"] = traffic("sq8_fetch", None, "sq8_scan_kernel", 4_000_000 * 768, {"workload": "4M x 768 one-byte codes, 1 query"})
i4 = traffic("i4_fetch", "i4_write", "int4_scan_tab_kernel<true>", 4_000_000 * 384 + 4_000_000 * 4, {"workload": "4M x 384 B INT4 codes, 1 query (+ 16 MB of distances written)"})
if i4:
    lds = {c: val("i4_lds", "int4_scan_tab_kernel<true>", c) for c in ("SQ_INSTS_VALU", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "GRBM_GUI_ACTIVE")}
    if lds["GRBM_GUI_ACTIVE"]:
        cyc = lds["GRBM_GUI_ACTIVE"] / 8
        lds["lds_array_busy_fraction"] = lds["SQ_LDS_IDX_ACTIVE"] / 256 / cyc
        lds["valu_busy_fraction"] = lds["SQ_INSTS_VALU"] * 4 / 1024 / cyc
    i4["counters"] = lds
    # issue slots (r04, tools/collect_pmc.sh i4_issue): one slot = one SIMD x 4 cycles; ANY = VALU + LDS + scalar + VMEM + misc
    iss = {c: val("i4_issue", "int4_scan_tab_kernel<true>", c) for c in ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
                                                                            "SQ_ACTIVE_INST_SCA", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE")}
    iss2 = {c: val("i4_wait", "int4_scan_tab_kernel<true>", c) for c in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_LDS", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE")}
    if iss.get("GRBM_GUI_ACTIVE") and iss.get("SQ_ACTIVE_INST_ANY"):
        slots = 1024 * (iss["GRBM_GUI_ACTIVE"] / 8) / 4
        iss["issue_slots"] = slots
        iss["issue_slots_used_fraction"] = iss["SQ_ACTIVE_INST_ANY"] / slots
        iss["valu_slot_fraction"] = iss["SQ_ACTIVE_INST_VALU"] / slots
        iss["lds_slot_fraction"] = iss["SQ_ACTIVE_INST_LDS"] / slots
    if iss2.get("SQ_WAVE_CYCLES"):
        iss2["waiting_for_an_issue_slot_fraction_of_wave_cycles"] = iss2["SQ_WAIT_INST_ANY"] / iss2["SQ_WAVE_CYCLES"]
        iss2["waiting_at_waitcnt_fraction_of_wave_cycles"] = iss2["SQ_WAIT_ANY"] / iss2["SQ_WAVE_CYCLES"]
    i4["issue"] = iss
    i4["waits"] = iss2
out["int4_scan"] = i4
g = traffic("gemm_fetch", "gemm_write", "flat_gemm_dma_kernel<false, 2, 0, false>", 1_000_000 * 768 * 4 + 1024 * 768 * 4, {"workload": "1024 queries x 1M x 768 per launch"})
if g:
    busy, gui = val("gemm_mfma", "flat_gemm_dma_kernel<false, 2, 0, false>", "SQ_VALU_MFMA_BUSY_CYCLES"), val("gemm_mfma", "flat_gemm_dma_kernel<false, 2, 0, false>", "GRBM_GUI_ACTIVE")
    g.update({"SQ_VALU_MFMA_BUSY_CYCLES": busy, "GRBM_GUI_ACTIVE_sum_over_8_xcds": gui,
              "mfma_busy_fraction": busy / (1024 * gui / 8) if busy and gui else None})
out["flat_gemm"] = g
for tag, sub, per_score in (("f32_128", "hnsw_search_kernel<0", 768 * 4), ("f32_2048", "hnsw_search_kernel<0", 768 * 4),
                            ("pq_128", "hnsw_search_kernel<2", 96), ("vamana_pq", "vamana_search_kernel<4", 96)):
    c = counts(f"walk_{tag}_fetch")
    alg = None
    if c:
        alg = (c["distance_computations"] + c.get("descent_distance_computations", 0)) * per_score + c["pops"] * 64 * 4
    t = traffic(f"walk_{tag}_fetch", f"walk_{tag}_write", sub, alg, {"workload": f"graph built on 1M x 768 (M0 = 64), 8192 queries per call, {tag}", "counts": c})
    if t:
        valu, gui = val(f"walk_{tag}_valu", sub, "SQ_INSTS_VALU"), val(f"walk_{tag}_valu", sub, "GRBM_GUI_ACTIVE")
        t.update({"SQ_INSTS_VALU": valu, "SQ_INSTS_SALU": val(f"walk_{tag}_valu", sub, "SQ_INSTS_SALU"),
                  "GRBM_GUI_ACTIVE_sum_over_8_xcds": gui, "valu_busy_fraction": valu * 4 / 1024 / (gui / 8) if valu and gui else None})
    out[{"f32_128": "hnsw_search", "f32_2048": "hnsw_search_ef2048", "pq_128": "hnsw_search_pq", "vamana_pq": "vamana_search_pq"}[tag]] = t

# ---- r06: the bfloat16 nomination GEMMs, the build-side matrix kernels -------------------------------------------------------------
def unit_counters(name, sub, keys):
    return {c: val(name, sub, c) for c in keys}


def gemm16(tag, sub):
    t = traffic(f"gemm16_{tag}_fetch", f"gemm16_{tag}_write", sub, 1_000_000 * 768 * 2 + 1024 * 768 * 2,
                {"workload": "bf16 nomination GEMM of the flat search, 1024 queries x 1M x 768 per launch"})
    if not t:
        return None
    m = unit_counters(f"gemm16_{tag}_mfma", sub, ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_INSTS_MFMA"))
    if m.get("GRBM_GUI_ACTIVE"):
        m["mfma_busy_fraction"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * m["GRBM_GUI_ACTIVE"] / 8)
    w = unit_counters(f"gemm16_{tag}_wait", sub, ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES"))
    if w.get("SQ_WAVE_CYCLES"):
        w["parked_at_waitcnt_or_barrier_fraction_of_wave_cycles"] = w["SQ_WAIT_ANY"] / w["SQ_WAVE_CYCLES"]
        w["issue_stall_fraction_of_wave_cycles"] = w["SQ_WAIT_INST_ANY"] / w["SQ_WAVE_CYCLES"]
    l = unit_counters(f"gemm16_{tag}_lds", sub, ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS"))
    c = unit_counters(f"gemm16_{tag}_tcc", sub, ("TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum"))
    if c.get("TCC_REQ_sum"):
        c["l2_hit_fraction"] = c["TCC_HIT_sum"] / c["TCC_REQ_sum"]
        c["l2_request_bytes_at_128"] = c["TCC_REQ_sum"] * 128
    t.update({"matrix_unit": m, "waits": w, "lds": l, "l2": c})
    return t


out["flat_gemm_bf16_big"] = gemm16("big", "flat_gemm_bf16_big_kernel<false, 3, 0>")
out["flat_gemm_bf16_tile128"] = gemm16("tile128", "flat_gemm_dma_kernel<false, 2, 0, true>")
# (r06: groups of up to 512 pairs take the tiles of up to four 32-query blocks: flat_gemm_dma32_grouped_kernel<.., MODE 2, RB 4>)
GROUPED = "flat_gemm_dma32_grouped_kernel<false, 2, 4, false>"
gr = traffic("grouped_fetch", None, GROUPED, None, {"workload": "partition-probed fp32 search: the grouped GEMM, mean over tools/probe_gemm_time.py's calls (nprobes 1 .. 32, 1024 and 8192 queries)"})
if gr:
    gr["matrix_unit"] = unit_counters("grouped_mfma", GROUPED, ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_INSTS_MFMA"))
    gr["waits"] = unit_counters("grouped_wait", GROUPED, ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES"))
out["flat_gemm_grouped"] = gr
km = traffic("kmeans_fetch", "kmeans_write", "km_gemm_kernel<false, true>", 1_000_000 * 768 * 4, {"workload": "k-means assignment GEMM, 1M x 768 x 122, bf16 [hi|lo] splits"})
if km:
    km["matrix_unit"] = unit_counters("kmeans_mfma", "km_gemm_kernel<false, true>", ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_INSTS_MFMA"))
    km["valu"] = unit_counters("kmeans_valu", "km_gemm_kernel<false, true>", ("SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY"))
out["km_gemm"] = km
out["km_update"] = traffic("kmeans_fetch", "kmeans_write", "km_update_kernel", 1_000_000 * 768 * 4, {"workload": "k-means update, 1M x 768 x 122: every row once, by member list"})
en = traffic("encode_fetch", None, "pq_nominate_bf16_kernel<true>", 348_160 * 768 * 4, {"workload": "ProductQuantizer.Encode, one launch = 348 160 rows x 768 (a 1M-row Encode is three)"})
if en:
    en["instructions"] = unit_counters("encode_inst", "pq_nominate_bf16_kernel<true>", ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES", "SQ_VALU_MFMA_BUSY_CYCLES"))
    en["waits"] = unit_counters("encode_wait", "pq_nominate_bf16_kernel<true>", ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE"))
    i, w = en["instructions"], en["waits"]
    if i.get("SQ_INSTS_VALU"):
        i["valu_instructions_per_block_of_32_rows_x_1_subquantizer"] = i["SQ_INSTS_VALU"] / (348_160 / 32 * 96)
    if w.get("GRBM_GUI_ACTIVE") and w.get("SQ_ACTIVE_INST_VALU"):
        w["valu_busy_fraction_at_4_cycles_per_instruction"] = w["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * w["GRBM_GUI_ACTIVE"] / 8)
out["pq_nominate_encode"] = en
print(json.dumps(out, indent=1))
