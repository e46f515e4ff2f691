"""The two frontier tables of DESIGN.md section 5, printed as markdown from a bench line
(default profiles/r03_bench_line.json): `python tools/design_tables.py [line.json]`."""
import json, sys
l = json.load(open(sys.argv[1] if len(sys.argv) > 1 else "profiles/r03_bench_line.json"))
h = l["hnsw_pq"]
cpu = {e["ef"]: e for e in h["cpu"]["sweep"]}
print("| pipeline | ef | recall@10 | ms per 1024 queries | queries/s | gathered TB/s | CPU twin queries/s (16 usable cores) |")
print("|---|---|---|---|---|---|---|")
for i, e in enumerate(h["frontier_f32"]):
    c = cpu.get(e["ef"])
    ctxt = f"{c['qps'] / 1e3:.1f} k (ids equal)" if c and c.get("ids_equal_gpu") else (f"{c['qps'] / 1e3:.1f} k (IDS DIFFER)" if c else "—")
    print(f"| {'HNSW fp32' if i == 0 else ''} | {e['ef']} | {e['recall_at_10']:.3f} | {e['ms_per_1024']:.2f} | {e['qps'] / 1e3:.0f} k | {e['gathered_gbs'] / 1e3:.2f} | {ctxt} |")
for i, e in enumerate(h["frontier_pq_rerank"]):
    print(f"| {'HNSW on PQ codes + exact rerank of ef' if i == 0 else ''} | {e['ef']} | {e['recall_at_10']:.3f} | {e['ms_per_1024']:.2f} (walk {e['walk_ms'] / 8:.2f} + rerank {e['rerank_ms'] / 8:.2f}) | "
          f"{e['qps'] / 1e3:.0f} k | {e['pq_scores_per_s'] / 1e9:.2f} G node scores/s | — |")
x = h["exact_path"]
print(f"| exact (operating point) | — | {x['recall_at_10']:.3f} | {x['ms_per_1024']:.2f} | {l['value'] / 1e3:.1f} k | MFMA {l['roofline']['frac']:.3f} of peak | {l['cpu_baseline']['value']:.0f} (flat scan, {l['cpu_baseline']['cores']} cores) |")
print()
s = l["structured_corpus"]
scpu = {e["ef"]: e for e in s["cpu"]["sweep"]}
f32, pq = s["frontier_f32"], s["frontier_pq_rerank"]
print("| pipeline | ef | recall@10 | queries/s |")
print("|---|---|---|---|")
print("| HNSW fp32 | " + " / ".join(str(e["ef"]) for e in f32) + " | " + " / ".join(f"{e['recall_at_10']:.3f}" for e in f32) + " | " + " / ".join(f"{e['qps'] / 1e3:.0f} k" for e in f32) + " |")
print("| HNSW on PQ + rerank | " + " / ".join(str(e["ef"]) for e in pq) + " | " + " / ".join(f"{e['recall_at_10']:.3f}" for e in pq) + " | " + " / ".join(f"{e['qps'] / 1e3:.0f} k" for e in pq) + " |")
print(f"| exact (MFMA) | — | 1.000 | {s['at_recall_0_95']['exact_qps'] / 1e3:.1f} k |")
es = sorted(scpu)
print("| CPU HNSW fp32, same graph, 16 cores | " + " / ".join(str(e) for e in es) + " | " + " / ".join(f"{scpu[e]['recall_at_10']:.3f}" for e in es) +
      (" (ids equal the GPU's)" if all(scpu[e].get("ids_equal_gpu") for e in es) else " (IDS DIFFER)") + " | " + " / ".join(f"{scpu[e]['qps'] / 1e3:.1f} k" for e in es) + " |")
print()
print(json.dumps(s["at_recall_0_95"], indent=1))
