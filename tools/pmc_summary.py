"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel, mean counter value per dispatch."""
import csv, glob, collections, sys
root = sys.argv[1]
f = glob.glob(root + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][-60:]
    if "vg::" not in r["Kernel_Name"]: continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
w = csv.writer(sys.stdout)   # kernel names carry commas (template arguments): quoted
w.writerow(["kernel", "counter", "dispatches", "mean_per_dispatch"])
for k, v in sorted(agg.items()):
    for a, b in sorted(v.items()):
        w.writerow([k, a, cnt[(k, a)], f"{b / cnt[(k, a)]:.6g}"])
