#!/bin/bash
# tools/kernel_stats.sh NAME PROGRAM ARGS... : rocprofv3 --kernel-trace --stats of one tool (through gpurun) ->
# gpurun_out/prof/NAME_kernel_stats.csv (name, calls, total ns, average ns, %) + the program's output in NAME.log
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
name=$1; shift
mkdir -p gpurun_out/prof
timeout -k 10 ${PROF_TIMEOUT:-420} rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/raw_$name -- "$@" > gpurun_out/prof/$name.log 2>&1 || echo "kernel_stats: $name failed" >&2
f=$(find gpurun_out/prof/raw_$name -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/prof/${name}_kernel_stats.csv
rm -rf gpurun_out/prof/raw_$name
python3 - "$name" <<'PY'
import csv, sys
rows = list(csv.reader(open(f"gpurun_out/prof/{sys.argv[1]}_kernel_stats.csv")))
for r in rows[1:25]:
    print(f"{r[0][:90]:90s} calls {r[1]:>6s} avg {float(r[3]) / 1e3:10.1f} us  {r[4]:>6s} %")
PY
tail -3 gpurun_out/prof/$name.log
