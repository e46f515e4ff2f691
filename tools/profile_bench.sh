#!/bin/bash
# The committed per-kernel summary of the bench: `rocprofv3 --kernel-trace --stats -- python3 bench.py` on the GPU box
# (through gpurun) -> gpurun_out/prof/${ROUND}_bench_kernel_stats.csv + the JSON line of that run.  bench.py's `roofline`
# (HIP events inside the library) must agree with the AverageNs of the same kernel here.
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run through gpurun}"
ROUND=${ROUND:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/raw -- python3 bench.py "$@" > gpurun_out/prof/bench_under_rocprof.log 2>&1
f=$(find gpurun_out/prof/raw -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/prof/${ROUND}_bench_kernel_stats.csv
grep '^{"metric"' gpurun_out/prof/bench_under_rocprof.log | tail -1 > gpurun_out/prof/${ROUND}_bench_line_under_rocprof.json
rm -rf gpurun_out/prof/raw
head -8 gpurun_out/prof/${ROUND}_bench_kernel_stats.csv | cut -c1-160
