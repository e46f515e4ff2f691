#!/bin/bash
# tools/pmc_run.sh NAME "COUNTER ..." PROGRAM ARGS... : one rocprofv3 --pmc pass (counters in their own run, with
# --kernel-trace only, as the pool requires) -> gpurun_out/pmc/NAME.csv = per-kernel mean counter values per dispatch.
# Run on the GPU box through gpurun.
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
name=$1; ctrs=$2; shift 2
mkdir -p gpurun_out/pmc
# (timeout: a counter set the hardware cannot collect makes rocprofv3 abort and then hang in its signal handler —
# one such pass cost 20 GPU-minutes in r04)
# a failing pass is recorded (its log stays) and the next one still runs; make_traffic_json.py fails if NO csv came out
timeout -k 10 ${PMC_TIMEOUT:-420} rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d gpurun_out/pmc/raw_$name -- "$@" > gpurun_out/pmc/$name.log 2>&1 || echo "pmc_run: $name failed, see gpurun_out/pmc/$name.log" >&2
python3 tools/pmc_summary.py gpurun_out/pmc/raw_$name > gpurun_out/pmc/$name.csv 2>> gpurun_out/pmc/$name.log || rm -f gpurun_out/pmc/$name.csv
rm -rf gpurun_out/pmc/raw_$name
