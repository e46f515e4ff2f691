#!/bin/bash
# texture-path counters of one graph-walk kernel, ONE hardware block per pass (a request that mixes TA and TCP blocks
# did not fit and made rocprofv3 abort, r04): tools/pmc_ta.sh TAG "MODE EF"   (through gpurun)
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
tag=$1; m=$2
export PMC_TIMEOUT=${PMC_TIMEOUT:-200}
tools/pmc_run.sh ta_${tag}_a "TA_TA_BUSY_sum GRBM_GUI_ACTIVE" python3 tools/walk_prof.py 1000000 $m
[ -s gpurun_out/pmc/ta_${tag}_a.csv ] || { echo "first pass failed: stopping"; tail -5 gpurun_out/pmc/ta_${tag}_a.log; exit 1; }
tools/pmc_run.sh ta_${tag}_b "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" python3 tools/walk_prof.py 1000000 $m
tools/pmc_run.sh ta_${tag}_c "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" python3 tools/walk_prof.py 1000000 $m
tools/pmc_run.sh ta_${tag}_d "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" python3 tools/walk_prof.py 1000000 $m
grep -h "hnsw_search\|vamana\|^kernel\|Kernel" gpurun_out/pmc/ta_${tag}_[a-d].csv | cut -c1-60,200-600
