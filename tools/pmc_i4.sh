#!/bin/bash
# issue-slot, wait and latency counters of int4_scan_tab_kernel (tools/int4_batch_time.py 4M x 768), one rocprofv3 --pmc
# pass per counter group -> gpurun_out/pmc/i4deep_*.csv.  usage (through gpurun): tools/pmc_i4.sh [tag]
set -uo pipefail
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
tag=${1:-i4deep}
P=tools/pmc_run.sh
$P ${tag}_a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" python3 tools/int4_batch_time.py 4000000
$P ${tag}_b "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT" python3 tools/int4_batch_time.py 4000000
$P ${tag}_c "SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_LDS_IDX_ACTIVE" python3 tools/int4_batch_time.py 4000000
$P ${tag}_d "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" python3 tools/int4_batch_time.py 4000000
grep -h "int4_scan_tab_kernel<false>" gpurun_out/pmc/${tag}_[a-d].csv
