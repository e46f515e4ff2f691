#!/bin/bash
tag=$1; m=$2
tools/pmc_run.sh deep_${tag}_a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" python3 tools/walk_prof.py 1000000 $m
tools/pmc_run.sh deep_${tag}_b "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT" python3 tools/walk_prof.py 1000000 $m
tools/pmc_run.sh deep_${tag}_c "SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAIT_INST_LDS" python3 tools/walk_prof.py 1000000 $m
tools/pmc_run.sh deep_${tag}_d "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" python3 tools/walk_prof.py 1000000 $m
tools/pmc_run.sh deep_${tag}_e "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INST_LEVEL_LDS" python3 tools/walk_prof.py 1000000 $m
grep -h "hnsw_search\|vamana\|^kernel\|Kernel" gpurun_out/pmc/deep_${tag}_[a-e].csv
tail -3 gpurun_out/pmc/deep_${tag}_c.log
