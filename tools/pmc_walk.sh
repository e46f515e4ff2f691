#!/bin/bash
# PMC passes over one graph-walk kernel (tools/walk_prof.py MODE EF), optionally for a library variant.
# usage: tools/pmc_walk.sh TAG "MODE EF" [LIB]
tag=$1; m=$2; export VECGO_HIP_LIB=$3
tools/pmc_run.sh walk_${tag}_a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE" python3 tools/walk_prof.py 1000000 $m
tools/pmc_run.sh walk_${tag}_b "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" python3 tools/walk_prof.py 1000000 $m
grep -h "hnsw_search\|vamana" gpurun_out/pmc/walk_${tag}_[ab].csv
