#!/bin/bash
# rocprofv3 PMC passes behind profiles/r02_pmc_*.csv (run on the GPU box through gpurun; counters in their own
# runs with --kernel-trace only, as the pool requires).  usage: tools/collect_pmc.sh [adc|graph|all]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
what=${1:-all}
run() {  # name, counters, command...
    name=$1; ctrs=$2; shift 2
    rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d gpurun_out/pmc_$name -- "$@" > gpurun_out/pmc_$name.log 2>&1
    python3 tools/pmc_summary.py gpurun_out/pmc_$name > gpurun_out/r02_pmc_$name.csv 2>> gpurun_out/pmc_$name.log
    rm -rf gpurun_out/pmc_$name
}
if [ "$what" = adc ] || [ "$what" = all ]; then
    run adc_lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" python3 tools/adc_prof.py 10000000 1 10
    run adc_valu "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" python3 tools/adc_prof.py 10000000 1 10
    run adc_wait "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAVES SQ_INSTS_VMEM_RD" python3 tools/adc_prof.py 10000000 1 10
    run adc_fetch "FETCH_SIZE" python3 tools/adc_prof.py 10000000 1 10
    run adc_write "WRITE_SIZE" python3 tools/adc_prof.py 10000000 1 10
fi
if [ "$what" = graph ] || [ "$what" = all ]; then
    run graph_fetch "FETCH_SIZE" python3 tools/graph_prof.py 200000
    run graph_write "WRITE_SIZE" python3 tools/graph_prof.py 200000
    run graph_valu "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" python3 tools/graph_prof.py 200000
    run graph_lds "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD" python3 tools/graph_prof.py 200000
    run sq8_fetch "FETCH_SIZE" python3 tools/sq8_prof.py 4000000 1 10
    python3 tools/graph_prof.py 200000 > gpurun_out/r02_graph_prof_counts.json 2>/dev/null
fi
ls -la gpurun_out/r02_pmc_*.csv
