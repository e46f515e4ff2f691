cd $GRAFT_REPO_ROOT
export ROUND=r06 PMC_TIMEOUT=300
tools/collect_pmc.sh ${1:-all} > gpurun_out/pmc_collect.log 2>&1
tail -5 gpurun_out/pmc_collect.log
ls gpurun_out/pmc/*.csv | wc -l
