#!/bin/bash
# Every rocprofv3 --pmc pass behind profiles/rNN_traffic.json (ROUND=r04 by default), at the bench's own shapes (run on the GPU box through
# gpurun; counters in their own runs with --kernel-trace only, as the pool requires).  Each pass leaves
# gpurun_out/pmc/NAME.csv (per-kernel mean counter values per dispatch, tools/pmc_run.sh); tools/make_traffic_json.py
# turns them into the JSON.  usage: tools/collect_pmc.sh [scans|gemm|walks|build|all]
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repository copy on the GPU box)}"
cd "$GRAFT_REPO_ROOT"
what=${1:-all}
ROUND=${ROUND:-r06}
P=tools/pmc_run.sh
if [ "$what" = scans ] || [ "$what" = all ]; then
    $P adc_fetch "FETCH_SIZE" python3 tools/adc_prof.py 10000000 1 10
    $P adc_write "WRITE_SIZE" python3 tools/adc_prof.py 10000000 1 10
    $P adc_lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" python3 tools/adc_prof.py 10000000 1 10
    $P rq_fetch "FETCH_SIZE" python3 tools/rabitq_prof.py 10000000 1 10
    $P rq_write "WRITE_SIZE" python3 tools/rabitq_prof.py 10000000 1 10
    $P rqmq_fetch "FETCH_SIZE" python3 tools/scan_batch_time.py rabitq 10000000 1024
    $P rqmq_valu "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_WAVES" python3 tools/scan_batch_time.py rabitq 10000000 1024
    $P adcmq_fetch "FETCH_SIZE" python3 tools/scan_batch_time.py adc 10000000 64
    $P sq8_fetch "FETCH_SIZE" python3 tools/sq8_prof.py 4000000 1 10
    $P i4_fetch "FETCH_SIZE" python3 tools/int4_batch_time.py 4000000
    $P i4_write "WRITE_SIZE" python3 tools/int4_batch_time.py 4000000
    $P i4_lds "SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" python3 tools/int4_batch_time.py 4000000
    $P i4_issue "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" python3 tools/int4_batch_time.py 4000000
    $P i4_wait "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" python3 tools/int4_batch_time.py 4000000
fi
if [ "$what" = gemm ] || [ "$what" = all ]; then
    $P gemm_fetch "FETCH_SIZE" python3 tools/flat_time.py
    $P gemm_write "WRITE_SIZE" python3 tools/flat_time.py
    $P gemm_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32" python3 tools/flat_time.py
fi
if [ "$what" = gemm16 ] || [ "$what" = all ]; then
    # r06: the bfloat16 nomination GEMMs — the persistent 256 x 256 tile (flat_gemm_bf16_big_kernel, the bf16 flat filter and the
    # SQ8 batch), the 128 x 128 tile it replaced above 128 queries (VG_FLAT_NO_BIG_TILE=1: tools/flat_bf16_time.py reads the hook
    # from its first argument) and the grouped GEMM of the partition-probed search
    M16="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA"
    W16="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
    L16="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"
    T16="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
    for v in big tile128; do
        a=""; [ $v = tile128 ] && a="no_big_tile"
        $P gemm16_${v}_fetch "FETCH_SIZE" python3 tools/flat_bf16_time.py $a
        $P gemm16_${v}_write "WRITE_SIZE" python3 tools/flat_bf16_time.py $a
        $P gemm16_${v}_mfma "$M16" python3 tools/flat_bf16_time.py $a
        $P gemm16_${v}_wait "$W16" python3 tools/flat_bf16_time.py $a
        $P gemm16_${v}_lds "$L16" python3 tools/flat_bf16_time.py $a
        $P gemm16_${v}_tcc "$T16" python3 tools/flat_bf16_time.py $a
    done
    $P grouped_fetch "FETCH_SIZE" python3 tools/probe_gemm_time.py
    $P grouped_mfma "$M16" python3 tools/probe_gemm_time.py
    $P grouped_wait "$W16" python3 tools/probe_gemm_time.py
    $P sq8nom_fetch "FETCH_SIZE" python3 tools/sq8_nominate_time.py
    $P sq8nom_mfma "$M16" python3 tools/sq8_nominate_time.py
fi
if [ "$what" = walks ] || [ "$what" = all ]; then
    for m in "f32 128" "f32 2048" "pq 128" "vamana_pq"; do
        tag=$(echo $m | tr ' ' '_')
        $P walk_${tag}_fetch "FETCH_SIZE" python3 tools/walk_prof.py 1000000 $m
        $P walk_${tag}_write "WRITE_SIZE" python3 tools/walk_prof.py 1000000 $m
        $P walk_${tag}_valu "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_WAVES" python3 tools/walk_prof.py 1000000 $m
    done
fi
if [ "$what" = build ] || [ "$what" = all ]; then
    # r05: the build-side kernels north_star names (k-means assignment, PQ Train, Encode): issue / wait / traffic / matrix-unit
    # counters at the bench's shapes (profiles/r05_pmc_{kmeans,pqtrain,encode}_*.csv)
    V="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
    M="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA"
    $P kmeans_valu "$V" python3 tools/kmeans_time.py
    $P kmeans_mfma "$M" python3 tools/kmeans_time.py
    $P kmeans_fetch "FETCH_SIZE" python3 tools/kmeans_time.py
    $P kmeans_write "WRITE_SIZE" python3 tools/kmeans_time.py
    $P pqtrain_valu "$V" python3 tools/pq_train_time.py
    $P pqtrain_mfma "$M" python3 tools/pq_train_time.py
    $P pqtrain_fetch "FETCH_SIZE" python3 tools/pq_train_time.py
    $P kmeans_wait "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" python3 tools/kmeans_time.py
    $P encode_valu "$V" python3 tools/encode_one.py
    $P encode_mfma "$M" python3 tools/encode_one.py
    $P encode_fetch "FETCH_SIZE" python3 tools/encode_one.py
    $P encode_inst "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES" python3 tools/encode_one.py
    $P encode_wait "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" python3 tools/encode_one.py
fi
python3 tools/make_traffic_json.py gpurun_out/pmc > gpurun_out/pmc/${ROUND}_traffic.json
ls gpurun_out/pmc/*.csv | wc -l
