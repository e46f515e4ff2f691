#!/bin/bash
# r06 GPU session 1: vector-ALU rates, the bf16 GEMM tiles against each other (+ their counters), parity of the searches on them
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s1
timeout 120 tools/ubench/valu_rate > gpurun_out/s1/valu_rate.txt 2>&1
timeout 300 tools/ubench/gemm_bf16_probe 1000000 > gpurun_out/s1/gemm_bf16_probe.txt 2>&1
timeout 900 python3 -m pytest tests/test_gpu_flat_bf16.py tests/test_gpu_sq8.py tests/test_gpu_flat.py -x -q -m gpu > gpurun_out/s1/pytest.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/s1/pytest.txt
P=tools/pmc_run.sh
export PMC_TIMEOUT=200
$P s1_gemm_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA" tools/ubench/gemm_bf16_probe 1000000
$P s1_gemm_lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" tools/ubench/gemm_bf16_probe 1000000
$P s1_gemm_wait "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" tools/ubench/gemm_bf16_probe 1000000
$P s1_gemm_tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" tools/ubench/gemm_bf16_probe 1000000
$P s1_gemm_fetch "FETCH_SIZE" tools/ubench/gemm_bf16_probe 1000000
cp gpurun_out/pmc/s1_*.csv gpurun_out/s1/ 2>/dev/null
ls gpurun_out/s1
