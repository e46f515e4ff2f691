cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s3
timeout 300 tools/ubench/gemm_bf16_probe 1000000 > gpurun_out/s3/gemm_bf16_probe.txt 2>&1
cat gpurun_out/s3/gemm_bf16_probe.txt
timeout 900 python3 -m pytest tests/test_gpu_flat_bf16.py tests/test_gpu_sq8.py tests/test_gpu_flat.py tests/test_gpu_probe.py tests/test_gpu_flat_filtered.py -x -q -m gpu 2>&1 | tail -5
timeout 300 python3 tools/flat_bf16_time.py 2>&1 | grep -v amdgpu.ids | tail -12
