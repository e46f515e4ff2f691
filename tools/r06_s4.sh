cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s4
timeout 1500 python3 -m pytest tests/test_gpu_sharded_2rank.py tests/test_gpu_bench_2rank.py -x -q -m gpu 2>&1 | tail -15
