cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_probe.py tests/test_gpu_sq8.py -x -q -m gpu 2>&1 | tail -8
