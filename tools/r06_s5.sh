cd $GRAFT_REPO_ROOT
timeout 600 python3 -m pytest tests/test_gpu_kmeans_mfma.py tests/test_gpu_kmeans_l0.py -x -q -m gpu 2>&1 | tail -4
PROF_TIMEOUT=300 tools/kernel_stats.sh kmeans6 python3 tools/kmeans_time.py 2>&1 | head -16
