cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s2
for v in "" $(ls variants/libvecgo_*.so 2>/dev/null); do
  VECGO_HIP_LIB=$v timeout 300 python3 tools/encode_one.py 1000000 2>&1 | grep -v amdgpu.ids >> gpurun_out/s2/encode_probe.txt
done
cat gpurun_out/s2/encode_probe.txt
timeout 600 python3 -m pytest tests/test_gpu_pq.py -x -q -m gpu 2>&1 | tail -3
