# FETCH_SIZE of the flat GEMM with and without the nontemporal hint on the row loads (run through gpurun)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in default bnt; do
  if [ $v = bnt ]; then export VECGO_HIP_LIB=variants/libvecgo_bnt.so; else unset VECGO_HIP_LIB; fi
  python3 tools/flat_time.py | tail -1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_gemm_$v -- python3 tools/flat_time.py > gpurun_out/pmc_gemm_$v.log 2>&1
  python3 tools/pmc_summary.py gpurun_out/pmc_gemm_$v | grep -i "gemm"
  rm -rf gpurun_out/pmc_gemm_$v
done
