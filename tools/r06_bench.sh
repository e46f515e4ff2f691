cd $GRAFT_REPO_ROOT
export ROUND=r06
mkdir -p gpurun_out/prof
timeout 1200 python3 bench.py > gpurun_out/prof/r06_bench_stdout.txt 2> gpurun_out/prof/r06_bench_stderr.txt; echo "bench rc=$?"
cp bench_full.json gpurun_out/prof/r06_bench_full.json 2>/dev/null
timeout 1500 tools/profile_bench.sh --steps 20 --warmup 3 2>&1 | tail -3
cut -c1-600 gpurun_out/prof/r06_bench_stdout.txt
