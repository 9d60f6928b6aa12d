set -x
O=gpurun_out/r05i; mkdir -p $O
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log; tail -6 $O/gpu_tests.log
grep -q "rc=0" $O/gpu_tests.log || exit 1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 bash scripts/collect_profiles.sh r05i > $O/collect.log 2>&1; tail -3 $O/collect.log
cp gpurun_out/r05i_bench_kernel_stats.csv gpurun_out/r05i_pmc_traffic_conv.json gpurun_out/r05i_bench_under_rocprof.json $O/ 2>/dev/null
head -12 gpurun_out/r05i_bench_kernel_stats.csv | cut -c1-150
