set -x
O=gpurun_out/r05m; mkdir -p $O
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log; tail -4 $O/gpu_tests.log
grep -q "rc=0" $O/gpu_tests.log || exit 1
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
timeout -k 10 600 python bench.py > $O/bench_default_stdout.txt 2> $O/bench_default.err; tail -n 1 $O/bench_default_stdout.txt | cut -c1-700
