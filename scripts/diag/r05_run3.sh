set -x
O=gpurun_out/r05h; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_wino4.py -x -q -k "f32 or fp32" -s > $O/test_f32_wino4.log 2>&1; echo "rc=$?" >> $O/test_f32_wino4.log; tail -15 $O/test_f32_wino4.log
grep -q "rc=0" $O/test_f32_wino4.log && timeout -k 10 500 python bench.py --precision fp32 --steps 10 --warmup 1 --no-secondary --no-cpu-baseline --no-other-configs --legs-file $O/bench_fp32_f43.json 2>$O/bench_fp32.err | tail -n 1 | cut -c1-1500
