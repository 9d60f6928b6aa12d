set -x
mkdir -p gpurun_out/r05a
timeout -k 10 500 python -m pytest tests/test_gpu_ops.py -x -q -k "gemm_f16x3" -s > gpurun_out/r05a/test_gemm.log 2>&1; echo "rc=$?" >> gpurun_out/r05a/test_gemm.log
tail -5 gpurun_out/r05a/test_gemm.log
grep -q "rc=0" gpurun_out/r05a/test_gemm.log && \
timeout -k 10 200 python scripts/bench_gemm_mix.py --set adm --sg -1 --only skip128,skip64,skip32,skip16,skip8 > gpurun_out/r05a/mix_xtile.log 2>&1 && \
timeout -k 10 200 python scripts/bench_gemm_mix.py --set adm --sg 0 --only skip128,skip64,skip32,skip16,skip8 > gpurun_out/r05a/mix_sgemm.log 2>&1 && \
timeout -k 10 200 python scripts/bench_gemm_mix.py --set adm --sg 4 --only skip128,skip64,skip32,skip16,skip8 > gpurun_out/r05a/mix_sgemm4.log 2>&1 && \
timeout -k 10 200 python scripts/bench_gemm_mix.py --set adm --sg 2 --only skip128,skip64,skip32,skip16,skip8 > gpurun_out/r05a/mix_sgemm2.log 2>&1
cat gpurun_out/r05a/mix_xtile.log gpurun_out/r05a/mix_sgemm.log
