#!/bin/bash
# packed GELU in k_hgemm: tests, GEMM mix against the previous library, U-ViT job A/B
O=gpurun_out/r05x; mkdir -p $O
PREV=scripts/diag/build/libdua_hip_prev_gelu.so
timeout -k 10 600 python -m pytest tests/test_gpu_hgemm.py tests/test_gpu_ops.py -x -q -k "hgemm or gelu or gemm or uvit" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
grep -q " passed" $O/tests.txt || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_models.py -x -q -k "uvit or G6 or G17" >> $O/tests.txt 2>&1; tail -3 $O/tests.txt
for rep in 1 2; do
for s in uvit320 uvit64; do
  echo "== prev $s" >> $O/gelu_ab.txt; timeout -k 10 200 python scripts/bench_gemm_mix.py --set $s --only fc1 --lib $PREV 2>/dev/null | grep -v amdgpu | tail -2 >> $O/gelu_ab.txt
  echo "== new $s" >> $O/gelu_ab.txt; timeout -k 10 200 python scripts/bench_gemm_mix.py --set $s --only fc1 2>/dev/null | grep -v amdgpu | tail -2 >> $O/gelu_ab.txt
done
echo "== prev job" >> $O/gelu_ab.txt; DUA_DIAG_LIB=$PREV timeout -k 10 300 python scripts/diag/uvit_job.py 5 2>/dev/null | tail -1 | cut -c1-200 >> $O/gelu_ab.txt
echo "== new job" >> $O/gelu_ab.txt; timeout -k 10 300 python scripts/diag/uvit_job.py 5 2>/dev/null | tail -1 | cut -c1-200 >> $O/gelu_ab.txt
done
grep "==\|fc1\|value" $O/gelu_ab.txt
