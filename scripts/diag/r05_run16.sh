#!/bin/bash
# k_hgemm: uniform item values in SGPRs, lane id formed in the epilogue: tests, GEMM mix and U-ViT job A/B against the previous library
O=gpurun_out/r05B; mkdir -p $O
PREV=scripts/diag/build/libdua_hip_prev_shfl.so
timeout -k 10 700 python -m pytest tests/test_gpu_hgemm.py tests/test_gpu_wino.py tests/test_gpu_wino4.py -x -q > $O/tests.txt 2>&1; tail -2 $O/tests.txt
grep -q " passed" $O/tests.txt || exit 1
for rep in 1 2; do
for s in uvit320 uvit64; do
  echo "== prev $s" >> $O/hg_ab.txt; timeout -k 10 200 python scripts/bench_gemm_mix.py --set $s --lib $PREV 2>/dev/null | grep -v amdgpu | tail -6 >> $O/hg_ab.txt
  echo "== new $s" >> $O/hg_ab.txt; timeout -k 10 200 python scripts/bench_gemm_mix.py --set $s 2>/dev/null | grep -v amdgpu | tail -6 >> $O/hg_ab.txt
done
echo "== prev job" >> $O/hg_ab.txt; DUA_DIAG_LIB=$PREV timeout -k 10 300 python scripts/diag/uvit_job.py 5 2>/dev/null | tail -1 | cut -c1-200 >> $O/hg_ab.txt
echo "== new job" >> $O/hg_ab.txt; timeout -k 10 300 python scripts/diag/uvit_job.py 5 2>/dev/null | tail -1 | cut -c1-200 >> $O/hg_ab.txt
done
grep "==\|mix\|value\|qkv\|fc1" $O/hg_ab.txt
