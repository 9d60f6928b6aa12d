#!/bin/bash
# range-check pointer as __constant__: tests that use it, then the U-ViT GEMM mix (old tree / new), then the U-ViT job
O=gpurun_out/r05t; mkdir -p $O
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "range or gemm_f16x3 or hgemm or uvit_block or attention" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
grep -q " passed" $O/tests.txt || exit 1
for rep in 1 2; do
for s in uvit320 uvit64; do
  echo "== old $s" >> $O/old_vs_new.txt; timeout -k 10 200 python scripts/diag/build/old_bdcd685/scripts/bench_gemm_mix.py --set $s 2>/dev/null | grep -v amdgpu | tail -6 >> $O/old_vs_new.txt
  echo "== new $s" >> $O/old_vs_new.txt; timeout -k 10 200 python scripts/bench_gemm_mix.py --set $s 2>/dev/null | grep -v amdgpu | tail -6 >> $O/old_vs_new.txt
done; done
grep "==\|mix\|fc1\|qkv" $O/old_vs_new.txt
timeout -k 10 300 python scripts/diag/uvit_job.py 5 2>/dev/null | tail -1 | cut -c1-260 | tee $O/uvit_job.txt
