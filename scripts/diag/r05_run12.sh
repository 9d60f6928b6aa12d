#!/bin/bash
# what the (switched-off) f16x3 range check still costs: product library against -DDUA_NO_RANGE_CHECK, GEMM mix + U-ViT job, alternating
O=gpurun_out/r05u; mkdir -p $O
NR=scripts/diag/build/libdua_hip_norange.so
for rep in 1 2; do
for s in uvit320 uvit64; do
  echo "== product $s" >> $O/norange_ab.txt; timeout -k 10 200 python scripts/bench_gemm_mix.py --set $s 2>/dev/null | grep -v amdgpu | tail -6 >> $O/norange_ab.txt
  echo "== norange $s" >> $O/norange_ab.txt; timeout -k 10 200 python scripts/bench_gemm_mix.py --set $s --lib $NR 2>/dev/null | grep -v amdgpu | tail -6 >> $O/norange_ab.txt
done
echo "== product job" >> $O/norange_ab.txt; timeout -k 10 300 python scripts/diag/uvit_job.py 5 2>/dev/null | tail -1 | cut -c1-200 >> $O/norange_ab.txt
echo "== norange job" >> $O/norange_ab.txt; DUA_DIAG_LIB=$NR timeout -k 10 300 python scripts/diag/uvit_job.py 5 2>/dev/null | tail -1 | cut -c1-200 >> $O/norange_ab.txt
done
grep "==\|mix\|value" $O/norange_ab.txt
