#!/bin/bash
# same-box comparison of the U-ViT GEMM mix: the tree at bdcd685 (scripts/diag/build/old_bdcd685, not committed) against the current one
O=gpurun_out/r05s; mkdir -p $O
for rep in 1 2; do
for s in uvit320 uvit64; do
  echo "== old $s" >> $O/old_vs_new.txt; timeout -k 10 200 python scripts/diag/build/old_bdcd685/scripts/bench_gemm_mix.py --set $s 2>/dev/null | grep -v amdgpu | tail -6 >> $O/old_vs_new.txt
  echo "== new $s" >> $O/old_vs_new.txt; timeout -k 10 200 python scripts/bench_gemm_mix.py --set $s 2>/dev/null | grep -v amdgpu | tail -6 >> $O/old_vs_new.txt
done; done
cat $O/old_vs_new.txt
