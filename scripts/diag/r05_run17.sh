#!/bin/bash
# V writer (k_gn_apply_wino4) compiled for 4 waves per SIMD (128 registers, a few spills) against 3 (138 registers), alternating
O=gpurun_out/r05C; mkdir -p $O
PREV=scripts/diag/build/libdua_hip_prev_writer.so
for rep in 1 2; do
  echo "== prev" >> $O/writer_ab.txt; DUA_DIAG_LIB=$PREV timeout -k 10 200 python scripts/bench_gn_apply.py f16x3w4 2>/dev/null | grep "wide=1" >> $O/writer_ab.txt
  echo "== new" >> $O/writer_ab.txt; timeout -k 10 200 python scripts/bench_gn_apply.py f16x3w4 2>/dev/null | grep "wide=1" >> $O/writer_ab.txt
done
cat $O/writer_ab.txt
