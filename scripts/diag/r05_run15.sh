#!/bin/bash
# lane-id-fresh column-sum folds: wino / hgemm tests, then conv mix A/B against the previous library (all F(4,3) layers + the F(2,3) ones)
O=gpurun_out/r05A; mkdir -p $O
PREV=scripts/diag/build/libdua_hip_prev_shfl.so
timeout -k 10 700 python -m pytest tests/test_gpu_wino4.py tests/test_gpu_wino.py tests/test_gpu_hgemm.py -x -q > $O/tests.txt 2>&1; tail -2 $O/tests.txt
grep -q " passed" $O/tests.txt || exit 1
for L in "--lib $PREV" "" "--lib $PREV" ""; do echo "== w4 $L" >> $O/shfl_ab.txt; timeout -k 10 200 python scripts/bench_conv_mix.py --fmt f16x3w4 --only 128x256x256,128x512x256,64x256x256,64x512x512,32x512x512,32x1280x512 $L 2>/dev/null | grep -v amdgpu >> $O/shfl_ab.txt; done
for L in "--lib $PREV" "" "--lib $PREV" ""; do echo "== w2 $L" >> $O/shfl_ab.txt; timeout -k 10 200 python scripts/bench_conv_mix.py --fmt f16x3w --only 16x768x768,16x1536x768,8x1024x1024,8x2048x1024 $L 2>/dev/null | grep -v amdgpu >> $O/shfl_ab.txt; done
grep "==\|mix" $O/shfl_ab.txt
