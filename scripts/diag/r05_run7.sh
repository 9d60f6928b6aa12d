set -x
O=gpurun_out/r05M; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_wino4.py tests/test_gpu_wino.py -x -q > $O/test_wino.log 2>&1; echo "rc=$?" >> $O/test_wino.log; tail -3 $O/test_wino.log
grep -q "rc=0" $O/test_wino.log || exit 1
for L in "--lib scripts/diag/build/libdua_hip_prev_patch.so" "" "--lib scripts/diag/build/libdua_hip_prev_patch.so" ""; do echo "== $L" >> $O/w4_residual_lds_ring_ab.txt; timeout -k 10 200 python scripts/bench_conv_mix.py --fmt f16x3w4 --only 128x256x256,128x512x256,64x256x256,64x512x512,32x512x512,32x1280x512 $L 2>/dev/null | grep -v amdgpu >> $O/w4_residual_lds_ring_ab.txt; done; grep "==\|mix" $O/w4_residual_lds_ring_ab.txt
