#!/bin/bash
# round 6, GPU call: (1) price the second-resident-wave rebuild of k_wconv4 with the timing-only builds (epilogue exposure of the 8-wave,
# 256-register F(2,3) kernel against the 4-wave, 512-register F(4,3) kernel on the same layers, same box); (2) MFMA-pipe / clock counters of
# k_wconv4 and k_gn_apply_wino4 at the final kernels; (3) rocprofv3 kernel stats of the configs[4] (SD) leg
set -o pipefail
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
L=128x256x256,64x512x512,32x512x512
{
for rep in 1 2; do
  echo "==== F(2,3) k_wconv, product (rep $rep)";          python3 scripts/bench_conv_mix.py --fmt f16x3w --only $L | grep -E "^H=|^mix"
  echo "==== F(2,3) k_wconv, WCONV_DIAG=64 no epilogue";   python3 scripts/bench_conv_mix.py --fmt f16x3w --only $L --lib scripts/diag/build/libdua_hip_wdiag64.so | grep -E "^H=|^mix"
  echo "==== F(4,3) k_wconv4, product (rep $rep)";         python3 scripts/bench_conv_mix.py --fmt f16x3w4 --only $L | grep -E "^H=|^mix"
  echo "==== F(4,3) k_wconv4, WCONV4_DIAG=2 no epilogue";  python3 scripts/bench_conv_mix.py --fmt f16x3w4 --only $L --lib scripts/diag/build/libdua_hip_w4diag2.so | grep -E "^H=|^mix"
done
} > $OUT/r06d_wconv_epilogue_exposure_f23_vs_f43.txt 2>&1
echo "pricing done"
bash scripts/pmc_kernel.sh k_wconv4 python3 scripts/bench_conv_mix.py --fmt f16x3w4 --only 128x256x256 --reps 2 > $OUT/r06d_pmc_mfma_lds_wconv4_128x256.json
bash scripts/pmc_kernel.sh k_wconv4 python3 scripts/bench_conv_mix.py --fmt f16x3w4 --only 64x512x512 --reps 2 > $OUT/r06d_pmc_mfma_lds_wconv4_64x512.json
bash scripts/pmc_kernel.sh k_gn_apply_wino4 python3 scripts/bench_gn_apply.py f16x3w4 > $OUT/r06d_pmc_gn_apply_wino4.json
echo "pmc done"
rocprofv3 --kernel-trace --stats -d $OUT/r06d_sd_stats -o r06d -- python3 scripts/diag/sd_job.py f16x3 graph > $OUT/r06d_sd_job_under_rocprof.txt 2>&1
python3 scripts/summarize_pmc.py stats $(find $OUT/r06d_sd_stats -name "*.db" | head -1) > $OUT/r06d_sd_job_kernel_stats.csv
echo "sd stats done"
