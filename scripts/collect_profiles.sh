#!/bin/bash
# Collects the rocprof evidence for one round tag on the GPU box (run through gpurun from the repo root):
#   scripts/collect_profiles.sh r02e [precision]
# 1. rocprofv3 --kernel-trace --stats of the headline job alone (python3 bench.py --no-cpu-baseline --no-secondary)
# 2. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) over a truncated job with the same batch and layer shapes
# Summaries land in gpurun_out/<tag>_*; copy the ones to be judged into profiles/.
set -e -o pipefail
TAG=${1:-rXX}
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_stats -o ${TAG} -- python3 bench.py --no-cpu-baseline --no-secondary --no-check > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_bench_under_rocprof.err
python3 scripts/summarize_pmc.py stats $(find $OUT/${TAG}_stats -name "*.db" | head -1) > $OUT/${TAG}_bench_kernel_stats.csv
PREC=${2:-f16x3}
# truncated job: 5 DDIM timesteps (one bench step), the last one a UC step with one Monte-Carlo forward: 6 denoiser forwards
TRUNC="--precision $PREC --generation-steps 5 --steps 1 --start-step-uc 4 --num-steps-uc 1 -M 1 --num-zigzag 1 --warmup 0 --no-cpu-baseline --no-profile --no-secondary --no-check"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace -d $OUT/${TAG}_pmc_$C -o ${TAG} -- python3 bench.py $TRUNC > $OUT/${TAG}_pmc_$C.json 2> $OUT/${TAG}_pmc_$C.err
done
python3 scripts/summarize_pmc.py pmc $OUT ${TAG} $PREC > $OUT/${TAG}_pmc_traffic_conv.json
echo done
