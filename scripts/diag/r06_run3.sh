#!/bin/bash
# round 6, final state (second call): the rocprof evidence (kernel stats of the headline job, HBM-side traffic of the conv launches from
# separate --pmc passes), the launcher form on one GPU (RCCL initialised), the U-ViT job on one stream under rocprofv3
set -o pipefail
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
bash scripts/collect_profiles.sh r06z; echo "collect rc=$?"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 1 --no-secondary --no-cpu-baseline > $OUT/r06z_torchrun1.log 2>&1; echo "torchrun rc=$?"
tail -1 $OUT/r06z_torchrun1.log > $OUT/r06z_torchrun1.json
DUA_UVIT_STREAMS=1 rocprofv3 --kernel-trace --stats -d $OUT/r06z_uvit_stats -o r06z -- python3 scripts/diag/uvit_job.py 5 > $OUT/r06z_uvit_job_single_stream.txt 2>&1
python3 scripts/summarize_pmc.py stats $(find $OUT/r06z_uvit_stats -name "*.db" | head -1) > $OUT/r06z_uvit_job_single_stream_kernel_stats.csv; echo "uvit stats rc=$?"
