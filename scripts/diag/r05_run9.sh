#!/bin/bash
# U-ViT: GEMM mixes at batch 32 / 64 / 160 / 320 and a single-stream job under rocprofv3 (exclusive kernel times)
set -o pipefail
O=gpurun_out/r05r; mkdir -p $O
export TMPDIR=/tmp
for s in uvit64 uvit320; do timeout -k 10 200 python scripts/bench_gemm_mix.py --set $s 2>/dev/null | grep -v amdgpu > $O/mix_$s.txt; tail -8 $O/mix_$s.txt; done
DUA_UVIT_STREAMS=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/uvit1 -o uvit1 -- python3 scripts/diag/uvit_job.py 5 > $O/uvit1_job.json 2> $O/uvit1_job.err
tail -1 $O/uvit1_job.json | cut -c1-300
python3 scripts/summarize_pmc.py stats $(find $O/uvit1 -name "*.db" | head -1) > $O/uvit1_kernel_stats.csv
head -14 $O/uvit1_kernel_stats.csv | cut -c1-150
