#!/bin/bash
# round 6, final state: full GPU test suite, the driver's bench invocation, the rocprof evidence (kernel stats of the headline job, HBM-side
# traffic of the conv launches from separate --pmc passes), the launcher form on one GPU (RCCL initialised), U-ViT job on one stream under rocprofv3
set -o pipefail
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
python -m pytest tests -m gpu -x -q > $OUT/r06z_tests.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/r06z_tests.log
python bench.py --steps 20 --warmup 5 > $OUT/r06z_bench_driver_style_steps20_warmup5.log 2> $OUT/r06z_bench.err; echo "bench rc=$?"
cp $OUT/bench_legs.json $OUT/r06z_bench_driver_style_full_record.json
tail -1 $OUT/r06z_bench_driver_style_steps20_warmup5.log > $OUT/r06z_bench_driver_style_steps20_warmup5.json
