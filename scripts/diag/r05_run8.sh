#!/bin/bash
# quantile kernel: several workgroups per sample - parity + timing + phase stamps
set -o pipefail
mkdir -p gpurun_out/r05p
timeout -k 10 420 python -m pytest tests/test_gpu_ops.py -x -q -k "quantile" -s > gpurun_out/r05p/quantile_tests.txt 2>&1
rc=$?
tail -5 gpurun_out/r05p/quantile_tests.txt
grep "quantile_mask (" gpurun_out/r05p/quantile_tests.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python scripts/diag/quantile_diag.py > gpurun_out/r05p/quantile_diag.txt 2>&1 || exit 1
cat gpurun_out/r05p/quantile_diag.txt
