#!/bin/bash
# headline job: product library against -DDUA_NO_RANGE_CHECK (all translation units), alternating, one job each
O=gpurun_out/r05w; mkdir -p $O
A="--steps 10 --warmup 5 --no-cpu-baseline --no-secondary --no-other-configs --no-check --no-profile"
for rep in 1 2; do
  timeout -k 10 400 python bench.py $A 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('product', d['value'])" >> $O/norange_headline.txt
  timeout -k 10 400 python scripts/diag/bench_with_lib.py scripts/diag/build/libdua_hip_norange.so $A 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('norange', d['value'])" >> $O/norange_headline.txt
done
cat $O/norange_headline.txt
