#!/bin/bash
# headline job: eager launches against the forward replayed from a HIP graph (captured in the warm-up), alternating
O=gpurun_out/r05D; mkdir -p $O
A="--steps 10 --warmup 10 --no-cpu-baseline --no-secondary --no-other-configs --no-check --no-profile"
for rep in 1 2; do
  timeout -k 10 400 python bench.py $A 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('eager', d['value'])" >> $O/graph_ab.txt
  DUA_ADM_HIP_GRAPH=1 timeout -k 10 400 python bench.py $A 2>$O/graph_err.txt | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('graph', d['value'])" >> $O/graph_ab.txt
done
cat $O/graph_ab.txt; tail -3 $O/graph_err.txt
