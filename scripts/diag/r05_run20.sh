#!/bin/bash
# U-ViT job: number of stream chunks with the tile width chosen by work
O=gpurun_out/r05R; mkdir -p $O
for rep in 1 2; do for n in 2 3 4 1; do
echo "== streams $n" >> $O/streams_ab.txt; DUA_UVIT_STREAMS=$n timeout -k 10 300 python scripts/diag/uvit_job.py 5 2>/dev/null | tail -1 | cut -c1-200 >> $O/streams_ab.txt
done; done
grep "==\|value" $O/streams_ab.txt | paste - - | sed 's/{"config.*"value"/value/' | cut -c1-80
