#!/bin/bash
# U-ViT job: stream chunks for every batch (default) against only for batches up to 64 (the plain steps)
O=gpurun_out/r05R; mkdir -p $O
for rep in 1 2 3; do
echo "== all" >> $O/maxb_ab.txt; timeout -k 10 300 python scripts/diag/uvit_job.py 5 2>/dev/null | tail -1 | cut -c1-200 >> $O/maxb_ab.txt
echo "== maxb64" >> $O/maxb_ab.txt; DUA_UVIT_STREAMS_MAXB=64 timeout -k 10 300 python scripts/diag/uvit_job.py 5 2>/dev/null | tail -1 | cut -c1-200 >> $O/maxb_ab.txt
done
grep "==\|value" $O/maxb_ab.txt | paste - - | sed 's/{"config.*"value"/value/' | cut -c1-80
