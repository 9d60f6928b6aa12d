#!/bin/bash
# U-ViT job: tile widths by whole rounds (DUA_HG_CONC=0) against by work while the batch runs in chunks on two streams; then the U-ViT / hgemm tests
O=gpurun_out/r05E; mkdir -p $O
for rep in 1 2; do
echo "== rounds" >> $O/conc_ab.txt; DUA_HG_CONC=0 timeout -k 10 300 python scripts/diag/uvit_job.py 5 2>/dev/null | tail -1 | cut -c1-200 >> $O/conc_ab.txt
echo "== work" >> $O/conc_ab.txt; timeout -k 10 300 python scripts/diag/uvit_job.py 5 2>/dev/null | tail -1 | cut -c1-200 >> $O/conc_ab.txt
done
grep "==\|value" $O/conc_ab.txt
