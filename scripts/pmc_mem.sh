#!/bin/bash
# Memory-side counters of the kernels whose name contains PATTERN, over one command (run on the GPU box through gpurun; separate --pmc
# passes with --kernel-trace only, as MI355X_MICROARCH.md prescribes):
#   bash scripts/pmc_mem.sh PATTERN python3 scripts/bench_conv_mix.py --fmt f16x3w --only 128x256x256 --reps 2 > gpurun_out/rXX_pmc_mem_<what>.json
# FETCH_SIZE / WRITE_SIZE are reported in KB; FETCH_SIZE is doubled in `derived` (gfx950 tallies the 128-B requests of a streaming read
# at 64 B).  A counter the part does not offer is skipped.
export TMPDIR=/tmp
PAT=$1; shift
OUT=/tmp/pmcmem
rm -rf $OUT
I=0
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCC_WRITE_sum TCC_EA0_WRREQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "GRBM_GUI_ACTIVE"; do
  I=$((I+1))
  rocprofv3 --pmc $C --kernel-trace -d $OUT/pass$I -o p -- "$@" > /dev/null 2>&1 || true
done
PAT="$PAT" CMD="$*" python3 - <<'PY'
import sqlite3, glob, json, os
pat = os.environ["PAT"]
res = {}
for db in sorted(glob.glob('/tmp/pmcmem/**/*.db', recursive=True)):
    c = sqlite3.connect(db)
    try:
        rows = list(c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like ? "
                              "group by kernel_name, counter_name", (f"%{pat}%",)))
    except sqlite3.Error:
        continue
    for k, n, v, cnt in rows:
        res.setdefault(k.split('(')[0].replace('void ', ''), {})[n] = dict(mean_per_launch=v, samples=cnt)
    for k, d, cnt in c.execute("select name, avg(duration), count(*) from kernels where name like ? group by name", (f"%{pat}%",)):
        res.setdefault(k.split('(')[0].replace('void ', ''), {})["mean_duration_ns"] = d
for k, d in res.items():
    g = lambda n: d.get(n, {}).get("mean_per_launch")
    der = {}
    if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
        rd, wr = 2 * g("FETCH_SIZE") * 1024, g("WRITE_SIZE") * 1024
        der.update(hbm_side_read_bytes=rd, hbm_side_write_bytes=wr, hbm_side_tb_per_s=round((rd + wr) / d["mean_duration_ns"] / 1e3, 3))
    if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None:
        der["l2_hit_rate"] = round(g("TCC_HIT_sum") / max(1.0, g("TCC_HIT_sum") + g("TCC_MISS_sum")), 4)
    if g("TCC_REQ_sum") is not None:
        der["l2_requests_per_ns"] = round(g("TCC_REQ_sum") / d["mean_duration_ns"], 3)
    if g("GRBM_GUI_ACTIVE") is not None:
        der["clock_ghz"] = round(g("GRBM_GUI_ACTIVE") / 8 / d["mean_duration_ns"], 3)
    d["derived"] = der
print(json.dumps(dict(command=os.environ["CMD"], kernels=res), indent=1))
PY
