#!/bin/bash
# MFMA / VALU / LDS / wait counters of the kernels whose name contains PATTERN, over one command (run on the GPU box through gpurun):
#   bash scripts/pmc_kernel.sh PATTERN python3 scripts/diag/<driver>.py args... > gpurun_out/rXX_pmc_<what>.json
export TMPDIR=/tmp
PAT=$1; shift
OUT=/tmp/pmckern
rm -rf $OUT
I=0
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_INSTS_LDS" "GRBM_GUI_ACTIVE"; do
  I=$((I+1))
  rocprofv3 --pmc $C --kernel-trace -d $OUT/pass$I -o p -- "$@" > /dev/null 2>&1
done
PAT="$PAT" CMD="$*" python3 - <<'PY'
import sqlite3, glob, json, os
pat = os.environ["PAT"]
res = {}
for db in sorted(glob.glob('/tmp/pmckern/**/*.db', recursive=True)):
    c = sqlite3.connect(db)
    for k, n, v, cnt in c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                                  "where kernel_name like ? group by kernel_name, counter_name", (f"%{pat}%",)):
        res.setdefault(k.split('(')[0].replace('void ', ''), {})[n] = dict(mean_per_launch=v, samples=cnt)
    for k, d, cnt in c.execute("select name, avg(duration), count(*) from kernels where name like ? group by name", (f"%{pat}%",)):
        res.setdefault(k.split('(')[0].replace('void ', ''), {})["mean_duration_ns"] = d
for k, d in res.items():
    try:
        cyc = d["GRBM_GUI_ACTIVE"]["mean_per_launch"] / 8
        d["derived"] = dict(clock_ghz=round(cyc / d["mean_duration_ns"], 3),
                            mfma_pipe_busy_frac=round(d["SQ_VALU_MFMA_BUSY_CYCLES"]["mean_per_launch"] / (cyc * 1024), 3),
                            valu_active_frac=round(d["SQ_ACTIVE_INST_VALU"]["mean_per_launch"] / (cyc * 1024), 3),
                            lds_array_busy_frac=round(d["SQ_LDS_IDX_ACTIVE"]["mean_per_launch"] / (cyc * 256), 3),
                            lds_bank_conflict_frac_of_lds=round(d["SQ_LDS_BANK_CONFLICT"]["mean_per_launch"] / max(d["SQ_LDS_IDX_ACTIVE"]["mean_per_launch"], 1), 3),
                            wave_wait_frac=round(d["SQ_WAIT_INST_ANY"]["mean_per_launch"] / d["SQ_WAVE_CYCLES"]["mean_per_launch"], 3))
    except KeyError as e:
        d["derived_error"] = str(e)
print(json.dumps(dict(command=os.environ["CMD"], kernels=res), indent=1))
PY
