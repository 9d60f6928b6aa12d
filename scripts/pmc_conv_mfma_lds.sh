#!/bin/bash
# MFMA-pipe and LDS counters of the conv kernel on one ADM-128 layer (run on the GPU box through gpurun from the repo root):
#   bash scripts/pmc_conv_mfma_lds.sh [fmt=f16x3] [layer=128x256x256] [engine=1] > gpurun_out/rXX_pmc_mfma_lds.json
# Separate --pmc passes per counter group (rocprofv3 --pmc with --kernel-trace only, see collect_profiles.sh).
export TMPDIR=/tmp
export FMT=${1:-f16x3} LAYER=${2:-128x256x256} ENGINE=${3:-1}
OUT=/tmp/pmcconv
rm -rf $OUT
I=0
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE"; do
  I=$((I+1))
  rocprofv3 --pmc $C --kernel-trace -d $OUT/pass$I -o p -- python3 scripts/bench_conv_mix.py --fmt $FMT --only $LAYER --engine $ENGINE --reps 2 > /dev/null 2>&1
done
python3 - <<'PY'
import sqlite3, glob, json, os
res = {}
for db in sorted(glob.glob('/tmp/pmcconv/**/*.db', recursive=True)):
    c = sqlite3.connect(db)
    for k, n, v, cnt in c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                                  "where kernel_name like '%k_xconv_dma%' or kernel_name like '%k_hconv%' group by kernel_name, counter_name"):
        res.setdefault(k.split('(')[0].replace('void ', ''), {})[n] = dict(mean_per_launch=v, samples=cnt)
    for k, d, cnt in c.execute("select name, avg(duration), count(*) from kernels where name like '%k_xconv_dma%' or name like '%k_hconv%' group by name"):
        res.setdefault(k.split('(')[0].replace('void ', ''), {})["mean_duration_ns"] = d
for k, d in res.items():
    try:      # derived: GRBM_GUI_ACTIVE sums the 8 XCDs; MFMA busy cycles sum the 1024 SIMDs; LDS cycles sum the 256 CUs
        cyc = d["GRBM_GUI_ACTIVE"]["mean_per_launch"] / 8
        d["derived"] = dict(clock_ghz=round(cyc / d["mean_duration_ns"], 3),
                            mfma_pipe_busy_frac=round(d["SQ_VALU_MFMA_BUSY_CYCLES"]["mean_per_launch"] / (cyc * 1024), 3),
                            lds_array_busy_frac=round(d["SQ_LDS_IDX_ACTIVE"]["mean_per_launch"] / (cyc * 256), 3),
                            lds_bank_conflict_cycles=d["SQ_LDS_BANK_CONFLICT"]["mean_per_launch"],
                            wave_wait_frac=round(d["SQ_WAIT_INST_ANY"]["mean_per_launch"] / d["SQ_WAVE_CYCLES"]["mean_per_launch"], 3))
    except KeyError:
        pass
E = os.environ
print(json.dumps(dict(command=f"rocprofv3 --pmc <group> --kernel-trace -- python3 scripts/bench_conv_mix.py --fmt {E['FMT']} --only {E['LAYER']} --engine {E['ENGINE']} --reps 2",
                      layer=f"conv3x3 HxCinxCout = {E['LAYER']}, batch 128, with and without residual ({E['FMT']})", kernels=res), indent=1))
PY
