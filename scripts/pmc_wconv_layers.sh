#!/bin/bash
# Per-launch HBM-side traffic of the Winograd conv kernel over the ADM-128 layer mix (run through gpurun from the repo root):
#   scripts/pmc_wconv_layers.sh r03b
# Separate --pmc passes (FETCH_SIZE, WRITE_SIZE) over scripts/bench_conv_mix.py --fmt f16x3w --reps 1; the launches of k_wconv are
# listed in dispatch order (two per layer line: the checked call and the timed one).
set -e -o pipefail
TAG=${1:-rXX}
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace -d $OUT/${TAG}_wl_$C -o ${TAG} -- python3 scripts/bench_conv_mix.py --fmt f16x3w --reps 1 > $OUT/${TAG}_wl_$C.log 2>&1
done
python3 - "$OUT" "$TAG" <<'PY'
import glob, sqlite3, sys
out, tag = sys.argv[1], sys.argv[2]
vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    db = glob.glob(f"{out}/{tag}_wl_{c}/**/*.db", recursive=True)[0]
    con = sqlite3.connect(db)
    rows = con.execute("select dispatch_id, kernel_name, value from counters_collection where counter_name = ? order by dispatch_id", (c,)).fetchall()
    vals[c] = [(d, v) for d, n, v in rows if "k_wconv" in n]
print("launch  FETCH_GB(x2)  WRITE_GB")
for (d, f), (_, w) in zip(vals["FETCH_SIZE"], vals["WRITE_SIZE"]):
    print(f"{d:6d}  {2 * f * 1024 / 1e9:10.3f}  {w * 1024 / 1e9:8.3f}")
PY
