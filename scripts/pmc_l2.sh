export TMPDIR=/tmp
for C in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
  rocprofv3 --pmc $C --kernel-trace -d gpurun_out/l2pmc -o l2 -- python3 scripts/bench_conv_mix.py --fmt f16x3w --only 128x256x256 --reps 2 > /dev/null 2>&1
  python3 - <<'PY'
import sqlite3,glob
db=sorted(glob.glob('gpurun_out/l2pmc/**/*.db',recursive=True))[-1]
c=sqlite3.connect(db)
rows=c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%k_wconv%' group by kernel_name, counter_name").fetchall()
for r in rows: print(r[0][:40], r[1], f"{r[2]:.4g}", r[3])
PY
  rm -rf gpurun_out/l2pmc
done
