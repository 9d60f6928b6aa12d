set -x
O=gpurun_out/r05j; mkdir -p $O
timeout -k 10 700 python bench.py --steps 20 --warmup 5 --legs-file $O/bench_full_record.json > $O/bench_stdout.txt 2> $O/bench.err; tail -n 1 $O/bench_stdout.txt | cut -c1-2200
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/uvit_stats -o r05j -- python3 scripts/diag/uvit_job.py 5 1 > $O/uvit_job.json 2> $O/uvit_job.err
python3 scripts/summarize_pmc.py stats $(find $O/uvit_stats -name "*.db" | head -1) > $O/r05j_uvit_job_kernel_stats.csv; head -8 $O/r05j_uvit_job_kernel_stats.csv | cut -c1-120; cat $O/uvit_job.json | cut -c1-300
rm -rf $O/uvit_stats
