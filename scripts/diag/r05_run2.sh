set -x
O=gpurun_out/r05g; mkdir -p $O
for L in "" "--lib scripts/diag/build/libdua_hip_prev_hgemm.so" "" "--lib scripts/diag/build/libdua_hip_prev_hgemm.so"; do
  echo "== lib=$L" >> $O/hgemm_epi1_ab.txt
  timeout -k 10 200 python scripts/bench_gemm_mix.py --set uvit64 $L 2>/dev/null | grep -v amdgpu >> $O/hgemm_epi1_ab.txt
  timeout -k 10 200 python scripts/bench_gemm_mix.py --set uvit320 $L 2>/dev/null | grep -v amdgpu >> $O/hgemm_epi1_ab.txt
done
cat $O/hgemm_epi1_ab.txt | grep "==\|fc1\|mix"
for G in 1 2 5 1 2 5; do
  timeout -k 10 300 python bench.py --steps 10 --warmup 1 --mc-group $G --no-secondary --no-cpu-baseline --no-other-configs --no-check --legs-file $O/mc_group_${G}.json 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mc_group $G', d['value'], d['ms_per_step'], d['roofline']['achieved'])" >> $O/mc_group_ab.txt
done
cat $O/mc_group_ab.txt
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 1 --no-secondary --no-cpu-baseline --no-other-configs --legs-file $O/torchrun1_legs.json > $O/torchrun1.txt 2> $O/torchrun1.err; tail -n 1 $O/torchrun1.txt | cut -c1-600
