#!/bin/bash
# rehearsal of `bench.py --gpus N` with gloo ranks sharing the one GPU of a gpurun box, and the same global box on one rank
N=$1; CELLS=${2:-64}; REP=${3:-5000}
gz=$((CELLS)); gy=$CELLS; gx=$CELLS
i=0; n=$N; while [ $n -gt 1 ]; do case $((i % 3)) in 0) gz=$((gz*2));; 1) gy=$((gy*2));; 2) gx=$((gx*2));; esac; n=$((n/2)); i=$((i+1)); done
MFMG_BENCH_BACKEND=gloo OMP_NUM_THREADS=2 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node=$N --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) bench.py --gpus $N --cells $CELLS --steps 5 --warmup 2 --amg-replicate-rows $REP > gpurun_out/rehearse_n$N.log 2>&1
echo "rc $?"; grep '"metric"' gpurun_out/rehearse_n$N.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['n_gpus'], d['ms_per_step'], d['config']['mean_residual_contraction_per_cycle'], d['config']['parallelism'])"
timeout -k 10 300 python bench.py --cells $CELLS --box $gx,$gy,$gz --steps 5 --warmup 2 --no-extras --no-cpu-baseline --no-smoother-512 > gpurun_out/rehearse_n${N}_one_rank.log 2>&1
echo "rc $?"; grep '"metric"' gpurun_out/rehearse_n${N}_one_rank.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['n_gpus'], d['ms_per_step'], d['config']['mean_residual_contraction_per_cycle'])"
