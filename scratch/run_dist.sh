#!/bin/bash
# usage: run_dist.sh world mesh  -- the shared-GPU distributed worker, output to gpurun_out/dist_<world>_<mesh>.log
w=$1; m=$2
OMP_NUM_THREADS=2 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node=$w --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 500)) tests/dist_worker.py --mode gpu --mesh $m > gpurun_out/dist_${w}_${m}.log 2>&1
echo "rc $? for $w $m"; grep -v "^\[W\|warn" gpurun_out/dist_${w}_${m}.log | tail -25
