#!/bin/bash
# Which SpMV layouts do the level matrices of a box rank get?  4 ranks sharing one GPU (gloo), rank 0's setup log,
# slabs beside boxes.  Usage: scratch/box_layouts.sh [cells per rank]
C=${1:-128}
export MFMG_BENCH_BACKEND=gloo MFMG_HIP_VERBOSE=1
for P in slab box; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29551 bench.py --gpus 4 --cells $C --steps 2 --warmup 1 --no-extras --partition $P > gpurun_out/lay_$P.log 2>&1
  echo "== $P"; grep "\[mfmg_hip\] matrix" gpurun_out/lay_$P.log | sort | uniq -c | sort -k5,5n -k1,1n | cut -c1-330 | head -40
done
