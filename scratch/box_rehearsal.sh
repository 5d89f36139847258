#!/bin/bash
# Rehearsal of bench.py on ranks that share one GPU (gloo, host transport): the distributed cycle -- slabs and boxes -- must
# contract like the one-rank cycle on the same global mesh.
# Usage: scratch/box_rehearsal.sh [cells per rank] [ranks] [grid px,py,pz] [global cells gx,gy,gz]
C=${1:-128}; N=${2:-4}; GRID=${3:-}; BOX=${4:-}
export MFMG_BENCH_BACKEND=gloo
show() { tail -1 $1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
c=d['config']
print('$2', 'GDoF/s %.3f' % (d['value']/1e9), 'ms %.3f' % d['ms_per_step'], 'contraction %.12f' % c['mean_residual_contraction_per_cycle'], 'setup %.2f' % c['setup_seconds'], c['parallelism'][:330])"; }
G=${BOX:-$(python -c "
import math
g=[$C]*3
for i in range(int(round(math.log2($N)))): g[2 - i % 3]*=2
print(','.join(map(str,g)))")}
python bench.py --gpus 1 --cells $C --box $G --steps 5 --warmup 2 --no-extras > gpurun_out/reh_one.log 2>&1 && show gpurun_out/reh_one.log one
for P in slab box; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus $N --cells $C --steps 5 --warmup 2 --no-extras --partition $P ${BOX:+--box $BOX} $( [ $P = box ] && [ -n "$GRID" ] && echo --grid $GRID ) > gpurun_out/reh_$P.log 2>&1 && show gpurun_out/reh_$P.log $P
done
