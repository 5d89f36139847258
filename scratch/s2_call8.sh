#!/bin/bash
# where the one-operator prolongation pays on a distributed rank: all levels (default) / like one rank / nowhere
R=${GRAFT_REPO_ROOT:?}
cd $R
for v in "MFMG_AMG_SMOOTHED_DISTRIBUTED=0" "MFMG_AMG_SMOOTHED_PROLONGATION=0"; do
  env $v AMG_REPLICATE_ROWS=20000 LOW_GHOST=4 DELAY_US=20 timeout -k 10 400 python scratch/rank_cycle_on_one_gpu.py 256 2,2,2 7 > gpurun_out/s2_c8_$v.log 2>&1 || { tail -20 gpurun_out/s2_c8_$v.log; exit 1; }
  echo "== $v"; tail -1 gpurun_out/s2_c8_$v.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_cycle_rank_alone_reflecting'], d['exchanges_per_cycle'], d['ms_per_cycle_rank_alone_with_wire_latency'], d['ms_per_cycle_one_rank_same_size'])"
done
