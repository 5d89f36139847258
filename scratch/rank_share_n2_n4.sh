cd $GRAFT_REPO_ROOT
for cfg in "1,1,2 1" "1,2,2 3"; do
  set -- $cfg
  AMG_REPLICATE_ROWS=20000 LOW_GHOST=4 DELAY_US=5,20 timeout -k 10 400 python scratch/rank_cycle_on_one_gpu.py 256 $1 $2 > gpurun_out/s2_c18_$1.log 2>&1 || { tail -5 gpurun_out/s2_c18_$1.log; exit 1; }
  tail -1 gpurun_out/s2_c18_$1.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['grid'], d['rank'], d['local_cells'], 'free', round(d['ms_per_cycle_rank_alone_reflecting'],3), 'one', round(d['ms_per_cycle_one_rank_same_size'],3), 'exch', d['exchanges_per_cycle'], d['ms_per_cycle_rank_alone_with_wire_latency'])"
done
