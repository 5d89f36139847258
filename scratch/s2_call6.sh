#!/bin/bash
# rehearsal of bench.py --gpus 2 / 4 with the new default (low ghost 4) on gloo ranks sharing the GPU; kernel sequence of rank 7
R=${GRAFT_REPO_ROOT:?}
cd $R
bash scratch/rehearse_bench.sh 2 64 || exit 1
bash scratch/rehearse_bench.sh 4 64 || exit 1
export TRACE=1 AMG_REPLICATE_ROWS=20000 LOW_GHOST=4
bash scratch/rank_cycle_trace.sh s2_rank7 256 2,2,2 7 > gpurun_out/s2_rank7_trace.log 2>&1 || { tail gpurun_out/s2_rank7_trace.log; exit 1; }
head -24 gpurun_out/s2_rank7_sequence.txt
unset TRACE
DELAY_US=5,10,20,40 timeout -k 10 400 python scratch/rank_cycle_on_one_gpu.py 256 2,2,2 7 > gpurun_out/s2_c6_rank7_delays.log 2>&1 || { tail gpurun_out/s2_c6_rank7_delays.log; exit 1; }
tail -1 gpurun_out/s2_c6_rank7_delays.log | cut -c1-400; tail -1 gpurun_out/s2_c6_rank7_delays.log | grep -o '"ms_per_cycle_rank_alone_with_wire_latency.*'
