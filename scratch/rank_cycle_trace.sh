#!/bin/bash
# kernel trace of scratch/rank_cycle_on_one_gpu.py (the timed rank runs alone at the end of the process), summarised on the box
# usage: rank_cycle_trace.sh tag [cells_per_rank] [grid] [rank]      env AMG_REPLICATE_ROWS
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
TAG=${1:-rank}; CELLS=${2:-256}; GRID=${3:-2,2,2}; RANK=${4:-0}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/$TAG
timeout -k 10 1050 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$TAG -o t -- python3 $R/scratch/rank_cycle_on_one_gpu.py $CELLS $GRID $RANK > $R/gpurun_out/$TAG/run.log 2>&1 || { echo "trace failed"; tail -5 $R/gpurun_out/$TAG/run.log; exit 1; }
F=$(find $R/gpurun_out/$TAG -name "*kernel_trace.csv" | head -1)
python3 $R/scratch/rank_cycle_trace_sum.py $F seq > $R/gpurun_out/${TAG}_sequence.txt
rm -f $F
tail -1 $R/gpurun_out/$TAG/run.log | cut -c1-700
cat $R/gpurun_out/${TAG}_sequence.txt
