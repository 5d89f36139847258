#!/bin/bash
# kernel sequence of one V-cycle (rocprofv3 kernel trace of scratch/cycle_trace.py); usage: cycle_seq.sh tag [cells] [material]
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
TAG=${1:-seq}; CELLS=${2:-256}; MAT=${3:-constant}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/$TAG
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$TAG -o t -- python3 $R/scratch/cycle_trace.py $CELLS $MAT > $R/gpurun_out/$TAG/run.log 2>&1 || { echo "trace failed"; tail -5 $R/gpurun_out/$TAG/run.log; exit 1; }
F=$(find $R/gpurun_out/$TAG -name "*kernel_trace.csv" | head -1)
python3 $R/scratch/cycle_trace_sum.py $F seq > $R/gpurun_out/${TAG}_sequence.txt
rm -f $F
tail -2 $R/gpurun_out/$TAG/run.log
cat $R/gpurun_out/${TAG}_sequence.txt
