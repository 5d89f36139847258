#!/bin/bash
# usage: pmc_cycle_mem.sh tag   ; HBM read / write counters of every kernel of scratch/cycle_trace.py (separate passes)
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmcc_$tag
for pass in "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $pass | cut -d' ' -f1)
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $R/gpurun_out/pmcc_$tag/$name -o p -- python3 $R/scratch/cycle_trace.py > $R/gpurun_out/pmcc_$tag/$name.log 2>&1 || echo "pass $name failed"
done
