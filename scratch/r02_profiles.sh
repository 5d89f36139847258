#!/bin/bash
# round-2 profiles: kernel stats of the default bench + PMC passes of the operator kernel (separate passes, kernel trace only)
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/prof_r02
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02 -o p -- python3 $R/bench.py --no-cpu-baseline --no-extras > $R/gpurun_out/prof_r02/bench.log 2>&1 || echo "stats run failed"
for cfg in "257 constant" "257 linear" "512 constant" "512 linear"; do
  set -- $cfg
  echo "pmc $1 $2"
  MATERIAL=$2 WAVES=0 bash $R/scratch/pmc.sh r02_$1_$2 $1 0 0 || echo "pmc $cfg failed"
done
cd $R && for cfg in 257_constant 257_linear 512_constant 512_linear; do echo "== $cfg"; python3 scratch/pmc_sum.py r02_$cfg 6; done > gpurun_out/pmc_r02_summary.txt 2>&1
cat gpurun_out/pmc_r02_summary.txt
