#!/bin/bash
# usage: pmc.sh tag n ty tz   (env: WAVES, MFMG_MF_VARIANT) ; separate --pmc passes, kernel-trace only
set -e
tag=$1; n=$2; ty=$3; tz=$4
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
mkdir -p "$R/gpurun_out/pmc_$tag"
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $R/gpurun_out/pmc_$tag/$name -o p -- python3 $R/scratch/smoother_only.py $n $ty $tz 2 > $R/gpurun_out/pmc_$tag/$name.log 2>&1 || echo "pass $name failed"
done
