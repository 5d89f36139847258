#!/bin/bash
# usage: pmc_cycle.sh tag   ; counters of every kernel of scratch/cycle_trace.py in separate --pmc passes (kernel-trace only)
# (FETCH_SIZE takes 3 of the 4 TCC slots and WRITE_SIZE 2, MI355X_MICROARCH.md "rocprofv3 PMC slots": a pass of their own each)
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmcc_$tag
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU" "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $R/gpurun_out/pmcc_$tag/$name -o p -- python3 $R/scratch/cycle_trace.py > $R/gpurun_out/pmcc_$tag/$name.log 2>&1 || echo "pass $name failed"
done
