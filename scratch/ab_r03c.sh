#!/bin/bash
# RECORD of a withdrawn experiment (profiles/README.md, "Withdrawn in round 3"): the switches and the second library it names
# belong to code that was taken out of the tree again; kept for the exact commands behind profiles/r03_f_withdrawn_experiments.txt.
# same-box A/B: one z-tile per XCD (a single round of workgroups) against the graded tiles; skewed XCD runs of the node kernels
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
cd $R
O=gpurun_out/ab2; mkdir -p $O
cp mfmg_amd/libmfmg_hip.so /tmp/new.so
for lib in new old; do
  [ $lib = old ] && cp scratch/ab/libmfmg_hip_old_op.so mfmg_amd/libmfmg_hip.so
  echo "== $lib: default tiles" | tee -a $O/summary.txt
  timeout -k 10 200 python scratch/op_time.py 257:constant 257:constant:3,10,4 2>&1 | grep '^{' | tee -a $O/summary.txt
  echo "== $lib: MFMG_MF_GRADE_M=0 (tz 32: one z-tile per XCD)" | tee -a $O/summary.txt
  MFMG_MF_GRADE_M=0 timeout -k 10 300 python scratch/op_time.py 257:constant:3,32,4 257:constant:4,32,4 257:constant:2,32,4 257:constant:4,32,2 257:constant:3,32,3 257:constant:3,16,4 257:constant:3,32,8 2>&1 | grep '^{' | tee -a $O/summary.txt
  cp /tmp/new.so mfmg_amd/libmfmg_hip.so
done
echo "== linear: default and one z-tile per XCD" | tee -a $O/summary.txt
timeout -k 10 200 python scratch/op_time.py 257:linear 2>&1 | grep '^{' | tee -a $O/summary.txt
MFMG_MF_GRADE_M=0 timeout -k 10 200 python scratch/op_time.py 257:linear:3,32,4 257:linear:2,32,4 257:linear:2,32,8 2>&1 | grep '^{' | tee -a $O/summary.txt
echo "== cycle sequence, plain order" | tee -a $O/summary.txt
bash scratch/cycle_seq.sh ab2/seq_xcd0 256 constant > $O/seq_xcd0.log 2>&1; grep "us/cycle" $O/seq_xcd0.log | tee -a $O/summary.txt
echo "== cycle sequence, skewed XCD runs" | tee -a $O/summary.txt
MFMG_HIP_XCD_RUNS=1 bash scratch/cycle_seq.sh ab2/seq_xcd1 256 constant > $O/seq_xcd1.log 2>&1; grep "us/cycle" $O/seq_xcd1.log | tee -a $O/summary.txt
echo "== linear, skewed XCD runs" | tee -a $O/summary.txt
MFMG_HIP_XCD_RUNS=1 bash scratch/cycle_seq.sh ab2/seql_xcd1 256 linear > $O/seql_xcd1.log 2>&1; grep "us/cycle" $O/seql_xcd1.log | tee -a $O/summary.txt
