#!/bin/bash
# RECORD of a withdrawn experiment (profiles/README.md, "Withdrawn in round 3"): the switches and the second library it names
# belong to code that was taken out of the tree again; kept for the exact commands behind profiles/r03_f_withdrawn_experiments.txt.
# Round 3, second half: same-box A/B of (1) the operator kernel in mode space against the corner form (library built from
# the previous source of mf_laplace.hip, scratch/ab/libmfmg_hip_old_op.so) and (2) XCD-contiguous block runs of the node
# kernels (MFMG_HIP_XCD_RUNS=0 switches them off).   usage (on the GPU box): bash scratch/ab_r03b.sh
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
cd $R
O=gpurun_out/ab; mkdir -p $O
cp mfmg_amd/libmfmg_hip.so /tmp/new.so
echo "== tests" | tee $O/summary.txt
timeout -k 10 500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_hierarchy.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?" | tee -a $O/summary.txt; tail -3 $O/tests.log | tee -a $O/summary.txt
grep -q "failed\|error" $O/tests.log && exit 1
echo "== operator kernel, new" | tee -a $O/summary.txt
timeout -k 10 200 python scratch/op_time.py 257:constant 512:constant 2>&1 | tee -a $O/summary.txt
echo "== operator kernel, old" | tee -a $O/summary.txt
cp scratch/ab/libmfmg_hip_old_op.so mfmg_amd/libmfmg_hip.so
timeout -k 10 200 python scratch/op_time.py 257:constant 512:constant 2>&1 | tee -a $O/summary.txt
echo "== operator kernel, new again" | tee -a $O/summary.txt
cp /tmp/new.so mfmg_amd/libmfmg_hip.so
timeout -k 10 200 python scratch/op_time.py 257:constant 512:constant 2>&1 | tee -a $O/summary.txt
echo "== cycle sequence, XCD runs on" | tee -a $O/summary.txt
bash scratch/cycle_seq.sh ab/seq_xcd1 256 constant > $O/seq_xcd1.log 2>&1; head -14 $O/seq_xcd1.log | tee -a $O/summary.txt
echo "== cycle sequence, XCD runs off" | tee -a $O/summary.txt
MFMG_HIP_XCD_RUNS=0 bash scratch/cycle_seq.sh ab/seq_xcd0 256 constant > $O/seq_xcd0.log 2>&1; head -14 $O/seq_xcd0.log | tee -a $O/summary.txt
echo "== linear material, XCD runs on / off" | tee -a $O/summary.txt
bash scratch/cycle_seq.sh ab/seql_xcd1 256 linear > $O/seql_xcd1.log 2>&1; head -12 $O/seql_xcd1.log | tee -a $O/summary.txt
MFMG_HIP_XCD_RUNS=0 bash scratch/cycle_seq.sh ab/seql_xcd0 256 linear > $O/seql_xcd0.log 2>&1; head -12 $O/seql_xcd0.log | tee -a $O/summary.txt
