#!/bin/bash
# XCD strips of the class kernel A/B (same box), parity at full size, setup phases at 513^3, kernel sequence of rank 7 (deep ghosts)
R=${GRAFT_REPO_ROOT:?}
cd $R
MFMG_XCD_STRIPS=0 bash scratch/cycle_seq.sh s2_seq_lex 256 constant > gpurun_out/s2_c5_seq_lex.log 2>&1 || { tail -5 gpurun_out/s2_c5_seq_lex.log; exit 1; }
bash scratch/cycle_seq.sh s2_seq_strips 256 constant > gpurun_out/s2_c5_seq_strips.log 2>&1 || { tail -5 gpurun_out/s2_c5_seq_strips.log; exit 1; }
echo "== lexicographic"; head -12 gpurun_out/s2_seq_lex_sequence.txt
echo "== strips"; head -12 gpurun_out/s2_seq_strips_sequence.txt
bash scratch/cycle_seq.sh s2_seq_linear 256 linear > gpurun_out/s2_c5_seq_linear.log 2>&1 || { tail -5 gpurun_out/s2_c5_seq_linear.log; exit 1; }
echo "== linear"; head -12 gpurun_out/s2_seq_linear_sequence.txt
cd $R
timeout -k 10 500 python -m pytest tests/test_gpu_hierarchy.py -x -q -k "full_size_vcycle and constant" > gpurun_out/s2_c5_tests.log 2>&1 || { tail -30 gpurun_out/s2_c5_tests.log; exit 1; }
tail -2 gpurun_out/s2_c5_tests.log
MFMG_HIP_VERBOSE=1 timeout -k 10 300 python scratch/setup_profile.py 512 constant > gpurun_out/s2_c5_setup512.log 2>&1 || { tail -20 gpurun_out/s2_c5_setup512.log; exit 1; }
grep -v "^$" gpurun_out/s2_c5_setup512.log | cut -c1-260 | tail -60
