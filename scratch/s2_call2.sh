#!/bin/bash
# session-2 call: parity of the masked coefficient loads, tiles of the eight-coefficient kernel, a rank's share of the 2x2x2 cycle
R=${GRAFT_REPO_ROOT:?}
cd $R
timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py -x -q -k "mf_" > gpurun_out/s2_c2_tests.log 2>&1 || { tail -20 gpurun_out/s2_c2_tests.log; exit 1; }
tail -2 gpurun_out/s2_c2_tests.log
timeout -k 10 300 python scratch/op_time.py 257:linear 257:linear:3,8,4 257:linear:3,16,4 257:linear:3,16,8 257:linear:2,16,8 257:linear:3,12,8 257:linear:4,8,4 512:linear 512:linear:3,16,8 512:linear:2,32,8 > gpurun_out/s2_c2_optime.log 2>&1 || { tail -20 gpurun_out/s2_c2_optime.log; exit 1; }
cat gpurun_out/s2_c2_optime.log
AMG_REPLICATE_ROWS=20000 timeout -k 10 500 python scratch/rank_cycle_on_one_gpu.py 256 2,2,2 0 > gpurun_out/s2_c2_rank.log 2>&1 || { tail -20 gpurun_out/s2_c2_rank.log; exit 1; }
tail -1 gpurun_out/s2_c2_rank.log | cut -c1-3000
