#!/bin/bash
# deep low ghosts (BoxPartition low_ghost_cells=4): the shared-GPU and thread tests, then a rank's share of the 2x2x2 cycle
R=${GRAFT_REPO_ROOT:?}
cd $R
timeout -k 10 900 python -m pytest tests/test_distributed.py tests/test_box_threads.py -x -q -m gpu -k "eight_ranks and constant" > gpurun_out/s2_c4_tests.log 2>&1 || { tail -40 gpurun_out/s2_c4_tests.log; exit 1; }
tail -2 gpurun_out/s2_c4_tests.log
AMG_REPLICATE_ROWS=20000 LOW_GHOST=4 timeout -k 10 400 python scratch/rank_cycle_on_one_gpu.py 256 2,2,2 7 > gpurun_out/s2_c4_rank7.log 2>&1 || { tail -20 gpurun_out/s2_c4_rank7.log; exit 1; }
tail -1 gpurun_out/s2_c4_rank7.log | cut -c1-1500
AMG_REPLICATE_ROWS=20000 LOW_GHOST=4 timeout -k 10 400 python scratch/rank_cycle_on_one_gpu.py 256 2,2,2 0 > gpurun_out/s2_c4_rank0.log 2>&1 || { tail -20 gpurun_out/s2_c4_rank0.log; exit 1; }
tail -1 gpurun_out/s2_c4_rank0.log | cut -c1-1500
