#!/bin/bash
# wider narrow last column (rest <= 32 - halo): bitwise sweep tests, distributed deep-ghost tests, a rank's share again
R=${GRAFT_REPO_ROOT:?}
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "sweep or tail_slab" > gpurun_out/s2_c7_tests.log 2>&1 || { tail -30 gpurun_out/s2_c7_tests.log; exit 1; }
tail -2 gpurun_out/s2_c7_tests.log
timeout -k 10 900 python -m pytest tests/test_distributed.py tests/test_box_threads.py tests/test_gpu_hierarchy.py -x -q -m gpu -k "two_ghost or (eight_ranks and constant) or sweep" > gpurun_out/s2_c7_tests2.log 2>&1 || { tail -40 gpurun_out/s2_c7_tests2.log; exit 1; }
tail -2 gpurun_out/s2_c7_tests2.log
AMG_REPLICATE_ROWS=20000 LOW_GHOST=4 DELAY_US=5,20 timeout -k 10 400 python scratch/rank_cycle_on_one_gpu.py 256 2,2,2 7 > gpurun_out/s2_c7_rank7.log 2>&1 || { tail -20 gpurun_out/s2_c7_rank7.log; exit 1; }
tail -1 gpurun_out/s2_c7_rank7.log | cut -c1-420; tail -1 gpurun_out/s2_c7_rank7.log | grep -o '"ms_per_cycle_rank_alone_with_wire_latency.*'
AMG_REPLICATE_ROWS=20000 LOW_GHOST=4 DELAY_US=5 timeout -k 10 400 python scratch/rank_cycle_on_one_gpu.py 256 2,2,2 0 > gpurun_out/s2_c7_rank0.log 2>&1 || { tail -20 gpurun_out/s2_c7_rank0.log; exit 1; }
tail -1 gpurun_out/s2_c7_rank0.log | cut -c1-420
