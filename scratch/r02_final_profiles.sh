#!/bin/bash
# end-of-round profiles: kernel stats of the default bench and the kernel sequence of one cycle (constant and linear)
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/prof_r02
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02 -o p -- python3 $R/bench.py --no-cpu-baseline --no-extras > $R/gpurun_out/prof_r02/bench.log 2>&1 || echo "stats run failed"
cp $R/gpurun_out/prof_r02/p_kernel_stats.csv $R/gpurun_out/r02_final_kernel_stats.csv
grep '"metric"' $R/gpurun_out/prof_r02/bench.log | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('bench under the profiler:', d['ms_per_step'], 'ms per cycle, operator launch', d['roofline']['avg_launch_ms'], 'ms over', d['roofline']['launches_in_timed_region'], 'launches, setup', d['config']['setup_seconds'], 's; 512^3 legs', d['north_star_512cubed_smoother']['ms_by_layout'])"
rm -rf $R/gpurun_out/prof_r02
cd $R
bash scratch/cycle_seq.sh r02_final_seq 256 constant > gpurun_out/seq.log 2>&1
bash scratch/cycle_seq.sh r02_final_seq_linear 256 linear > gpurun_out/seq_linear.log 2>&1
rm -rf gpurun_out/r02_final_seq gpurun_out/r02_final_seq_linear
head -3 gpurun_out/r02_final_seq_sequence.txt gpurun_out/r02_final_seq_linear_sequence.txt
