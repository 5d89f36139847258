#!/bin/bash
# round-3 profiles (run on the GPU box through gpurun; every rocprofv3 pass is kernel-trace only, --pmc passes separate):
#   a  kernel stats of the default bench                              -> gpurun_out/r03_a_bench_kernel_stats.csv
#   b  PMC passes of the operator kernel (257^3 / 512^3, both layouts) -> gpurun_out/r03_b_mf_kernel_pmc.txt
#   c  kernel sequence of one V-cycle (constant, linear)               -> gpurun_out/r03_c_cycle_kernel_sequence*.txt
#   d  HBM bytes per launch of every kernel of the cycle               -> gpurun_out/r03_d_cycle_hbm_bytes_per_launch.txt
#   e  counters of the coarse-part kernels                             -> gpurun_out/r03_e_cycle_kernels_pmc.txt
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
STEPS=${1:-abcde}
cd /tmp && export TMPDIR=/tmp
if [[ $STEPS == *a* ]]; then
  mkdir -p $R/gpurun_out/prof_r03
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03 -o p -- python3 $R/bench.py --no-cpu-baseline --no-extras > $R/gpurun_out/prof_r03/bench.log 2>&1 || echo "stats run failed"
  cp $R/gpurun_out/prof_r03/p_kernel_stats.csv $R/gpurun_out/r03_a_bench_kernel_stats.csv
  grep '"metric"' $R/gpurun_out/prof_r03/bench.log > $R/gpurun_out/r03_a_bench_line.json
  python3 -c "
import sys, json
d = json.loads(open('$R/gpurun_out/r03_a_bench_line.json').read()); print('bench under the profiler:', d['ms_per_step'], 'ms per cycle, operator launch', d['roofline']['avg_launch_ms'], 'ms over', d['roofline']['launches_in_timed_region'], 'launches, setup', d['config']['setup_seconds'], 's; 512^3 legs', d['north_star_512cubed_smoother']['ms_by_layout'])"
  rm -rf $R/gpurun_out/prof_r03
fi
if [[ $STEPS == *b* ]]; then
  for cfg in "257 constant" "257 linear" "512 constant" "512 linear"; do
    set -- $cfg
    echo "pmc $1 $2"
    MATERIAL=$2 WAVES=0 bash $R/scratch/pmc.sh r03_$1_$2 $1 0 0 || echo "pmc $cfg failed"
  done
  cd $R && for cfg in 257_constant 257_linear 512_constant 512_linear; do echo "== $cfg"; python3 scratch/pmc_sum.py r03_$cfg 6; done > gpurun_out/r03_b_mf_kernel_pmc.txt 2>&1
  rm -rf $R/gpurun_out/pmc_r03_*
  cat $R/gpurun_out/r03_b_mf_kernel_pmc.txt
fi
cd $R
if [[ $STEPS == *c* ]]; then
  bash scratch/cycle_seq.sh r03_seq 256 constant > gpurun_out/seq.log 2>&1
  bash scratch/cycle_seq.sh r03_seq_linear 256 linear > gpurun_out/seq_linear.log 2>&1
  mv gpurun_out/r03_seq_sequence.txt gpurun_out/r03_c_cycle_kernel_sequence.txt
  mv gpurun_out/r03_seq_linear_sequence.txt gpurun_out/r03_c_cycle_kernel_sequence_linear.txt
  rm -rf gpurun_out/r03_seq gpurun_out/r03_seq_linear
  head -14 gpurun_out/r03_c_cycle_kernel_sequence.txt
fi
if [[ $STEPS == *d* ]]; then
  bash scratch/pmc_cycle_mem.sh r03mem > gpurun_out/pmc_cycle_mem.log 2>&1
  python3 scratch/pmc_cycle_mem_sum.py r03mem > gpurun_out/r03_d_cycle_hbm_bytes_per_launch.txt 2>&1
  rm -rf gpurun_out/pmcc_r03mem
  cat gpurun_out/r03_d_cycle_hbm_bytes_per_launch.txt
fi
if [[ $STEPS == *e* ]]; then
  bash scratch/pmc_cycle.sh r03cyc > gpurun_out/pmc_cycle.log 2>&1
  for pat in residual_restriction sr_prolong bdia_class_node bdia_node_split mf_laplace; do echo "== $pat"; python3 scratch/pmc_cycle_sum.py r03cyc $pat; done > gpurun_out/r03_e_cycle_kernels_pmc.txt 2>&1
  rm -rf gpurun_out/pmcc_r03cyc
  head -40 gpurun_out/r03_e_cycle_kernels_pmc.txt
fi
