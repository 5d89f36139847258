#!/bin/bash
# round-4 profiles of the LAST state (second session), one gpurun call; every rocprofv3 pass is kernel-trace only, --pmc passes separate:
#   a  kernel stats of the default bench                                   -> r04_j_bench_kernel_stats.csv, r04_j_bench_line.json
#   b  HBM bytes per launch of the sweep (257^3, 512^3) and of the eight-coefficient one-term kernel (masked loads)
#                                                                            -> r04_k_sweep_traffic.txt, r04_k_general_kernel_pmc.txt
#   c  kernel sequence of one V-cycle (constant, linear)                     -> r04_l_cycle_kernel_sequence*.txt
#   d  HBM bytes per launch of every kernel of the cycle                     -> r04_m_cycle_hbm_bytes_per_launch.txt
#   e  counters of the three large kernels of the coarse part                -> r04_n_coarse_kernels_pmc.txt
#   f  kernel sequence of rank 7's share of the 2x2x2 cycle                  -> r04_h_rank_cycle_on_one_gpu.txt
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
STEPS=${1:-abcdef}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
if [[ $STEPS == *a* ]]; then
  mkdir -p $O/prof_r04
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r04 -o p -- python3 $R/bench.py --no-cpu-baseline --no-extras > $O/prof_r04/bench.log 2>&1 || echo "stats run failed"
  cp $O/prof_r04/p_kernel_stats.csv $O/r04_j_bench_kernel_stats.csv
  grep '"metric"' $O/prof_r04/bench.log > $O/r04_j_bench_line.json
  python3 -c "
import json
d = json.loads(open('$O/r04_j_bench_line.json').read()); print('bench under the profiler:', d['ms_per_step'], 'ms per cycle;', d['roofline']['kernel'], d['roofline']['avg_launch_ms'], 'ms over', d['roofline']['launches_in_timed_region'], 'launches, setup', d['config']['setup_seconds'], 's; 512^3 legs', d['north_star_512cubed_smoother']['ms_by_layout'])"
  rm -rf $O/prof_r04
fi
if [[ $STEPS == *b* ]]; then
  : > $O/r04_k_sweep_traffic.txt
  for cfg in "257" "512"; do
    for pass in FETCH_SIZE WRITE_SIZE; do
      d=$O/pmc_r04_${cfg}_$pass
      timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $d -o p -- python3 $R/scratch/sweep_only.py $cfg 3 0 0 0 3 > $d.log 2>&1 || echo "pass $pass $cfg failed"
    done
    python3 - "$R" "$cfg" >> $O/r04_k_sweep_traffic.txt <<'PY'
import csv, glob, sys, collections, re
R, n = sys.argv[1], sys.argv[2]
vals = {}
for pass_ in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(float)
    for f in glob.glob(f"{R}/gpurun_out/pmc_r04_{n}_{pass_}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "mf_cheb_fused" in r["Kernel_Name"] and r["Counter_Name"] == pass_:
                acc[int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    ds = sorted(acc)[-4:]
    vals[pass_] = sum(acc[d] for d in ds) / max(len(ds), 1)
tile = re.search(r"\((\d+), (\d+), (\d+)\)", open(f"{R}/gpurun_out/pmc_r04_{n}_FETCH_SIZE.log").read())
print(f"{n}^3 DoFs, sweep of 3 terms, tile {tile.groups() if tile else '?'}: FETCH_SIZE {vals['FETCH_SIZE']:.6g} kB, WRITE_SIZE {vals['WRITE_SIZE']:.6g} kB per launch -> "
      f"HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE = {(2 * vals['FETCH_SIZE'] + vals['WRITE_SIZE']) * 1e3:.6g}")
PY
  done
  rm -rf $O/pmc_r04_*
  cat $O/r04_k_sweep_traffic.txt
  for cfg in "257 linear" "512 linear"; do
    set -- $cfg
    MATERIAL=$2 WAVES=0 bash $R/scratch/pmc.sh r04_$1_$2 $1 0 0 || echo "pmc $cfg failed"
  done
  cd $R && for cfg in 257_linear 512_linear; do echo "== $cfg"; python3 scratch/pmc_sum.py r04_$cfg 6; done > $O/r04_k_general_kernel_pmc.txt 2>&1
  rm -rf $O/pmc_r04_*
  cat $O/r04_k_general_kernel_pmc.txt
fi
cd $R
if [[ $STEPS == *c* ]]; then
  bash scratch/cycle_seq.sh r04_seq 256 constant > $O/seq.log 2>&1
  bash scratch/cycle_seq.sh r04_seq_linear 256 linear > $O/seq_linear.log 2>&1
  mv $O/r04_seq_sequence.txt $O/r04_l_cycle_kernel_sequence.txt
  mv $O/r04_seq_linear_sequence.txt $O/r04_l_cycle_kernel_sequence_linear.txt
  rm -rf $O/r04_seq $O/r04_seq_linear
  head -12 $O/r04_l_cycle_kernel_sequence.txt
fi
if [[ $STEPS == *d* ]]; then
  bash scratch/pmc_cycle_mem.sh r04mem > $O/pmc_cycle_mem.log 2>&1
  python3 scratch/pmc_cycle_mem_sum.py r04mem > $O/r04_m_cycle_hbm_bytes_per_launch.txt 2>&1
  rm -rf $O/pmcc_r04mem
  cat $O/r04_m_cycle_hbm_bytes_per_launch.txt
fi
if [[ $STEPS == *e* ]]; then
  python3 scratch/r04_coarse_pmc.py fin > $O/coarse_pmc.log 2>&1 || tail -5 $O/coarse_pmc.log
  mv $O/r04_n_coarse_kernels_pmc_fin.txt $O/r04_n_coarse_kernels_pmc.txt
  rm -rf $O/r04cpmc_fin
  cat $O/r04_n_coarse_kernels_pmc.txt
fi
if [[ $STEPS == *f* ]]; then
  TRACE=1 AMG_REPLICATE_ROWS=20000 LOW_GHOST=4 bash scratch/rank_cycle_trace.sh r04_rank7 256 2,2,2 7 > $O/r04_rank7_trace.log 2>&1 || tail -5 $O/r04_rank7_trace.log
  mv $O/r04_rank7_sequence.txt $O/r04_h_rank_cycle_sequence.txt
  rm -rf $O/r04_rank7
  head -20 $O/r04_h_rank_cycle_sequence.txt
fi
