#!/bin/bash
# round-4 profiles (run on the GPU box through gpurun; every rocprofv3 pass is kernel-trace only, --pmc passes separate):
#   a  kernel stats of the default bench                                   -> gpurun_out/r04_c_bench_kernel_stats.csv
#   b  HBM bytes per launch of the multi-term sweep (257^3, 512^3)           -> gpurun_out/r04_d_sweep_traffic.txt (+ json entries)
#   c  kernel sequence of one V-cycle (constant, linear)                     -> gpurun_out/r04_e_cycle_kernel_sequence*.txt
#   d  HBM bytes per launch of every kernel of the cycle                     -> gpurun_out/r04_f_cycle_hbm_bytes_per_launch.txt
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
STEPS=${1:-abcd}
cd /tmp && export TMPDIR=/tmp
if [[ $STEPS == *a* ]]; then
  mkdir -p $R/gpurun_out/prof_r04
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r04 -o p -- python3 $R/bench.py --no-cpu-baseline --no-extras > $R/gpurun_out/prof_r04/bench.log 2>&1 || echo "stats run failed"
  cp $R/gpurun_out/prof_r04/p_kernel_stats.csv $R/gpurun_out/r04_c_bench_kernel_stats.csv
  grep '"metric"' $R/gpurun_out/prof_r04/bench.log > $R/gpurun_out/r04_c_bench_line.json
  python3 -c "
import json
d = json.loads(open('$R/gpurun_out/r04_c_bench_line.json').read()); print('bench under the profiler:', d['ms_per_step'], 'ms per cycle;', d['roofline']['kernel'], d['roofline']['avg_launch_ms'], 'ms over', d['roofline']['launches_in_timed_region'], 'launches, setup', d['config']['setup_seconds'], 's; 512^3 legs', d['north_star_512cubed_smoother']['ms_by_layout'])"
  rm -rf $R/gpurun_out/prof_r04
fi
if [[ $STEPS == *b* ]]; then
  : > $R/gpurun_out/r04_d_sweep_traffic.txt
  for cfg in "257 0 0 0" "512 0 0 0"; do
    set -- $cfg
    for pass in FETCH_SIZE WRITE_SIZE; do
      d=$R/gpurun_out/pmc_r04_$1_$pass
      timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $d -o p -- python3 $R/scratch/sweep_only.py $1 3 $2 $3 $4 3 > $d.log 2>&1 || echo "pass $pass $1 failed"
    done
    python3 - "$R" "$1" >> $R/gpurun_out/r04_d_sweep_traffic.txt <<'PY'
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
  rm -rf $R/gpurun_out/pmc_r04_*
  cat $R/gpurun_out/r04_d_sweep_traffic.txt
fi
cd $R
if [[ $STEPS == *c* ]]; then
  bash scratch/cycle_seq.sh r04_seq 256 constant > gpurun_out/seq.log 2>&1
  bash scratch/cycle_seq.sh r04_seq_linear 256 linear > gpurun_out/seq_linear.log 2>&1
  mv gpurun_out/r04_seq_sequence.txt gpurun_out/r04_e_cycle_kernel_sequence.txt
  mv gpurun_out/r04_seq_linear_sequence.txt gpurun_out/r04_e_cycle_kernel_sequence_linear.txt
  rm -rf gpurun_out/r04_seq gpurun_out/r04_seq_linear
  head -14 gpurun_out/r04_e_cycle_kernel_sequence.txt
fi
if [[ $STEPS == *d* ]]; then
  bash scratch/pmc_cycle_mem.sh r04mem > gpurun_out/pmc_cycle_mem.log 2>&1
  python3 scratch/pmc_cycle_mem_sum.py r04mem > gpurun_out/r04_f_cycle_hbm_bytes_per_launch.txt 2>&1
  rm -rf gpurun_out/pmcc_r04mem
  cat gpurun_out/r04_f_cycle_hbm_bytes_per_launch.txt
fi
