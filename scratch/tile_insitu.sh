#!/bin/bash
# in-situ comparison of operator tiles: one hierarchy setup per tile (bench.py --tile ty,tz,waves)
for t in "$@"; do
  timeout -k 10 300 python bench.py --no-smoother-512 --no-cpu-baseline --tile $t > gpurun_out/insitu_$t.log 2>&1 || exit 1
  echo "$t $(grep -o '"avg_launch_ms": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/insitu_$t.log | tr '\n' ' ')" >> gpurun_out/insitu_summary.txt
done
