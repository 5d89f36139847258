#!/bin/bash
# cycle time and contraction for different deep-level aggregate sizes (host AMG setup)
for deep in "3,4" "3,3" "4,4" "3,8" "4,8"; do
  echo "== deep '$deep'"
  timeout -k 10 280 python3 bench.py --no-cpu-baseline --no-extras --no-smoother-512 --amg-setup host ${deep:+--amg-deep $deep} 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); c = d['config']; print(d['ms_per_step'], d['value'], c.get('mean_residual_contraction_per_cycle'), c.get('setup_seconds'), c.get('coarse_amg_levels_rows_nnzA_nnzP'))
" || echo failed
done
