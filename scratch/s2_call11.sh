#!/bin/bash
R=${GRAFT_REPO_ROOT:?}
cd $R
bash scratch/rehearse_bench.sh 2 64 || exit 1
timeout -k 10 500 python bench.py --no-cpu-baseline --no-smoother-512 --no-vcycle-513 > gpurun_out/s2_c11_bench.log 2>&1 || { tail -20 gpurun_out/s2_c11_bench.log; exit 1; }
python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/s2_c11_bench.log') if l.startswith('{"metric"')][-1])
print(d['ms_per_step']); print(json.dumps(d.get('cg_solve_256cubed'))[:900])
PY
