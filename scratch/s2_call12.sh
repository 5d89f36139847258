#!/bin/bash
R=${GRAFT_REPO_ROOT:?}
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_hierarchy.py -x -q -k "from_zero or without_zeroing or outer_cg or preconditioner_mode or sweep_equals or hierarchy_driver" > gpurun_out/s2_c12_tests.log 2>&1 || { tail -40 gpurun_out/s2_c12_tests.log; exit 1; }
tail -2 gpurun_out/s2_c12_tests.log
timeout -k 10 500 python bench.py --no-cpu-baseline --no-smoother-512 --no-vcycle-513 > gpurun_out/s2_c12_bench.log 2>&1 || { tail -20 gpurun_out/s2_c12_bench.log; exit 1; }
python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/s2_c12_bench.log') if l.startswith('{"metric"')][-1])
print(d['ms_per_step']); print(json.dumps(d.get('cg_solve_256cubed'))[:900])
PY
