import os, subprocess
print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
for f in ['/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us', '/sys/fs/cgroup/cpu/cpu.cfs_period_us']:
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, 'n/a')
print(subprocess.run('nproc; lscpu | head -20; free -g | head -2; echo OMP=$OMP_NUM_THREADS', shell=True, capture_output=True, text=True).stdout)
import sys, time, torch
sys.path.insert(0,'.')
import mfmg_amd as M
ctx = M.Context()
for n in (64, 128):
    t=time.perf_counter()
    prob = M.LaplaceProblem((n-1,)*3, device='cuda')
    params = {"eigensolver": {"number of eigenvectors": 2}, "agglomeration": {"nx": 2, "ny": 2, "nz": 2},
              "smoother": {"type": "Chebyshev", "degree": 3, "smoothing_range": 20.0}, "solver": {"type": "pcg", "n_iterations": 10}, "is preconditioner": False}
    h = M.Hierarchy(ctx, "HipMatrixFreeMeshEvaluator", prob, params)
    print(n, 'setup', time.perf_counter()-t)
    print(h.timer_report())
