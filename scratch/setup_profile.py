"""Where the setup time of the bench hierarchy goes (Hierarchy timer report)."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import mfmg_amd as M
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 256
material = sys.argv[2] if len(sys.argv) > 2 else "constant"
ctx = M.Context()
t0 = time.perf_counter()
prob = M.LaplaceProblem((cells,) * 3, material, device='cuda')
torch.cuda.synchronize(); t1 = time.perf_counter()
params = {"eigensolver": {"number of eigenvectors": 2}, "agglomeration": {"nx": 2, "ny": 2, "nz": 2},
          "smoother": {"type": "Chebyshev", "degree": 3, "smoothing_range": 20.0},
          "solver": {"type": "amg"}, "is preconditioner": False}
h = M.Hierarchy(ctx, "HipMatrixFreeMeshEvaluator", prob, params)
torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"problem {t1-t0:.2f} s, hierarchy {t2-t1:.2f} s  ({material})")
print(h.timer_report())
