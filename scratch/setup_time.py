# setup time of the hierarchy: where the seconds go (timer report); usage: setup_time.py [material] [eigensolver where] [cells] [float]
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import mfmg_amd as M
material = sys.argv[1] if len(sys.argv) > 1 else "constant"
where = sys.argv[2] if len(sys.argv) > 2 else "device"
cells = int(sys.argv[3]) if len(sys.argv) > 3 else 256
ctx = M.Context()
prob = M.LaplaceProblem((cells,) * 3, material, device="cuda")
params = {"eigensolver": {"number of eigenvectors": 2}, "agglomeration": {"nx": 2, "ny": 2, "nz": 2},
          "smoother": {"type": "Chebyshev", "degree": 3, "smoothing_range": 20.0},
          "solver": {"type": "amg", "amg": {"smoothing_range": 4.0}}, "restrictor": {"eigensolver": where},
          "is preconditioner": False, "max levels": 2}
if len(sys.argv) > 4 and sys.argv[4] == "float":
    params["setup value precision"] = "float"
t = time.perf_counter()
h = M.Hierarchy(ctx, os.environ.get("EVALUATOR", "HipMatrixFreeMeshEvaluator"), prob, params)
ctx.synchronize()
print(f"material {material} eigensolver {where}: setup {time.perf_counter() - t:.2f} s")
print(h.timer_report())
