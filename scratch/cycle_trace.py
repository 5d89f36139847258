"""Ten V-cycles of the bench problem, nothing else after the setup: run under `rocprofv3 --kernel-trace` and feed the
kernel trace to scratch/cycle_trace_sum.py for time per kernel and idle time per cycle.  usage: cycle_trace.py [cells] [graph]"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mfmg_amd as M
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 256
material = sys.argv[2] if len(sys.argv) > 2 else "constant"
structured = (sys.argv[3] != "0") if len(sys.argv) > 3 else True
ctx = M.Context()
prob = M.LaplaceProblem((cells,) * 3, material, device="cuda")
params = {"eigensolver": {"number of eigenvectors": 2}, "agglomeration": {"partitioner": "block", "nx": 2, "ny": 2, "nz": 2},
          "smoother": {"type": "Chebyshev", "degree": 3, "smoothing_range": 20.0, "n_smoothing_steps": 1},
          "solver": {"type": "amg", "amg": {"smoother_degree": 1, "smoothing_range": 4.0, "n_cycles": 1, "aggregate_block": 2,
                                            **({} if os.environ.get("AMG_V11") == "1" else {"pre_smoothing_levels": 0})}},
          "is preconditioner": False, "max levels": 2, "restrictor": {"structured": structured}}
if os.environ.get("SETUP_FLOAT") == "1":
    params["setup value precision"] = "float"
h = M.Hierarchy(ctx, "HipMatrixFreeMeshEvaluator", prob, params)
x = torch.rand(h.level_size(0), dtype=torch.float64, device="cuda"); b = torch.zeros_like(x)
for _ in range(3): h.apply(b, x)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(10): h.apply(b, x)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
print(f"ms/cycle {dt*1e3:.3f}")
