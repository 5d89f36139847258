"""ms per V-cycle with the fine level in FP32 (Hierarchy.apply_f32), 256^3 cells; env MFMG_MF_NARROW=0 for the A/B of the narrow
last chunk column.  usage: fp32_cycle_time.py [cells]"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mfmg_amd as M
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx = M.Context()
prob = M.LaplaceProblem((cells,) * 3, "constant", device="cuda")
params = {"eigensolver": {"number of eigenvectors": 2}, "agglomeration": {"partitioner": "block", "nx": 2, "ny": 2, "nz": 2},
          "smoother": {"type": "Chebyshev", "degree": 3, "smoothing_range": 20.0, "n_smoothing_steps": 1},
          "solver": {"type": "amg", "amg": {"smoother_degree": 1, "smoothing_range": 4.0, "n_cycles": 1, "aggregate_block": 2, "pre_smoothing_levels": 0}},
          "is preconditioner": False, "max levels": 2, "fine level precision": "float"}
h = M.Hierarchy(ctx, "HipMatrixFreeMeshEvaluator", prob, params)
n = h.level_size(0)
x = torch.rand(n, dtype=torch.float32, device="cuda"); b = torch.zeros_like(x)
xd = x.double(); bd = b.double()
def block(f, k=10):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(k): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / k * 1e3
for _ in range(3): h.apply_f32(b, x); h.apply(bd, xd)
print(f"narrow={os.environ.get('MFMG_MF_NARROW', '1')}: FP32 fine level {sorted(block(lambda: h.apply_f32(b, x)) for _ in range(5))[2]:.3f} ms per cycle, FP64 {sorted(block(lambda: h.apply(bd, xd)) for _ in range(5))[2]:.3f}")
