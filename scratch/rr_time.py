# time of the one-pass residual restriction at 256^3 cells against residual + restriction
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import mfmg_amd as M
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx = M.Context()
prob = M.LaplaceProblem((cells,) * 3, "constant", device="cuda")
params = {"eigensolver": {"number of eigenvectors": 2}, "agglomeration": {"nx": 2, "ny": 2, "nz": 2},
          "smoother": {"type": "Chebyshev", "degree": 3, "smoothing_range": 20.0},
          "solver": {"type": "pcg", "n_iterations": 2}, "is preconditioner": False, "max levels": 2}
h = M.Hierarchy(ctx, "HipMatrixFreeMeshEvaluator", prob, params)
print("classes", h.residual_restriction_classes(), flush=True)
nf, nc = h.level_size(0), h.level_size(1)
x = torch.rand(nf, dtype=torch.float64, device="cuda"); b = torch.rand(nf, dtype=torch.float64, device="cuda")
y = torch.empty(nc, dtype=torch.float64, device="cuda"); res = torch.empty_like(x); y2 = torch.empty_like(y)
def timed(f, n=20):
    for _ in range(3): f()
    ctx.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    ctx.synchronize(); return (time.perf_counter() - t) / n * 1e6
t1 = timed(lambda: h.restrict_residual(x, b, y))
def two():
    h.operator_apply(0, x, res); h.restrictor_apply(1, res, y2)
t2 = timed(two)
print(f"MFMG_RR_WAVES={os.environ.get('MFMG_RR_WAVES','-')}: one pass {t1:.1f} us ({(16*nf+8*nc)/t1/1e6:.2f} TB/s on x, b, b_c), apply + restriction {t2:.1f} us")
