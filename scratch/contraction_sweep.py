# contraction of the bench's cycle for different smoothing ranges (fine Chebyshev(3), coarse Chebyshev(1)); time is unaffected
import os, sys, math, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import mfmg_amd as M
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ctx = M.Context()
prob = M.LaplaceProblem((cells,) * 3, "constant", device="cuda")
op = M.MatrixFreeLaplace(ctx, prob)
for fine_range, coarse_range, omega in [(20., 4., None), (10., 4., None), (30., 4., None), (15., 4., None), (20., 2., None), (20., 3., None), (20., 6., None),
                                        (20., 8., None), (15., 3., None), (30., 6., None)]:
    params = {"eigensolver": {"number of eigenvectors": 2}, "agglomeration": {"nx": 2, "ny": 2, "nz": 2},
              "smoother": {"type": "Chebyshev", "degree": 3, "smoothing_range": fine_range},
              "solver": {"type": "amg", "amg": {"smoother_degree": 1, "smoothing_range": coarse_range, "pre_smoothing_levels": 0}},
              "is preconditioner": False, "max levels": 2}
    h = M.Hierarchy(ctx, "HipMatrixFreeMeshEvaluator", prob, params)
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.rand(prob.n_dofs, dtype=torch.float64, device="cuda", generator=g) * (prob.constrained != 1)
    b = torch.zeros_like(x); r = torch.empty_like(x)
    norms = []
    for _ in range(11):
        op.vmult(r, x); norms.append(ctx.l2_norm(r)); h.apply(b, x)
    rates = [norms[i + 1] / norms[i] for i in range(10)]
    print(f"fine range {fine_range:5.1f} coarse range {coarse_range:4.1f}: mean contraction {(norms[10] / norms[0]) ** 0.1:.4f}, last {rates[-1]:.4f}", flush=True)
    del h
