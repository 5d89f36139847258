"""Time of the agglomerate-wise restriction / prolongation on the bench mesh."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import mfmg_amd as M
from mfmg_amd import lib as L
ctx = M.Context()
prob = M.LaplaceProblem((256,) * 3, device='cuda')
params = {"eigensolver": {"number of eigenvectors": 2}, "agglomeration": {"nx": 2, "ny": 2, "nz": 2},
          "smoother": {"type": "Chebyshev", "degree": 3}, "solver": {"type": "pcg", "n_iterations": 1}}
h = M.Hierarchy(ctx, "HipMatrixFreeMeshEvaluator", prob, params)
nf, nc = h.level_size(0), h.level_size(1)
g = torch.Generator(device='cuda').manual_seed(3)
y = torch.rand(nc, dtype=torch.float64, device='cuda', generator=g); x = torch.rand(nf, dtype=torch.float64, device='cuda', generator=g)
out = torch.empty(nf, dtype=torch.float64, device='cuda'); r = torch.empty(nc, dtype=torch.float64, device='cuda')
for name, f in (("prolong", lambda: h.restrictor_apply(1, y, out, L.TRANS)), ("restrict", lambda: h.restrictor_apply(1, x, r))):
    f(); ctx.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); e1.synchronize()
    print(name, e0.elapsed_time(e1) / 10 * 1e3, "us", flush=True)
print("checksum %.15e %.15e" % (float(out.sum()), float(r.sum())))
