# usage: sweep_only.py n K nw ty tz reps   -- the multi-term smoother sweep alone (profiling target)
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import mfmg_amd as M
from bench import smoother_coefficients
n, K, nw, ty, tz, reps = (int(v) for v in sys.argv[1:7])
ctx = M.Context()
prob = M.LaplaceProblem((n - 1,) * 3, os.environ.get('MATERIAL', 'constant'), device='cuda')
op = M.MatrixFreeLaplace(ctx, prob)
N = prob.n_dofs
del prob; torch.cuda.empty_cache()
op.set_sweep_tile(nw, ty, tz)
x = torch.rand(N, dtype=torch.float64, device='cuda'); b = torch.zeros_like(x); o = torch.empty_like(x)
coefs = smoother_coefficients(3, 0.09, 1.8)
al = [0.0] + [c[0] for c in coefs][1:]; be = [c[1] for c in coefs]
for _ in range(reps):
    op.smoother_sweep(al[:K], be[:K], b, x, o, None)
    op.smoother_sweep(al[:K], be[:K], b, o, x, None)
ctx.synchronize()
print('done', N, op.get_sweep_tile(K))
