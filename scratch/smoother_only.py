# usage: smoother_only.py n ty tz reps [calib]
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import mfmg_amd as M
from bench import smoother_coefficients
n, ty, tz, reps = (int(v) for v in sys.argv[1:5])
ctx = M.Context()
prob = M.LaplaceProblem((n-1,)*3, os.environ.get('MATERIAL', 'constant'), device='cuda')
op = M.MatrixFreeLaplace(ctx, prob)
N = prob.n_dofs
del prob; torch.cuda.empty_cache()
op.set_tile(ty, tz, int(os.environ.get('WAVES', '1')))
x = torch.rand(N, dtype=torch.float64, device='cuda'); b = torch.zeros_like(x); s1=torch.empty_like(x); s2=torch.empty_like(x)
coefs = smoother_coefficients(3, 0.09, 1.8)
for _ in range(reps):
    op.smoother_step(b, x, None, coefs[0][0], coefs[0][1], s2)
    op.smoother_step(b, s2, x, coefs[1][0], coefs[1][1], s1)
    op.smoother_step(b, s1, s2, coefs[2][0], coefs[2][1], x)
ctx.synchronize()
if len(sys.argv) > 5:
    # calibration of the HBM counters on known byte counts: 1 GiB read + 1 GiB write each
    a = torch.empty(1 << 27, dtype=torch.float64, device='cuda'); c = torch.empty_like(a)
    c.copy_(a)                      # torch vectorised copy (16 B per lane)
    ctx.add(c, 1.0, a)              # mfmg vec add: 8 B per lane, reads a and c, writes c
    ctx.synchronize()
print('done', N)
