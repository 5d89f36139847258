# tiles of the one-term kernel for the eight-coefficient layout at 257^3 (halo re-reads of the coefficients: 1.34 x at (4,3,8))
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import mfmg_amd as M
from bench import smoother_coefficients
n = int(sys.argv[1]); material = sys.argv[2]
ctx = M.Context()
if len(sys.argv) > 3:
    ctx.set_mf_fused_terms(int(sys.argv[3]))
prob = M.LaplaceProblem((n - 1,) * 3, material, device='cuda')
op = M.MatrixFreeLaplace(ctx, prob)
N = prob.n_dofs
x = torch.rand(N, dtype=torch.float64, device='cuda'); b = torch.rand_like(x); s1 = torch.empty_like(x); s2 = torch.empty_like(x)
coefs = smoother_coefficients(3, 0.09, 1.8)
def apply():
    op.smoother_step(b, x, None, coefs[0][0], coefs[0][1], s2)
    op.smoother_step(b, s2, x, coefs[1][0], coefs[1][1], s1)
    op.smoother_step(b, s1, s2, coefs[2][0], coefs[2][1], x)
def timeit(reps=10):
    apply(); ctx.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        apply()
    ctx.synchronize()
    return (time.perf_counter() - t) / reps * 1e3
print('default tile', op.get_tile(), f'{timeit():.3f} ms per Chebyshev(3) apply', flush=True)
for nw, ty, tz in [(4, 3, 8), (4, 3, 16), (8, 2, 8), (8, 2, 16), (8, 3, 8), (8, 3, 16), (4, 4, 16), (8, 2, 32), (8, 4, 16), (4, 2, 16), (8, 3, 32), (8, 2, 11), (8, 3, 11)]:
    try:
        op.set_tile(ty, tz, nw)
        print((nw, ty, tz), f'{timeit():.3f}', flush=True)
    except Exception as e:
        print((nw, ty, tz), 'failed', str(e)[:80])
