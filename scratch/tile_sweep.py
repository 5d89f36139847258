import sys, time, torch
sys.path.insert(0,'.')
import mfmg_amd as M
from bench import smoother_coefficients
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx = M.Context()
prob = M.LaplaceProblem((n-1,)*3, device='cuda')
op = M.MatrixFreeLaplace(ctx, prob)
N = prob.n_dofs
del prob; torch.cuda.empty_cache()
x = torch.rand(N, dtype=torch.float64, device='cuda'); b = torch.zeros_like(x); s1=torch.empty_like(x); s2=torch.empty_like(x)
coefs = smoother_coefficients(3, 0.09, 1.8)
def apply():
    op.smoother_step(b, x, None, coefs[0][0], coefs[0][1], s2)
    op.smoother_step(b, s2, x, coefs[1][0], coefs[1][1], s1)
    op.smoother_step(b, s1, s2, coefs[2][0], coefs[2][1], x)
tiles = [(16,64),(16,32),(16,16),(8,64),(8,32),(8,16),(4,32),(4,64),(32,32),(2,64)]
if len(sys.argv) > 2:
    tiles = [tuple(int(v) for v in t.split('x')) for t in sys.argv[2:]]
for (ty,tz) in tiles:
    try:
        op.set_tile(ty,tz)
        apply(); ctx.synchronize()
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): apply()
        e1.record(); e1.synchronize()
        ms=e0.elapsed_time(e1)/5
        print(f"n={n} tile=({ty},{tz}) smoother {ms:.3f} ms  {N*400/ms/1e6:.0f} GB/s  frac {N*400/ms/1e6/8000:.3f}", flush=True)
    except Exception as e:
        print(ty,tz,'ERR',e)
