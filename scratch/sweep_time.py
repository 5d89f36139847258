# usage: sweep_time.py n K "nw,ty,tz;nw,ty,tz;..."  -- time of the multi-term sweep for a list of tiles (one process per env setting)
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import mfmg_amd as M
from bench import smoother_coefficients
n, K = int(sys.argv[1]), int(sys.argv[2])
tiles = [tuple(int(v) for v in t.split(',')) for t in sys.argv[3].split(';')]
ctx = M.Context()
prob = M.LaplaceProblem((n - 1,) * 3, os.environ.get('MATERIAL', 'constant'), device='cuda')
op = M.MatrixFreeLaplace(ctx, prob)
N = prob.n_dofs
x = torch.rand(N, dtype=torch.float64, device='cuda'); b = torch.rand_like(x); o = torch.empty_like(x)
coefs = smoother_coefficients(3, 0.09, 1.8)
al = [0.0] + [c[0] for c in coefs][1:]; be = [c[1] for c in coefs]
def timeit(f, reps=20):
    f(); ctx.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        f()
    ctx.synchronize()
    return (time.perf_counter() - t) / reps * 1e3
for tile in tiles:
    op.set_sweep_tile(*tile)
    t = timeit(lambda: op.smoother_sweep(al[:K], be[:K], b, x, o, None))
    print(f'{n}^3 K={K} dbg={os.environ.get("MFMG_MF_FUSED_DBG", "0")} tile {op.get_sweep_tile(K)}: {t:.3f} ms', flush=True)
