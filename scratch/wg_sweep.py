"""Bit-exactness of the workgroup-cooperative operator kernel against the one-wavefront kernel, then a tile sweep."""
import os, sys, torch
sys.path.insert(0, '.')
import mfmg_amd as M
from bench import smoother_coefficients

ctx = M.Context()
# ---- correctness on awkward sizes
for cells in [(5, 3, 4), (70, 9, 6), (64, 33, 17), (130, 40, 21)]:
    prob = M.LaplaceProblem(cells, "linear", device='cuda')
    op = M.MatrixFreeLaplace(ctx, prob)
    N = prob.n_dofs
    torch.manual_seed(1)
    x = torch.rand(N, dtype=torch.float64, device='cuda'); b = torch.rand_like(x); xp = torch.rand_like(x)
    ref = torch.empty_like(x); out = torch.empty_like(x)
    op.set_tile(4, 8, 1)
    op.smoother_step(b, x, xp, 0.3, 0.4, ref); ctx.synchronize()
    for nw in (1, 2, 3, 4, 8):
        for ty in (1, 2, 3, 5):
            for tz in (1, 3, 8):
                if nw * ty < 2:
                    continue
                out.fill_(float('nan'))
                op.set_tile(ty, tz, nw)
                op.smoother_step(b, x, xp, 0.3, 0.4, out); ctx.synchronize()
                if not torch.equal(out, ref):
                    d = (out - ref).abs()
                    print("MISMATCH", cells, nw, ty, tz, float(d.nan_to_num(nan=1e30).max()), int(torch.isnan(out).sum()))
                    sys.exit(1)
    print("ok", cells, flush=True)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
prob = M.LaplaceProblem((n,) * 3, device='cuda')
op = M.MatrixFreeLaplace(ctx, prob)
N = prob.n_dofs
del prob; torch.cuda.empty_cache()
x = torch.rand(N, dtype=torch.float64, device='cuda'); b = torch.zeros_like(x); s1 = torch.empty_like(x); s2 = torch.empty_like(x)
coefs = smoother_coefficients(3, 0.09, 1.8)
def apply():
    op.smoother_step(b, x, None, coefs[0][0], coefs[0][1], s2)
    op.smoother_step(b, s2, x, coefs[1][0], coefs[1][1], s1)
    op.smoother_step(b, s1, s2, coefs[2][0], coefs[2][1], x)
cfgs = [(4, 3, 8), (4, 4, 8), (4, 3, 8), (4, 4, 16)]
for (nw, ty, tz) in cfgs:
    op.set_tile(ty, tz, nw)
    apply(); ctx.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8): apply()
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / 8
    print(f"n={n} waves={nw} ty={ty} tz={tz} smoother {ms:.3f} ms  {N*400/ms/1e6:.0f} GB/s  frac {N*400/ms/1e6/8000:.3f}  ns/kDoF {ms*1e6/N*1e3:.2f}", flush=True)
