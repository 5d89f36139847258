# ps per DoF and Chebyshev(3) apply of the operator kernel on boxes of different shapes: separates the tail-column
# cost from the ramp-up / drain of a launch.   usage: size_effect.py nx,ny,nz[:ty,tz,waves] ...   (DoFs per direction)
import os, sys, json, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import mfmg_amd as M
from bench import smoother_coefficients, smoother_bytes_per_dof
ctx = M.Context()
coefs = smoother_coefficients(3, 0.09, 1.8)
material = os.environ.get('MATERIAL', 'constant')
for spec in sys.argv[1:]:
    parts = spec.split(':')
    N3 = tuple(int(v) for v in parts[0].split(','))
    torch.cuda.empty_cache()
    prob = M.LaplaceProblem(tuple(v - 1 for v in N3), material, device='cuda')
    op = M.MatrixFreeLaplace(ctx, prob)
    N = prob.n_dofs
    del prob
    torch.cuda.empty_cache()
    x = torch.rand(N, dtype=torch.float64, device='cuda'); b = torch.zeros_like(x)
    s1 = torch.empty_like(x); s2 = torch.empty_like(x)
    if len(parts) > 1:
        t = [int(v) for v in parts[1].split(',')]
        op.set_tile(t[0], t[1], t[2] if len(t) > 2 else None)
    def sweep():
        op.smoother_step(b, x, None, coefs[0][0], coefs[0][1], s2)
        op.smoother_step(b, s2, x, coefs[1][0], coefs[1][1], s1)
        op.smoother_step(b, s1, s2, coefs[2][0], coefs[2][1], x)
    for _ in range(3):
        sweep()
    ctx.synchronize()
    reps = 15
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for r in range(reps):
        sweep(); ev[r + 1].record()
    ev[-1].synchronize()
    ts = sorted(ev[r].elapsed_time(ev[r + 1]) for r in range(reps))
    ms = ts[reps // 2]
    req = smoother_bytes_per_dof(3, 8, op.cell_constant_layout(), survey=False, ids_computed=op.ids_computed())
    print(json.dumps({"N": N3, "tile": list(op.get_tile()), "ms_per_apply": round(ms, 4), "ps_per_dof": round(ms * 1e9 / N, 2),
                      "required_frac_of_8TBs": round(N * req / (ms * 1e-3) / 8e12, 4)}), flush=True)
    del op, x, b, s1, s2
