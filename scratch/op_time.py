# Smoother-apply timing of the operator kernel: median of per-apply HIP-event times.
# usage: op_time.py "n:material[:ty,tz,waves]" ...     e.g.  op_time.py 257:constant 512:constant 257:linear:3,8,4
import os, sys, json, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import mfmg_amd as M
from bench import smoother_coefficients, smoother_bytes_per_dof
ctx = M.Context()
coefs = smoother_coefficients(3, 0.09, 1.8)
last = None
for spec in sys.argv[1:]:
    parts = spec.split(':')
    n, material = int(parts[0]), parts[1]
    if last is None or last[0] != (n, material):
        last = None
        torch.cuda.empty_cache()
        prob = M.LaplaceProblem((n - 1,) * 3, material, device='cuda')
        op = M.MatrixFreeLaplace(ctx, prob)
        N = prob.n_dofs
        del prob
        torch.cuda.empty_cache()
        x = torch.rand(N, dtype=torch.float64, device='cuda'); b = torch.zeros_like(x)
        s1 = torch.empty_like(x); s2 = torch.empty_like(x)
        last = ((n, material), op, N, x, b, s1, s2)
    _, op, N, x, b, s1, s2 = last
    if len(parts) > 2:
        t = [int(v) for v in parts[2].split(',')]
        op.set_tile(t[0], t[1], t[2] if len(t) > 2 else None)
    else:
        op.set_tile(0, 0, 0)
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
    req = smoother_bytes_per_dof(3, 8, op.cell_constant_layout(), survey=False)
    print(json.dumps({"n": n, "material": material, "tile": list(op.get_tile()), "ms_per_apply": round(ms, 4), "min": round(ts[0], 4),
                      "ms_per_launch": round(ms / 3, 4), "required_frac_of_8TBs": round(N * req / (ms * 1e-3) / 8e12, 4)}), flush=True)
