import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import mfmg_amd as M
from bench import smoother_coefficients
ctx = M.Context()
ctx.set_mf_fused_terms(1)
mat = os.environ.get('MFMG_MF_F32_TIME', 'linear')
for dtype, cls in ((torch.float32, M.MatrixFreeLaplaceF32), (torch.float64, M.MatrixFreeLaplace)):
    prob = M.LaplaceProblem((256,) * 3, mat, device='cuda')
    op = cls(ctx, prob)
    N = prob.n_dofs
    x = torch.rand(N, dtype=dtype, device='cuda'); b = torch.rand_like(x); s1 = torch.empty_like(x); s2 = torch.empty_like(x)
    c = smoother_coefficients(3, 0.09, 1.8)
    def apply():
        op.smoother_step(b, x, None, c[0][0], c[0][1], s2); op.smoother_step(b, s2, x, c[1][0], c[1][1], s1); op.smoother_step(b, s1, s2, c[2][0], c[2][1], x)
    for _ in range(3): apply()
    ctx.synchronize(); t = time.perf_counter()
    for _ in range(20): apply()
    ctx.synchronize()
    print(f"  {sys.argv[1]:28s} {str(dtype):14s} cell-constant layout {op.cell_constant_layout()}: {(time.perf_counter() - t) / 20 * 1e3:.3f} ms per apply", flush=True)
    del op, prob, x, b, s1, s2
