# construction time of the matrix-free operators (FP64 and FP32 instances) at 257^3 DoFs
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import mfmg_amd as M
ctx = M.Context()
for material in ("constant", "linear"):
    prob = M.LaplaceProblem((256,) * 3, material, device="cuda")
    torch.cuda.synchronize()
    for cls in (M.MatrixFreeLaplace, M.MatrixFreeLaplaceF32, M.MatrixFreeLaplace, M.MatrixFreeLaplaceF32):
        t = time.perf_counter(); op = cls(ctx, prob); ctx.synchronize(); print(material, cls.__name__, f"{time.perf_counter() - t:.3f} s", flush=True)
        del op
