"""Time the SpMV kernel variants on the real matrices of the bench hierarchy (A_c, R, R^T, AMG levels)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import ctypes as C
import mfmg_amd as M
from mfmg_amd.api import SparseMatrixDevice, check
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx = M.Context()
prob = M.LaplaceProblem((cells,) * 3, device='cuda')
params = {"eigensolver": {"number of eigenvectors": 2}, "agglomeration": {"nx": 2, "ny": 2, "nz": 2},
          "smoother": {"type": "Chebyshev", "degree": 3},
          "solver": {"type": "amg", "amg": {"smoother_degree": 1, "smoothing_range": 4.0, "n_cycles": 1}}}
h = M.Hierarchy(ctx, "HipMatrixFreeMeshEvaluator", prob, params)
def bench(name, A):
    m, k = A.shape; nnz = A.nnz
    x = torch.rand(k, dtype=torch.float64, device='cuda'); y = torch.empty(m, dtype=torch.float64, device='cuda')
    lpr0, kind0 = A.get_kernel()
    print(f"{name}: {m} x {k}, nnz {nnz} ({nnz/m:.1f}/row)  default lpr={lpr0} kind={kind0}", flush=True)
    for kind in (kind0 if kind0 >= 2 else 2, 1, 0):
        for lpr in ((4, 8, 16, 32, 64) if kind != 2 else (0,)):
            A.set_kernel(lpr, kind)
            if A.get_kernel()[1] != kind:
                break
            A.vmult(y, x); ctx.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): A.vmult(y, x)
            e1.record(); e1.synchronize()
            ms = e0.elapsed_time(e1) / 5
            byt = nnz * 12 + 8 * (m + k) + 4 * m
            print(f"   kind={kind} lpr={lpr:2d}  {ms*1e3:8.1f} us  {byt/ms/1e6:6.0f} GB/s (12 B/nnz)", flush=True)
    A.set_kernel(lpr0, kind0)

R = h.restrictor(); bench("R", R); bench("Rt", R.transpose()); del R
bench("A_c", h.coarse_operator())
n = C.c_int32(); check(h._lib.mfmg_hip_hierarchy_coarse_amg_levels(h.handle, C.byref(n)))
for l in range(n.value - 1):
    hp = C.c_void_p(); check(h._lib.mfmg_hip_hierarchy_coarse_amg_get(h.handle, l, 1, C.byref(hp)))
    P = SparseMatrixDevice(ctx, _handle=hp, _borrowed=True, _keepalive=h)
    bench(f"P{l+1}", P); bench(f"P{l+1}t", P.transpose()); del P
    if l + 1 < n.value - 1:
        ha = C.c_void_p(); check(h._lib.mfmg_hip_hierarchy_coarse_amg_get(h.handle, l + 1, 0, C.byref(ha)))
        bench(f"A_{l+1}", SparseMatrixDevice(ctx, _handle=ha, _borrowed=True, _keepalive=h))
