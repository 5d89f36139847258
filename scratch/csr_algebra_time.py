# device vs host transpose / product of a prolongator-like and a 27-point matrix (csr_algebra.hip)
import os, sys, time, numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import mfmg_amd as M
ctx = M.Context()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
N = n ** 3
I = sp.identity(n, format="csr")
T = sp.diags([-np.ones(n - 1), 2.5 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1]).tocsr()
A = (sp.kron(sp.kron(T, T), T)).tocsr()           # 27-point pattern
rng = np.random.default_rng(0)
agg = (np.arange(N) // 8).astype(np.int64)       # tentative prolongator, two columns per aggregate
P = sp.csr_matrix((rng.standard_normal(2 * N), (np.repeat(np.arange(N), 2), np.stack([2 * agg, 2 * agg + 1], 1).ravel())),
                  shape=(N, 2 * (N // 8 + 1)))
Ad, Pd = M.SparseMatrixDevice(ctx, A), M.SparseMatrixDevice(ctx, P)
for mode in ("device_only", "host", "device_only", "host"):
    os.environ["MFMG_CSR_ALGEBRA"] = mode
    t0 = time.perf_counter(); Pt = Pd.transpose(); ctx.synchronize(); t1 = time.perf_counter()
    AP = Ad.multiply(Pd); ctx.synchronize(); t2 = time.perf_counter()
    C = Pt.multiply(AP); ctx.synchronize(); t3 = time.perf_counter()
    print(f"{mode:12s} N={N} nnz(A)={A.nnz}: transpose {t1 - t0:.3f} s, A*P {t2 - t1:.3f} s (nnz {AP.nnz}), Pt*(AP) {t3 - t2:.3f} s (nnz {C.nnz})", flush=True)
