# device vs host algebra on the assembled hierarchy's own matrices: which product differs, and where
import os, sys, numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import mfmg_amd as M
ctx = M.Context()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
N = n ** 3
T = sp.diags([-np.ones(n - 1), 2.5 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1]).tocsr()
A = (sp.kron(sp.kron(T, T), T)).tocsr()
rng = np.random.default_rng(0)
agg = (np.arange(N) // 8).astype(np.int64)
P = sp.csr_matrix((rng.standard_normal(2 * N), (np.repeat(np.arange(N), 2), np.stack([2 * agg, 2 * agg + 1], 1).ravel())),
                  shape=(N, 2 * (N // 8 + 1)))
Ad, Pd = M.SparseMatrixDevice(ctx, A), M.SparseMatrixDevice(ctx, P)
res = {}
for mode in ("device_only", "host"):
    os.environ["MFMG_CSR_ALGEBRA"] = mode
    Pd2 = M.SparseMatrixDevice(ctx, P)
    Pt = Pd2.transpose()
    AP = Ad.multiply(Pd2)
    C = Pt.multiply(AP)
    res[mode] = [m.to_scipy() for m in (Pt, AP, C)]
for name, a, b in zip(("Pt", "AP", "C"), res["device_only"], res["host"]):
    same = np.array_equal(a.indptr, b.indptr) and np.array_equal(a.indices, b.indices) and np.array_equal(a.data, b.data)
    print(name, a.shape, a.nnz, b.nnz, "identical" if same else "DIFFERENT", flush=True)
    if not same:
        d = np.flatnonzero(np.diff(a.indptr) != np.diff(b.indptr))
        print("  rows with different lengths:", d[:10], len(d))
        if len(d) == 0:
            w = np.flatnonzero((a.indices != b.indices) | (a.data != b.data))
            print("  entries differing:", w[:10], len(w), a.data[w[:5]], b.data[w[:5]], a.indices[w[:5]], b.indices[w[:5]])
