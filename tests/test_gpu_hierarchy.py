"""V-cycle parity: mfmg::Hierarchy on the GPU against the CPU oracle on the same problem,
same R, same smoother parameters -- residual history equal to 1e-10 relative (BASELINE.json)."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

import mfmg_amd as M
from mfmg_amd import lib as L
import mfmg_oracle as O

pytestmark = pytest.mark.gpu
HIST_TOL = 1e-10
HIST_ATOL = 1e-12   # relative residuals below this sit on the FP64 rounding floor of b - A x


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()


def gpu_history(ctx, h, apply_monitor, b, x0, n_cycles=20):
    """tests/test_hierarchy.cc:95-123 with the residual monitored by `apply_monitor(y, x)`."""
    x = dev(x0)
    bd = dev(b)
    r = torch.empty_like(x)
    apply_monitor(r, x)
    ctx.sadd(r, -1.0, 1.0, bd)
    r0 = ctx.l2_norm(r)
    res = [1.0]
    for _ in range(n_cycles):
        h.apply(bd, x)
        apply_monitor(r, x)
        ctx.sadd(r, -1.0, 1.0, bd)
        res.append(ctx.l2_norm(r) / r0)
    ctx.synchronize()
    return np.array(res), x.cpu().numpy()


def base_params(n_eig=2, **extra):
    p = {"eigensolver": {"number of eigenvectors": n_eig}, "agglomeration": {"nx": 2, "ny": 2, "nz": 2},
         "is preconditioner": False, "max levels": 2}
    p.update(extra)
    return p


@pytest.mark.parametrize("n,material,degree", [((4, 4, 4), "constant", 1), ((8, 8, 8), "linear", 3),
                                               ((16, 16, 16), "constant", 3), ((12, 10, 6), "discontinuous", 2)])
def test_matrix_free_chebyshev_vcycle_history(ctx, n, material, degree):
    mesh = O.StructuredMesh(n)
    coef = O.coefficient_table(mesh, material)
    con = mesh.constrained_mask()
    mf = O.MatrixFreeLaplace(mesh, coef)
    prob = M.LaplaceProblem(n, material, device="cuda")
    params = base_params(smoother={"type": "Chebyshev", "degree": degree, "smoothing_range": 20.0})
    h = M.Hierarchy(ctx, "HipMatrixFreeMeshEvaluator", prob, params)
    deg, lmin, lmax = h.smoother_info()
    assert deg == degree
    # oracle with the same R (the product's own, downloaded) and the same polynomial
    R = h.restrictor().to_scipy()
    Ro = O.build_restrictor(mesh, coef, mf.diagonal(), n_eig=2, variant="mf", eig_mode="krylov").csr
    assert abs(R - Ro).max() < 1e-11          # product setup == oracle setup (unique selection rule)
    Ac = O.galerkin_coarse_matrix(mf.vmult, R)
    assert abs(h.coarse_operator().to_scipy() - Ac).max() < 1e-11 * abs(Ac).max()
    p = O.ChebyshevParams(degree=degree, lambda_max=lmax, lambda_min=lmin)
    dinv = mf.diagonal_inverse()
    smoother = lambda b, x: O.chebyshev_smoother_apply(mf.vmult, dinv, p, b, x)
    ho = O.TwoLevelHierarchy(mf.vmult, smoother, R, O.direct_coarse_solver(Ac), 1, False)
    x0 = O.random_initial_guess(mesh.n_dofs, con)
    b = np.zeros(mesh.n_dofs)
    res_o, rate_o, x_o = O.vcycle_history(ho, mf.vmult, b, x0)
    op = M.MatrixFreeLaplace(ctx, prob)
    res_g, x_g = gpu_history(ctx, h, lambda y, x: op.vmult(y, x), b, x0)
    np.testing.assert_allclose(res_g, res_o, rtol=HIST_TOL, atol=HIST_ATOL)
    assert res_g[-1] / res_g[-2] == pytest.approx(rate_o, rel=1e-8)
    assert rate_o < 0.5


@pytest.mark.parametrize("n,material,degree", [((16, 16, 16), "constant", 3), ((12, 10, 6), "linear", 2), ((66, 8, 6), "constant", 3),
                                               ((132, 10, 8), "constant", 3)])   # (the last two: b_c = R (A x - b) in one pass from FP32 x, b)
def test_fine_level_in_fp32_against_the_fp64_oracle(ctx, n, material, degree):
    """BASELINE.json configs[4] (FP32): the cycle with the fine level in float (operator, smoother and residual through
    the FP32 instance of the matrix-free kernel, coarse levels in FP64) against the FP64 oracle with the same R and
    the same polynomial: residual history to 1e-4 relative, down to the rounding floor of a float iterate."""
    mesh = O.StructuredMesh(n)
    coef = O.coefficient_table(mesh, material)
    con = mesh.constrained_mask()
    mf = O.MatrixFreeLaplace(mesh, coef)
    prob = M.LaplaceProblem(n, material, device="cuda")
    params = base_params(smoother={"type": "Chebyshev", "degree": degree, "smoothing_range": 20.0})
    params["fine level precision"] = "float"
    h = M.Hierarchy(ctx, "HipMatrixFreeMeshEvaluator", prob, params)
    deg, lmin, lmax = h.smoother_info()
    R = h.restrictor().to_scipy()
    Ac = O.galerkin_coarse_matrix(mf.vmult, R)
    p = O.ChebyshevParams(degree=degree, lambda_max=lmax, lambda_min=lmin)
    dinv = mf.diagonal_inverse()
    ho = O.TwoLevelHierarchy(mf.vmult, lambda b, x: O.chebyshev_smoother_apply(mf.vmult, dinv, p, b, x), R,
                             O.direct_coarse_solver(Ac), 1, False)
    x0 = O.random_initial_guess(mesh.n_dofs, con).astype(np.float32).astype(np.float64)   # representable in float
    b = np.zeros(mesh.n_dofs)
    res_o, _, _ = O.vcycle_history(ho, mf.vmult, b, x0, n_cycles=8)
    # the float cycle, residual monitored in FP64
    op = M.MatrixFreeLaplace(ctx, prob)
    xf = torch.from_numpy(x0.astype(np.float32)).cuda()
    bf = torch.zeros_like(xf)
    r = torch.empty(mesh.n_dofs, dtype=torch.float64, device="cuda")

    def norm():
        op.vmult(r, xf.double())
        return ctx.l2_norm(r)
    r0 = norm()
    res_f = [1.0]
    for _ in range(8):
        h.apply_f32(bf, xf)
        res_f.append(norm() / r0)
    np.testing.assert_allclose(res_f, res_o, rtol=1e-4, atol=2e-6)
    assert res_f[-1] < 1e-2                                          # and it does converge
    # the FP64 cycle of the same hierarchy object is untouched
    res_g, _ = gpu_history(ctx, h, lambda y, x: op.vmult(y, x), b, x0, n_cycles=8)
    np.testing.assert_allclose(res_g, res_o, rtol=HIST_TOL, atol=HIST_ATOL)
    with pytest.raises(L.MfmgInvalidArgument, match="fine level precision"):
        M.Hierarchy(ctx, "HipMatrixFreeMeshEvaluator", prob, base_params()).apply_f32(bf, xf)


@pytest.mark.parametrize("n,material,numbering", [((12, 10, 6), "linear", "lexicographic"), ((8, 6, 4), "discontinuous", "lexicographic"),
                                                  ((8, 8, 2), "constant", "lexicographic"), ((6, 8, 10), "linear", "random")])
def test_galerkin_product_on_device_equals_host(ctx, n, material, numbering):
    """R A R^T of the matrix-free operator by probing on the device (27 n_eig applications of R^T, A, R over colour
    classes of agglomerates; fewer colours where a direction has fewer than three agglomerates) against the host
    triple product and the oracle's Galerkin matrix."""
    nd = int(np.prod([v + 1 for v in n]))
    perm = torch.from_numpy(np.random.default_rng(3).permutation(nd)) if numbering == "random" else None
    params = base_params(smoother={"type": "Chebyshev", "degree": 2})
    mats = {}
    try:
        for on_device in (True, False):
            ctx.set_galerkin_on_device(on_device)
            prob = M.LaplaceProblem(n, material, device="cuda", dof_numbering=perm)
            mats[on_device] = M.Hierarchy(ctx, "HipMatrixFreeMeshEvaluator", prob, params).coarse_operator().to_scipy()
    finally:
        ctx.set_galerkin_on_device(True)
    scale = abs(mats[False]).max()
    assert abs(mats[True] - mats[False]).max() < 1e-12 * scale
    assert mats[True].shape == mats[False].shape and mats[True].nnz >= mats[False].nnz - 0
    if numbering == "lexicographic":
        mesh = O.StructuredMesh(n)
        coef = O.coefficient_table(mesh, material)
        mf = O.MatrixFreeLaplace(mesh, coef)
        R = O.build_restrictor(mesh, coef, mf.diagonal(), n_eig=2, variant="mf", eig_mode="krylov").csr
        Ac = O.galerkin_coarse_matrix(mf.vmult, R)
        assert abs(mats[True] - Ac).max() < 1e-11 * abs(Ac).max()


@pytest.mark.parametrize("n,numbering", [((8, 6, 4), "lexicographic"), ((6, 6, 6), "random"), ((10, 4, 2), "lexicographic")])
def test_agglomerate_wise_restrictor_equals_csr(ctx, n, numbering):
    """The agglomerate-wise evaluation of R and R^T (structured_restrictor.hpp) against the CSR kernels and
    scipy on the same R, for a lexicographic and a renumbered DoF set; the V-cycle agrees to rounding."""
    rng = np.random.default_rng(12)
    nd = int(np.prod([v + 1 for v in n]))
    perm = torch.from_numpy(rng.permutation(nd)) if numbering == "random" else None
    hs = {}
    for structured in (True, False):
        prob = M.LaplaceProblem(n, "linear", device="cuda", dof_numbering=perm)
        params = base_params(smoother={"type": "Chebyshev", "degree": 2}, restrictor={"structured": structured})
        hs[structured] = (M.Hierarchy(ctx, "HipMatrixFreeMeshEvaluator", prob, params), prob)
    R = hs[False][0].restrictor().to_scipy()
    assert abs(hs[True][0].restrictor().to_scipy() - R).max() == 0.0
    xf, xc = rng.random(R.shape[1]), rng.random(R.shape[0])
    for structured in (True, False):
        h = hs[structured][0]
        yc = torch.empty(R.shape[0], dtype=torch.float64, device="cuda")
        h.restrictor_apply(1, dev(xf), yc)
        ctx.synchronize()
        np.testing.assert_allclose(yc.cpu().numpy(), R @ xf, rtol=1e-13, atol=1e-14)
        yf = torch.full((R.shape[1],), np.nan, dtype=torch.float64, device="cuda")
        h.restrictor_apply(1, dev(xc), yf, L.TRANS)
        ctx.synchronize()
        np.testing.assert_allclose(yf.cpu().numpy(), R.T @ xc, rtol=1e-13, atol=1e-14)
    b = np.zeros(nd)
    x0 = rng.random(nd) * (hs[True][1].constrained.cpu().numpy() != 1)
    out = []
    for structured in (True, False):
        h, prob = hs[structured]
        op = M.MatrixFreeLaplace(ctx, prob)
        res, _ = gpu_history(ctx, h, lambda yy, xx: op.vmult(yy, xx), b, x0, n_cycles=8)
        out.append(res)
    np.testing.assert_allclose(out[0], out[1], rtol=1e-10, atol=HIST_ATOL)


@pytest.mark.parametrize("name", ["mf_cheb3_8x8x8_linear", "mf_cheb3_12x6x4_constant"])
def test_vcycle_against_committed_vectors(ctx, name):
    """The HIP path against tests/golden/oracle_vcycle_*.npz (inputs and expected outputs frozen by
    tests/golden/make_fixtures.py): operator, setup (R, A_c, smoother bounds) and residual history."""
    import os
    f = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"oracle_vcycle_{name}.npz"))
    n = tuple(int(v) for v in f["cells"])
    prob = M.LaplaceProblem(n, str(f["material"]), device="cuda")
    op = M.MatrixFreeLaplace(ctx, prob)
    y = torch.empty(prob.n_dofs, dtype=torch.float64, device="cuda")
    op.vmult(y, dev(f["x0"]))
    ctx.synchronize()
    np.testing.assert_allclose(y.cpu().numpy(), f["vmult_x0"], rtol=1e-12, atol=1e-13)
    params = base_params(smoother={"type": "Chebyshev", "degree": int(f["degree"]), "smoothing_range": 20.0})
    h = M.Hierarchy(ctx, "HipMatrixFreeMeshEvaluator", prob, params)
    deg, lmin, lmax = h.smoother_info()
    assert lmax == pytest.approx(float(f["lambda_max"]), rel=1e-10)
    assert lmin == pytest.approx(float(f["lambda_min"]), rel=1e-10)
    R = h.restrictor().to_scipy()
    Rf = sp.csr_matrix((f["R_data"], f["R_indices"], f["R_indptr"]), shape=tuple(f["R_shape"]))
    assert abs(R - Rf).max() < 1e-11
    Acf = sp.csr_matrix((f["Ac_data"], f["Ac_indices"], f["Ac_indptr"]), shape=(Rf.shape[0],) * 2)
    assert abs(h.coarse_operator().to_scipy() - Acf).max() < 1e-11 * abs(Acf).max()
    res_g, x_g = gpu_history(ctx, h, lambda yy, xx: op.vmult(yy, xx), f["b"], f["x0"])
    np.testing.assert_allclose(res_g, f["history"], rtol=HIST_TOL, atol=HIST_ATOL)


@pytest.mark.parametrize("start", ["dealii", "hashed"])
def test_eigenvalue_estimate_matches_dealii_restatement(ctx, start):
    n = (8, 8, 8)
    mesh = O.StructuredMesh(n)
    coef = O.coefficient_table(mesh, "linear")
    mf = O.MatrixFreeLaplace(mesh, coef)
    h = M.Hierarchy(ctx, "HipMatrixFreeMeshEvaluator", M.LaplaceProblem(n, "linear", device="cuda"),
                    base_params(smoother={"type": "Chebyshev", "degree": 2, "eig_start_vector": start}))
    deg, lmin, lmax = h.smoother_info()
    p = O.dealii_chebyshev_params(mf.vmult, mf.diagonal_inverse(), mesh.n_dofs, degree=2, start=start)
    assert lmax == pytest.approx(p.lambda_max, rel=1e-10)
    assert lmin == pytest.approx(p.lambda_min, rel=1e-10)


def test_gold_cuda_jacobi_on_gpu(ctx):
    """tests/test_hierarchy_device.cu:359-420 replayed on the MI355X: assembled operator, Jacobi,
    dense coarse solve, x0 random on all DoFs in deal.II DoF-id order; R from the LAPACK-based oracle
    (the gold depends on LAPACK's basis of a degenerate eigenspace)."""
    n = (4, 4, 4)
    mesh = O.StructuredMesh(n)
    coef = O.coefficient_table(mesh)
    con = mesh.constrained_mask()
    A = O.assemble_csr(mesh, coef)
    R = O.build_restrictor(mesh, np.ones_like(coef), A.diagonal(), n_eig=2, variant="device", eig_mode="lapack").csr
    prob = M.LaplaceProblem(n, device="cuda")
    params = base_params(smoother={"type": "Jacobi"}, solver={"type": "lu_dense"})
    h = M.Hierarchy(ctx, "HipMeshEvaluator", prob, params)
    assert h.n_levels == 2 and h.level_size(0) == 125 and h.level_size(1) == 16
    h.set_restrictor(R)
    dn = O.dealii_global_numbering(mesh)
    x0 = O.random_initial_guess(mesh.n_dofs, con, order=np.argsort(dn), zero_constrained=False)
    Ad = M.SparseMatrixDevice(ctx, A)
    res, _ = gpu_history(ctx, h, lambda y, x: Ad.vmult(y, x), np.zeros(mesh.n_dofs), x0)
    assert res[-1] / res[-2] == pytest.approx(0.14933479171507894, rel=1e-8)   # 1e-6 %
    # and the oracle history on the same data
    dinv = 1.0 / A.diagonal()
    Ac = (R @ A @ R.T).tocsr()
    smoother = lambda b, x: O.smoother_wrapper(lambda v: A @ v, lambda r: dinv * r, b, x)
    ho = O.TwoLevelHierarchy(lambda v: A @ v, smoother, R, O.direct_coarse_solver(Ac), 1, False)
    res_o, _, _ = O.vcycle_history(ho, lambda v: A @ v, np.zeros(mesh.n_dofs), x0)
    np.testing.assert_allclose(res, res_o, rtol=HIST_TOL, atol=HIST_ATOL)


@pytest.mark.parametrize("dim", [2, 3])
def test_assembled_jacobi_own_setup(ctx, dim):
    """Assembled CSR path end to end with the product's own restrictor (2-D: BASELINE configs[0] mesh)."""
    n = (8, 8) if dim == 2 else (8, 8, 8)
    mesh = O.StructuredMesh(n)
    coef = O.coefficient_table(mesh, "linear")
    con = mesh.constrained_mask()
    A = O.assemble_csr(mesh, coef)
    prob = M.LaplaceProblem(n, "linear", device="cuda")
    h = M.Hierarchy(ctx, "HipMeshEvaluator", prob, base_params(smoother={"type": "Jacobi"}))
    R = h.restrictor().to_scipy()
    Ac = (R @ A @ R.T).tocsr()
    assert abs(h.coarse_operator().to_scipy() - Ac).max() < 1e-12 * abs(Ac).max()
    dinv = 1.0 / A.diagonal()
    smoother = lambda b, x: O.smoother_wrapper(lambda v: A @ v, lambda r: dinv * r, b, x)
    ho = O.TwoLevelHierarchy(lambda v: A @ v, smoother, R, O.direct_coarse_solver(Ac), 1, False)
    rng = np.random.default_rng(0)
    x0 = np.where(con, 0.0, rng.random(mesh.n_dofs))
    b = np.where(con, 0.0, rng.random(mesh.n_dofs))     # non-zero right-hand side
    res_o, rate, _ = O.vcycle_history(ho, lambda v: A @ v, b, x0, n_cycles=12)
    Ad = M.SparseMatrixDevice(ctx, A)
    res_g, _ = gpu_history(ctx, h, lambda y, x: Ad.vmult(y, x), b, x0, n_cycles=12)
    np.testing.assert_allclose(res_g, res_o, rtol=HIST_TOL, atol=HIST_ATOL)
    assert np.all(np.diff(res_g) < 0)


def test_components_and_preconditioner_mode(ctx):
    """Level accessors (include/mfmg/common/level.hpp:30-48) one by one + 'is preconditioner' zeroing."""
    n = (8, 6, 6)
    mesh = O.StructuredMesh(n)
    coef = O.coefficient_table(mesh, "linear_x")
    mf = O.MatrixFreeLaplace(mesh, coef)
    prob = M.LaplaceProblem(n, "linear_x", device="cuda")
    params = base_params(smoother={"type": "Chebyshev", "degree": 3, "lambda_max": 1.8, "lambda_min": 0.12})
    params["is preconditioner"] = True
    h = M.Hierarchy(ctx, "HipMatrixFreeMeshEvaluator", prob, params)
    R = h.restrictor().to_scipy()
    Ac = h.coarse_operator().to_scipy()
    nf, nc = h.level_size(0), h.level_size(1)
    rng = np.random.default_rng(7)
    xf, bf, xc = rng.random(nf), rng.random(nf), rng.random(nc)
    out_f = torch.empty(nf, dtype=torch.float64, device="cuda")
    out_c = torch.empty(nc, dtype=torch.float64, device="cuda")
    h.operator_apply(0, dev(xf), out_f)
    np.testing.assert_allclose(out_f.cpu().numpy(), mf.vmult(xf), rtol=1e-12, atol=1e-13)
    h.operator_apply(1, dev(xc), out_c)
    np.testing.assert_allclose(out_c.cpu().numpy(), Ac @ xc, rtol=1e-12, atol=1e-13)
    h.restrictor_apply(1, dev(xf), out_c)
    np.testing.assert_allclose(out_c.cpu().numpy(), R @ xf, rtol=1e-12, atol=1e-13)
    h.restrictor_apply(1, dev(xc), out_f, L.TRANS)
    np.testing.assert_allclose(out_f.cpu().numpy(), R.T @ xc, rtol=1e-12, atol=1e-13)
    h.coarse_apply(dev(xc), out_c)
    np.testing.assert_allclose(out_c.cpu().numpy(), np.linalg.solve(Ac.toarray(), xc), rtol=1e-9)
    p = O.ChebyshevParams(3, 1.8, 0.12)
    xs = dev(xf)
    h.smoother_apply(0, dev(bf), xs)
    np.testing.assert_allclose(xs.cpu().numpy(), O.chebyshev_smoother_apply(mf.vmult, mf.diagonal_inverse(), p, bf, xf),
                               rtol=1e-11, atol=1e-12)
    with pytest.raises(L.MfmgNotImplementedError):
        h.operator_apply(0, dev(xf), out_f, L.TRANS)      # cuda_matrix_free_operator.cu:64-71
    # preconditioner mode: garbage in x must not matter (hierarchy.hpp:253-259)
    smoother = lambda b, x: O.chebyshev_smoother_apply(mf.vmult, mf.diagonal_inverse(), p, b, x)
    ho = O.TwoLevelHierarchy(mf.vmult, smoother, R, O.direct_coarse_solver(Ac), 1, True)
    xg = dev(1e6 * rng.random(nf))
    h.vmult(xg, dev(bf))
    np.testing.assert_allclose(xg.cpu().numpy(), ho.apply(bf, np.zeros(nf)), rtol=1e-10, atol=1e-12)


def test_pcg_coarse_solver_parity(ctx):
    n = (16, 16, 16)
    mesh = O.StructuredMesh(n)
    coef = O.coefficient_table(mesh, "constant")
    con = mesh.constrained_mask()
    mf = O.MatrixFreeLaplace(mesh, coef)
    prob = M.LaplaceProblem(n, device="cuda")
    params = base_params(smoother={"type": "Chebyshev", "degree": 3, "smoothing_range": 20.0},
                         solver={"type": "pcg", "n_iterations": 12})
    h = M.Hierarchy(ctx, "HipMatrixFreeMeshEvaluator", prob, params)
    deg, lmin, lmax = h.smoother_info()
    R = h.restrictor().to_scipy()
    Ac = h.coarse_operator().to_scipy()
    p = O.ChebyshevParams(deg, lmax, lmin)
    smoother = lambda b, x: O.chebyshev_smoother_apply(mf.vmult, mf.diagonal_inverse(), p, b, x)
    ho = O.TwoLevelHierarchy(mf.vmult, smoother, R, O.pcg_coarse_solver(Ac, 12), 1, False)
    x0 = O.random_initial_guess(mesh.n_dofs, con)
    b = np.zeros(mesh.n_dofs)
    res_o, _, _ = O.vcycle_history(ho, mf.vmult, b, x0, n_cycles=10)
    op = M.MatrixFreeLaplace(ctx, prob)
    res_g, _ = gpu_history(ctx, h, lambda y, x: op.vmult(y, x), b, x0, n_cycles=10)
    np.testing.assert_allclose(res_g, res_o, rtol=1e-9, atol=HIST_ATOL)


@pytest.mark.parametrize("cells", [(64, 128, 128), (128, 64, 96), (96, 96, 96)])
def test_smoother_bounds_on_large_lexicographic_meshes(ctx, cells):
    """The eigenvalue estimate behind the Chebyshev bounds starts from a hash of the DoF id.  A multiplicative hash of
    consecutive ids is a low-discrepancy sequence (smooth in index space): on a 65 x 129 x 129 mesh it gave
    lambda_max = 1.41 where the truth is 1.5, and the cycle contracted at 0.78 instead of 0.2.  The splitmix64 finaliser
    must not under-estimate on such meshes."""
    prob = M.LaplaceProblem(cells, device="cuda", cell_size=(1.0 / 64,) * 3)       # cubic cells: lambda_max(D^-1 A) = 1.5
    op = M.MatrixFreeLaplace(ctx, prob)
    h = M.Hierarchy(ctx, "HipMatrixFreeMeshEvaluator", prob,
                    base_params(smoother={"type": "Chebyshev", "degree": 3, "smoothing_range": 20.0}, solver={"type": "amg"}))
    _, lmin, lmax = h.smoother_info()
    assert 1.47 < lmax < 1.9, lmax
    x = torch.rand(prob.n_dofs, dtype=torch.float64, device="cuda") * (prob.constrained == 0)
    b = torch.zeros_like(x)
    r = torch.empty_like(x)
    norms = []
    for _ in range(5):
        op.vmult(r, x)
        norms.append(ctx.l2_norm(r))
        h.apply(b, x)
    assert all(norms[i + 1] < 0.4 * norms[i] for i in range(4)), norms


def test_smoother_bounds_cover_the_spectrum_at_row_length_128(ctx):
    """lambda_max(D^-1 A) of the Q1 Laplacian is 1.5; the hashed start vector must not under-estimate
    it on a lexicographic mesh with 128 DoFs per row (deal.II's i % 11 pattern does: 1.08)."""
    prob = M.LaplaceProblem((127, 15, 15), device="cuda")
    h = M.Hierarchy(ctx, "HipMatrixFreeMeshEvaluator", prob,
                    base_params(smoother={"type": "Chebyshev", "degree": 3, "smoothing_range": 20.0},
                                solver={"type": "pcg", "n_iterations": 5}))
    _, lmin, lmax = h.smoother_info()
    assert lmax > 1.45
    # and the cycle contracts
    x = torch.rand(prob.n_dofs, dtype=torch.float64, device="cuda") * (prob.constrained == 0)
    b = torch.zeros_like(x)
    op = M.MatrixFreeLaplace(ctx, prob)
    r = torch.empty_like(x)
    norms = []
    for _ in range(4):
        op.vmult(r, x)
        norms.append(ctx.l2_norm(r))
        h.apply(b, x)
    assert norms[3] < norms[2] < norms[1] < norms[0]


@pytest.mark.parametrize("n_cycles,cells,pre_levels", [(1, 32, None), (2, 32, None), (1, 64, None), (1, 128, None),
                                                        (1, 32, 0), (2, 32, 1), (1, 64, 0), (1, 128, 0)])
def test_amg_coarse_solver_parity(ctx, n_cycles, cells, pre_levels):
    """solver.type amg: the V-cycle over the aggregation hierarchy on the GPU against its oracle restatement,
    run on the level matrices downloaded from the product.  64 cells per direction: A_c (65536 rows) in block-diagonal
    storage with regular rows, stencil classes and listed rows, its prolongator in node classes, the restrictor with
    block classes -- the table-driven kernels against the oracle's scipy products of the same matrices."""
    n = (cells,) * 3
    mesh = O.StructuredMesh(n)
    coef = O.coefficient_table(mesh, "constant")
    con = mesh.constrained_mask()
    mf = O.MatrixFreeLaplace(mesh, coef)
    prob = M.LaplaceProblem(n, device="cuda")
    amg = {"coarsest_size": 600, "n_cycles": n_cycles}
    if pre_levels is not None:          # V(0,1) from that level of the aggregation hierarchy on (the bench runs 0)
        amg["pre_smoothing_levels"] = pre_levels
    params = base_params(smoother={"type": "Chebyshev", "degree": 3, "smoothing_range": 20.0},
                         solver={"type": "amg", "amg": amg})
    h = M.Hierarchy(ctx, "HipMatrixFreeMeshEvaluator", prob, params)
    levels = h.coarse_amg_levels()
    assert [A.shape[0] for A, _, _ in levels] == {32: [8192, 1024, 128], 64: [65536, 8192, 1024, 128], 128: [524288, 65536, 8192, 1024, 128]}[cells]
    if cells >= 64:
        kernels = {(l, w): (kind, classes) for l, w, _, kind, classes, _ in h.coarse_amg_kernels()}
        assert kernels[(0, 0)][0] == 3 and kernels[(0, 0)][1] >= 20 and kernels[(0, 1)][0] == 5
    deg, lmin, lmax = h.smoother_info()
    R = h.restrictor().to_scipy()
    # coarse solve alone
    solve = O.amg_coarse_solver(levels, n_cycles, pre_smoothing_levels=pre_levels)
    bc = np.random.default_rng(1).random(R.shape[0])
    xc = torch.empty(R.shape[0], dtype=torch.float64, device="cuda")
    h.coarse_apply(dev(bc), xc)
    ref = solve(bc)
    assert np.abs(xc.cpu().numpy() - ref).max() < 1e-10 * np.abs(ref).max()
    # whole cycle
    p = O.ChebyshevParams(deg, lmax, lmin)
    smoother = lambda b, x: O.chebyshev_smoother_apply(mf.vmult, mf.diagonal_inverse(), p, b, x)
    ho = O.TwoLevelHierarchy(mf.vmult, smoother, R, solve, 1, False)
    x0 = O.random_initial_guess(mesh.n_dofs, con)
    b = np.zeros(mesh.n_dofs)
    n_hist = 10 if cells <= 64 else 4          # (the numpy operator takes a second per application at 129^3 DoFs)
    res_o, rate, _ = O.vcycle_history(ho, mf.vmult, b, x0, n_cycles=n_hist)
    op = M.MatrixFreeLaplace(ctx, prob)
    res_g, _ = gpu_history(ctx, h, lambda y, x: op.vmult(y, x), b, x0, n_cycles=n_hist)
    np.testing.assert_allclose(res_g, res_o, rtol=HIST_TOL, atol=HIST_ATOL)     # 1e-10, BASELINE.json's tolerance
    # sanity band of the contraction (res[n] / res[n-1] of the LAST cycle, which is larger than the mean: measured 0.368 at
    # 64 cells after 10 cycles with the round-2/3 setup); the parity statement is the line above
    assert rate < 0.4, rate


def test_error_conventions(ctx):
    prob = M.LaplaceProblem((4, 4, 4), device="cuda")
    with pytest.raises(L.MfmgError, match="Unknown smoother name"):       # cuda_smoother.cu:110
        M.Hierarchy(ctx, "HipMeshEvaluator", prob, base_params(smoother={"type": "Gauss-Seidel"}))
    with pytest.raises(L.MfmgError, match="Unknown solver name"):         # cuda_solver.cu:71
        M.Hierarchy(ctx, "HipMeshEvaluator", prob, base_params(solver={"type": "multifrontal"}))
    with pytest.raises(L.MfmgNotImplementedError):                        # no AmgX shim
        M.Hierarchy(ctx, "HipMeshEvaluator", prob, base_params(solver={"type": "amgx"}))
    with pytest.raises(L.MfmgNotImplementedError):                        # hierarchy.hpp:49-107 string switch
        M.Hierarchy(ctx, "DealIIMeshEvaluator", prob, base_params())
    with pytest.raises(L.MfmgError, match="must be positive"):            # hierarchy.hpp:173-175
        M.Hierarchy(ctx, "HipMeshEvaluator", prob, base_params(**{"max levels": 0}))
    h = M.Hierarchy(ctx, "HipMeshEvaluator", prob, base_params())
    with pytest.raises(ValueError):
        h.apply(torch.zeros(125, dtype=torch.float64), torch.zeros(125, dtype=torch.float64))   # host tensors
    assert "Setup: build restrictor" in h.timer_report()


def test_single_level_hierarchy_is_coarse_solve(ctx):
    n = (4, 4, 4)
    mesh = O.StructuredMesh(n)
    A = O.assemble_csr(mesh, O.coefficient_table(mesh))
    h = M.Hierarchy(ctx, "HipMeshEvaluator", M.LaplaceProblem(n, device="cuda"), base_params(**{"max levels": 1}))
    assert h.n_levels == 1
    b = np.random.default_rng(0).random(125)
    x = dev(np.zeros(125))
    h.apply(dev(b), x)
    np.testing.assert_allclose(x.cpu().numpy(), np.linalg.solve(A.toarray(), b), rtol=1e-10)


@pytest.mark.parametrize("material", ["constant", "linear"])
def test_full_size_hierarchy_properties(ctx, material):
    """BASELINE.json's per-GPU size (256^3 cells = 257^3 DoFs, Chebyshev(3), AMG coarse solve), where the numpy
    oracle does not finish in seconds: size-independent properties of every operator on the path --
    symmetry of A and A_c, adjointness of restriction and prolongation, R (weights) partition of unity on the
    interior, the smoother as an affine map, contraction of the cycle -- and the kernel variants in use."""
    n = (256, 256, 256)
    prob = M.LaplaceProblem(n, material, device="cuda")
    params = base_params(smoother={"type": "Chebyshev", "degree": 3, "smoothing_range": 20.0},
                         solver={"type": "amg"})
    h = M.Hierarchy(ctx, "HipMatrixFreeMeshEvaluator", prob, params)
    nf, nc = h.level_size(0), h.level_size(1)
    assert nf == 257 ** 3 and nc == 2 * 128 ** 3
    constant = material == "constant"          # linear: eight coefficients per cell, every coarse row stored
    g = torch.Generator(device="cuda").manual_seed(3)
    free = (prob.constrained == 0).to(torch.float64)
    rnd = lambda m: torch.rand(m, dtype=torch.float64, device="cuda", generator=g)
    x, y = rnd(nf) * free, rnd(nf) * free
    u, v = rnd(nc), rnd(nc)
    ax, ay = torch.empty_like(x), torch.empty_like(x)
    h.operator_apply(0, x, ax)
    h.operator_apply(0, y, ay)
    assert abs(ctx.dot(ax, y) - ctx.dot(x, ay)) < 1e-11 * abs(ctx.dot(ax, y))           # A symmetric
    au, av = torch.empty_like(u), torch.empty_like(u)
    h.operator_apply(1, u, au)
    h.operator_apply(1, v, av)
    assert abs(ctx.dot(au, v) - ctx.dot(u, av)) < 1e-11 * abs(ctx.dot(au, v))           # A_c symmetric
    assert ctx.dot(au, u) > 0.0                                                          # ... and positive
    rx, rtu = torch.empty_like(u), torch.empty_like(x)
    h.restrictor_apply(1, x, rx)
    h.restrictor_apply(1, u, rtu, L.TRANS)
    assert abs(ctx.dot(rx, u) - ctx.dot(x, rtu)) < 1e-11 * abs(ctx.dot(rx, u))          # <R x, u> = <x, R^T u>
    # Galerkin: <R A R^T u, v> = <A_c u, v>
    t1, t2, t3 = torch.empty_like(x), torch.empty_like(x), torch.empty_like(u)
    h.restrictor_apply(1, u, t1, L.TRANS)
    h.operator_apply(0, t1, t2)
    h.restrictor_apply(1, t2, t3)
    assert abs(ctx.dot(t3, v) - ctx.dot(au, v)) < 1e-9 * abs(ctx.dot(au, v))
    # smoother: S(b, x) is affine in (b, x): S(b, x) - S(0, 0) linear
    b1, b2 = rnd(nf) * free, rnd(nf) * free
    def smooth(b, x0):
        xx = x0.clone()
        h.smoother_apply(0, b, xx)
        return xx
    s1, s2 = smooth(b1, x), smooth(b2, y)
    s12 = smooth((2.0 * b1 - 3.0 * b2).contiguous(), (2.0 * x - 3.0 * y).contiguous())
    ctx.synchronize()
    assert (s12 - (2.0 * s1 - 3.0 * s2)).abs().max().item() < 1e-11 * s1.abs().max().item()
    # cycle: contraction of the residual for b = 0
    xx = x.clone()
    b = torch.zeros_like(xx)
    r = torch.empty_like(xx)
    norms = []
    for _ in range(5):
        h.operator_apply(0, xx, r)
        norms.append(ctx.l2_norm(r))
        h.apply(b, xx)
    assert all(norms[i + 1] < 0.4 * norms[i] for i in range(4)), norms
    # the layouts the design relies on were actually chosen at this size
    assert h.coarse_operator().get_kernel()[1] == 3          # block diagonals for A_c, upper half stored
    kernels = {(l, w): (rows, kind, classes, listed) for l, w, rows, kind, classes, listed in h.coarse_amg_kernels()}
    if constant:
        nw_, ty_, tz_ = h.operator_tile()                      # a compiled row count, at least two wavefronts
        assert ty_ in (2, 3, 4) and nw_ * ty_ >= 2 and tz_ >= 1
        for l in (0, 1, 2):                                  # A_c and the two levels below: tables for >= 99.9 % of the rows
            rows, kind, classes, listed = kernels[(l, 0)]
            assert kind == 3 and classes >= 50 and listed <= 0.02 * rows, kernels[(l, 0)]
        for key in ((0, 1), (0, 2), (1, 1), (1, 2)):         # the first two prolongators and their transposes: node classes
            rows, kind, classes, listed = kernels[key]
            assert kind == 5 and classes >= 10 and listed <= 0.02 * rows, (key, kernels[key])
    else:
        assert not h.coarse_operator().regular_rows()        # no two rows alike: stored planes
        assert kernels[(1, 0)][1] == 3 and kernels[(2, 0)][1] == 1   # 125 diagonals stored, 343 left to the CSR kernel
    # the table-driven paths against the stored values / plain CSR kernels of the same matrices: same cycle to rounding
    h.coarse_amg_kernels(regular_rows=False)
    xx2 = x.clone()
    norms2 = []
    for _ in range(5):
        h.operator_apply(0, xx2, r)
        norms2.append(ctx.l2_norm(r))
        h.apply(b, xx2)
    h.coarse_amg_kernels(regular_rows=True)
    np.testing.assert_allclose(norms2, norms, rtol=1e-10)


@pytest.mark.parametrize("material", ["constant", "linear"])
def test_full_size_assembled_hierarchy_properties(ctx, material):
    """BASELINE.json configs[2] at size: 256^3 cells with the ASSEMBLED fine operator (HipMeshEvaluator: 17 M rows,
    27 entries per row in CSR; restriction / prolongation as SpMV).  The numpy oracle does not finish at this size,
    so the size-independent properties of every operator on the path are checked, and the table-driven variants of
    the fine and coarse matrices against their stored values / plain CSR kernels."""
    n = (256, 256, 256)
    prob = M.LaplaceProblem(n, material, device="cuda")
    params = base_params(smoother={"type": "Chebyshev", "degree": 3, "smoothing_range": 20.0, "lambda_max": 1.8, "lambda_min": 0.09},
                         solver={"type": "amg"})
    h = M.Hierarchy(ctx, "HipMeshEvaluator", prob, params)
    nf, nc = h.level_size(0), h.level_size(1)
    assert nf == 257 ** 3 and nc == 2 * 128 ** 3
    g = torch.Generator(device="cuda").manual_seed(5)
    free = (prob.constrained == 0).to(torch.float64)
    rnd = lambda m: torch.rand(m, dtype=torch.float64, device="cuda", generator=g)
    x, y = rnd(nf) * free, rnd(nf) * free
    u, v = rnd(nc), rnd(nc)
    ax, ay = torch.empty_like(x), torch.empty_like(x)
    h.operator_apply(0, x, ax)
    h.operator_apply(0, y, ay)
    assert abs(ctx.dot(ax, y) - ctx.dot(x, ay)) < 1e-11 * abs(ctx.dot(ax, y))           # A symmetric on the free DoFs
    # the assembled operator and the matrix-free one agree on vectors that vanish on the Dirichlet DoFs
    # (tests/test_hierarchy.cc:644-695: matrix-free vs assembled, < 1e-9)
    mf = M.MatrixFreeLaplace(ctx, prob)
    amf = torch.empty_like(x)
    mf.vmult(amf, x)
    ctx.synchronize()
    assert ((ax - amf) * free).abs().max().item() < 1e-9 * ax.abs().max().item()
    del mf, amf
    au, av = torch.empty_like(u), torch.empty_like(u)
    h.operator_apply(1, u, au)
    h.operator_apply(1, v, av)
    assert abs(ctx.dot(au, v) - ctx.dot(u, av)) < 1e-11 * abs(ctx.dot(au, v))           # A_c symmetric
    assert ctx.dot(au, u) > 0.0
    rx, rtu = torch.empty_like(u), torch.empty_like(x)
    h.restrictor_apply(1, x, rx)
    h.restrictor_apply(1, u, rtu, L.TRANS)
    assert abs(ctx.dot(rx, u) - ctx.dot(x, rtu)) < 1e-11 * abs(ctx.dot(rx, u))          # <R x, u> = <x, R^T u>
    t1, t2, t3 = torch.empty_like(x), torch.empty_like(x), torch.empty_like(u)
    h.restrictor_apply(1, u, t1, L.TRANS)
    h.operator_apply(0, t1, t2)
    h.restrictor_apply(1, t2, t3)
    assert abs(ctx.dot(t3, v) - ctx.dot(au, v)) < 1e-9 * abs(ctx.dot(au, v))            # Galerkin: R A R^T = A_c
    # the smoother is an affine map
    b1, b2 = rnd(nf) * free, rnd(nf) * free
    def smooth(b, x0):
        xx = x0.clone()
        h.smoother_apply(0, b, xx)
        return xx
    s1, s2 = smooth(b1, x), smooth(b2, y)
    s12 = smooth((2.0 * b1 - 3.0 * b2).contiguous(), (2.0 * x - 3.0 * y).contiguous())
    ctx.synchronize()
    assert (s12 - (2.0 * s1 - 3.0 * s2)).abs().max().item() < 1e-11 * s1.abs().max().item()
    del s1, s2, s12, b1, b2, t1, t2, t3
    # the cycle contracts; the same cycle with every table-driven matrix switched back to its stored values agrees
    def cycle_norms():
        xx = x.clone()
        b = torch.zeros_like(xx)
        r = torch.empty_like(xx)
        out = []
        for _ in range(4):
            h.operator_apply(0, xx, r)
            out.append(ctx.l2_norm(r))
            h.apply(b, xx)
        return out
    norms = cycle_norms()
    assert all(norms[i + 1] < 0.5 * norms[i] for i in range(3)), norms
    h.coarse_amg_kernels(regular_rows=False)
    np.testing.assert_allclose(cycle_norms(), norms, rtol=1e-10)
    h.coarse_amg_kernels(regular_rows=True)


@pytest.mark.parametrize("degree", [4, 5])
@pytest.mark.parametrize("evaluator", ["HipMatrixFreeMeshEvaluator", "HipMeshEvaluator"])
def test_chebyshev_degree_4_and_5(ctx, evaluator, degree):
    """Polynomial degrees above three: the fused kernels write a term into the vector that holds x_{k-1} (out aliases
    x_prev), for the matrix-free and the CSR smoother step alike; one smoother application against the oracle."""
    n = (10, 9, 8)
    mesh = O.StructuredMesh(n)
    coef = O.coefficient_table(mesh, "linear")
    con = mesh.constrained_mask()
    prob = M.LaplaceProblem(n, "linear", device="cuda")
    lmax, lmin = 1.8, 0.09
    h = M.Hierarchy(ctx, evaluator, prob, base_params(smoother={"type": "Chebyshev", "degree": degree, "lambda_max": lmax,
                                                                 "lambda_min": lmin}, solver={"type": "lu_dense"}))
    if evaluator == "HipMatrixFreeMeshEvaluator":
        mf = O.MatrixFreeLaplace(mesh, coef)
        apply_a, dinv = mf.vmult, mf.diagonal_inverse()
    else:
        A = O.assemble_csr(mesh, coef)
        apply_a, dinv = (lambda z: A @ z), 1.0 / A.diagonal()
    rng = np.random.default_rng(11)
    b = np.where(con, 0.0, rng.random(mesh.n_dofs))
    x0 = np.where(con, 0.0, rng.random(mesh.n_dofs))
    ref = O.chebyshev_smoother_apply(apply_a, dinv, O.ChebyshevParams(degree, lmax, lmin), b, x0.copy())
    x = dev(x0)
    h.smoother_apply(0, dev(b), x)
    ctx.synchronize()
    assert np.abs(x.cpu().numpy() - ref).max() < 1e-12 * np.abs(ref).max()


def test_outer_cg_driver_matches_oracle(ctx):
    """tests/hierarchy_driver.cc:103-116: CG on the matrix-free operator preconditioned by Hierarchy::vmult
    ("is preconditioner" true) -- iteration count and residual history against the oracle's CG with the
    oracle's V-cycle as preconditioner."""
    n = (12, 10, 8)
    mesh = O.StructuredMesh(n)
    coef = O.coefficient_table(mesh, "linear")
    con = mesh.constrained_mask()
    mf = O.MatrixFreeLaplace(mesh, coef)
    prob = M.LaplaceProblem(n, "linear", device="cuda")
    params = base_params(smoother={"type": "Chebyshev", "degree": 2, "smoothing_range": 20.0})
    params["is preconditioner"] = True
    h = M.Hierarchy(ctx, "HipMatrixFreeMeshEvaluator", prob, params)
    deg, lmin, lmax = h.smoother_info()
    R = h.restrictor().to_scipy()
    Ac = O.galerkin_coarse_matrix(mf.vmult, R)
    p = O.ChebyshevParams(degree=2, lambda_max=lmax, lambda_min=lmin)
    dinv = mf.diagonal_inverse()
    smoother = lambda b, x: O.chebyshev_smoother_apply(mf.vmult, dinv, p, b, x)
    ho = O.TwoLevelHierarchy(mf.vmult, smoother, R, O.direct_coarse_solver(Ac), 1, True)
    rng = np.random.default_rng(5)
    b = np.where(con, 0.0, rng.random(mesh.n_dofs))
    x0 = np.zeros(mesh.n_dofs)
    tol = 1e-9 * np.linalg.norm(b)
    xo, hist_o = O.pcg_solve(mf.vmult, lambda r: ho.apply(r, np.zeros_like(r)), b, x0, tol, 100)
    x = dev(x0)
    iters, hist_g = h.solve_cg(dev(b), x, tol, 100)
    ctx.synchronize()
    assert iters == len(hist_o) - 1 and 3 <= iters <= 25
    np.testing.assert_allclose(hist_g, hist_o, rtol=1e-7, atol=1e-3 * tol)
    assert relerr(x.cpu().numpy(), xo) < 1e-9
    # without enough iterations the driver reports no convergence, like SolverControl::NoConvergence
    with pytest.raises(L.MfmgError, match="did not reach"):
        h.solve_cg(dev(b), dev(x0), tol, 2)


def relerr(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("n,material,evaluator,variant,selection,agg", [
    ((8, 8, 8), "linear", "HipMatrixFreeMeshEvaluator", "mf", "krylov", (2, 2, 2)),
    ((9, 7, 5), "discontinuous", "HipMatrixFreeMeshEvaluator", "mf", "krylov", (2, 2, 2)),     # clipped agglomerates
    ((8, 8, 8), "linear", "HipMeshEvaluator", "device", "lapack", (2, 2, 2)),
    ((6, 6, 6), "linear_x", "HipMeshEvaluator", "host", "krylov", (2, 2, 2)),
    ((9, 6, 6), "linear", "HipMatrixFreeMeshEvaluator", "mf", "krylov", (3, 3, 3)),            # 64 nodes per agglomerate
    ((12, 10), "linear", "HipMeshEvaluator", "device", "lapack", (2, 2)),                       # 2-D, configs[0]
])
def test_restrictor_eigenproblems_on_device(ctx, n, material, evaluator, variant, selection, agg):
    """the agglomerate eigenproblems solved on the GPU (one wavefront per agglomerate, cyclic
    Jacobi in LDS, selection rules included) against the host path of the same rules (to rounding) and against the
    oracle's restrictor (numpy eigh + the same selection; 1e-11, the degenerate eigenspaces of the Krylov rule are
    basis independent)."""
    dim = len(n)
    mesh = O.StructuredMesh(n)
    coef = O.coefficient_table(mesh, material)
    prob = M.LaplaceProblem(n, material, device="cuda")
    prm = lambda where: {"eigensolver": {"number of eigenvectors": 2, "variant": variant, "selection": selection},
                         "agglomeration": dict(zip(("nx", "ny", "nz"), agg)), "is preconditioner": False, "max levels": 2,
                         "restrictor": {"eigensolver": where}, "solver": {"type": "pcg", "n_iterations": 2}}
    Rd = M.Hierarchy(ctx, evaluator, prob, prm("device")).restrictor().to_scipy()
    Rh = M.Hierarchy(ctx, evaluator, prob, prm("host")).restrictor().to_scipy()
    assert Rd.shape == Rh.shape and np.array_equal(Rd.indices, Rh.indices)
    scale = abs(Rh).max()
    assert abs(Rd - Rh).max() < 1e-12 * scale
    if evaluator == "HipMatrixFreeMeshEvaluator":
        diag = O.MatrixFreeLaplace(mesh, coef).diagonal()
    else:
        diag = O.assemble_csr(mesh, coef).diagonal()
    Ro = O.build_restrictor(mesh, coef, diag, agg=agg[:dim], n_eig=2, variant=variant, eig_mode=selection).csr
    assert Ro.shape == Rd.shape
    # rows of one agglomerate may come in either sign from a dense eigensolver: compare row by row up to sign
    D, Oo = Rd.toarray(), Ro.toarray()
    for r in range(D.shape[0]):
        e = min(np.abs(D[r] - Oo[r]).max(), np.abs(D[r] + Oo[r]).max())
        assert e < 1e-11 * scale, (r, e)


@pytest.mark.parametrize("n,material", [((16, 12, 10), "constant"), ((12, 10, 6), "discontinuous"), ((8, 8, 8), "linear")])
def test_shared_agglomerate_eigensolves_change_no_bit(ctx, monkeypatch, n, material):
    """Agglomerates with equal shape, constraint flags and cell coefficients share one eigensolve on the device
    (amge_device.hip: two 64-bit hashes of exactly that input); the restrictor must come out bit for bit as with a solve
    per agglomerate."""
    prob = M.LaplaceProblem(n, material, device="cuda")
    params = base_params(solver={"type": "pcg", "n_iterations": 2})
    R1 = M.Hierarchy(ctx, "HipMatrixFreeMeshEvaluator", prob, params).restrictor().to_scipy()
    monkeypatch.setenv("MFMG_AMGE_MEMO", "0")
    R0 = M.Hierarchy(ctx, "HipMatrixFreeMeshEvaluator", prob, params).restrictor().to_scipy()
    assert np.array_equal(R1.indptr, R0.indptr) and np.array_equal(R1.indices, R0.indices)
    assert np.array_equal(R1.data, R0.data)


@pytest.mark.parametrize("n,material,evaluator,expect", [
    ((8, 8, 8), "constant", "HipMatrixFreeMeshEvaluator", None),         # too small for classes of agglomerates
    ((4, 16, 20), "constant", "HipMatrixFreeMeshEvaluator", None),       # two agglomerates in x: everything in the list
    ((180, 12, 6), "constant", "HipMatrixFreeMeshEvaluator", True),      # 90 agglomerates in x: a full and a partial run
    ((130, 36, 6), "constant", "HipMatrixFreeMeshEvaluator", True),      # tiles of 8 agglomerate rows: 18 rows (a ragged tile), 3 layers
    ((132, 10, 8), "constant", "HipMeshEvaluator", True),                # assembled fine operator
    ((130, 2, 4), "constant", "HipMatrixFreeMeshEvaluator", None),       # one agglomerate in y (both faces), two in z
    ((128, 6, 2), "constant", "HipMatrixFreeMeshEvaluator", None),       # one agglomerate layer in z
    ((70, 66, 4), "discontinuous", "HipMatrixFreeMeshEvaluator", None),  # blocks that repeat in patterns
    ((16, 16, 16), "linear", "HipMatrixFreeMeshEvaluator", False),       # every agglomerate its own block: two steps
    ((12, 10, 6), "discontinuous", "HipMatrixFreeMeshEvaluator", None),  # whatever the classes allow: must agree
])
def test_residual_restriction_in_one_pass(ctx, monkeypatch, n, material, evaluator, expect):
    """b_c = R (A x - b) (include/mfmg/common/hierarchy.hpp:281-290) as one kernel over x and b where the rows of R A
    repeat themselves (residual_restriction.hip), against the two steps on the same restrictor, and the cycle built on
    it against the cycle without it."""
    prob = M.LaplaceProblem(n, material, device="cuda")
    params = base_params(smoother={"type": "Chebyshev", "degree": 3, "smoothing_range": 20.0})
    h = M.Hierarchy(ctx, evaluator, prob, params)
    classes = h.residual_restriction_classes()
    if expect is not None:
        assert (classes > 0) == expect
    nf, nc = h.level_size(0), h.level_size(1)
    rng = np.random.default_rng(5)
    x, b = dev(rng.standard_normal(nf)), dev(rng.standard_normal(nf))
    one = torch.empty(nc, dtype=torch.float64, device="cuda")
    h.restrict_residual(x, b, one)
    res = torch.empty(nf, dtype=torch.float64, device="cuda")
    h.operator_apply(0, x, res)
    res -= b
    two = torch.empty(nc, dtype=torch.float64, device="cuda")
    h.restrictor_apply(1, res, two)
    ctx.synchronize()
    assert (one - two).abs().max().item() <= 1e-12 * two.abs().max().item()
    # the cycle with and without it
    x0 = rng.random(nf)
    hist = []
    for fused in ("1", "0"):
        monkeypatch.setenv("MFMG_FUSED_RESIDUAL", fused)
        hh = M.Hierarchy(ctx, evaluator, prob, params)
        assert fused == "1" or hh.residual_restriction_classes() == 0
        xx, bb = dev(x0), dev(np.zeros(nf))
        for _ in range(5):
            hh.apply(bb, xx)
        ctx.synchronize()
        hist.append(xx.cpu().numpy())
    assert np.abs(hist[0] - hist[1]).max() <= 1e-12 * np.abs(x0).max()


@pytest.mark.parametrize("material", ["constant", "linear", "discontinuous", "linear_x"])
def test_full_size_vcycle_history_against_the_native_oracle(ctx, material):
    """Parity AT THE SIZE THE BENCH QUOTES (BASELINE.json: 256^3 cells = 257^3 DoFs per GPU, Chebyshev(3), multilevel
    coarse solve): the harness of /root/reference/tests/test_hierarchy.cc:76-123 (x0 random with the constrained
    entries zero, b = 0, residual norm after every cycle) run by the GPU product and by oracle/oracle_kernels.cpp
    (pinned against the numpy oracle in tests/test_oracle_native.py) on the host cores.  The oracle takes the
    product's own R, A_c and aggregation levels (downloaded), computes the operator, its diagonal, the smoother and
    every coarse product itself with plain CSR loops: 5-cycle residual history and final iterate to 1e-10 relative."""
    import oracle_native as ON

    n = (256, 256, 256)
    prob = M.LaplaceProblem(n, material, device="cuda")
    # constant: the bench's configuration (the aggregation hierarchy runs V(0,1): post-smoothing only); linear: V(1,1)
    pre_levels = 0 if material == "constant" else None
    params = base_params(smoother={"type": "Chebyshev", "degree": 3, "smoothing_range": 20.0},
                         solver={"type": "amg", "amg": {} if pre_levels is None else {"pre_smoothing_levels": pre_levels}})
    h = M.Hierarchy(ctx, "HipMatrixFreeMeshEvaluator", prob, params)
    deg, lmin, lmax = h.smoother_info()
    cd = prob.cell_dofs.cpu().numpy()
    co = prob.coefficient.cpu().numpy()
    cn = prob.constrained.cpu().numpy()
    # the operator's diagonal: oracle's own (compute_diagonal), and the product's against it
    d_o = ON.mf_diagonal(n, prob.h, cd, co, cn)
    op = M.MatrixFreeLaplace(ctx, prob)
    d_g = op.diagonal_inverse().cpu().numpy()
    np.testing.assert_allclose(d_g, 1.0 / d_o, rtol=1e-13)
    R = h.restrictor().to_scipy()
    Ac = h.coarse_operator().to_scipy()
    levels = h.coarse_amg_levels()
    x0 = np.where(cn.astype(bool), 0.0, np.random.default_rng(5).random(prob.n_dofs))
    b = np.zeros(prob.n_dofs)
    cycles = 5
    x_o, res_o = ON.vcycles(n, prob.h, cd, co, cn, 1.0 / d_o, deg, lmin, lmax, R, Ac, 0, b, x0, cycles,
                            amg_levels=levels, amg_pre_smoothing_levels=pre_levels)
    del R, Ac, levels
    res_g, x_g = gpu_history(ctx, h, lambda y, x: op.vmult(y, x), b, x0, n_cycles=cycles)
    np.testing.assert_allclose(res_g, res_o, rtol=HIST_TOL, atol=HIST_ATOL)
    assert np.abs(x_g - x_o).max() <= 1e-10 * np.abs(x0).max()
    assert res_o[-1] / res_o[-2] < (0.4 if material != "discontinuous" else 0.7)
    del h
    torch.cuda.empty_cache()
    if material in ("discontinuous", "linear_x"):
        # (the other two of the reference's four materials, /root/reference/tests/test_hierarchy_helpers.hpp:75-188 as exercised at
        # tests/test_hierarchy.cc:644-695: the matrix-free cycle at full size; the assembled and FP32 legs run on the first two)
        return
    # BASELINE.json configs[2] at its size: the SAME cycle with the ASSEMBLED fine operator (HipMeshEvaluator: CSR matrix of
    # 27 entries per row formed on the device, Chebyshev(3) through the SpMV kernels, R A R^T by probing).  With b = 0
    # and x0 = 0 on the Dirichlet DoFs the assembled rows (diagonal kept) and the matrix-free ones (identity) generate the
    # same iterates, so the matrix-free oracle run above is the reference for it as well -- given the same smoother bounds.
    pa = dict(params)
    pa["smoother"] = dict(params["smoother"], lambda_min=lmin, lambda_max=lmax)
    ha = M.Hierarchy(ctx, "HipMeshEvaluator", prob, pa)
    # (the oracle takes THIS hierarchy's R, A_c and aggregation levels: the assembled evaluator's AMGe variant differs)
    x_oa, res_oa = ON.vcycles(n, prob.h, cd, co, cn, 1.0 / d_o, deg, lmin, lmax, ha.restrictor().to_scipy(),
                              ha.coarse_operator().to_scipy(), 0, b, x0, cycles, amg_levels=ha.coarse_amg_levels(),
                              amg_pre_smoothing_levels=pre_levels)
    res_a, x_a = gpu_history(ctx, ha, lambda y, x: op.vmult(y, x), b, x0, n_cycles=cycles)
    np.testing.assert_allclose(res_a, res_oa, rtol=HIST_TOL, atol=HIST_ATOL)
    assert np.abs(x_a - x_oa).max() <= 1e-10 * np.abs(x0).max()
    del ha
    if material == "constant":
        # BASELINE.json configs[4] at size: the fine level in FP32 against the FP64 oracle, 1e-4 relative
        torch.cuda.empty_cache()
        pf = dict(params)
        pf["fine level precision"] = "float"
        hf = M.Hierarchy(ctx, "HipMatrixFreeMeshEvaluator", prob, pf)
        x0f = x0.astype(np.float32)
        x_of, res_of = ON.vcycles(n, prob.h, cd, co, cn, 1.0 / d_o, deg, lmin, lmax, hf.restrictor().to_scipy(),
                                  hf.coarse_operator().to_scipy(), 0, b, x0f.astype(np.float64), 4,
                                  amg_levels=hf.coarse_amg_levels(), amg_pre_smoothing_levels=pre_levels)
        xf = torch.from_numpy(x0f).cuda()
        bf = torch.zeros_like(xf)
        r = torch.empty(prob.n_dofs, dtype=torch.float64, device="cuda")
        op.vmult(r, xf.double())
        r0 = ctx.l2_norm(r)
        res_f = [1.0]
        for _ in range(4):
            hf.apply_f32(bf, xf)
            op.vmult(r, xf.double())
            res_f.append(ctx.l2_norm(r) / r0)
        np.testing.assert_allclose(res_f, res_of, rtol=1e-4, atol=2e-6)


@pytest.mark.parametrize("material,stored_diagonal", [("constant", False), ("constant", True), ("linear", False)])
def test_smoother_apply_at_512cubed_against_the_native_oracle(ctx, material, stored_diagonal):
    """Oracle parity WHERE north_star QUOTES ITS TARGET: the fine-level Chebyshev(3) smoother apply at 512^3 DoFs, the three
    layouts of the bench (one coefficient per cell with D^-1 derived in the kernel / kept in the records; eight coefficients per
    cell), as the product runs it (one sweep over the mesh where the operator offers it, else a launch per term), against
    oracle_native's operator and diagonal composed into the reference's smoother wrapper
    (/root/reference/source/dealii/dealii_matrix_free_smoother.cc:63-76: r = A x - b, tmp = B^-1 r, x -= tmp, B^-1 the Chebyshev
    polynomial; the three-term form on x is that order to rounding, tests/test_oracle_fixtures.py) -- three operator applications
    on the host cores, 1e-12 of the iterate's max-norm."""
    import psutil
    import oracle_native as ON
    if psutil.virtual_memory().available < 60e9:
        pytest.skip("needs ~50 GB of host memory for the 511^3-cell mesh arrays and the vectors of the oracle")
    from bench import smoother_coefficients
    n = (511, 511, 511)
    prob = M.LaplaceProblem(n, material, device="cuda")
    ctx.set_stored_diagonal(stored_diagonal)
    try:
        op = M.MatrixFreeLaplace(ctx, prob)
    finally:
        ctx.set_stored_diagonal(False)
    assert op.cell_constant_layout() == (material == "constant")
    assert op.diagonal_in_record() == (stored_diagonal or material != "constant")
    N = prob.n_dofs
    coefs = smoother_coefficients(3, 0.09, 1.8)
    rng = np.random.default_rng(3)
    x0 = rng.random(N)          # nonzero on the Dirichlet DoFs too: identity rows
    b = rng.random(N)
    xg, bg = dev(x0), dev(b)
    out = torch.empty_like(xg)
    if op.sweep_available(3):
        op.smoother_sweep([c[0] for c in coefs], [c[1] for c in coefs], bg, xg, out)
    else:
        s1, s2 = torch.empty_like(xg), torch.empty_like(xg)
        op.smoother_step(bg, xg, None, coefs[0][0], coefs[0][1], s1)
        op.smoother_step(bg, s1, xg, coefs[1][0], coefs[1][1], s2)
        op.smoother_step(bg, s2, s1, coefs[2][0], coefs[2][1], out)
        del s1, s2
    ctx.synchronize()
    got = out.cpu().numpy()
    del xg, bg, out, op
    torch.cuda.empty_cache()
    cd = prob.cell_dofs.cpu().numpy()
    co = prob.coefficient.cpu().numpy()
    cn = prob.constrained.cpu().numpy()
    del prob
    torch.cuda.empty_cache()
    dinv = 1.0 / ON.mf_diagonal(n, (1.0 / 511,) * 3, cd, co, cn)
    cur, prev = x0, None
    for al, be in coefs:
        r = ON.mf_apply(n, (1.0 / 511,) * 3, cd, co, cn, cur)
        r -= b
        r *= dinv
        nxt = cur - be * r
        if prev is not None:
            nxt += al * (cur - prev)
        prev, cur = cur, nxt
    assert np.abs(got - cur).max() <= 1e-12 * np.abs(cur).max()


@pytest.mark.parametrize("evaluator,n,material", [("HipMeshEvaluator", (8, 8), "constant"), ("HipMeshEvaluator", (8, 6, 4), "linear"),
                                                   ("HipMatrixFreeMeshEvaluator", (8, 6, 4), "linear"),
                                                   ("HipMatrixFreeMeshEvaluator", (16, 16, 16), "constant")])
def test_fast_ap(ctx, evaluator, n, material):
    """`fast_ap = true` -- the parameter set of the reference's driver (/root/reference/tests/hierarchy_driver.cc:270-272:
    fast_ap, eigensolver.type anasazi, eigensolver.tolerance 1e-3) -- takes `fast_multiply_transpose()` of the helpers
    (include/mfmg/common/hierarchy.hpp:214-221) instead of `a->multiply_transpose(restrictor)`.  The reference's own check
    (tests/test_hierarchy.cc:507-642): A R^T of the fast path against the plain one, entry by entry, 1e-9; here also
    against the oracle's assembled A times R^T, and the hierarchies built either way give the same cycle."""
    mesh = O.StructuredMesh(n)
    coef = O.coefficient_table(mesh, material)
    con = mesh.constrained_mask()
    mf = O.MatrixFreeLaplace(mesh, coef) if len(n) == 3 else None
    matrix_free = evaluator == "HipMatrixFreeMeshEvaluator"
    prob = M.LaplaceProblem(n, material, device="cuda")
    smoother = {"type": "Chebyshev", "degree": 2, "smoothing_range": 20.0} if matrix_free else {"type": "Jacobi"}
    plain = base_params(smoother=smoother, keep_ap=True)
    fast = base_params(smoother=smoother, keep_ap=True, fast_ap=True)
    fast["eigensolver"].update({"type": "anasazi", "tolerance": 1e-3})
    h_plain = M.Hierarchy(ctx, evaluator, prob, plain)
    h_fast = M.Hierarchy(ctx, evaluator, prob, fast)
    assert "Setup: fast_ap" in h_fast.timer_report() and "Setup: fast_ap" not in h_plain.timer_report()
    R = h_plain.restrictor().to_scipy()
    assert abs(h_fast.restrictor().to_scipy() - R).max() == 0.0
    nf, nc = h_plain.level_size(0), h_plain.level_size(1)
    # A R^T column by column (the matrices of the reference's test, read through the operators)
    A = O.assemble_csr(mesh, coef)
    ref = (A @ R.T).toarray() if not matrix_free else np.stack([mf.vmult(R.T[:, [j]].toarray().ravel()) for j in range(nc)], axis=1)
    cols_plain, cols_fast = np.empty((nf, nc)), np.empty((nf, nc))
    out = torch.empty(nf, dtype=torch.float64, device="cuda")
    for j in range(nc):
        e = np.zeros(nc)
        e[j] = 1.0
        h_plain.ap_apply(1, dev(e), out)
        cols_plain[:, j] = out.cpu().numpy()
        h_fast.ap_apply(1, dev(e), out)
        cols_fast[:, j] = out.cpu().numpy()
    scale = np.abs(ref).max()
    assert np.abs(cols_fast - cols_plain).max() <= 1e-12 * scale      # (reference: 1e-9 relative per entry)
    assert np.abs(cols_fast - ref).max() <= 1e-12 * scale
    # the coarse operators and the cycles of the two hierarchies
    Ac_plain, Ac_fast = h_plain.coarse_operator().to_scipy(), h_fast.coarse_operator().to_scipy()
    assert abs(Ac_fast - Ac_plain).max() <= 1e-12 * abs(Ac_plain).max()
    x0 = np.where(con, 0.0, np.random.default_rng(4).random(mesh.n_dofs))
    b = np.zeros(mesh.n_dofs)
    Ad = M.SparseMatrixDevice(ctx, A)
    res_p, x_p = gpu_history(ctx, h_plain, lambda y, x: Ad.vmult(y, x), b, x0, n_cycles=8)
    res_f, x_f = gpu_history(ctx, h_fast, lambda y, x: Ad.vmult(y, x), b, x0, n_cycles=8)
    np.testing.assert_allclose(res_f, res_p, rtol=1e-12, atol=HIST_ATOL)
    assert np.abs(x_f - x_p).max() <= 1e-12
    # a hierarchy without keep_ap holds no A R^T
    h = M.Hierarchy(ctx, evaluator, prob, base_params(smoother=smoother, fast_ap=True))
    with pytest.raises(L.MfmgInvalidArgument, match="keep_ap"):
        h.ap_apply(1, dev(np.zeros(nc)), out)


def test_global_problem_of_config3_on_one_gpu(ctx):
    """BASELINE.json configs[3]'s GLOBAL problem (512^3 cells = 513^3 DoFs) as one V-cycle hierarchy on ONE GPU -- the
    strong-scaling anchor of the 8-GPU run (`bench.py: vcycle_513cubed_1gpu`).  The oracle does not reach this size in
    a test's time (8 x the 257^3 case of test_full_size_vcycle_history_against_the_native_oracle), so: the
    size-independent properties of the operators on the path and the contraction of the cycle, and the table-driven
    layouts chosen (the first coarse operator, 33.5 M rows, must not fall back to stored values)."""
    n = (512, 512, 512)
    prob = M.LaplaceProblem(n, "constant", device="cuda")
    params = base_params(smoother={"type": "Chebyshev", "degree": 3, "smoothing_range": 20.0}, solver={"type": "amg"})
    h = M.Hierarchy(ctx, "HipMatrixFreeMeshEvaluator", prob, params)
    nf, nc = h.level_size(0), h.level_size(1)
    assert nf == 513 ** 3 and nc == 2 * 256 ** 3
    g = torch.Generator(device="cuda").manual_seed(11)
    free = (prob.constrained == 0).to(torch.float64)
    rnd = lambda m: torch.rand(m, dtype=torch.float64, device="cuda", generator=g)
    x, y = rnd(nf) * free, rnd(nf) * free
    u, v = rnd(nc), rnd(nc)
    ax, ay = torch.empty_like(x), torch.empty_like(x)
    h.operator_apply(0, x, ax)
    h.operator_apply(0, y, ay)
    assert abs(ctx.dot(ax, y) - ctx.dot(x, ay)) < 1e-11 * abs(ctx.dot(ax, y))           # A symmetric
    au, av = torch.empty_like(u), torch.empty_like(u)
    h.operator_apply(1, u, au)
    h.operator_apply(1, v, av)
    assert abs(ctx.dot(au, v) - ctx.dot(u, av)) < 1e-11 * abs(ctx.dot(au, v))           # A_c symmetric
    t1, t2, t3 = torch.empty_like(x), torch.empty_like(x), torch.empty_like(u)
    h.restrictor_apply(1, u, t1, L.TRANS)
    h.operator_apply(0, t1, t2)
    h.restrictor_apply(1, t2, t3)
    assert abs(ctx.dot(t3, v) - ctx.dot(au, v)) < 1e-9 * abs(ctx.dot(au, v))             # <R A R^T u, v> = <A_c u, v>
    del ax, ay, t1, t2, t3, au, av
    assert h.coarse_operator().get_kernel()[1] == 3 and h.coarse_operator().regular_rows()
    xx = x
    b = torch.zeros_like(xx)
    r = torch.empty_like(xx)
    norms = []
    for _ in range(4):
        h.operator_apply(0, xx, r)
        norms.append(ctx.l2_norm(r))
        h.apply(b, xx)
    assert all(norms[i + 1] < 0.4 * norms[i] for i in range(3)), norms


@pytest.mark.parametrize("n,material,degree", [((8, 8), "constant", 1), ((16, 16), "linear", 3), ((32, 32), "constant", 2)])
def test_matrix_free_hierarchy_in_two_dimensions(ctx, n, material, degree):
    """The matrix-free evaluator on a 2-D mesh (the reference: LaplaceMatrixFree<2> behind DealIIMatrixFreeMeshEvaluator<2>,
    tests/test_hierarchy.cc:276-330 with hierarchy_input.info's laplace.n_refinements; BASELINE.json configs[0] is the 2-D
    unit square): own restrictor against the oracle's, R A R^T against the column-by-column product, and the 20-cycle
    residual history to 1e-10."""
    mesh = O.StructuredMesh(n)
    coef = O.coefficient_table(mesh, material)
    con = mesh.constrained_mask()
    mf = O.MatrixFreeLaplace(mesh, coef)
    prob = M.LaplaceProblem(n, material, device="cuda")
    params = base_params(smoother={"type": "Chebyshev", "degree": degree, "smoothing_range": 20.0})
    h = M.Hierarchy(ctx, "HipMatrixFreeMeshEvaluator", prob, params)
    deg, lmin, lmax = h.smoother_info()
    R = h.restrictor().to_scipy()
    Ro = O.build_restrictor(mesh, coef, mf.diagonal(), agg=(2, 2), n_eig=2, variant="mf", eig_mode="krylov").csr
    assert abs(R - Ro).max() < 1e-11
    Ac = O.galerkin_coarse_matrix(mf.vmult, R)
    assert abs(h.coarse_operator().to_scipy() - Ac).max() < 1e-11 * abs(Ac).max()
    p = O.ChebyshevParams(degree=degree, lambda_max=lmax, lambda_min=lmin)
    dinv = mf.diagonal_inverse()
    smoother = lambda b, x: O.chebyshev_smoother_apply(mf.vmult, dinv, p, b, x)
    ho = O.TwoLevelHierarchy(mf.vmult, smoother, R, O.direct_coarse_solver(Ac), 1, False)
    x0 = O.random_initial_guess(mesh.n_dofs, con)
    b = np.zeros(mesh.n_dofs)
    res_o, rate_o, _ = O.vcycle_history(ho, mf.vmult, b, x0)
    op = M.MatrixFreeLaplace(ctx, prob)
    res_g, _ = gpu_history(ctx, h, lambda y, x: op.vmult(y, x), b, x0)
    np.testing.assert_allclose(res_g, res_o, rtol=HIST_TOL, atol=HIST_ATOL)
    assert rate_o < 0.6


@pytest.mark.parametrize("n,material,solver", [((32, 32, 32), "linear", {"type": "lu_dense"}),
                                               ((64, 64, 64), "linear", {"type": "amg"}),
                                               ((64, 64, 64), "discontinuous", {"type": "amg"})])
def test_setup_value_precision_float(ctx, n, material, solver):
    """"setup value precision" float: the matrices the setup forms (R, R A R^T, the aggregation hierarchy) are rounded to
    float-representable values, and the stored blocks of a variable-coefficient problem are then KEPT in float (half the
    bytes per application) -- arithmetic stays FP64.  The hierarchy is the exact cycle of the rounded matrices: the oracle
    run on the product's own (downloaded) R, A_c and aggregation levels reproduces the residual history to 1e-10; the
    rounded R is within float rounding of the FP64 one; the cycle contracts like the FP64 hierarchy."""
    mesh = O.StructuredMesh(n)
    coef = O.coefficient_table(mesh, material)
    con = mesh.constrained_mask()
    mf = O.MatrixFreeLaplace(mesh, coef)
    prob = M.LaplaceProblem(n, material, device="cuda")
    smoother = {"type": "Chebyshev", "degree": 3, "smoothing_range": 20.0}
    p64 = base_params(smoother=smoother, solver=solver)
    p32 = base_params(smoother=smoother, solver=solver, **{"setup value precision": "float"})
    h64 = M.Hierarchy(ctx, "HipMatrixFreeMeshEvaluator", prob, p64)
    h32 = M.Hierarchy(ctx, "HipMatrixFreeMeshEvaluator", prob, p32)
    R64, R32 = h64.restrictor().to_scipy(), h32.restrictor().to_scipy()
    assert np.array_equal(R32.data, R32.data.astype(np.float32).astype(np.float64))       # float-representable
    assert abs(R32 - R64).max() <= 2.0 ** -23 * abs(R64).max()                             # ... and R rounded, nothing else
    Ac32 = h32.coarse_operator().to_scipy()
    assert np.array_equal(Ac32.data, Ac32.data.astype(np.float32).astype(np.float64))
    assert abs(Ac32 - Ac32.T).max() == 0.0                                                 # rounded symmetrically
    # (R vanishes on the constrained DoFs, so the assembled matrix gives the same product as the matrix-free operator)
    A = O.assemble_csr(mesh, coef)
    G = (R32 @ A @ R32.T).tocsr()
    assert abs(Ac32 - G).max() <= 2.0 ** -22 * abs(G).max()                                # = fl32(R A R^T) of the rounded R
    if n[0] >= 64:
        # the stored layouts hold floats where the FP64 hierarchy holds doubles
        assert h32.restrictor().float_storage() and not h64.restrictor().float_storage()
        assert h32.coarse_operator().float_storage() and not h64.coarse_operator().float_storage()
        assert h32.coarse_operator().get_kernel()[1] == 3                                  # symmetric half, as in FP64
    deg, lmin, lmax = h32.smoother_info()
    p = O.ChebyshevParams(deg, lmax, lmin)
    dinv = mf.diagonal_inverse()
    sm = lambda b, x: O.chebyshev_smoother_apply(mf.vmult, dinv, p, b, x)
    coarse = O.amg_coarse_solver(h32.coarse_amg_levels(), 1) if solver["type"] == "amg" else O.direct_coarse_solver(Ac32)
    ho = O.TwoLevelHierarchy(mf.vmult, sm, R32, coarse, 1, False)
    x0 = O.random_initial_guess(mesh.n_dofs, con)
    b = np.zeros(mesh.n_dofs)
    n_hist = 8
    res_o, _, x_o = O.vcycle_history(ho, mf.vmult, b, x0, n_cycles=n_hist)
    op = M.MatrixFreeLaplace(ctx, prob)
    res_32, x_32 = gpu_history(ctx, h32, lambda y, x: op.vmult(y, x), b, x0, n_cycles=n_hist)
    np.testing.assert_allclose(res_32, res_o, rtol=HIST_TOL, atol=HIST_ATOL)
    res_64, _ = gpu_history(ctx, h64, lambda y, x: op.vmult(y, x), b, x0, n_cycles=n_hist)
    np.testing.assert_allclose(res_32, res_64, rtol=1e-4)        # the same preconditioner up to the rounding of its matrices
    if solver["type"] == "lu_dense":
        # the precision belongs to the hierarchy, not to the context: the FP64 hierarchy, its restrictor replaced AFTER the float
        # one was built on the same context, forms its Galerkin operator unrounded
        h64.set_restrictor(R64)
        Ac64 = h64.coarse_operator().to_scipy()
        assert not np.array_equal(Ac64.data, Ac64.data.astype(np.float32).astype(np.float64))


@pytest.mark.parametrize("material", ["constant", "linear"])
def test_release_setup_matrices(ctx, material):
    """"release setup matrices" true: the table-driven operators (A_c, the operators of the aggregation levels) free their CSR
    arrays once the hierarchy stands -- 12 B per entry that only the setup algebra and the exports read; the rows their tables
    do not cover stay as a compact CSR.  The cycle is the same cycle bit for bit, the library holds less memory, and the exports
    of the released matrices fail loudly instead of handing out nothing."""
    n = (64, 64, 64)
    prob = M.LaplaceProblem(n, material, device="cuda")
    params = base_params(smoother={"type": "Chebyshev", "degree": 3, "smoothing_range": 20.0},
                         solver={"type": "amg", "amg": {"smoother_degree": 1, "smoothing_range": 4.0, "pre_smoothing_levels": 0}})

    def library_bytes():
        return float(M.memory_inventory().strip().splitlines()[-1].split()[0])
    x0 = np.random.default_rng(8).random(prob.n_dofs) * (prob.constrained.cpu().numpy() != 1)
    out, held = [], []
    for release in (False, True):
        before = library_bytes()
        p = dict(params)
        p["release setup matrices"] = release
        h = M.Hierarchy(ctx, "HipMatrixFreeMeshEvaluator", prob, p)
        held.append(library_bytes() - before)
        x, b = dev(x0), dev(np.zeros(prob.n_dofs))
        for _ in range(4):
            h.apply(b, x)
        ctx.synchronize()
        out.append(x.cpu().numpy())
        if release:
            if material == "constant":                                 # A_c is table-driven here: its CSR arrays are gone
                assert h.coarse_operator().regular_rows()
                with pytest.raises(L.MfmgError, match="released"):
                    h.coarse_operator().to_scipy()
            else:                                                      # stored blocks are the layout: nothing to release
                h.coarse_operator().to_scipy()
            h.restrictor().to_scipy()                                  # (the restrictor keeps its host copy)
        del h
    assert np.array_equal(out[0], out[1])
    if material == "constant":
        assert held[1] < 0.8 * held[0]


@pytest.mark.parametrize("n,material,numbering", [((8, 8), "linear", "lexicographic"), ((6, 5, 4), "discontinuous", "random"),
                                                   ((16, 12, 10), "linear", "lexicographic"), ((3, 3, 3), "constant", "random")])
def test_fine_operator_assembled_on_the_device(ctx, n, material, numbering):
    """`HipMeshEvaluator::evaluate_global` (the reference: the user's assembled system matrix, tests/laplace.hpp:154-204,
    Dirichlet rows and columns eliminated by AffineConstraints::distribute_local_to_global) forms the matrix in a kernel:
    the same rows as the host assembly bit for bit, and the oracle's matrix to rounding."""
    mesh = O.StructuredMesh(n)
    coef = O.coefficient_table(mesh, material)
    rng = np.random.default_rng(3)
    perm = rng.permutation(mesh.n_dofs) if numbering == "random" else np.arange(mesh.n_dofs)
    prob = M.LaplaceProblem(n, material, device="cuda", dof_numbering=torch.from_numpy(perm))
    h = M.Hierarchy(ctx, "HipMeshEvaluator", prob, base_params(smoother={"type": "Jacobi"}, **{"max levels": 1}))
    A_dev = h.fine_operator().to_scipy()
    A_host = M.host_assemble_matrix(M.LaplaceProblem(n, material, device="cpu", dof_numbering=torch.from_numpy(perm)))
    assert np.array_equal(A_dev.indptr, A_host.indptr) and np.array_equal(A_dev.indices, A_host.indices)
    assert np.array_equal(A_dev.data, A_host.data)                                       # bit for bit
    A_o = O.assemble_csr(mesh, coef).tocsr()[:, :]
    # the oracle numbers nodes lexicographically: value of node i lives at DoF perm[i]
    P = sp.csr_matrix((np.ones(mesh.n_dofs), (perm, np.arange(mesh.n_dofs))), shape=(mesh.n_dofs,) * 2)
    assert abs(A_dev - P @ A_o @ P.T).max() <= 1e-14 * abs(A_o).max()


@pytest.mark.parametrize("dim,matrix_free", [(2, 0), (2, 1), (3, 0), (3, 1)])
@pytest.mark.parametrize("preconditioner", [False, True])
def test_hierarchy_driver_example(ctx, tmp_path, dim, matrix_free, preconditioner):
    """examples/hierarchy_driver.py: the reference's driver workflow (tests/hierarchy_driver.cc; its CTest entries run it for
    -m 0 / 1 and only ask that it completes, tests/CMakeLists.txt:18-42) on the reference's own input file -- 20 cycles with
    a contracting residual, or CG to the tolerance."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("hierarchy_driver", os.path.join(root, "examples", "hierarchy_driver.py"))
    drv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(drv)
    text = open(os.path.join(root, "tests", "golden", "reference_hierarchy_input.info")).read()
    p = M.info_to_params(text)
    p["is preconditioner"] = preconditioner
    p["laplace"]["n_refinements"] = 4            # (5 in the file: 33^3 DoFs; 4 keeps the eight runs short)
    f = tmp_path / "input.info"
    f.write_text(M.params_to_info(p))
    out = drv.main(["-f", str(f), "-d", str(dim), "-m", str(matrix_free), "-t", "1e-8"])
    if preconditioner:
        assert 1 <= out <= 40                    # CG iterations
    else:
        assert 0.0 < out < 0.7                   # res[20] / res[19]


@pytest.mark.parametrize("n,material", [((140, 40, 30), "constant"), ((70, 30, 20), "linear"), ((130, 70, 12), "constant")])   # (the last: tail columns)
def test_interior_and_shell_launches_change_no_bit(ctx, monkeypatch, n, material):
    """A rank of a box decomposition applies its fine operator as two launches -- the tiles that read no ghost plane, and the
    shell around them (one launch over a compact tile list, on the exchange stream beside the interior tiles).  The same
    launches can be produced on ONE rank (Context.set_mf_emulate_split: the corner rank of a 2 x 2 x 2 grid, no exchange; the
    measurement of DESIGN.md section 7 uses it): every variant -- concurrent, one after the other, the slab-by-slab launches
    of the first box version -- must reproduce the single launch bit for bit, cycle after cycle."""
    prob = M.LaplaceProblem(n, material, device="cuda")
    # (one smoother term per launch and chunk records with one halo lane: the layout and launches of a distributed rank)
    params = base_params(smoother={"type": "Chebyshev", "degree": 3, "smoothing_range": 20.0, "fused_terms": 1},
                         solver={"type": "pcg", "n_iterations": 4})
    ctx.set_mf_fused_terms(1)
    try:
        h = M.Hierarchy(ctx, "HipMatrixFreeMeshEvaluator", prob, params)
    finally:
        ctx.set_mf_fused_terms(3)
    h.set_operator_tile(2, 2, 4)          # several column, y- and z-tiles on these small meshes
    rng = np.random.default_rng(12)
    x0 = dev(rng.random(prob.n_dofs) * (prob.constrained.cpu().numpy() != 1))
    b = dev(rng.random(prob.n_dofs) * (prob.constrained.cpu().numpy() != 1))

    def run():
        x = x0.clone()
        for _ in range(3):
            h.apply(b, x)
        r = torch.empty_like(x)
        h.operator_apply(0, x, r)
        ctx.synchronize()
        return x.clone(), r.clone()

    ref = run()
    try:
        ctx.set_mf_emulate_split("xyz")
        for variant in ("beside", "after", "slabs"):
            ctx.set_mf_shell(variant)
            out = run()
            assert torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1]), variant
    finally:
        ctx.set_mf_emulate_split(None)
        ctx.set_mf_shell("beside")


@pytest.mark.parametrize("n,degree,steps", [((40, 36, 30), 3, 1), ((70, 20, 24), 2, 1), ((33, 32, 31), 5, 1), ((36, 36, 36), 3, 2), ((24, 24, 24), 4, 3)])
def test_cycle_with_the_multi_term_sweep_is_the_term_by_term_cycle(ctx, n, degree, steps):
    """The fine-level Chebyshev smoother as ONE sweep over the mesh (the hierarchy alternates between x and a workspace vector,
    degree <= 3; beyond that three terms per sweep and a launch per further term) against `smoother.fused_terms 1`, a launch per
    term on chunk records with one halo lane: the same V-cycle bit for bit -- iterates and residual history, any number of
    smoothing steps (an odd number of out-of-place applications ends in the workspace vector and is copied back)."""
    prob = M.LaplaceProblem(n, "constant", device="cuda")
    sm = {"type": "Chebyshev", "degree": degree, "smoothing_range": 20.0, "n_smoothing_steps": steps, "lambda_max": 1.9, "lambda_min": 0.095,
          "sweep_arithmetic": "reference"}
    amg = {"coarsest_size": 300, "pre_smoothing_levels": 0}
    h_sweep = M.Hierarchy(ctx, "HipMatrixFreeMeshEvaluator", prob, base_params(smoother=dict(sm), solver={"type": "amg", "amg": dict(amg)}))
    assert h_sweep.smoother_sweep_terms() == (min(degree - 1, 3) if degree >= 3 else 0, min(degree, 3))
    ctx.set_mf_fused_terms(1)
    try:
        h_terms = M.Hierarchy(ctx, "HipMatrixFreeMeshEvaluator", prob,
                              base_params(smoother=dict(sm, fused_terms=1), solver={"type": "amg", "amg": dict(amg)}))
    finally:
        ctx.set_mf_fused_terms(3)
    assert h_terms.smoother_sweep_terms() == (0, 0)
    rng = np.random.default_rng(8)
    free = prob.constrained.cpu().numpy() != 1
    x0 = rng.random(prob.n_dofs) * free
    b = rng.random(prob.n_dofs) * free
    runs = []
    for h in (h_sweep, h_terms):
        x, bb = dev(x0), dev(b)
        for _ in range(4):
            h.apply(bb, x)
        # the smoother on its own, in place (two terms per sweep + a launch) and through the C ABI's smoother_apply
        y = dev(x0)
        h.smoother_apply(0, bb, y)
        ctx.synchronize()
        runs.append((x.clone(), y.clone()))
    assert torch.equal(runs[0][0], runs[1][0])
    assert torch.equal(runs[0][1], runs[1][1])


@pytest.mark.parametrize("cells", [32, 64])
def test_smoothed_prolongation_equals_two_steps(ctx, cells):
    """V(0,1) levels of the aggregation hierarchy with a damped-Jacobi post-smoother: prolongation and post-smoothing as ONE
    operator, x' = P~ x_c + beta D^-1 b with P~ = (I - beta D^-1 A) P formed at setup by probing (a distributed run saves a
    blocking exchange per level with it; one rank takes it on request).  The same cycle as the two steps, to rounding."""
    prob = M.LaplaceProblem((cells,) * 3, "constant", device="cuda")
    amg = {"smoother_degree": 1, "smoothing_range": 4.0, "n_cycles": 1, "pre_smoothing_levels": 0, "coarsest_size": 300}
    params = base_params(smoother={"type": "Chebyshev", "degree": 3, "smoothing_range": 20.0}, solver={"type": "amg", "amg": dict(amg)})
    h2 = M.Hierarchy(ctx, "HipMatrixFreeMeshEvaluator", prob, params)
    params["solver"]["amg"]["smoothed_prolongation"] = True
    h1 = M.Hierarchy(ctx, "HipMatrixFreeMeshEvaluator", prob, params)
    rng = np.random.default_rng(5)
    free = prob.constrained.cpu().numpy() != 1
    x0, b = rng.random(prob.n_dofs) * free, rng.random(prob.n_dofs) * free
    hist = []
    for h in (h2, h1):
        res, x = gpu_history(ctx, h, lambda y, xx: h.operator_apply(0, xx, y), b, x0, n_cycles=8)
        hist.append((res, x))
    np.testing.assert_allclose(hist[1][0], hist[0][0], rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(hist[1][1], hist[0][1], rtol=0, atol=1e-11 * np.abs(hist[0][1]).max())
    # ... and it is a different sequence of launches: the coarse solve of the one-step hierarchy applies no level operator
    xc = torch.rand(h1.level_size(1), dtype=torch.float64, device="cuda")
    y1, y2 = torch.empty_like(xc), torch.empty_like(xc)
    h1.coarse_apply(xc, y1)
    h2.coarse_apply(xc, y2)
    ctx.synchronize()
    assert relerr(y1.cpu().numpy(), y2.cpu().numpy()) < 1e-12


@pytest.mark.parametrize("n", [(16, 12, 10), (70, 8, 6)])
def test_preconditioner_application_starts_from_zero_without_zeroing(ctx, n):
    """'is preconditioner' true (include/mfmg/common/hierarchy.hpp:253-259): x is zeroed before the cycle.  With the one-sweep
    smoother the first pre-smoothing step neither zeroes nor reads x (Smoother::apply_from_zero: x_1 = beta_1 D^-1 b, two operator
    applications instead of three) -- garbage in x must not matter, the cycle equals the oracle's from x = 0 and the cycle of a
    hierarchy that takes the generic path (reference arithmetic of the sweep: x zeroed, three applications)."""
    mesh = O.StructuredMesh(n)
    coef = O.coefficient_table(mesh, "constant")
    mf = O.MatrixFreeLaplace(mesh, coef)
    rng = np.random.default_rng(11)
    bf = rng.random(mesh.n_dofs) * (~mesh.constrained_mask())
    outs = {}
    for arithmetic in ("modes", "reference"):
        prob = M.LaplaceProblem(n, "constant", device="cuda")
        params = base_params(smoother={"type": "Chebyshev", "degree": 3, "smoothing_range": 20.0, "sweep_arithmetic": arithmetic})
        params["is preconditioner"] = True
        h = M.Hierarchy(ctx, "HipMatrixFreeMeshEvaluator", prob, params)
        assert h.smoother_sweep_terms() == (2, 3)
        xg = dev(1e6 * rng.random(mesh.n_dofs))
        h.vmult(xg, dev(bf))
        ctx.synchronize()
        outs[arithmetic] = xg.cpu().numpy().copy()
        if arithmetic == "modes":
            deg, lmin, lmax = h.smoother_info()
            R = h.restrictor().to_scipy()
            Ac = h.coarse_operator().to_scipy()
    assert np.isfinite(outs["modes"]).all()
    assert relerr(outs["modes"], outs["reference"]) < 1e-12
    p = O.ChebyshevParams(deg, lmax, lmin)
    dinv = mf.diagonal_inverse()
    smoother = lambda b, x: O.chebyshev_smoother_apply(mf.vmult, dinv, p, b, x)
    ho = O.TwoLevelHierarchy(mf.vmult, smoother, R, O.direct_coarse_solver(Ac), 1, True)
    np.testing.assert_allclose(outs["modes"], ho.apply(bf, np.zeros(mesh.n_dofs)), rtol=1e-10, atol=1e-12 * np.abs(outs["modes"]).max())
