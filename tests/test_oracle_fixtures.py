"""Pins the CPU oracle (oracle/mfmg_oracle.py) against the reference's own fixtures:
known-answer tests and gold numbers held by /root/reference/tests (cited per test).
No GPU, no product code."""
import os
import sys

import numpy as np
import pytest
import scipy.sparse as sp

import mfmg_oracle as O


# ---- tests/test_smoother_device.cu:28-119 : one Jacobi step on tridiag(-1,4,-1), n = 30 ----
def test_jacobi_step_tridiagonal_known_answer():
    n = 30
    A = sp.diags([-np.ones(n - 1), 4 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1]).tocsr()
    b = np.ones(n)          # domain vector = 1
    x = np.zeros(n)         # range vector = 0
    dinv = O.jacobi_inverse_diagonal_csr(A)
    x_new = O.smoother_wrapper(lambda v: A @ v, lambda r: dinv * r, b, x)
    # reference: res = A*0 - b; tmp = D^-1 res; x = 0 - tmp  -> 1/4 everywhere
    np.testing.assert_allclose(x_new, 0.25 * np.ones(n), rtol=1e-14)


# ---- tests/test_sparse_matrix_device_operator.cu:23-137 : banded 30x39, a_ij = i + j ----
def _banded():
    n_rows, nnz_per_row = 30, 10
    n_cols = n_rows + nnz_per_row - 1
    rows = np.repeat(np.arange(n_rows), nnz_per_row)
    cols = rows + np.tile(np.arange(nnz_per_row), n_rows)
    vals = (rows + cols).astype(float)
    return sp.csr_matrix((vals, (rows, cols)), shape=(n_rows, n_cols))


def test_banded_operator_apply_transpose_multiply_exact():
    A = _banded()
    dense = A.toarray()
    ones_c = np.ones(A.shape[1])
    y = O.csr_spmv(A.indptr, A.indices, A.data, ones_c)
    assert np.array_equal(y, dense @ ones_c)            # BOOST_CHECK_EQUAL (exact, integers)
    At = A.T.tocsr()
    At.sort_indices()
    ones_r = np.ones(A.shape[0])
    yt = O.csr_spmv(At.indptr, At.indices, At.data, ones_r)
    assert np.array_equal(yt, dense.T @ ones_r)
    C = (A @ At).tocsr()
    C.sort_indices()
    yc = O.csr_spmv(C.indptr, C.indices, C.data, ones_r)
    assert np.array_equal(yc, dense @ (dense.T @ ones_r))


# ---- tests/test_sparse_matrix_device.cu:24-114 : random-pattern SpMV, std::default_random_engine(i) ----
def random_pattern_matrix(size=10):
    rows, cols, vals = [], [], []
    for i in range(size):
        gen = O.MinstdRand0(i)
        entries = {}
        for j in range(5):
            c = gen.uniform_int(0, size - 1)
            entries[c] = float(i + j)          # sparse_matrix.set overwrites
        for c, v in entries.items():
            rows.append(i)
            cols.append(c)
            vals.append(v)
    return sp.csr_matrix((vals, (rows, cols)), shape=(size, size))


def test_random_pattern_spmv():
    A = random_pattern_matrix()
    A.sort_indices()
    x = np.arange(A.shape[0], dtype=float)
    y = O.csr_spmv(A.indptr, A.indices, A.data, x)
    np.testing.assert_allclose(y, A.toarray() @ x, rtol=1e-16, atol=0)   # BOOST_CHECK_CLOSE 1e-14 %


# ---- tests/test_restriction_matrix.cc:62-168 : R(i, dof) = diag_elem * eigvec ----
def test_restriction_matrix_entries():
    n_dofs = 25                 # 2-D, 2 refinements, Q1
    n_rows = n_dofs
    esize = 3
    eigenvectors = [np.array([n_rows * esize + i * esize + j for j in range(esize)], dtype=float)
                    for i in range(n_rows)]
    gen = O.MinstdRand0()
    dof_maps = []
    for i in range(n_rows):
        seen = []
        while len(seen) < esize:
            d = gen.uniform_int(0, n_dofs - 1)
            if d not in seen:
                seen.append(d)
        dof_maps.append(seen)
    count = {}
    for i in range(n_rows):
        for j in range(esize):
            count[dof_maps[i][j]] = count.get(dof_maps[i][j], 0.0) + 1.0
    diag_elements = [[1.0 / count[dof_maps[i][j]] for j in range(esize)] for i in range(n_rows)]
    R = O.restriction_from_eigenvectors(eigenvectors, diag_elements, dof_maps, [1] * n_rows,
                                        np.ones(n_dofs), n_dofs)
    for i in range(n_rows):
        for j in range(esize):
            assert R[i, dof_maps[i][j]] == pytest.approx(diag_elements[i][j] * eigenvectors[i][j], rel=1e-14)


# ---- tests/test_restriction_matrix.cc:293-354 : partition of unity of the weights ----
def test_weight_sum_partition_of_unity():
    mesh = O.StructuredMesh((16, 16))
    coef = O.coefficient_table(mesh, "constant")
    A = O.assemble_csr(mesh, coef)
    no_bc = np.zeros(mesh.n_dofs, dtype=bool)     # "The test requires us not to put boundary conditions"
    R = O.build_restrictor(mesh, coef, A.diagonal(), agg=(2, 2), n_eig=1, variant="host",
                           eig_mode="lapack", constrained=no_bc).csr
    col_l1 = np.asarray(abs(3.0 * R).sum(axis=0)).ravel()
    np.testing.assert_allclose(col_l1, 1.0, rtol=2e-4)


# ---- tests/test_hierarchy.cc:644-695 : matrix-free vmult == assembled vmult, 4 coefficients ----
@pytest.mark.parametrize("material", ["constant", "linear", "linear_x", "discontinuous"])
@pytest.mark.parametrize("dim", [2, 3])
def test_matrix_free_equals_assembled(material, dim):
    mesh = O.StructuredMesh((32, 32) if dim == 2 else (8, 8, 8))
    coef = O.coefficient_table(mesh, material)
    con = mesh.constrained_mask()
    x = O.random_initial_guess(mesh.n_dofs, con)
    mf = O.MatrixFreeLaplace(mesh, coef)
    A = O.assemble_csr(mesh, coef)
    diff = mf.vmult(x) - A @ x
    assert np.linalg.norm(diff) < 1e-9          # BOOST_TEST(rhs_ref.l2_norm() < 1.e-9)
    # diagonal: constrained entries one (tests/laplace_matrix_free.hpp:88), free entries = matrix diagonal
    d = mf.diagonal()
    np.testing.assert_allclose(d[~con], A.diagonal()[~con], rtol=1e-13)
    assert np.all(d[con] == 1.0)


def test_chebyshev_fused_form_equals_reference_order():
    mesh = O.StructuredMesh((6, 6, 6))
    coef = O.coefficient_table(mesh, "linear")
    mf = O.MatrixFreeLaplace(mesh, coef)
    dinv = mf.diagonal_inverse()
    p = O.ChebyshevParams(degree=3, lambda_max=1.9, lambda_min=0.2)
    rng = np.random.default_rng(0)
    x = rng.random(mesh.n_dofs)
    b = rng.random(mesh.n_dofs)
    ref = O.chebyshev_smoother_apply(mf.vmult, dinv, p, b, x)
    fused = O.chebyshev_smoother_apply_fused(mf.vmult, dinv, p, b, x)
    np.testing.assert_allclose(fused, ref, rtol=1e-12, atol=1e-13)


# ---- convergence-rate golds -------------------------------------------------------------
def _gold_setup():
    mesh = O.StructuredMesh((4, 4, 4))        # hyper_cube, laplace.n_refinements = 2
    coef = O.coefficient_table(mesh, "constant")
    con = mesh.constrained_mask()
    A = O.assemble_csr(mesh, coef)
    dn = O.dealii_global_numbering(mesh)
    return mesh, coef, con, A, dn


def test_gold_cuda_jacobi_hyper_cube():
    """tests/test_hierarchy_device.cu:359-420: 0.14933479171507894, tolerance 1e-6 %."""
    mesh, coef, con, A, dn = _gold_setup()
    R = O.build_restrictor(mesh, np.ones_like(coef), A.diagonal(), n_eig=2, variant="device",
                           eig_mode="lapack").csr
    Ac = (R @ A @ R.T).tocsr()
    dinv = O.jacobi_inverse_diagonal_csr(A)
    smoother = lambda b, x: O.smoother_wrapper(lambda v: A @ v, lambda r: dinv * r, b, x)
    h = O.TwoLevelHierarchy(lambda v: A @ v, smoother, R, O.direct_coarse_solver(Ac), 1, False)
    x0 = O.random_initial_guess(mesh.n_dofs, con, order=np.argsort(dn), zero_constrained=False)
    _, rate, _ = O.vcycle_history(h, lambda v: A @ v, np.zeros(mesh.n_dofs), x0)
    assert rate == pytest.approx(0.14933479171507894, rel=1e-8)


def test_gold_cpu_gauss_seidel_hyper_cube():
    """tests/test_hierarchy.cc:343,352: 0.0235237332 (arpack and lanczos), tolerance 1e-2."""
    mesh, coef, con, A, dn = _gold_setup()
    R = O.build_restrictor(mesh, coef, A.diagonal(), n_eig=2, variant="host", eig_mode="krylov").csr
    Ac = (R @ A @ R.T).tocsr()
    n = mesh.n_dofs
    P = sp.csr_matrix((np.ones(n), (dn, np.arange(n))), shape=(n, n))   # lexicographic -> deal.II ids
    Ad = (P @ A @ P.T).tocsr()
    gs = lambda r: P.T @ O.gauss_seidel_from_zero(Ad, P @ r)            # the sweep runs in DoF-id order
    smoother = lambda b, x: O.smoother_wrapper(lambda v: A @ v, gs, b, x)
    h = O.TwoLevelHierarchy(lambda v: A @ v, smoother, R, O.direct_coarse_solver(Ac), 1, False)
    x0 = O.random_initial_guess(n, con, order=np.argsort(dn))
    _, rate, _ = O.vcycle_history(h, lambda v: A @ v, np.zeros(n), x0)
    assert rate == pytest.approx(0.0235237332, rel=1e-8)


def test_gold_cpu_matrix_free_chebyshev_hyper_cube():
    """tests/test_hierarchy.cc:353: 0.0880045475 (lanczos), the reference's own tolerance 1e-2.
    deal.II's eigenvalue estimate is third-party arithmetic restated from memory of 9.1."""
    mesh, coef, con, A, dn = _gold_setup()
    n = mesh.n_dofs
    mf = O.MatrixFreeLaplace(mesh, coef)
    dinv = mf.diagonal_inverse()
    R = O.build_restrictor(mesh, coef, mf.diagonal(), n_eig=2, variant="mf", eig_mode="krylov").csr
    Ac = O.galerkin_coarse_matrix(mf.vmult, R)
    P = sp.csr_matrix((np.ones(n), (dn, np.arange(n))), shape=(n, n))
    p = O.dealii_chebyshev_params(lambda v: P @ mf.vmult(P.T @ v), P @ dinv, n, degree=1)
    smoother = lambda b, x: O.chebyshev_smoother_apply(mf.vmult, dinv, p, b, x)
    h = O.TwoLevelHierarchy(mf.vmult, smoother, R, O.direct_coarse_solver(Ac), 1, False)
    x0 = O.random_initial_guess(n, con, order=np.argsort(dn))
    _, rate, _ = O.vcycle_history(h, mf.vmult, np.zeros(n), x0)
    assert rate == pytest.approx(0.0880045475, rel=1e-2)


# ---- BASELINE.json configs[0]: 2-D plumbing case (tests/test_hierarchy.cc:203-211) ----
@pytest.mark.parametrize("smoother_type", ["Gauss-Seidel", "Jacobi"])
def test_config1_2d_plumbing(smoother_type):
    mesh = O.StructuredMesh((8, 8))           # 3 uniform refinements, 81 DoFs
    coef = O.coefficient_table(mesh, "constant")
    con = mesh.constrained_mask()
    A = O.assemble_csr(mesh, coef)
    R = O.build_restrictor(mesh, coef, A.diagonal(), agg=(2, 2), n_eig=2, variant="host",
                           eig_mode="krylov").csr
    assert R.shape == (32, 81)
    Ac = (R @ A @ R.T).tocsr()
    if smoother_type == "Jacobi":
        dinv = 1.0 / A.diagonal()
        Binv = lambda r: dinv * r
    else:
        Binv = lambda r: O.gauss_seidel_from_zero(A, r)
    smoother = lambda b, x: O.smoother_wrapper(lambda v: A @ v, Binv, b, x)
    h = O.TwoLevelHierarchy(lambda v: A @ v, smoother, R, O.direct_coarse_solver(Ac), 1, False)
    x0 = O.random_initial_guess(mesh.n_dofs, con)
    res, rate, _ = O.vcycle_history(h, lambda v: A @ v, np.zeros(mesh.n_dofs), x0)
    assert rate < 1.0
    assert np.all(np.diff(res) < 0)           # monotone residual history


def test_minstd_rand0_known_answer():
    # the 10000th output of std::minstd_rand0 seeded with 1 is 1043618065 ([rand.predef])
    g = O.MinstdRand0()
    v = 0
    for _ in range(10000):
        v = g.next_u32()
    assert v == 1043618065


# ---- committed fixtures (tests/golden) ------------------------------------------------------
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_reference_gold_table_is_what_the_tests_above_check():
    """tests/golden/reference_golds.json lists the reference's known answers; the three gold tests above use
    exactly these numbers."""
    import json
    golds = {g["name"]: g for g in json.load(open(os.path.join(GOLDEN, "reference_golds.json")))["convergence_rates"]}
    assert golds["cuda_jacobi_hyper_cube"]["value"] == 0.14933479171507894
    assert golds["cpu_gauss_seidel_hyper_cube"]["value"] == 0.0235237332
    assert golds["cpu_matrix_free_chebyshev_hyper_cube"]["value"] == 0.0880045475


@pytest.mark.parametrize("name", ["mf_cheb3_8x8x8_linear", "mf_cheb3_12x6x4_constant"])
def test_oracle_reproduces_committed_vectors(name):
    """The oracle of the day against the vectors frozen by tests/golden/make_fixtures.py."""
    sys.path.insert(0, GOLDEN)
    import make_fixtures
    f = np.load(os.path.join(GOLDEN, f"oracle_vcycle_{name}.npz"))
    out = make_fixtures.build(*make_fixtures.CASES[name])
    np.testing.assert_array_equal(out["x0"], f["x0"])
    np.testing.assert_allclose(out["vmult_x0"], f["vmult_x0"], rtol=1e-13, atol=1e-14)
    np.testing.assert_allclose(out["R_data"], f["R_data"], rtol=0, atol=1e-11)
    np.testing.assert_array_equal(out["R_indices"], f["R_indices"])
    assert out["lambda_max"] == pytest.approx(float(f["lambda_max"]), rel=1e-12)
    np.testing.assert_allclose(out["history"], f["history"], rtol=1e-9, atol=1e-13)


# ---- tests/test_agglomerate.cc:69-230 : block agglomerates 2 x 3 (x 4) on 8 x 8 (x 8) cells, one rank ----
@pytest.mark.parametrize("dim", [2, 3])
def test_block_agglomerate_ids_literal_arrays(dim):
    import json
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_agglomerate_ids.json")))
    mesh = O.StructuredMesh((8,) * dim)
    ids = O.dealii_block_agglomerate_ids(mesh, (2, 3, 4)[:dim])
    assert ids.tolist() == gold["agglomerate_%dd" % dim]                 # BOOST_TEST(agglomerates == ref_agglomerates)
    # and they are the boxes the restrictor is built on: same cells per agglomerate as block_agglomerates
    aggs, counts = O.block_agglomerates(mesh, (2, 3, 4)[:dim])
    assert len(aggs) == ids.max() == int(np.prod(counts))
    sizes = sorted(int(np.prod([hi[d] - lo[d] for d in range(dim)])) for lo, hi in aggs)
    assert sizes == sorted(np.bincount(ids)[1:].tolist())


# ---- tests/test_eigenvectors.cc:74-130 : a diagonal agglomerate matrix diag(1 .. n), the first five eigenpairs ----
@pytest.mark.parametrize("mode", ["lapack", "krylov"])
def test_eigenvectors_of_a_diagonal_agglomerate_matrix(mode):
    import scipy.linalg as sla
    n, n_eig = 81, 5                                   # 8 x 8 cells of Q1: 81 DoFs, n_eigenvectors = 5
    w, V = sla.eigh(np.diag(np.arange(1.0, n + 1.0)))
    v0 = np.random.default_rng(0).random(n)            # (a Krylov start vector with a component on every eigenvector)
    vals, vecs = O._select_eigenvectors(w, V, n_eig, mode, v0)
    np.testing.assert_allclose(vals, np.arange(1.0, n_eig + 1.0), rtol=1e-12)      # eigenvalues i + 1
    ref = np.zeros((n, n_eig))
    ref[np.arange(n_eig), np.arange(n_eig)] = 1.0
    np.testing.assert_allclose(np.abs(vecs), ref, atol=1e-12)                      # |eigenvector_i| = e_i
