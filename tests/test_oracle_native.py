"""Pins oracle/oracle_kernels.cpp (the C++17/OpenMP restatement, bench.py's `cpu_baseline` "port" and the checker
of the full-size GPU parity test) against the numpy oracle, which itself is pinned by the reference's fixtures
(tests/test_oracle_fixtures.py).  Harness: /root/reference/tests/test_hierarchy.cc:76-123 (random initial guess with
the constrained entries zeroed, b = 0, residual history of n cycles).  No GPU, no product code."""
import numpy as np
import pytest
import scipy.sparse as sp

import mfmg_oracle as O
import oracle_native as ON

TOL = 1e-13


def _problem(n, material):
    mesh = O.StructuredMesh(n)
    coef = O.coefficient_table(mesh, material)
    mf = O.MatrixFreeLaplace(mesh, coef)
    return mesh, coef, mf


@pytest.mark.parametrize("n", [(4, 4, 4), (5, 3, 6), (12, 12, 12), (16, 16, 16)])
@pytest.mark.parametrize("material", ["constant", "linear", "discontinuous"])
def test_native_matrix_free_operator(n, material):
    """tests/laplace_matrix_free.hpp:121-156: y = A x, constrained rows y = x."""
    mesh, coef, mf = _problem(n, material)
    con = mesh.constrained_mask()
    rng = np.random.default_rng(7)
    for zero_constrained in (True, False):
        x = rng.standard_normal(mesh.n_dofs)
        if zero_constrained:
            x[con] = 0.0
        ref = mf.vmult(x)
        got = ON.mf_apply(n, mesh.h, mesh.cell_dofs(), coef, con, x)
        assert np.abs(got - ref).max() <= TOL * np.abs(ref).max()
    # compute_diagonal (tests/laplace_matrix_free.hpp:75-98,158-199)
    d = ON.mf_diagonal(n, mesh.h, mesh.cell_dofs(), coef, con)
    np.testing.assert_allclose(d, mf.diagonal(), rtol=TOL, atol=0.0)
    assert np.all(d[con] == 1.0)


def test_native_matrix_free_operator_is_thread_count_independent():
    """The layer-parity schedule of the port must give the same bits for any number of threads."""
    n = (9, 10, 11)
    mesh, coef, mf = _problem(n, "linear")
    con = mesh.constrained_mask()
    x = np.random.default_rng(3).standard_normal(mesh.n_dofs)
    before = ON.num_threads()
    try:
        outs = []
        for t in (1, 2, 5):
            ON.set_num_threads(t)
            outs.append(ON.mf_apply(n, mesh.h, mesh.cell_dofs(), coef, con, x))
    finally:
        ON.set_num_threads(before)
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])


def test_native_csr_spmv():
    """source/dealii/dealii_trilinos_matrix_operator.cc:28-35; the banded 30 x 39 matrix a_ij = i + j of
    tests/test_sparse_matrix_device_operator.cu:31-133 (exact) and a random pattern."""
    n_rows, w = 30, 10
    rows = np.repeat(np.arange(n_rows), w)
    cols = rows + np.tile(np.arange(w), n_rows)
    A = sp.csr_matrix(((rows + cols).astype(float), (rows, cols)), shape=(n_rows, n_rows + w - 1))
    ones = np.ones(A.shape[1])
    assert np.array_equal(ON.csr_spmv(A, ones), A.toarray() @ ones)
    rng = np.random.default_rng(11)
    B = sp.random(500, 300, density=0.05, random_state=5, format="csr")
    B.sort_indices()
    x = rng.standard_normal(300)
    ref = O.csr_spmv(B.indptr, B.indices, B.data, x)
    got = ON.csr_spmv(B, x)
    assert np.abs(got - ref).max() <= TOL * np.abs(ref).max()


def _two_level(n, material, degree):
    mesh, coef, mf = _problem(n, material)
    con = mesh.constrained_mask()
    dinv = mf.diagonal_inverse()
    R = O.build_restrictor(mesh, coef, mf.diagonal(), agg=(2, 2, 2), n_eig=2, variant="mf", eig_mode="krylov",
                           constrained=con, initial_guess="lexicographic").csr
    Ac = O.galerkin_coarse_matrix(mf.vmult, R).tocsr()
    p = O.dealii_chebyshev_params(mf.vmult, dinv, mesh.n_dofs, degree=degree, smoothing_range=20.0)
    x0 = O.random_initial_guess(mesh.n_dofs, con)
    return mesh, coef, mf, con, dinv, R, Ac, p, x0


def _aggregation_levels(Ac, n_levels=2):
    """A small aggregation hierarchy on A_c in the oracle's `levels` format: piecewise-constant prolongators over
    runs of 4 rows, Galerkin products, Chebyshev(1) bounds from the Gershgorin circle of D^-1 A."""
    levels = []
    A = Ac.tocsr()
    for _ in range(n_levels):
        n = A.shape[0]
        nc = -(-n // 4)
        P = sp.csr_matrix((np.ones(n), (np.arange(n), np.arange(n) // 4)), shape=(n, nc))
        d = A.diagonal()
        lmax = float((abs(A).sum(axis=1).A1 / d).max())
        levels.append((A, P, (1, lmax / 4.0, lmax)))
        A = (P.T @ A @ P).tocsr()
        A.sort_indices()
    levels.append((A, None, None))
    return levels


@pytest.mark.parametrize("n,material,degree", [((8, 8, 8), "constant", 3), ((8, 8, 8), "linear", 3),
                                               ((12, 10, 8), "linear", 2), ((16, 16, 16), "constant", 3),
                                               ((16, 16, 16), "discontinuous", 1)])
def test_native_vcycles_pcg_coarse_solve(n, material, degree):
    """include/mfmg/common/hierarchy.hpp:246-309 with the matrix-free Chebyshev smoother
    (source/dealii/dealii_matrix_free_smoother.cc:63-76) and `coarse_iters` Jacobi-PCG steps as the coarse solve:
    residual history and final iterate of the port against the numpy oracle."""
    mesh, coef, mf, con, dinv, R, Ac, p, x0 = _two_level(n, material, degree)
    smoother = lambda b, x: O.chebyshev_smoother_apply(mf.vmult, dinv, p, b, x)
    ho = O.TwoLevelHierarchy(mf.vmult, smoother, R, O.pcg_coarse_solver(Ac, 7), 1, False)
    b = np.zeros(mesh.n_dofs)
    cycles = 6
    res_o, _, x_o = O.vcycle_history(ho, mf.vmult, b, x0, n_cycles=cycles)
    x_n, res_n = ON.vcycles(n, mesh.h, mesh.cell_dofs(), coef, con, dinv, p.degree, p.lambda_min, p.lambda_max, R, Ac,
                            7, b, x0, cycles)
    np.testing.assert_allclose(res_n, res_o, rtol=1e-11, atol=1e-15)
    assert np.abs(x_n - x_o).max() <= 1e-12 * np.abs(x0).max()
    # a non-zero right-hand side as well (the harness of the reference uses b = 0 only)
    b = np.random.default_rng(2).standard_normal(mesh.n_dofs)
    b[con] = 0.0
    res_o, _, x_o = O.vcycle_history(ho, mf.vmult, b, x0, n_cycles=3)
    x_n, res_n = ON.vcycles(n, mesh.h, mesh.cell_dofs(), coef, con, dinv, p.degree, p.lambda_min, p.lambda_max, R, Ac,
                            7, b, x0, 3)
    np.testing.assert_allclose(res_n, res_o, rtol=1e-11, atol=1e-15)
    assert np.abs(x_n - x_o).max() <= TOL * 10 * max(np.abs(x_o).max(), 1.0)


@pytest.mark.parametrize("pre_levels", [None, 0, 1])
@pytest.mark.parametrize("n,material", [((8, 8, 8), "constant"), ((16, 16, 16), "linear")])
def test_native_vcycles_amg_coarse_solve(n, material, pre_levels):
    """The multilevel coarse 'solve' (the role of ML / AMGx, source/dealii/dealii_solver.cc:48-66): the port's
    amg_cycle against mfmg_oracle.amg_coarse_solver on the same level matrices."""
    mesh, coef, mf, con, dinv, R, Ac, p, x0 = _two_level(n, material, 3)
    levels = _aggregation_levels(Ac)
    solve = O.amg_coarse_solver(levels, 1, pre_smoothing_levels=pre_levels)
    smoother = lambda b, x: O.chebyshev_smoother_apply(mf.vmult, dinv, p, b, x)
    ho = O.TwoLevelHierarchy(mf.vmult, smoother, R, solve, 1, False)
    b = np.zeros(mesh.n_dofs)
    cycles = 6
    res_o, _, x_o = O.vcycle_history(ho, mf.vmult, b, x0, n_cycles=cycles)
    x_n, res_n = ON.vcycles(n, mesh.h, mesh.cell_dofs(), coef, con, dinv, p.degree, p.lambda_min, p.lambda_max, R, Ac,
                            0, b, x0, cycles, amg_levels=levels, amg_pre_smoothing_levels=pre_levels)
    np.testing.assert_allclose(res_n, res_o, rtol=1e-11, atol=1e-15)
    assert np.abs(x_n - x_o).max() <= 1e-12 * np.abs(x0).max()
