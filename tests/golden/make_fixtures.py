"""Generates tests/golden/oracle_vcycle_*.npz: seeded inputs and the oracle's outputs for small cases of the
V-cycle path (restrictor, coarse operator, smoother bounds, residual history).

The reference itself (C++ on deal.II / Trilinos / CUDA) cannot be built in this image, so these vectors
are produced by the CPU oracle (oracle/mfmg_oracle.py), which is pinned against the reference's own gold
numbers (reference_golds.json, tests/test_oracle_fixtures.py).  The fixtures freeze the oracle's answers:
tests compare (a) the oracle of the day against them (regression of the checker) and (b) the HIP path
against them on the GPU box, where only the committed data travels.

    python tests/golden/make_fixtures.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import mfmg_oracle as O  # noqa: E402

CASES = {
    # name: (cells, material, smoother degree, n_eig)
    "mf_cheb3_8x8x8_linear": ((8, 8, 8), "linear", 3, 2),
    "mf_cheb3_12x6x4_constant": ((12, 6, 4), "constant", 3, 2),
}


def build(cells, material, degree, n_eig):
    """The setup of tests/test_gpu_hierarchy.py::test_matrix_free_chebyshev_vcycle_history."""
    mesh = O.StructuredMesh(cells)
    coef = O.coefficient_table(mesh, material)
    con = mesh.constrained_mask()
    mf = O.MatrixFreeLaplace(mesh, coef)
    dinv = mf.diagonal_inverse()
    R = O.build_restrictor(mesh, coef, mf.diagonal(), n_eig=n_eig, variant="mf", eig_mode="krylov").csr
    Ac = O.galerkin_coarse_matrix(mf.vmult, R).tocsr()
    p = O.dealii_chebyshev_params(mf.vmult, dinv, mesh.n_dofs, degree=degree, smoothing_range=20.0, start="hashed")
    smoother = lambda b, x: O.chebyshev_smoother_apply(mf.vmult, dinv, p, b, x)
    h = O.TwoLevelHierarchy(mf.vmult, smoother, R, O.direct_coarse_solver(Ac), 1, False)
    x0 = O.random_initial_guess(mesh.n_dofs, con)
    b = np.zeros(mesh.n_dofs)
    res, rate, x = O.vcycle_history(h, mf.vmult, b, x0)
    return dict(cells=np.array(cells), material=material, degree=degree, n_eig=n_eig, x0=x0, b=b,
                vmult_x0=mf.vmult(x0), diagonal=mf.diagonal(), lambda_max=p.lambda_max, lambda_min=p.lambda_min,
                R_indptr=R.indptr, R_indices=R.indices, R_data=R.data, R_shape=np.array(R.shape),
                Ac_indptr=Ac.indptr, Ac_indices=Ac.indices, Ac_data=Ac.data,
                history=np.asarray(res), rate=rate, x_final=x)


if __name__ == "__main__":
    for name, args in CASES.items():
        out = build(*args)
        np.savez_compressed(os.path.join(HERE, f"oracle_vcycle_{name}.npz"), **out)
        print(name, "history", out["history"])
