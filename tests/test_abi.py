"""CPU-side checks of the C-ABI boundary: the library loads, exports every symbol that
include/mfmg_hip.h declares, fails loudly without a GPU, and its host-side setup pieces
(INFO parameter reader, assembly, AMGe restrictor, Galerkin product) agree with the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

import mfmg_amd as M
from mfmg_amd import lib as L
from mfmg_amd.api import params_get
import mfmg_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "mfmg_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mfmg_hip_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(mfmg_lib):
    names = _declared_symbols()
    assert len(names) > 50
    raw = C.CDLL(L.LIB_PATH)
    missing = [n for n in names if not hasattr(raw, n)]
    assert not missing, missing
    # and the Python binding declares a signature for each of them
    assert set(names) == set(mfmg_lib._declared)


def test_abi_version_of_header_and_library_agree(mfmg_lib):
    text = open(os.path.join(ROOT, "include", "mfmg_hip.h")).read()
    declared = int(re.search(r"#define\s+MFMG_HIP_ABI_VERSION\s+(\d+)", text).group(1))
    assert mfmg_lib.mfmg_hip_abi_version() == declared == 3


@pytest.mark.skipif(torch.cuda.is_available(), reason="only meaningful on a box without a GPU")
def test_context_fails_loudly_without_gpu(mfmg_lib):
    with pytest.raises(L.MfmgDeviceError, match="no CPU fallback"):
        M.Context()


def test_info_parameter_reader(mfmg_lib):
    # the key set of tests/data/hierarchy_input.info (restated, not copied)
    info = M.params_to_info({
        "eigensolver": {"number of eigenvectors": 2, "tolerance": 1e-14},
        "smoother": {"type": "Gauss-Seidel"},
        "is preconditioner": False,
        "agglomeration": {"partitioner": "block", "nx": 2, "ny": 2, "nz": 2},
        "hidden": {"coarse": {"params": {"smoother: type": "symmetric Gauss-Seidel"}}},
    })
    info = "; a comment line\n" + info
    assert params_get(info, "eigensolver.number of eigenvectors") == "2"
    assert params_get(info, "smoother.type") == "Gauss-Seidel"
    assert params_get(info, "is preconditioner") == "false"
    assert params_get(info, "hidden.coarse.params.smoother: type") == "symmetric Gauss-Seidel"
    with pytest.raises(L.MfmgError, match="No such node"):
        params_get(info, "smoother.degree")
    with pytest.raises(L.MfmgError, match="missing"):
        params_get("a {\n b 1\n", "a.b")


PRM = {"eigensolver": {"number of eigenvectors": 2}, "agglomeration": {"nx": 2, "ny": 2, "nz": 2}}


@pytest.mark.parametrize("n,material", [((4, 4, 4), "constant"), ((6, 5, 4), "linear"), ((8, 8), "discontinuous")])
def test_host_assembly_matches_oracle(mfmg_lib, n, material):
    p = M.LaplaceProblem(n, material)
    mesh = O.StructuredMesh(n)
    coef = O.coefficient_table(mesh, material)
    np.testing.assert_allclose(p.coefficient.numpy(), coef, rtol=1e-15)
    assert np.array_equal(p.cell_dofs.numpy(), mesh.cell_dofs())
    assert np.array_equal(p.constrained.numpy().astype(bool), mesh.constrained_mask())
    A = M.host_assemble_matrix(p, "assembled")
    Ao = O.assemble_csr(mesh, coef)
    assert abs(A - Ao).max() < 1e-14 * abs(Ao).max()
    # matrix-free semantics: identity rows on constrained DoFs
    Amf = M.host_assemble_matrix(p, "matrix_free")
    x = np.random.default_rng(1).random(mesh.n_dofs)
    np.testing.assert_allclose(Amf @ x, O.MatrixFreeLaplace(mesh, coef).vmult(x), rtol=1e-12, atol=1e-13)


@pytest.mark.parametrize("n,material", [((4, 4, 4), "constant"), ((6, 6, 4), "linear_x"), ((5, 4, 3), "linear")])
def test_host_restrictor_matrix_free_matches_oracle(mfmg_lib, n, material):
    """'mf' agglomerate operator + Krylov selection is unique, so the product's own setup
    must reproduce the oracle's R (include/mfmg/dealii/amge_host.templates.hpp:278-350)."""
    p = M.LaplaceProblem(n, material)
    mesh = O.StructuredMesh(n)
    coef = O.coefficient_table(mesh, material)
    R = M.host_build_restrictor(p, PRM, matrix_free=True)
    mf = O.MatrixFreeLaplace(mesh, coef)
    Ro = O.build_restrictor(mesh, coef, mf.diagonal(), n_eig=2, variant="mf", eig_mode="krylov").csr
    assert R.shape == Ro.shape
    assert abs(R - Ro).max() < 1e-11
    Ac = M.host_galerkin(p, R, "matrix_free")
    Aco = O.galerkin_coarse_matrix(mf.vmult, Ro)
    assert abs(Ac - Aco).max() < 1e-11 * abs(Aco).max()


def test_host_restrictor_device_variant_properties(mfmg_lib):
    """The dense 'device' variant is not unique inside degenerate eigenspaces (the reference's own
    gold depends on the LAPACK in use); check structure and the Rayleigh quotients."""
    p = M.LaplaceProblem((4, 4, 4))
    mesh = O.StructuredMesh((4, 4, 4))
    coef = O.coefficient_table(mesh)
    A = O.assemble_csr(mesh, coef)
    prm = {"eigensolver": {"number of eigenvectors": 2, "selection": "lapack"}, "agglomeration": PRM["agglomeration"]}
    R = M.host_build_restrictor(p, prm, matrix_free=False)
    ref = O.build_restrictor(mesh, coef, A.diagonal(), n_eig=2, variant="device", eig_mode="lapack")
    assert R.shape == ref.csr.shape == (16, 125)
    assert np.array_equal(R.indptr, ref.csr.indptr) and np.array_equal(R.indices, ref.csr.indices)
    Ac = M.host_galerkin(p, R, "assembled")
    assert abs(Ac - R @ A @ R.T).max() < 1e-15
    # the lowest eigenvalue of every agglomerate is simple: that row agrees up to sign
    for a in range(8):
        r_mine = R[2 * a].toarray().ravel()
        r_ref = ref.csr[2 * a].toarray().ravel()
        assert min(np.abs(r_mine - r_ref).max(), np.abs(r_mine + r_ref).max()) < 1e-12


@pytest.mark.parametrize("n", [(4, 4, 4), (8, 8), (6, 5, 4)])
def test_host_restrictor_assembled_default_matches_oracle(mfmg_lib, n):
    """Default of the assembled evaluator: unshifted dense eigenproblem + the (unique) Krylov selection."""
    p = M.LaplaceProblem(n, "linear")
    mesh = O.StructuredMesh(n)
    coef = O.coefficient_table(mesh, "linear")
    A = O.assemble_csr(mesh, coef)
    R = M.host_build_restrictor(p, PRM, matrix_free=False)
    Ro = O.build_restrictor(mesh, coef, A.diagonal(), agg=(2, 2, 2)[:len(n)], n_eig=2, variant="device",
                            eig_mode="krylov").csr
    assert R.shape == Ro.shape and abs(R - Ro).max() < 1e-11
    Ac = M.host_galerkin(p, R, "assembled")
    assert abs(Ac - Ro @ A @ Ro.T).max() < 1e-11 * abs(Ac).max()


def test_host_amg_hierarchy_properties(mfmg_lib):
    """Smoothed-aggregation hierarchy of the multilevel coarse solver: Galerkin consistency, coarsening by
    8 per level with the components kept apart, constants reproduced by the tentative space."""
    n = (16, 16, 16)
    p = M.LaplaceProblem(n)
    R = M.host_build_restrictor(p, PRM, matrix_free=True)
    Ac = M.host_galerkin(p, R, "matrix_free")
    rows = np.arange(Ac.shape[0])
    B = np.where(rows % 2 == 0, np.asarray(R.sum(axis=1)).ravel(), 1.0)
    levels = M.host_amg_build(Ac, B, {"solver": {"amg": {"coarsest_size": 100}}}, grid_dims=[8, 8, 8],
                              node_of_row=rows // 2, component_of_row=rows % 2)
    sizes = [A.shape[0] for A, _ in levels]
    assert sizes == [1024, 128, 16]
    for l in range(len(levels) - 1):
        A, P = levels[l]
        An = levels[l + 1][0]
        assert P.shape == (A.shape[0], An.shape[0])
        G = (P.T @ A @ P).tocsr()
        assert abs(G - An).max() < 1e-13 * abs(An).max()
        assert abs(An - An.T).max() < 1e-13 * abs(An).max()
    assert levels[-1][1] is None
    # the multilevel cycle (restated in the oracle) is a contraction for the coarse problem
    lv = []
    for (A, P) in levels:
        if P is None:
            lv.append((A, None, None))
        else:
            mn, mx = O.dealii_chebyshev_eigen_estimate(lambda z: A @ z, 1.0 / A.diagonal(), A.shape[0], n_iter=10,
                                                       start="hashed")
            lv.append((A, P, (2, 1.2 * mx / 10.0, 1.2 * mx)))
    solve = O.amg_coarse_solver(lv)
    rhs = Ac @ np.random.default_rng(0).random(Ac.shape[0])
    x = solve(rhs)
    e0 = np.linalg.norm(rhs)
    assert np.linalg.norm(rhs - Ac @ x) < 0.6 * e0
    # without the geometric hint the greedy strength-of-connection aggregation is used
    alg = M.host_amg_build(Ac, B, {"solver": {"amg": {"coarsest_size": 100}}})
    assert alg[0][0].shape[0] == 1024 and alg[1][0].shape[0] < 1024


def test_host_rejects_unstructured_index_array(mfmg_lib):
    p = M.LaplaceProblem((3, 3, 3))
    p.cell_dofs[5, 2] = p.cell_dofs[5, 3]          # break the shared-vertex structure
    with pytest.raises(L.MfmgError, match="not a logically structured"):
        M.host_assemble_matrix(p)


def test_renumbered_dofs_give_permuted_matrix(mfmg_lib):
    n = (4, 3, 5)
    mesh = O.StructuredMesh(n)
    perm = torch.from_numpy(np.random.default_rng(3).permutation(mesh.n_dofs))
    p = M.LaplaceProblem(n, "linear", dof_numbering=perm)
    A = M.host_assemble_matrix(p).toarray()
    A0 = M.host_assemble_matrix(M.LaplaceProblem(n, "linear")).toarray()
    pn = perm.numpy()
    np.testing.assert_allclose(A[np.ix_(pn, pn)], A0, rtol=1e-14, atol=1e-15)


def test_info_parser_reads_the_reference_data_file():
    """tests/golden/reference_hierarchy_input.info is the reference's tests/data/hierarchy_input.info; the
    values the reference's tests read from it (tests/test_hierarchy.cc:60-75) come out of the INFO parser."""
    import json
    import os
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    info = open(os.path.join(golden, "reference_hierarchy_input.info")).read()
    want = json.load(open(os.path.join(golden, "reference_golds.json")))["ptree_defaults"]
    for key, value in want.items():
        if key == "reference":
            continue
        got = params_get(info, key)
        if isinstance(value, bool):
            assert got == ("true" if value else "false")
        elif isinstance(value, (int, float)):
            assert float(got) == float(value)
        else:
            assert got == value
    assert params_get(info, "hidden.coarse.params.coarse: type") == "Amesos-KLU"


def test_info_text_to_params_and_back():
    """The INFO reader of the Python side (examples/hierarchy_driver.py reads the reference's input file with it):
    the reference's data file, and a round trip through params_to_info."""
    text = open(os.path.join(ROOT, "tests", "golden", "reference_hierarchy_input.info")).read()
    p = M.info_to_params(text)
    assert p["eigensolver"] == {"number of eigenvectors": 2, "tolerance": 1e-14}
    assert p["smoother"]["type"] == "Gauss-Seidel" and p["is preconditioner"] is False
    assert p["agglomeration"] == {"partitioner": "block", "nx": 2, "ny": 2, "nz": 2}
    assert p["laplace"]["n_refinements"] == 5 and p["laplace"]["reordering"] == "None"
    assert p["hidden"]["coarse"]["params"]["smoother: type"] == "symmetric Gauss-Seidel"
    assert M.info_to_params(M.params_to_info(p)) == p
    # the library's own parser reads what params_to_info writes
    assert params_get(M.params_to_info(p), "hidden.coarse.params.coarse: type") == "Amesos-KLU"
