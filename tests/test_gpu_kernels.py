"""Parity of the hand-written HIP kernels against the CPU oracle, through the C ABI.
FP64 tolerance: 1e-12 relative to the vector's max-norm per kernel (the V-cycle level
requirement of BASELINE.json is 1e-10 relative on the residual history)."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

import mfmg_amd as M
from mfmg_amd import lib as L
import mfmg_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-12


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()


def host(t, ctx):
    ctx.synchronize()
    return t.cpu().numpy()


def relerr(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


# ---- matrix-free operator --------------------------------------------------------------
CASES = [((4, 4, 4), "constant"), ((7, 5, 6), "linear"), ((16, 16, 16), "discontinuous"),
         ((64, 9, 11), "linear_x"), ((70, 12, 10), "linear"), ((130, 8, 7), "constant"),
         # rows with a nearly empty last chunk and >= 64 node rows, one coefficient per cell: the tail columns run
         # as a rotated slab inside the same launch
         ((65, 70, 5), "constant"), ((64, 64, 2), "constant"), ((141, 63, 3), "constant"), ((128, 65, 4), "constant")]


@pytest.mark.parametrize("n,material", CASES)
def test_mf_vmult_matches_oracle(ctx, n, material):
    mesh = O.StructuredMesh(n)
    coef = O.coefficient_table(mesh, material)
    ref = O.MatrixFreeLaplace(mesh, coef)
    prob = M.LaplaceProblem(n, material, device="cuda")
    op = M.MatrixFreeLaplace(ctx, prob)
    x = np.random.default_rng(0).random(mesh.n_dofs)   # nonzero on constrained DoFs too
    y = torch.empty(mesh.n_dofs, dtype=torch.float64, device="cuda")
    op.vmult(y, dev(x))
    assert relerr(host(y, ctx), ref.vmult(x)) < TOL
    np.testing.assert_allclose(host(op.diagonal(), ctx), ref.diagonal(), rtol=1e-13)
    np.testing.assert_allclose(host(op.diagonal_inverse(), ctx), ref.diagonal_inverse(), rtol=1e-13)


@pytest.mark.parametrize("nw,ty,tz", [(1, 2, 1), (1, 5, 3), (2, 3, 8), (4, 4, 16), (8, 1, 2), (3, 2, 5), (4, 12, 2),
                                      (8, 5, 64), (0, 1, 1)])
def test_mf_vmult_independent_of_tile(ctx, nw, ty, tz):
    n = (20, 13, 9)
    mesh = O.StructuredMesh(n)
    coef = O.coefficient_table(mesh, "linear")
    ref = O.MatrixFreeLaplace(mesh, coef)
    op = M.MatrixFreeLaplace(ctx, M.LaplaceProblem(n, "linear", device="cuda"))
    x = np.random.default_rng(1).random(mesh.n_dofs)
    y0 = torch.empty(mesh.n_dofs, dtype=torch.float64, device="cuda")
    op.vmult(y0, dev(x))
    y0 = host(y0, ctx).copy()
    op.set_tile(ty, tz, nw)
    y = torch.empty(mesh.n_dofs, dtype=torch.float64, device="cuda")
    op.vmult(y, dev(x))
    y = host(y, ctx)
    assert relerr(y, ref.vmult(x)) < TOL
    # owner-computes without atomics: the tiling must not change a single bit
    assert np.array_equal(y, y0)


def test_mf_vmult_host_arrays_and_renumbered_dofs(ctx):
    n = (9, 6, 5)
    mesh = O.StructuredMesh(n)
    coef = O.coefficient_table(mesh, "linear")
    ref = O.MatrixFreeLaplace(mesh, coef)
    perm = np.random.default_rng(5).permutation(mesh.n_dofs)
    prob = M.LaplaceProblem(n, "linear", device="cpu", dof_numbering=torch.from_numpy(perm))
    op = M.MatrixFreeLaplace(ctx, prob)          # host arrays are staged by the library
    x_lex = np.random.default_rng(2).random(mesh.n_dofs)
    x = np.empty_like(x_lex)
    x[perm] = x_lex                               # value of node i lives at DoF perm[i]
    y = torch.empty(mesh.n_dofs, dtype=torch.float64, device="cuda")
    op.vmult(y, dev(x))
    assert relerr(host(y, ctx)[perm], ref.vmult(x_lex)) < TOL


@pytest.mark.parametrize("n,tile", [((12, 10, 9), None), ((70, 30, 12), (3, 4, 4)), ((20, 13, 9), (2, 3, 2))])
def test_mf_computed_ids_change_no_bit(ctx, monkeypatch, n, tile):
    """A lexicographic numbering with Dirichlet faces is computed by the eight-coefficient kernels instead of read from
    the records (mfmg_hip_mf_laplace_ids_computed); a renumbered mesh and the one-coefficient kernels keep the stored
    ids.  Every mode of the kernel must give the same bits either way."""
    prob = M.LaplaceProblem(n, "linear", device="cuda")
    op = M.MatrixFreeLaplace(ctx, prob)
    assert op.ids_computed()
    monkeypatch.setenv("MFMG_MF_AFFINE_IDS", "0")
    op0 = M.MatrixFreeLaplace(ctx, prob)
    assert not op0.ids_computed()
    monkeypatch.delenv("MFMG_MF_AFFINE_IDS")
    assert not M.MatrixFreeLaplace(ctx, M.LaplaceProblem(n, "constant", device="cuda")).ids_computed()
    perm = torch.from_numpy(np.random.default_rng(1).permutation(prob.n_dofs))
    assert not M.MatrixFreeLaplace(ctx, M.LaplaceProblem(n, "linear", device="cpu", dof_numbering=perm)).ids_computed()
    rng = np.random.default_rng(4)
    x, b, xp = (dev(rng.standard_normal(prob.n_dofs)) for _ in range(3))
    outs = []
    for o in (op, op0):
        if tile:
            o.set_tile(*tile)
        y = [torch.empty_like(x) for _ in range(4)]
        o.vmult(y[0], x)
        o.residual(x, b, y[1])
        o.smoother_step(b, x, None, 0.0, 0.7, y[2])
        o.smoother_step(b, x, xp, 0.3, 0.7, y[3])
        ctx.synchronize()
        outs.append(y)
    for a, c in zip(*outs):
        assert torch.equal(a, c)


@pytest.mark.parametrize("n,material,tile", [((12, 10, 9), "discontinuous", None), ((12, 10, 9), "constant", None),
                                             ((70, 30, 12), "constant", (3, 4, 4)), ((70, 30, 12), "linear", (3, 4, 4)),
                                             ((20, 13, 9), "constant", (4, 2, 2)), ((20, 13, 9), "constant", (2, 3, 2)),
                                             ((66, 67, 6), "constant", (3, 2, 4))])
def test_mf_fused_epilogues(ctx, n, material, tile):
    """Every epilogue of the operator kernel, for both coefficient layouts, with one and several wavefronts per
    workgroup (the hand-over row between wavefronts has its own epilogue path) and with the tail slab."""
    mesh = O.StructuredMesh(n)
    coef = O.coefficient_table(mesh, material)
    ref = O.MatrixFreeLaplace(mesh, coef)
    dinv = ref.diagonal_inverse()
    op = M.MatrixFreeLaplace(ctx, M.LaplaceProblem(n, material, device="cuda"))
    if tile:
        op.set_tile(*tile)
    if material == "constant":
        # the cell-constant layout derives D^-1 in the kernel by default; the stored form must give the same smoother step
        assert not op.diagonal_in_record()
        ctx.set_stored_diagonal(True)
        op_s = M.MatrixFreeLaplace(ctx, M.LaplaceProblem(n, material, device="cuda"))
        ctx.set_stored_diagonal(False)
        assert op_s.diagonal_in_record()
        if tile:
            op_s.set_tile(*tile)
        r0 = np.random.default_rng(4)
        xs, bs, xps = r0.random(mesh.n_dofs), r0.random(mesh.n_dofs), r0.random(mesh.n_dofs)
        o1 = torch.empty(mesh.n_dofs, dtype=torch.float64, device="cuda")
        o2 = torch.empty_like(o1)
        op.smoother_step(dev(bs), dev(xs), dev(xps), 0.3, 0.45, o1)
        op_s.smoother_step(dev(bs), dev(xs), dev(xps), 0.3, 0.45, o2)
        assert relerr(host(o1, ctx), host(o2, ctx)) < 1e-14
    rng = np.random.default_rng(3)
    x, b, xp = rng.random(mesh.n_dofs), rng.random(mesh.n_dofs), rng.random(mesh.n_dofs)
    out = torch.empty(mesh.n_dofs, dtype=torch.float64, device="cuda")
    op.residual(dev(x), dev(b), out)
    assert relerr(host(out, ctx), ref.vmult(x) - b) < TOL
    op.smoother_step(dev(b), dev(x), None, 0.0, 0.7, out)
    assert relerr(host(out, ctx), x - 0.7 * dinv * (ref.vmult(x) - b)) < TOL
    op.smoother_step(dev(b), dev(x), dev(xp), 0.3, 0.45, out)
    assert relerr(host(out, ctx), x + 0.3 * (x - xp) - 0.45 * dinv * (ref.vmult(x) - b)) < TOL
    # the polynomial term may overwrite its own x_{k-1}
    xp_d = dev(xp)
    op.smoother_step(dev(b), dev(x), xp_d, 0.3, 0.45, xp_d)
    assert relerr(host(xp_d, ctx), x + 0.3 * (x - xp) - 0.45 * dinv * (ref.vmult(x) - b)) < TOL
    with pytest.raises(L.MfmgError, match="in place"):
        xd = dev(x)
        op.vmult(xd, xd)


@pytest.mark.parametrize("n,material", [((20, 12, 9), "linear"), ((20, 12, 9), "constant"), ((66, 67, 4), "constant")])
def test_mf_fp32_instance(ctx, n, material):
    """BASELINE.json configs[4] (FP32): same kernels instantiated for float (eight coefficients per cell, one per
    cell, one per cell with the tail columns as a slab); tolerance 1e-4 relative against the FP64 oracle."""
    mesh = O.StructuredMesh(n)
    coef = O.coefficient_table(mesh, material)
    ref = O.MatrixFreeLaplace(mesh, coef)
    op = M.MatrixFreeLaplaceF32(ctx, M.LaplaceProblem(n, material, device="cuda"))
    assert op.cell_constant_layout() == (material == "constant")
    rng = np.random.default_rng(8)
    x, b, xp = rng.random(mesh.n_dofs), rng.random(mesh.n_dofs), rng.random(mesh.n_dofs)
    f = lambda a: torch.from_numpy(a.astype(np.float32)).cuda()
    out = torch.empty(mesh.n_dofs, dtype=torch.float32, device="cuda")
    op.vmult(out, f(x))
    ctx.synchronize()
    assert relerr(out.cpu().numpy().astype(float), ref.vmult(x)) < 1e-4
    dinv = ref.diagonal_inverse()
    np.testing.assert_allclose(op.diagonal_inverse().cpu().numpy(), dinv, rtol=1e-5)
    op.smoother_step(f(b), f(x), f(xp), 0.3, 0.45, out)
    ctx.synchronize()
    assert relerr(out.cpu().numpy().astype(float), x + 0.3 * (x - xp) - 0.45 * dinv * (ref.vmult(x) - b)) < 1e-4
    op.residual(f(x), f(b), out)
    ctx.synchronize()
    assert relerr(out.cpu().numpy().astype(float), ref.vmult(x) - b) < 1e-4


@pytest.mark.parametrize("n", [(1, 1, 1), (2, 1, 3), (1, 5, 1), (62, 1, 1), (63, 1, 1), (64, 2, 1), (125, 2, 1), (126, 1, 2)])
def test_mf_degenerate_meshes(ctx, n):
    """One-cell-thick meshes and chunk-boundary widths (63 owned columns per chunk): every mode against the oracle."""
    mesh = O.StructuredMesh(n)
    coef = O.coefficient_table(mesh, "linear")
    ref = O.MatrixFreeLaplace(mesh, coef)
    op = M.MatrixFreeLaplace(ctx, M.LaplaceProblem(n, "linear", device="cuda"))
    rng = np.random.default_rng(21)
    x, b, xp = rng.random(mesh.n_dofs), rng.random(mesh.n_dofs), rng.random(mesh.n_dofs)
    out = torch.empty(mesh.n_dofs, dtype=torch.float64, device="cuda")
    op.vmult(out, dev(x))
    assert relerr(host(out, ctx), ref.vmult(x)) < TOL
    dinv = ref.diagonal_inverse()
    op.smoother_step(dev(b), dev(x), dev(xp), 0.3, 0.45, out)
    assert relerr(host(out, ctx), x + 0.3 * (x - xp) - 0.45 * dinv * (ref.vmult(x) - b)) < TOL


def test_csr_degenerate_shapes(ctx):
    """Empty matrix rows, a matrix without entries, single row / column."""
    import scipy.sparse as sp
    for A in (sp.csr_matrix((5, 7)), sp.csr_matrix(np.array([[0.0, 2.0, 0.0]])), sp.csr_matrix(np.array([[3.0], [0.0], [1.0]]))):
        Ad = M.SparseMatrixDevice(ctx, A)
        x = np.arange(1, A.shape[1] + 1, dtype=np.float64)
        y = torch.full((A.shape[0],), np.nan, dtype=torch.float64, device="cuda")
        Ad.vmult(y, dev(x))
        np.testing.assert_array_equal(host(y, ctx), A @ x)
        assert Ad.transpose().shape == (A.shape[1], A.shape[0])


def test_csr_few_long_rows(ctx):
    """About a thousand rows of about a thousand entries (the bottom of the aggregation hierarchy: transfer operators,
    dense triangular inverses): a workgroup per row; every fused mode, and against a wavefront per row."""
    import scipy.sparse as sp
    rng = np.random.default_rng(41)
    n = 700
    A = sp.random(n, n, density=0.6, random_state=np.random.default_rng(5), format="csr") + sp.diags(np.full(n, 50.0))
    A = A.tocsr()
    A[13, :] = 0.0                                          # an empty row and a short one
    A[14, :] = 0.0
    A[14, 3] = 2.0
    A.eliminate_zeros()
    A.sort_indices()
    Ad = M.SparseMatrixDevice(ctx, A)
    assert Ad.get_kernel() == (256, 0)
    x, b, xp = rng.random(n), rng.random(n), rng.random(n)
    dinv = rng.random(n)
    ref = A @ x
    out = torch.empty(n, dtype=torch.float64, device="cuda")
    for lanes in (256, 64):
        Ad.set_kernel(lanes, -1)
        Ad.vmult(out, dev(x))
        assert relerr(host(out, ctx), ref) < TOL
        Ad.residual(dev(x), dev(b), out)
        assert relerr(host(out, ctx), ref - b) < TOL
        Ad.smoother_step(dev(dinv), dev(b), dev(x), dev(xp), 0.25, 0.6, out)
        assert relerr(host(out, ctx), x + 0.25 * (x - xp) - 0.6 * dinv * (ref - b)) < TOL
    T = sp.random(900, 5000, density=0.1, random_state=np.random.default_rng(6), format="csr")   # rectangular
    Td = M.SparseMatrixDevice(ctx, T)
    assert Td.get_kernel()[0] == 256
    xw = rng.random(5000)
    yw = torch.empty(900, dtype=torch.float64, device="cuda")
    Td.vmult(yw, dev(xw))
    assert relerr(host(yw, ctx), T @ xw) < TOL


@pytest.mark.parametrize("material,expect", [("constant", True), ("linear", False), ("cellwise", True)])
def test_mf_cell_constant_layout(ctx, material, expect):
    """One coefficient per cell where a cell's eight quadrature values are equal: chosen automatically, same
    operator as the general layout to rounding, every fused mode against the oracle."""
    n = (70, 11, 9)
    mesh = O.StructuredMesh(n)
    rng = np.random.default_rng(17)
    if material == "cellwise":
        coef = np.repeat(1.0 + rng.random((mesh.n_cells, 1)), 8, axis=1)       # piecewise constant, varies by cell
        prob = M.LaplaceProblem(n, "constant", device="cuda")
        prob.coefficient = torch.from_numpy(coef).cuda()
    else:
        coef = O.coefficient_table(mesh, material)
        prob = M.LaplaceProblem(n, material, device="cuda")
    ref = O.MatrixFreeLaplace(mesh, coef)
    op = M.MatrixFreeLaplace(ctx, prob)
    assert op.cell_constant_layout() == expect
    ctx.set_cell_constant_layout(False)
    try:
        op_general = M.MatrixFreeLaplace(ctx, prob)
    finally:
        ctx.set_cell_constant_layout(True)
    assert not op_general.cell_constant_layout()
    x, b, xp = rng.random(mesh.n_dofs), rng.random(mesh.n_dofs), rng.random(mesh.n_dofs)
    dinv = ref.diagonal_inverse()
    np.testing.assert_allclose(host(op.diagonal_inverse(), ctx), dinv, rtol=1e-13)
    outs = []
    for o in (op, op_general):
        out = torch.empty(mesh.n_dofs, dtype=torch.float64, device="cuda")
        o.vmult(out, dev(x))
        assert relerr(host(out, ctx), ref.vmult(x)) < TOL
        o.residual(dev(x), dev(b), out)
        assert relerr(host(out, ctx), ref.vmult(x) - b) < TOL
        o.smoother_step(dev(b), dev(x), dev(xp), 0.3, 0.45, out)
        assert relerr(host(out, ctx), x + 0.3 * (x - xp) - 0.45 * dinv * (ref.vmult(x) - b)) < TOL
        outs.append(host(out, ctx).copy())
    assert relerr(outs[0], outs[1]) < 1e-13


@pytest.mark.parametrize("halo_lanes", [1, 3])
@pytest.mark.parametrize("nw,ty,tz", [(0, 0, 0), (1, 2, 1), (4, 3, 8), (2, 5, 2), (4, 4, 3)])
def test_mf_tail_slab_all_modes_and_renumbering(ctx, nw, ty, tz, halo_lanes):
    """The rotated slab of the tail columns (cell-wise constant coefficient, 67 node columns = 63 + 4; chunk records with one
    halo lane -- with three, the layout of the multi-term sweep, the columns are spread evenly and there is no slab): every
    fused mode, a random DoF numbering, Dirichlet and interior nodes, tile independence."""
    ctx.set_mf_fused_terms(halo_lanes)
    try:
        _tail_slab_case(ctx, nw, ty, tz)
    finally:
        ctx.set_mf_fused_terms(3)


def _tail_slab_case(ctx, nw, ty, tz):
    n = (66, 67, 4)
    mesh = O.StructuredMesh(n)
    rng = np.random.default_rng(31)
    coef = np.repeat(1.0 + rng.random((mesh.n_cells, 1)), 8, axis=1)
    ref = O.MatrixFreeLaplace(mesh, coef)
    perm = rng.permutation(mesh.n_dofs)
    prob = M.LaplaceProblem(n, "constant", device="cuda", dof_numbering=torch.from_numpy(perm))
    prob.coefficient = torch.from_numpy(coef).cuda()
    op = M.MatrixFreeLaplace(ctx, prob)
    assert op.cell_constant_layout()
    op.set_tile(ty, tz, nw)
    x, b, xp = rng.random(mesh.n_dofs), rng.random(mesh.n_dofs), rng.random(mesh.n_dofs)

    def to_dof(v):                        # value of node i lives at DoF perm[i]
        o = np.empty_like(v)
        o[perm] = v
        return o

    out = torch.empty(mesh.n_dofs, dtype=torch.float64, device="cuda")
    op.vmult(out, dev(to_dof(x)))
    assert relerr(host(out, ctx)[perm], ref.vmult(x)) < TOL
    dinv = ref.diagonal_inverse()
    np.testing.assert_allclose(host(op.diagonal_inverse(), ctx)[perm], dinv, rtol=1e-13)
    op.residual(dev(to_dof(x)), dev(to_dof(b)), out)
    assert relerr(host(out, ctx)[perm], ref.vmult(x) - b) < TOL
    op.smoother_step(dev(to_dof(b)), dev(to_dof(x)), dev(to_dof(xp)), 0.3, 0.45, out)
    assert relerr(host(out, ctx)[perm], x + 0.3 * (x - xp) - 0.45 * dinv * (ref.vmult(x) - b)) < TOL
    op.smoother_step(dev(to_dof(b)), dev(to_dof(x)), None, 0.0, 0.45, out)
    first = host(out, ctx).copy()
    assert relerr(first[perm], x - 0.45 * dinv * (ref.vmult(x) - b)) < TOL
    op.set_tile(3, 2, 2)
    op.smoother_step(dev(to_dof(b)), dev(to_dof(x)), None, 0.0, 0.45, out)
    assert np.array_equal(host(out, ctx), first)


# ---- several smoother terms in one sweep (mf_cheb_fused.hip) -----------------------------------------------------------
def _cellwise_problem(n, seed=7):
    prob = M.LaplaceProblem(n, "constant", device="cuda")
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    prob.coefficient = (0.5 + torch.rand(prob.n_cells_total, 1, dtype=torch.float64, device="cuda", generator=g)).expand(-1, 8).contiguous()
    return prob


SWEEP_COEFS = [(0.0, 0.61), (0.23, 0.87), (0.31, 0.79)]     # (alpha, beta) of three Chebyshev-like terms


@pytest.mark.parametrize("n,material", [((6, 5, 7), "constant"), ((20, 17, 9), "cellwise"), ((70, 30, 20), "constant"),
                                        ((130, 40, 33), "cellwise"), ((86, 19, 11), "cellwise")])   # (87 = 58 + 29 columns: the widest narrow last chunk)
@pytest.mark.parametrize("n_terms", [2, 3])
@pytest.mark.parametrize("tile", [None, (4, 3, 8), (8, 3, 5), (2, 4, 7), (8, 2, 64), (1, 4, 3)])
def test_smoother_sweep_equals_term_by_term_bit_for_bit(ctx, n, material, n_terms, tile):
    """2 or 3 smoother terms in one sweep over the mesh == the same terms as one launch each, bit for bit, for every
    tiling of the sweep (owner computes; same cell kernel and summation order; source/dealii/dealii_matrix_free_smoother.cc:63-76)."""
    if tile is not None and tile[0] * tile[1] - 2 * n_terms + 1 < 1:
        pytest.skip("tile smaller than its halo rows")
    prob = _cellwise_problem(n) if material == "cellwise" else M.LaplaceProblem(n, material, device="cuda")
    op = M.MatrixFreeLaplace(ctx, prob)
    assert op.sweep_available(n_terms)
    op.set_sweep_reference(True)          # the arithmetic of the one-term kernel (the default, mode space, rounds differently)
    N = prob.n_dofs
    g = torch.Generator(device="cuda")
    g.manual_seed(1234)
    x = torch.rand(N, dtype=torch.float64, device="cuda", generator=g)          # nonzero on the Dirichlet DoFs too
    b = torch.rand(N, dtype=torch.float64, device="cuda", generator=g)
    al = [c[0] for c in SWEEP_COEFS][:n_terms]
    be = [c[1] for c in SWEEP_COEFS][:n_terms]
    its = [x]
    for k in range(n_terms):
        o = torch.full_like(x, float("nan"))
        op.smoother_step(b, its[-1], its[-2] if k > 0 else None, al[k], be[k], o)
        its.append(o)
    if tile is not None:
        op.set_sweep_tile(*tile)
    out = torch.full_like(x, float("nan"))
    outp = torch.full_like(x, float("nan"))
    op.smoother_sweep(al, be, b, x, out, outp)
    ctx.synchronize()
    assert torch.equal(out, its[-1])
    assert torch.equal(outp, its[-2])
    out2 = torch.full_like(x, float("nan"))
    op.smoother_sweep(al, be, b, x, out2, None)           # without the second output
    ctx.synchronize()
    assert torch.equal(out2, its[-1])
    # the default arithmetic -- the cell matrix in mode space, butterflies shared between neighbouring cells -- is the same
    # operator with its own rounding, and it does not depend on the tiling either (bit for bit)
    op.set_sweep_reference(False)
    outm = torch.full_like(x, float("nan"))
    outmp = torch.full_like(x, float("nan"))
    op.smoother_sweep(al, be, b, x, outm, outmp)
    ctx.synchronize()
    assert (outm - its[-1]).abs().max().item() <= 1e-13 * its[-1].abs().max().item()
    assert (outmp - its[-2]).abs().max().item() <= 1e-13 * its[-2].abs().max().item()
    op.set_sweep_tile(3, 3, 4)
    outm2 = torch.full_like(x, float("nan"))
    op.smoother_sweep(al, be, b, x, outm2, None)
    ctx.synchronize()
    assert torch.equal(outm2, outm)


def test_smoother_sweep_against_the_oracle_and_other_layouts(ctx):
    """The three-term sweep against the oracle's operator (1e-12), with D^-1 kept in the records, in FP32, and refused where
    the operator cannot run it (eight coefficients per cell, a random numbering, one halo lane)."""
    n = (40, 33, 21)
    mesh = O.StructuredMesh(n)
    coef = O.coefficient_table(mesh, "constant")
    ref = O.MatrixFreeLaplace(mesh, coef)
    rng = np.random.default_rng(5)
    x, b = rng.random(mesh.n_dofs), rng.random(mesh.n_dofs)
    dinv = ref.diagonal_inverse()
    al = [c[0] for c in SWEEP_COEFS]
    be = [c[1] for c in SWEEP_COEFS]
    want, prev = x, None
    for k in range(3):
        nxt = want + (al[k] * (want - prev) if prev is not None else 0.0) - be[k] * dinv * (ref.vmult(want) - b)
        prev, want = want, nxt
    prob = M.LaplaceProblem(n, "constant", device="cuda")
    op = M.MatrixFreeLaplace(ctx, prob)
    out = torch.empty(mesh.n_dofs, dtype=torch.float64, device="cuda")
    op.smoother_sweep(al, be, dev(b), dev(x), out)
    assert relerr(host(out, ctx), want) < TOL
    ctx.set_stored_diagonal(True)
    try:
        op_stored = M.MatrixFreeLaplace(ctx, prob)
    finally:
        ctx.set_stored_diagonal(False)
    assert op_stored.diagonal_in_record()
    out_s = torch.empty_like(out)
    op_stored.smoother_sweep(al, be, dev(b), dev(x), out_s)
    assert relerr(host(out_s, ctx), want) < TOL
    op32 = M.MatrixFreeLaplaceF32(ctx, prob)
    assert op32.sweep_available(3)
    op32.set_sweep_reference(True)
    o32 = torch.empty(mesh.n_dofs, dtype=torch.float32, device="cuda")
    x32, b32 = dev(x).float(), dev(b).float()
    op32.smoother_sweep(al, be, b32, x32, o32)
    t32 = [x32]
    for k in range(3):
        o = torch.empty_like(x32)
        op32.smoother_step(b32, t32[-1], t32[-2] if k > 0 else None, al[k], be[k], o)
        t32.append(o)
    ctx.synchronize()
    assert torch.equal(o32, t32[-1])
    # refused: eight coefficients per cell, a numbering the kernel cannot compute, records with one halo lane
    assert not M.MatrixFreeLaplace(ctx, M.LaplaceProblem(n, "linear", device="cuda")).sweep_available(3)
    perm = torch.from_numpy(rng.permutation(mesh.n_dofs))
    assert not M.MatrixFreeLaplace(ctx, M.LaplaceProblem(n, "constant", device="cuda", dof_numbering=perm)).sweep_available(2)
    ctx.set_mf_fused_terms(1)
    try:
        op1 = M.MatrixFreeLaplace(ctx, prob)
    finally:
        ctx.set_mf_fused_terms(3)
    assert not op1.sweep_available(2)
    with pytest.raises(L.MfmgNotImplementedError):
        op1.smoother_sweep(al, be, dev(b), dev(x), out)


@pytest.mark.parametrize("n,material,numbering", [((8, 8), "constant", "lexicographic"), ((12, 7), "linear", "random"),
                                                   ((33, 5), "discontinuous", "lexicographic"), ((1, 1), "constant", "lexicographic")])
def test_mf_operator_in_two_dimensions(ctx, n, material, numbering):
    """The matrix-free operator in 2-D (the reference runs LaplaceMatrixFree<2>, tests/test_hierarchy.cc:276-330,416-443):
    y = A x with identity rows for the constrained DoFs, the diagonal, and every fused mode against the oracle."""
    mesh = O.StructuredMesh(n)
    coef = O.coefficient_table(mesh, material)
    ref = O.MatrixFreeLaplace(mesh, coef)
    rng = np.random.default_rng(5)
    perm = rng.permutation(mesh.n_dofs) if numbering == "random" else np.arange(mesh.n_dofs)
    prob = M.LaplaceProblem(n, material, device="cuda", dof_numbering=torch.from_numpy(perm))
    op = M.MatrixFreeLaplace(ctx, prob)
    x, b, xp = rng.standard_normal(mesh.n_dofs), rng.random(mesh.n_dofs), rng.random(mesh.n_dofs)

    def to_dof(v):
        o = np.empty_like(v)
        o[perm] = v
        return o

    out = torch.empty(mesh.n_dofs, dtype=torch.float64, device="cuda")
    op.vmult(out, dev(to_dof(x)))
    assert relerr(host(out, ctx)[perm], ref.vmult(x)) < TOL
    np.testing.assert_allclose(host(op.diagonal(), ctx)[perm], ref.diagonal(), rtol=1e-13)
    dinv = ref.diagonal_inverse()
    np.testing.assert_allclose(host(op.diagonal_inverse(), ctx)[perm], dinv, rtol=1e-13)
    op.residual(dev(to_dof(x)), dev(to_dof(b)), out)
    assert relerr(host(out, ctx)[perm], ref.vmult(x) - b) < TOL
    op.smoother_step(dev(to_dof(b)), dev(to_dof(x)), dev(to_dof(xp)), 0.3, 0.45, out)
    assert relerr(host(out, ctx)[perm], x + 0.3 * (x - xp) - 0.45 * dinv * (ref.vmult(x) - b)) < TOL
    op.smoother_step(dev(to_dof(b)), dev(to_dof(x)), None, 0.0, 0.45, out)
    assert relerr(host(out, ctx)[perm], x - 0.45 * dinv * (ref.vmult(x) - b)) < TOL


def test_mf_rejects_bad_input(ctx):
    p2 = M.LaplaceProblem((4, 4), device="cuda")
    p2.cell_dofs[5, 2] = p2.cell_dofs[5, 3]
    with pytest.raises(L.MfmgError, match="not a logically structured"):
        M.MatrixFreeLaplace(ctx, p2)
    p = M.LaplaceProblem((3, 3, 3), device="cuda")
    p.cell_dofs[5, 2] = p.cell_dofs[5, 3]
    with pytest.raises(L.MfmgError, match="not a logically structured"):
        M.MatrixFreeLaplace(ctx, p)


def test_mf_operator_properties_large(ctx):
    """Size-independent properties at a size the numpy oracle would not finish quickly:
    symmetry <Ax,y> = <x,Ay>, linearity, constants in the kernel away from the boundary."""
    n = (127, 127, 127)
    prob = M.LaplaceProblem(n, "linear", device="cuda")
    op = M.MatrixFreeLaplace(ctx, prob)
    N = prob.n_dofs
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.rand(N, dtype=torch.float64, device="cuda", generator=g)
    y = torch.rand(N, dtype=torch.float64, device="cuda", generator=g)
    free = (prob.constrained == 0).to(torch.float64)
    x, y = x * free, y * free
    ax, ay, axy = (torch.empty_like(x) for _ in range(3))
    op.vmult(ax, x)
    op.vmult(ay, y)
    assert abs(ctx.dot(ax, y) - ctx.dot(x, ay)) < 1e-12 * abs(ctx.dot(ax, y))
    z = (2.0 * x - 3.0 * y).contiguous()
    op.vmult(axy, z)
    ctx.synchronize()
    assert (axy - (2.0 * ax - 3.0 * ay)).abs().max().item() < 1e-12 * ax.abs().max().item()
    ones = torch.ones(N, dtype=torch.float64, device="cuda")
    a1 = torch.empty_like(ones)
    op.vmult(a1, ones)           # constrained entries are *read as zero*: only rows next to the boundary see them
    ctx.synchronize()
    a1 = a1.reshape(128, 128, 128)
    assert a1[2:-2, 2:-2, 2:-2].abs().max().item() < 1e-12


# ---- CSR kernels --------------------------------------------------------------------------
def test_csr_fixture_banded_operator(ctx):
    """tests/test_sparse_matrix_device_operator.cu:23-137 (exact equality on integer data)."""
    n_rows, nnz_per_row = 30, 10
    rows = np.repeat(np.arange(n_rows), nnz_per_row)
    cols = rows + np.tile(np.arange(nnz_per_row), n_rows)
    A = sp.csr_matrix(((rows + cols).astype(float), (rows, cols)), shape=(n_rows, n_rows + nnz_per_row - 1))
    Ad = M.SparseMatrixDevice(ctx, A)
    assert Ad.shape == (30, 39)                     # build_range/domain_vector sizes
    y = torch.empty(30, dtype=torch.float64, device="cuda")
    Ad.apply(dev(np.ones(39)), y)
    assert np.array_equal(host(y, ctx), A.toarray() @ np.ones(39))
    At = Ad.transpose()
    assert At.shape == (39, 30)
    yt = torch.empty(39, dtype=torch.float64, device="cuda")
    At.apply(dev(np.ones(30)), yt)
    assert np.array_equal(host(yt, ctx), A.toarray().T @ np.ones(30))
    Ad.apply(dev(np.ones(30)), yt, L.TRANS)
    assert np.array_equal(host(yt, ctx), A.toarray().T @ np.ones(30))
    C = Ad.multiply(At)
    yc = torch.empty(30, dtype=torch.float64, device="cuda")
    C.apply(dev(np.ones(30)), yc)
    assert np.array_equal(host(yc, ctx), A.toarray() @ (A.toarray().T @ np.ones(30)))
    assert abs(C.to_scipy() - A @ A.T).max() == 0.0


@pytest.mark.parametrize("shape,density,dense_col", [((300, 211), 0.04, False), ((2000, 1500), 0.01, True),
                                                      ((64, 5000), 0.02, False), ((1, 1), 1.0, False)])
def test_csr_transpose_and_product_on_device(ctx, monkeypatch, shape, density, dense_col):
    """The device setup algebra (csr_algebra.hip) against the host algorithms, BIT FOR BIT: transposed rows longer
    than a wavefront (the workgroup sort), empty rows, rectangular shapes; the product sums in the
    order of the host Gustavson product (reference: cusparseDcsrgemm,
    include/mfmg/cuda/sparse_matrix_device.templates.cuh:373-434, and the Epetra transpose,
    source/cuda/cuda_matrix_operator.cu:93-130)."""
    rng = np.random.default_rng(shape[0] * 7 + shape[1])
    A = sp.random(*shape, density=density, random_state=rng, format="lil", dtype=np.float64)
    if dense_col:
        A[:, 3] = rng.standard_normal((shape[0], 1))      # a transposed row of 2000 entries: LDS sort
        A[5, :] = 0.0                                      # an empty row
    A = sp.csr_matrix(A)
    A.sort_indices()
    B = sp.random(shape[1], 97, density=0.05, random_state=rng, format="csr", dtype=np.float64)
    B.sort_indices()

    def both(fn):
        out = []
        for mode in ("device_only", "host"):
            monkeypatch.setenv("MFMG_CSR_ALGEBRA", mode)
            out.append(fn())
        return out

    Ad, Bd = M.SparseMatrixDevice(ctx, A), M.SparseMatrixDevice(ctx, B)
    t_dev, t_host = both(lambda: Ad.transpose().to_scipy())
    ref = sp.csr_matrix(A.T)
    ref.sort_indices()
    for t in (t_dev, t_host):
        assert np.array_equal(t.indptr, ref.indptr) and np.array_equal(t.indices, ref.indices)
        assert np.array_equal(t.data, ref.data)
    c_dev, c_host = both(lambda: Ad.multiply(Bd).to_scipy())
    assert np.array_equal(c_dev.indptr, c_host.indptr) and np.array_equal(c_dev.indices, c_host.indices)
    assert np.array_equal(c_dev.data, c_host.data)          # same summation order: the same bits
    ref = (A @ B).toarray()
    np.testing.assert_allclose(c_dev.toarray(), ref, rtol=0, atol=1e-13 * max(1.0, np.abs(ref).max()))
    for r in range(c_dev.shape[0]):                         # columns strictly increasing within every row
        assert np.all(np.diff(c_dev.indices[c_dev.indptr[r]:c_dev.indptr[r + 1]]) > 0)


def test_csr_algebra_falls_back_for_rows_beyond_lds(ctx, monkeypatch):
    """Rows with more candidate columns than the LDS tables hold (hash: 2048 candidates; addressed by the column: 8192 columns
    of B) or transposed rows beyond the LDS sort take the host algorithm: the same result, and an error where the test insists on
    the device path.  Between the two limits the column-addressed table serves the row on the device."""
    n = 9000
    A = sp.csr_matrix(np.ones((1, n)))
    B = sp.identity(n, format="csr") * 2.0
    Ad, Bd = M.SparseMatrixDevice(ctx, A), M.SparseMatrixDevice(ctx, B)
    monkeypatch.setenv("MFMG_CSR_ALGEBRA", "device")
    assert np.array_equal(Ad.multiply(Bd).to_scipy().toarray(), 2.0 * np.ones((1, n)))
    At = M.SparseMatrixDevice(ctx, sp.csr_matrix(np.ones((n, 1)))).transpose().to_scipy()
    assert np.array_equal(At.toarray(), np.ones((1, n)))
    monkeypatch.setenv("MFMG_CSR_ALGEBRA", "device_only")
    with pytest.raises(RuntimeError, match="LDS tables"):
        Ad.multiply(Bd)
    with pytest.raises(RuntimeError, match="LDS tables"):
        M.SparseMatrixDevice(ctx, sp.csr_matrix(np.ones((n, 1)))).transpose()
    # 5000 candidates in a row, 5000 columns: beyond the hash table, within the column-addressed one -- on the device, and the
    # bits of the host product (a dense-ish product with sums of several terms per entry)
    rng = np.random.default_rng(4)
    m = 5000
    A2 = sp.random(40, 600, density=0.5, random_state=rng, format="csr", dtype=np.float64)
    B2 = sp.random(600, m, density=0.3, random_state=rng, format="csr", dtype=np.float64)
    C2 = M.SparseMatrixDevice(ctx, A2).multiply(M.SparseMatrixDevice(ctx, B2)).to_scipy()
    monkeypatch.setenv("MFMG_CSR_ALGEBRA", "host")
    C2h = M.SparseMatrixDevice(ctx, A2).multiply(M.SparseMatrixDevice(ctx, B2)).to_scipy()
    assert np.array_equal(C2.indptr, C2h.indptr) and np.array_equal(C2.indices, C2h.indices) and np.array_equal(C2.data, C2h.data)
    assert abs(C2 - A2 @ B2).max() <= 1e-13 * abs(C2h).max()


def test_csr_fixture_random_pattern_and_jacobi(ctx):
    from test_oracle_fixtures import random_pattern_matrix
    A = random_pattern_matrix()                      # tests/test_sparse_matrix_device.cu:43-56
    x = np.arange(10, dtype=float)
    y = torch.empty(10, dtype=torch.float64, device="cuda")
    M.SparseMatrixDevice(ctx, A).vmult(y, dev(x))
    np.testing.assert_allclose(host(y, ctx), A @ x, rtol=1e-15)
    # tests/test_smoother_device.cu:28-119
    n = 30
    T = sp.diags([-np.ones(n - 1), 4 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1]).tocsr()
    Td = M.SparseMatrixDevice(ctx, T)
    dinv = torch.empty(n, dtype=torch.float64, device="cuda")
    Td.inverse_diagonal(dinv)
    out = torch.empty(n, dtype=torch.float64, device="cuda")
    Td.smoother_step(dinv, dev(np.ones(n)), dev(np.zeros(n)), None, 0.0, 1.0, out)
    np.testing.assert_allclose(host(out, ctx), 0.25 * np.ones(n), rtol=1e-14)


@pytest.mark.parametrize("shape,density", [((1, 1), 1.0), ((200, 300), 0.02), ((1000, 1000), 0.03), ((500, 64), 0.5),
                                          ((64, 5000), 0.2)])
def test_csr_spmv_ragged(ctx, shape, density):
    rng = np.random.default_rng(11)
    A = sp.random(*shape, density=density, random_state=rng, format="csr", dtype=np.float64)
    A.sort_indices()
    x = rng.random(shape[1])
    y = torch.empty(shape[0], dtype=torch.float64, device="cuda")
    M.SparseMatrixDevice(ctx, A).vmult(y, dev(x))
    ref = O.csr_spmv(A.indptr, A.indices, A.data, x)    # rows without entries give 0
    assert np.abs(host(y, ctx) - ref).max() <= 1e-13 * max(np.abs(ref).max(), 1.0)


def test_csr_fused_modes_27pt(ctx):
    mesh = O.StructuredMesh((10, 9, 8))
    coef = O.coefficient_table(mesh, "linear")
    A = O.assemble_csr(mesh, coef)
    Ad = M.SparseMatrixDevice(ctx, A)
    n = mesh.n_dofs
    rng = np.random.default_rng(4)
    x, b, xp = rng.random(n), rng.random(n), rng.random(n)
    dinv = 1.0 / A.diagonal()
    out = torch.empty(n, dtype=torch.float64, device="cuda")
    Ad.residual(dev(x), dev(b), out)
    assert relerr(host(out, ctx), A @ x - b) < TOL
    Ad.smoother_step(dev(dinv), dev(b), dev(x), dev(xp), 0.25, 0.6, out)
    assert relerr(host(out, ctx), x + 0.25 * (x - xp) - 0.6 * dinv * (A @ x - b)) < TOL
    with pytest.raises(L.MfmgError, match="out of range"):
        M.SparseMatrixDevice(ctx, (np.array([0, 2], dtype=np.int32), np.array([0, 7], dtype=np.int32),
                                   np.array([1.0, 1.0]), (1, 3))).shape


@pytest.mark.parametrize("comps,symmetric", [(1, False), (2, False), (3, False), (1, True), (2, True), (4, True)])
def test_csr_kernel_variants_on_stencil_matrix(ctx, comps, symmetric):
    """A stencil matrix with `comps` unknowns per grid node (the shape of the AMGe coarse operators): the
    block-diagonal storage is chosen and every kernel variant / fused mode agrees with the CPU SpMV."""
    import scipy.sparse as sp
    dims = (37, 33, 29)
    rng = np.random.default_rng(11)

    def t1(n):
        return sp.diags([np.ones(n - 1), np.ones(n), np.ones(n - 1)], [-1, 0, 1])
    pattern = sp.kron(t1(dims[2]), sp.kron(t1(dims[1]), t1(dims[0]))).tocsr()
    A = sp.kron(pattern, np.ones((comps, comps))).tocsr()
    A.data = rng.random(A.nnz) - 0.3
    # a few rows lose entries (boundary-like irregularity stays inside the stencil)
    A.data[rng.integers(0, A.nnz, 500)] = 0.0
    A.eliminate_zeros()
    if symmetric:                            # the coarse operators are: half of the diagonals is stored
        A = (0.5 * (A + A.T)).tocsr()
    A = (A + sp.diags(np.full(A.shape[0], 30.0))).tocsr()
    A.sort_indices()
    n = A.shape[0]
    Ad = M.SparseMatrixDevice(ctx, A)
    lpr, kind = Ad.get_kernel()
    assert kind == (3 if symmetric else 2), "block-diagonal storage expected for a stencil matrix"
    x, b, xp = rng.random(n), rng.random(n), rng.random(n)
    dinv = 1.0 / A.diagonal()
    out = torch.empty(n, dtype=torch.float64, device="cuda")
    ref = O.csr_spmv(A.indptr, A.indices, A.data, x)
    for k in (2, 1, 0):
        Ad.set_kernel(0, k)
        if k == 2:
            assert Ad.get_kernel()[1] == kind
        if k == 1 and Ad.get_kernel()[1] != 1:
            continue                         # LDS lists are only built for large matrices
        Ad.vmult(out, dev(x))
        assert relerr(host(out, ctx), ref) < TOL
        Ad.residual(dev(x), dev(b), out)
        assert relerr(host(out, ctx), ref - b) < TOL
        Ad.smoother_step(dev(dinv), dev(b), dev(x), dev(xp), 0.25, 0.6, out)
        assert relerr(host(out, ctx), x + 0.25 * (x - xp) - 0.6 * dinv * (ref - b)) < TOL
        Ad.smoother_step(dev(dinv), dev(b), dev(x), None, 0.0, 0.6, out)
        assert relerr(host(out, ctx), x - 0.6 * dinv * (ref - b)) < TOL
    # a rectangular matrix with the same stencil in every row (a prolongator-like shape): row-base storage;
    # with entries missing at random the fill is too low for it and the LDS-cached CSR kernel takes over
    if comps == 1:
        Bw = sp.vstack([sp.hstack([A, 0.5 * A])] * 6).tocsr()       # 212 k rows: above the row-base threshold
        out = torch.empty(Bw.shape[0], dtype=torch.float64, device="cuda")
        keep = rng.random(Bw.nnz) > 0.45
        Bs = sp.csr_matrix((Bw.data[keep], Bw.indices[keep], np.concatenate([[0], np.cumsum(
            np.add.reduceat(keep.astype(np.int64), Bw.indptr[:-1]))])), shape=Bw.shape)
        for Bm, want in ((Bw, 4), (Bs, 1)):
            Bd = M.SparseMatrixDevice(ctx, Bm)
            assert Bd.get_kernel()[1] == want
            xw = rng.random(Bm.shape[1])
            Bd.vmult(out, dev(xw))
            assert relerr(host(out, ctx), Bm @ xw) < TOL
            Bd.set_kernel(0, 0)
            Bd.vmult(out, dev(xw))
            assert relerr(host(out, ctx), Bm @ xw) < TOL
    # an unstructured matrix of the same size keeps the CSR kernels
    cols = rng.integers(0, n, size=(n, 20))
    B = sp.csr_matrix((rng.random(n * 20), cols.ravel(), np.arange(0, 20 * n + 1, 20)), shape=(n, n))
    B.sum_duplicates()
    assert M.SparseMatrixDevice(ctx, B).get_kernel()[1] != 2


@pytest.mark.parametrize("comps,symmetric,reach", [(1, True, 1), (2, True, 1), (2, False, 1), (2, True, 2), (1, False, 2)])
def test_csr_regular_rows_of_translation_invariant_operator(ctx, comps, symmetric, reach):
    """A symmetric stencil matrix whose interior rows repeat one stencil (what the coarse operators of a
    constant-coefficient problem look like): interior rows come from the stencil table, boundary rows and a
    few perturbed rows from the stored planes; every fused mode against scipy, and against the path switched off.
    reach 2: 125 block diagonals (the second level of the aggregation hierarchy), stencils split over wavefronts."""
    import scipy.sparse as sp
    dims = (37, 33, 29) if reach == 1 else (35, 33, 31)
    rng = np.random.default_rng(23)

    def t1(n, a, b):
        if reach == 2:
            return sp.diags([np.full(n - 2, 0.3 * b), np.full(n - 1, b), np.full(n, a), np.full(n - 1, b),
                             np.full(n - 2, 0.3 * b)], [-2, -1, 0, 1, 2])
        return sp.diags([np.full(n - 1, b), np.full(n, a), np.full(n - 1, b)], [-1, 0, 1])
    pattern = sp.kron(t1(dims[2], 2.0, -0.3), sp.kron(t1(dims[1], 1.5, -0.25), t1(dims[0], 1.0, -0.2))).tocsr()
    blk = np.array([[3.0, 0.4], [0.4, 2.0]])[:comps, :comps]
    A = sp.kron(pattern, blk).tocsr()
    A.sort_indices()
    # a few interior rows differ (on the diagonal: still symmetric): they must take the stored-value path
    n = A.shape[0]
    bump = np.zeros(n)
    rows = rng.integers(n // 3, 2 * n // 3, 20)
    bump[rows] = 0.5 * A.diagonal()[rows]
    A = (A + sp.diags(bump)).tocsr()
    if not symmetric:                        # e.g. the coarse operator of one rank: rows of the neighbours emptied
        keep = np.ones(n)
        keep[:400] = 0.0
        A = (sp.diags(keep) @ A + sp.diags(1.0 - keep)).tocsr()
    A.eliminate_zeros()
    A.sort_indices()
    n = A.shape[0]
    Ad = M.SparseMatrixDevice(ctx, A)
    assert Ad.get_kernel()[1] == (3 if symmetric else 2) and Ad.regular_rows()
    # faces and edges of the box are stencil classes of their own; the corners and the perturbed rows stay listed
    n_classes, listed = Ad.stencil_classes()
    assert n_classes >= (18 if reach == 1 else 60) and listed <= (100 if reach == 1 else 600)
    x, b, xp = rng.random(n), rng.random(n), rng.random(n)
    dinv = 1.0 / A.diagonal()
    ref = A @ x
    res = {}
    for on in (True, False):
        Ad.set_regular_rows(on)
        assert Ad.regular_rows() == on
        out = torch.empty(n, dtype=torch.float64, device="cuda")
        Ad.vmult(out, dev(x))
        assert relerr(host(out, ctx), ref) < TOL
        Ad.residual(dev(x), dev(b), out)
        assert relerr(host(out, ctx), ref - b) < TOL
        Ad.smoother_step(dev(dinv), dev(b), dev(x), dev(xp), 0.25, 0.6, out)
        assert relerr(host(out, ctx), x + 0.25 * (x - xp) - 0.6 * dinv * (ref - b)) < TOL
        res[on] = host(out, ctx).copy()
    assert relerr(res[True], res[False]) < 1e-13


def test_csr_row_classes_of_translation_invariant_prolongator(ctx):
    """A rectangular stencil matrix whose rows repeat a few value tuples (what the smoothed prolongators of a
    constant-coefficient problem look like): nodes sorted by class and evaluated from per-class tables, the rows of
    their own one wavefront each over the CSR entries; every fused mode, and against the plain CSR kernel."""
    import scipy.sparse as sp
    dims = (37, 33, 29)
    rng = np.random.default_rng(29)

    def t1(n, a, b):
        return sp.diags([np.full(n - 1, b), np.full(n, a), np.full(n - 1, b)], [-1, 0, 1])
    pattern = sp.kron(t1(dims[2], 2.0, -0.3), sp.kron(t1(dims[1], 1.5, -0.25), t1(dims[0], 1.0, -0.2))).tocsr()
    P = sp.vstack([sp.hstack([pattern, 0.5 * pattern])] * 6).tolil()      # 212 k rows x 71 k columns
    for r in rng.integers(0, P.shape[0], 30):                             # a few rows of their own
        P[r, P.rows[r][0]] = 7.5
    P = P.tocsr()
    Pd = M.SparseMatrixDevice(ctx, P)
    assert Pd.get_kernel()[1] == 5 and Pd.regular_rows()
    n_classes, listed = Pd.stencil_classes()
    assert n_classes >= 19 and 30 <= listed <= 400          # interior, faces, edges; corners and the 30 rows listed
    x = rng.random(P.shape[1])
    y0 = rng.random(P.shape[0])
    res = {}
    for on in (True, False):
        Pd.set_regular_rows(on)
        assert Pd.regular_rows() == on
        out = torch.empty(P.shape[0], dtype=torch.float64, device="cuda")
        Pd.vmult(out, dev(x))
        assert relerr(host(out, ctx), P @ x) < TOL
        res[on] = host(out, ctx).copy()
        Pd.residual(dev(x), dev(y0), out)
        assert relerr(host(out, ctx), P @ x - y0) < TOL
    assert relerr(res[True], res[False]) < 1e-13


def test_vector_kernels(ctx):
    rng = np.random.default_rng(9)
    for n in (1, 63, 64, 1000, 1 << 20):
        x, v = rng.random(n), rng.random(n)
        xd, vd = dev(x), dev(v)
        assert ctx.dot(xd, vd) == pytest.approx(float(x @ v), rel=1e-13)
        assert ctx.l2_norm(xd) == pytest.approx(float(np.linalg.norm(x)), rel=1e-13)
        ctx.add(xd, -1.0, vd)                         # res->add(-1., b), hierarchy.hpp:286
        np.testing.assert_array_equal(host(xd, ctx), x - v)
        ctx.sadd(xd, -1.0, 1.0, vd)                   # residual.sadd(-1., 1., rhs), tests/test_hierarchy.cc:104
        np.testing.assert_array_equal(host(xd, ctx), -(x - v) + v)
        ctx.set(xd, 0.0)                              # x = 0., hierarchy.hpp:258
        assert not host(xd, ctx).any()
    # deterministic reduction: same bits on repeat
    xd = dev(rng.random(1 << 20))
    assert ctx.dot(xd, xd) == ctx.dot(xd, xd)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-6), (torch.float64, 1e-14)])
@pytest.mark.parametrize("variant", ["valu", "mfma"])
@pytest.mark.parametrize("n_cells", [1, 15, 64, 1000, 70001])
def test_cell_contraction_valu_and_mfma(ctx, dtype, tol, variant, n_cells):
    """BASELINE.json configs[4]: the cell-local evaluation as a batched dense contraction, on the vector ALU and on the
    matrix cores (v_mfma_f32_16x16x4_f32 / v_mfma_f64_16x16x4_f64), against the oracle's cell matrix: the reference
    matrix K_ref of a Cartesian cell is what the oracle's operator applies per cell for a unit coefficient."""
    h = (0.25, 0.5, 0.125)       # anisotropic on purpose: K_ref is not symmetric under corner permutations
    mesh = O.StructuredMesh((1, 1, 1), length=1.0)
    mesh.h = h
    Ke = O.cell_matrices(mesh, np.ones((1, 8)))[0]            # 8 x 8, unit coefficient
    rng = np.random.default_rng(n_cells)
    U = rng.standard_normal((8, n_cells))                    # asymmetric data: a row/column swap would show
    c = 1.0 + rng.random(n_cells)
    ref = (Ke @ U) * c
    u = torch.from_numpy(U).to(dtype).cuda().contiguous()
    cc = torch.from_numpy(c).to(dtype).cuda().contiguous()
    v = torch.full((8, n_cells), float("nan"), dtype=dtype, device="cuda")
    ctx.cell_contraction(u, cc, v, h, variant=variant)
    ctx.synchronize()
    got = v.cpu().numpy().astype(float)
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() < tol * np.abs(ref).max() * 8


def test_direct_solvers_on_the_reference_tridiagonal_system(ctx):
    """/root/reference/tests/test_direct_solver_device.cu:23-110: the 30 x 30 matrix tridiag(-1, 4, -1), a solution drawn
    from N(10, 2), rhs = A x; CudaSolver with solver.type cholesky / lu_dense / lu_sparse_host must return x to 1e-12 %
    (here: the three names share the dense LU factored at construction); the iterative types on the same system."""
    n = 30
    A = sp.diags([-np.ones(n - 1), 4 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1]).tocsr()
    sol = 10.0 + 2.0 * np.random.default_rng(0).standard_normal(n)
    rhs = A @ sol
    Ad = M.SparseMatrixDevice(ctx, A)
    x = torch.empty(n, dtype=torch.float64, device="cuda")
    for solver in ("cholesky", "lu_dense", "lu_sparse_host"):
        x.fill_(123.0)                                           # the solvers start from zero by themselves
        Ad.solve({"solver": {"type": solver}}, dev(rhs), x)
        np.testing.assert_allclose(host(x, ctx), sol, rtol=1e-14)          # (BOOST_CHECK_CLOSE 1e-12 per cent)
    Ad.solve({"solver": {"type": "pcg", "n_iterations": 30}}, dev(rhs), x)  # CG on 30 unknowns is exact after 30 steps
    np.testing.assert_allclose(host(x, ctx), sol, rtol=1e-10)
    with pytest.raises(L.MfmgError, match="Unknown solver name"):           # cuda_solver.cu:71
        Ad.solve({"solver": {"type": "qr"}}, dev(rhs), x)
    with pytest.raises(L.MfmgNotImplementedError):                          # no AmgX shim
        Ad.solve({"solver": {"type": "amgx"}}, dev(rhs), x)


@pytest.mark.parametrize("n,material,stored_dinv", [((6, 5, 7), "constant", False), ((70, 30, 20), "constant", False), ((86, 19, 11), "cellwise", False),
                                                    ((130, 40, 33), "cellwise", True)])
@pytest.mark.parametrize("tile", [None, (8, 3, 5), (4, 3, 8)])
def test_smoother_sweep_from_zero_guess(ctx, n, material, stored_dinv, tile):
    """The three-term sweep from x_0 = 0 WITHOUT reading x_0 (x = None: the pre-smoother of a preconditioner application,
    include/mfmg/common/hierarchy.hpp:253-259; first term x_1 = beta_1 D^-1 b, no operator application) == the sweep run on a
    zeroed vector (values equal; only the sign of an exact zero may differ), with a wide and a narrow last chunk column, the
    diagonal derived in the kernel or stored; refused for two terms and for the reference arithmetic."""
    ctx.set_stored_diagonal(stored_dinv)
    try:
        prob = _cellwise_problem(n) if material == "cellwise" else M.LaplaceProblem(n, material, device="cuda")
        op = M.MatrixFreeLaplace(ctx, prob)
    finally:
        ctx.set_stored_diagonal(False)
    assert op.sweep_available(3)
    if tile is not None:
        op.set_sweep_tile(*tile)
    N = prob.n_dofs
    g = torch.Generator(device="cuda")
    g.manual_seed(77)
    b = torch.rand(N, dtype=torch.float64, device="cuda", generator=g)
    al = [c[0] for c in SWEEP_COEFS]
    be = [c[1] for c in SWEEP_COEFS]
    ref = torch.full_like(b, float("nan"))
    op.smoother_sweep(al, be, b, torch.zeros_like(b), ref)
    out = torch.full_like(b, float("nan"))
    op.smoother_sweep(al, be, b, None, out)
    ctx.synchronize()
    assert torch.isfinite(out).all() and torch.equal(out, ref)
    with pytest.raises(L.MfmgNotImplementedError):
        op.smoother_sweep(al[:2], be[:2], b, None, out)
    op.set_sweep_reference(True)
    with pytest.raises(L.MfmgNotImplementedError):
        op.smoother_sweep(al, be, b, None, out)
