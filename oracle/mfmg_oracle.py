"""CPU oracle for the mfmg V-cycle apply path -- TEST INFRASTRUCTURE ONLY.

This module is a numpy/scipy restatement of the algorithm of ORNL-CEES/mfmg's
two-level spectral-AMGe V-cycle on a *structured* Q1 hyper-cube, written from
the reference sources (cited per function as ``path:line`` relative to
``/root/reference``).  It is the checker for the HIP path: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  The product package ``mfmg_amd`` never does.

Pinning status: the reference's own translation units need deal.II, Trilinos
and Boost (absent here, no network), so the reference cannot be compiled
(``oracle/_ref`` does not exist).  The oracle is pinned by the reference's own
fixtures that are pure data (see ``tests/test_oracle_fixtures.py``):
tridiag Jacobi known answer (tests/test_smoother_device.cu:39-114), banded
30x39 apply/transpose/multiply (tests/test_sparse_matrix_device_operator.cu:31-133),
R(i,dof)=diag*eigvec (tests/test_restriction_matrix.cc:62-168), partition of
unity (tests/test_restriction_matrix.cc:293-354), matrix-free == assembled
(tests/test_hierarchy.cc:644-695) and the convergence-rate golds
(tests/test_hierarchy.cc:343-405, tests/test_hierarchy_device.cu:359-420).
The third-party arithmetic of deal.II's PreconditionChebyshev is restated from
its published algorithm (deal.II 9.1 ``lac/precondition.h``).

Conventions: domain [0,L]^dim, ``n[d]`` cells per
dimension, Q1, DoF id lexicographic ``i + Nx*(j + Ny*k)`` with ``N = n+1``;
cell id lexicographic; cell corner ``c = a + 2b + 4d`` (x fastest, the deal.II
vertex order); quadrature point ``q = qa + 2qb + 4qc`` (deal.II QGauss<dim>(2)
tensor order).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Optional, Sequence

import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp
import scipy.sparse.linalg as spla

# Two-point Gauss rule on [0,1] (dealii::QGauss<1>(fe_degree+1), fe_degree=1;
# tests/laplace_matrix_free.hpp:303)
_G0 = 0.5 - 0.5 / math.sqrt(3.0)
_G1 = 0.5 + 0.5 / math.sqrt(3.0)
GAUSS_PTS = np.array([_G0, _G1])


# ----------------------------------------------------------------------------
# RNG helpers: libstdc++ std::default_random_engine == std::minstd_rand0 and
# std::uniform_real_distribution<double>(0,1) == generate_canonical<double,53>
# (tests/test_hierarchy.cc:76-87, source/dealii/dealii_mesh_evaluator.cc:44-56)
# ----------------------------------------------------------------------------
class MinstdRand0:
    """std::minstd_rand0: x <- 16807 x mod (2^31-1), default seed 1."""

    M = 2147483647
    A = 16807

    def __init__(self, seed: int = 1):
        s = seed % self.M
        self.state = s if s != 0 else 1

    def next_u32(self) -> int:
        self.state = (self.A * self.state) % self.M
        return self.state

    def uniform01(self) -> float:
        """libstdc++ generate_canonical<double,53>: k = 2 draws, range R = M-1."""
        R = float(self.M - 1)  # max - min + 1 = 2147483646
        s = float(self.next_u32() - 1)
        s += float(self.next_u32() - 1) * R
        r = s / (R * R)
        if r >= 1.0:
            r = math.nextafter(1.0, 0.0)
        return r

    def uniform_int(self, lo: int, hi: int) -> int:
        """libstdc++ uniform_int_distribution<int>(lo,hi) for a generator whose
        range (M-2) is larger than the requested range: rejection downscaling
        (bits/uniform_int_dist.h)."""
        urng_range = self.M - 2  # max - min = 2147483646 - 1
        urange = hi - lo
        uerange = urange + 1
        scaling = urng_range // uerange
        past = uerange * scaling
        while True:
            ret = self.next_u32() - 1
            if ret < past:
                break
        return ret // scaling + lo


# ----------------------------------------------------------------------------
# Structured Q1 mesh (restates what tests/laplace_matrix_free.hpp:243-313 and
# tests/laplace.hpp:87-152 get from deal.II on GridGenerator::hyper_cube)
# ----------------------------------------------------------------------------
@dataclass
class StructuredMesh:
    n: tuple  # cells per dimension
    length: float = 1.0
    dirichlet: bool = True  # boundary id 1 everywhere, homogeneous Dirichlet

    def __post_init__(self):
        self.n = tuple(int(v) for v in self.n)
        self.dim = len(self.n)
        assert self.dim in (2, 3)
        self.N = tuple(v + 1 for v in self.n)
        self.h = tuple(self.length / v for v in self.n)
        self.n_dofs = int(np.prod(self.N))
        self.n_cells = int(np.prod(self.n))

    # -- numbering -------------------------------------------------------
    def dof_id(self, *ijk):
        idx = 0
        stride = 1
        for d in range(self.dim):
            idx = idx + ijk[d] * stride
            stride *= self.N[d]
        return idx

    def cell_dofs(self) -> np.ndarray:
        """int32 [n_cells][2^dim] cell->DoF index array (the array deal.II's
        cell->get_dof_indices fills, tests/laplace.hpp:196)."""
        rng = [np.arange(v) for v in self.n]
        grids = np.meshgrid(*rng, indexing="ij")  # [i][j][k]
        # lexicographic cell order: x fastest -> transpose to (k,j,i) raveling
        base = self.dof_id(*grids)
        base = np.transpose(base, tuple(reversed(range(self.dim)))).ravel()
        out = np.empty((self.n_cells, 2 ** self.dim), dtype=np.int64)
        for c in range(2 ** self.dim):
            off = 0
            stride = 1
            for d in range(self.dim):
                off += ((c >> d) & 1) * stride
                stride *= self.N[d]
            out[:, c] = base + off
        return out.astype(np.int32)

    def constrained_mask(self) -> np.ndarray:
        """bool [n_dofs]: DoFs on the boundary (interpolate_boundary_values with
        boundary id 1 on every boundary face, tests/laplace_matrix_free.hpp:255-291)."""
        m = np.zeros(self.N[::-1], dtype=bool)  # index [k][j][i]
        if self.dirichlet:
            for d in range(self.dim):
                ax = self.dim - 1 - d
                sl = [slice(None)] * self.dim
                sl[ax] = 0
                m[tuple(sl)] = True
                sl[ax] = -1
                m[tuple(sl)] = True
        return m.ravel()

    def quadrature_points(self) -> np.ndarray:
        """float64 [n_cells][2^dim][dim] physical coordinates of the Gauss points."""
        rng = [np.arange(v) for v in self.n]
        grids = np.meshgrid(*rng, indexing="ij")
        org = [np.transpose(g, tuple(reversed(range(self.dim)))).ravel() * self.h[d]
               for d, g in enumerate(grids)]
        nq = 2 ** self.dim
        pts = np.empty((self.n_cells, nq, self.dim))
        for q in range(nq):
            for d in range(self.dim):
                pts[:, q, d] = org[d] + GAUSS_PTS[(q >> d) & 1] * self.h[d]
        return pts


def material_property(kind: str, pts: np.ndarray) -> np.ndarray:
    """Coefficient at points [..., dim] (tests/test_hierarchy_helpers.hpp:75-188)."""
    dim = pts.shape[-1]
    if kind == "constant":
        return np.ones(pts.shape[:-1])
    if kind == "linear_x":
        return 1.0 + np.abs(pts[..., 0])
    if kind == "linear":
        val = np.ones(pts.shape[:-1])
        for d in range(dim):
            val = val + (1.0 + d) * np.abs(pts[..., d])
        return val
    if kind == "discontinuous":
        s = np.zeros(pts.shape[:-1], dtype=np.int64)
        for d in range(dim):
            s += np.floor(pts[..., d] * 100).astype(np.int64) % 2
        return np.where(s == dim, 100.0, 10.0)
    raise NotImplementedError(kind)  # ASSERT_THROW_NOT_IMPLEMENTED (:203)


def coefficient_table(mesh: StructuredMesh, kind: str = "constant") -> np.ndarray:
    """_coefficient(cell,q) of LaplaceOperator::evaluate_coefficient
    (tests/laplace_matrix_free.hpp:100-119)."""
    return material_property(kind, mesh.quadrature_points())


# ----------------------------------------------------------------------------
# Q1 reference element
# ----------------------------------------------------------------------------
def reference_gradients(dim: int) -> np.ndarray:
    """G[q, d, i] = d(phi_i)/d(xi_d) at Gauss point q on [0,1]^dim."""
    nq = 2 ** dim
    G = np.zeros((nq, dim, nq))
    for q in range(nq):
        xi = [GAUSS_PTS[(q >> d) & 1] for d in range(dim)]
        for i in range(nq):
            for d in range(dim):
                v = 1.0
                for e in range(dim):
                    bit = (i >> e) & 1
                    if e == d:
                        v *= 1.0 if bit else -1.0
                    else:
                        v *= xi[e] if bit else (1.0 - xi[e])
                G[q, d, i] = v
    return G


def geometry_factors(mesh: StructuredMesh) -> np.ndarray:
    """JxW * (1/h_d)^2 per direction for the Cartesian cell (J = diag(h))."""
    vol = float(np.prod(mesh.h))
    w = vol / (2 ** mesh.dim)  # Gauss weight 1/2 per direction
    return np.array([w / (mesh.h[d] ** 2) for d in range(mesh.dim)])


def cell_matrices(mesh: StructuredMesh, coef: np.ndarray) -> np.ndarray:
    """A_e[c,i,j] = sum_q coef(c,q) grad phi_i . grad phi_j JxW
    (tests/laplace.hpp:160-195; tests/test_hierarchy_helpers.hpp:262-283)."""
    G = reference_gradients(mesh.dim)
    f = geometry_factors(mesh)
    # K[q,i,j] = sum_d f_d G[q,d,i] G[q,d,j]
    K = np.einsum("d,qdi,qdj->qij", f, G, G)
    return np.einsum("cq,qij->cij", coef, K)


# ----------------------------------------------------------------------------
# Matrix-free operator  (tests/laplace_matrix_free.hpp:121-156 + deal.II
# MatrixFreeOperators::Base::vmult constrained-row semantics)
# ----------------------------------------------------------------------------
@dataclass
class MatrixFreeLaplace:
    mesh: StructuredMesh
    coef: np.ndarray
    cell_dofs: np.ndarray = None
    constrained: np.ndarray = None

    def __post_init__(self):
        if self.cell_dofs is None:
            self.cell_dofs = self.mesh.cell_dofs()
        if self.constrained is None:
            self.constrained = self.mesh.constrained_mask()
        self.G = reference_gradients(self.mesh.dim)
        self.f = geometry_factors(self.mesh)
        self.n = self.mesh.n_dofs

    def vmult(self, x: np.ndarray) -> np.ndarray:
        cd = self.cell_dofs.astype(np.int64)
        xr = np.where(self.constrained, 0.0, x)  # read_dof_values: constrained -> 0
        u = xr[cd]  # [c, i]
        grad = np.einsum("qdi,ci->cqd", self.G, u)  # evaluate(gradients)
        flux = grad * self.coef[:, :, None] * self.f[None, None, :]  # submit_gradient
        v = np.einsum("qdi,cqd->ci", self.G, flux)  # integrate
        y = np.zeros(self.n)
        np.add.at(y, cd.ravel(), v.ravel())  # distribute_local_to_global
        y[self.constrained] = x[self.constrained]  # Base::vmult: dst_c = src_c
        return y

    def diagonal(self) -> np.ndarray:
        """compute_diagonal (tests/laplace_matrix_free.hpp:75-98,158-199):
        per-cell unit-vector applies; constrained entries set to one."""
        cd = self.cell_dofs.astype(np.int64)
        K = np.einsum("d,qdi,qdi->qi", self.f, self.G, self.G)
        dloc = np.einsum("cq,qi->ci", self.coef, K)
        d = np.zeros(self.n)
        np.add.at(d, cd.ravel(), dloc.ravel())
        d[self.constrained] = 1.0
        return d

    def diagonal_inverse(self) -> np.ndarray:
        return 1.0 / self.diagonal()


# ----------------------------------------------------------------------------
# Assembled operator (tests/laplace.hpp:154-204) with
# AffineConstraints::distribute_local_to_global elimination of homogeneous
# Dirichlet rows/columns: off-diagonals dropped, the summed local diagonal kept.
# ----------------------------------------------------------------------------
def assemble_csr(mesh: StructuredMesh, coef: np.ndarray,
                 cell_dofs: Optional[np.ndarray] = None,
                 constrained: Optional[np.ndarray] = None) -> sp.csr_matrix:
    cd = (mesh.cell_dofs() if cell_dofs is None else cell_dofs).astype(np.int64)
    con = mesh.constrained_mask() if constrained is None else constrained
    Ae = cell_matrices(mesh, coef)
    nl = cd.shape[1]
    rows = np.repeat(cd, nl, axis=1).ravel()
    cols = np.tile(cd, (1, nl)).ravel()
    vals = Ae.reshape(len(cd), -1).ravel().copy()
    rc = con[rows]
    cc = con[cols]
    keep = ~(rc | cc) | (rows == cols)
    A = sp.coo_matrix((vals[keep], (rows[keep], cols[keep])),
                      shape=(mesh.n_dofs, mesh.n_dofs)).tocsr()
    A.sum_duplicates()
    A.sort_indices()
    return A


# ----------------------------------------------------------------------------
# CSR helpers mirroring SparseMatrixDevice / CudaMatrixOperator semantics
# (include/mfmg/cuda/sparse_matrix_device.templates.cuh:351-371,
#  source/cuda/cuda_matrix_operator.cu:80-149)
# ----------------------------------------------------------------------------
def csr_spmv(row_ptr, col, val, x):
    """Plain row-by-row CSR y = A x (what cusparseDcsrmv computes)."""
    n = len(row_ptr) - 1
    prod = val * x[col]
    y = np.add.reduceat(np.concatenate([prod, [0.0]]), row_ptr[:-1].astype(np.int64))
    empty = row_ptr[1:] == row_ptr[:-1]
    y = y[:n]
    y[empty] = 0.0
    return y


# ----------------------------------------------------------------------------
# Smoothers
# ----------------------------------------------------------------------------
def smoother_wrapper(apply_A: Callable, Binv: Callable, b, x):
    """x <- x - B^{-1}(A x - b)  (source/dealii/dealii_matrix_free_smoother.cc:63-76
    == source/dealii/dealii_smoother.cc:69-81 == source/cuda/cuda_smoother.cu:48-59)."""
    r = apply_A(x) - b
    tmp = Binv(r)
    return x - tmp


def jacobi_inverse_diagonal_csr(A: sp.csr_matrix) -> np.ndarray:
    """extract_inv_diag (source/cuda/cuda_smoother.cu:86-96)."""
    return 1.0 / A.diagonal()


def gauss_seidel_from_zero(A: sp.csr_matrix, r: np.ndarray) -> np.ndarray:
    """One forward SOR(omega=1) sweep from a zero guess = (D+L)^{-1} r
    (TrilinosWrappers::PreconditionSOR defaults, source/dealii/dealii_smoother.cc:45-50)."""
    L = sp.tril(A, format="csr")
    return spla.spsolve_triangular(L, r, lower=True)


def sgs_from_zero(A: sp.csr_matrix, r: np.ndarray) -> np.ndarray:
    """PreconditionSSOR(omega=1): (D+L)^{-1} D (D+U)^{-1} ... Ifpack SGS one sweep
    from zero: forward then backward sweep (source/dealii/dealii_smoother.cc:39-44)."""
    L = sp.tril(A, format="csr")
    U = sp.triu(A, format="csr")
    y = spla.spsolve_triangular(L, r, lower=True)
    # backward sweep on the residual-updated system: x = y + (D+U)^{-1}(r - A y)
    return y + spla.spsolve_triangular(U, r - A @ y, lower=False)


@dataclass
class ChebyshevParams:
    """Explicit parameters of the Chebyshev polynomial smoother.  theta/delta as
    in deal.II PreconditionChebyshev (lac/precondition.h, 9.1):
    theta = (lmax+lmin)/2, delta = (lmax-lmin)/2."""
    degree: int
    lambda_max: float
    lambda_min: float

    @property
    def theta(self):
        return 0.5 * (self.lambda_max + self.lambda_min)

    @property
    def delta(self):
        return 0.5 * (self.lambda_max - self.lambda_min)

    def step_coefficients(self):
        """[(alpha_k, beta_k)] so that x_{k+1} = x_k + alpha_k (x_k - x_{k-1})
        - beta_k D^{-1}(A x_k - b), k = 0..degree-1 (alpha_0 = 0)."""
        theta, delta = self.theta, self.delta
        out = [(0.0, 1.0 / theta)]
        if self.degree < 2 or abs(delta) < 1e-40:
            return out
        rhok = delta / theta
        sigma = theta / delta
        for _ in range(self.degree - 1):
            rhokp = 1.0 / (2.0 * sigma - rhok)
            out.append((rhokp * rhok, 2.0 * rhokp / delta))
            rhok = rhokp
        return out


def chebyshev_vmult(apply_A: Callable, dinv: np.ndarray, p: ChebyshevParams, src):
    """dst = P(A) src exactly in deal.II's update1/update2 form
    (PreconditionChebyshev::vmult; called at
    source/dealii/dealii_matrix_free_smoother.cc:74)."""
    coefs = p.step_coefficients()
    update1 = coefs[0][1] * (dinv * src)
    dst = update1.copy()
    for (f1, f2) in coefs[1:]:
        update2 = apply_A(dst) - src
        update1 = f1 * update1 - f2 * (dinv * update2)
        dst = dst + update1
    return dst


def chebyshev_smoother_apply(apply_A, dinv, p: ChebyshevParams, b, x):
    """mfmg wrapper + deal.II polynomial, reference order of operations."""
    return smoother_wrapper(apply_A, lambda r: chebyshev_vmult(apply_A, dinv, p, r), b, x)


def chebyshev_smoother_apply_fused(apply_A, dinv, p: ChebyshevParams, b, x):
    """Algebraically identical three-term form on x itself: the form the HIP
    kernels fuse (SURVEY.md section 8d 'fused Chebyshev/Jacobi step')."""
    xm = None
    for (alpha, beta) in p.step_coefficients():
        r = apply_A(x) - b
        if xm is None:
            xn = x - beta * (dinv * r)
        else:
            xn = x + alpha * (x - xm) - beta * (dinv * r)
        xm, x = x, xn
    return x


def dealii_chebyshev_initial_guess(n: int, first_local: int = 0) -> np.ndarray:
    """deal.II 9.1 PreconditionChebyshevImplementation::set_initial_guess for
    LinearAlgebra::distributed::Vector: (i + first) % 11, made mean-free."""
    v = ((np.arange(n) + first_local) % 11).astype(float)
    return v - v.mean()


def hashed_initial_guess(n: int) -> np.ndarray:
    """Numbering-independent start vector used by the HIP build by default (splitmix64 finaliser of
    the DoF id, mean-free): deal.II's (i % 11) pattern degenerates to a plane wave for lexicographic
    numberings with row lengths such as 128 or 256 and then under-estimates lambda_max."""
    with np.errstate(over="ignore"):
        z = np.arange(n, dtype=np.uint64) + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    v = (z >> np.uint64(11)).astype(float) / 9007199254740992.0
    return v - v.mean()


def dealii_chebyshev_eigen_estimate(apply_A, dinv, n, n_iter=8, residual=1e-2,
                                    constrained=None, start="dealii"):
    """Restates PreconditionChebyshev::estimate_eigenvalues (deal.II 9.1): CG on
    A with preconditioner D^{-1}, rhs = initial guess, at most eig_cg_n_iterations
    steps (or until |r| < eig_cg_residual*|rhs|), eigenvalues of the Lanczos
    tridiagonal; returns (min_est, 1.2*max_est)."""
    rhs = dealii_chebyshev_initial_guess(n) if start == "dealii" else hashed_initial_guess(n)
    if constrained is not None:
        rhs = np.where(constrained, 0.0, rhs)
    x = np.zeros(n)
    r = rhs - apply_A(x)
    z = dinv * r
    p = z.copy()
    rz = r @ z
    alphas, betas = [], []
    tol = max(residual * np.linalg.norm(rhs), 1e-300)
    for _ in range(n_iter):
        Ap = apply_A(p)
        alpha = rz / (p @ Ap)
        x = x + alpha * p
        r = r - alpha * Ap
        alphas.append(alpha)
        if np.linalg.norm(r) < tol:
            break
        z = dinv * r
        rz_new = r @ z
        beta = rz_new / rz
        betas.append(beta)
        rz = rz_new
        p = z + beta * p
    m = len(alphas)
    T = np.zeros((m, m))
    for i in range(m):
        T[i, i] = 1.0 / alphas[i] + (betas[i - 1] / alphas[i - 1] if i > 0 else 0.0)
        if i + 1 < m:
            T[i, i + 1] = T[i + 1, i] = math.sqrt(betas[i]) / alphas[i]
    ev = np.linalg.eigvalsh(T)
    return float(ev[0]), 1.2 * float(ev[-1])


def dealii_chebyshev_params(apply_A, dinv, n, degree=1, smoothing_range=0.0,
                            constrained=None, start="dealii") -> ChebyshevParams:
    """AdditionalData defaults of deal.II 9.1 (degree=1, smoothing_range=0,
    eig_cg_n_iterations=8, eig_cg_residual=1e-2) as consumed by
    source/dealii/dealii_matrix_free_smoother.cc:34-56."""
    mn, mx = dealii_chebyshev_eigen_estimate(apply_A, dinv, n, constrained=constrained, start=start)
    alpha = mx / smoothing_range if smoothing_range > 1.0 else min(0.9 * mx, mn)
    return ChebyshevParams(degree=degree, lambda_max=mx, lambda_min=alpha)


# ----------------------------------------------------------------------------
# AMGe restrictor on block agglomerates
# ----------------------------------------------------------------------------
def block_agglomerates(mesh: StructuredMesh, agg: Sequence[int]):
    """build_agglomerates_block (include/mfmg/common/amge.templates.hpp:412-499):
    boxes of agg[d] cells anchored at the origin, clipped at the far boundary.
    Returns list of (cell_lo, cell_hi) tuples in x-fastest order of agglomerates."""
    dim = mesh.dim
    counts = [-(-mesh.n[d] // agg[d]) for d in range(dim)]
    out = []
    rngs = [range(c) for c in counts]
    if dim == 2:
        order = [(i, j) for j in rngs[1] for i in rngs[0]]
    else:
        order = [(i, j, k) for k in rngs[2] for j in rngs[1] for i in rngs[0]]
    for a in order:
        lo = tuple(a[d] * agg[d] for d in range(dim))
        hi = tuple(min(lo[d] + agg[d], mesh.n[d]) for d in range(dim))
        out.append((lo, hi))
    return out, counts


def dealii_block_agglomerate_ids(mesh: StructuredMesh, agg: Sequence[int]) -> np.ndarray:
    """Agglomerate id of every cell as AMGe::build_agglomerates_block leaves it in the cells' user indices
    (include/mfmg/common/amge.templates.hpp:412-499): cells in deal.II's iteration order (refine_global of a
    hyper_cube: Morton order, x fastest inside a parent), agglomerates numbered from 1 in the order in which the
    traversal first meets them.  Requires 2^r cells per direction.  Pinned by the literal arrays of
    tests/test_agglomerate.cc:69-120,122-230 (one rank)."""
    dim = mesh.dim
    n = mesh.n[0]
    assert all(v == n for v in mesh.n) and (n & (n - 1)) == 0
    levels = int(math.log2(n))
    seen = {}
    out = []
    for m in range(n ** dim):
        c = [0] * dim
        for l in range(levels):
            for d in range(dim):
                c[d] |= ((m >> (dim * l + d)) & 1) << l
        key = tuple(c[d] // agg[d] for d in range(dim))
        if key not in seen:
            seen[key] = len(seen) + 1
        out.append(seen[key])
    return np.array(out, dtype=np.int64)


def agglomerate_local(mesh: StructuredMesh, lo, hi):
    """Local lexicographic DoF numbering of one agglomerate patch: returns
    (global dof ids [nloc], local cell->local-dof [ncell_loc, 2^dim],
    global cell ids [ncell_loc])."""
    dim = mesh.dim
    ln = tuple(hi[d] - lo[d] for d in range(dim))
    lN = tuple(v + 1 for v in ln)
    # local dof (a,b,c) -> global
    grids = np.meshgrid(*[np.arange(v) for v in lN], indexing="ij")
    gl = mesh.dof_id(*[grids[d] + lo[d] for d in range(dim)])
    gl = np.transpose(gl, tuple(reversed(range(dim)))).ravel()
    sub = StructuredMesh(ln, length=1.0)
    lcd = sub.cell_dofs()
    cg = np.meshgrid(*[np.arange(v) for v in ln], indexing="ij")
    cid = 0
    stride = 1
    for d in range(dim):
        cid = cid + (cg[d] + lo[d]) * stride
        stride *= mesh.n[d]
    cid = np.transpose(cid, tuple(reversed(range(dim)))).ravel()
    return gl.astype(np.int64), lcd.astype(np.int64), cid.astype(np.int64), lN


def first_touch_numbering(lcd: np.ndarray, n_loc: int) -> np.ndarray:
    """deal.II DoF numbering of a patch triangulation
    (GridTools::build_triangulation_from_patch + DoFHandler::distribute_dofs for
    FE_Q(1)): DoFs numbered at first touch walking cells in order, vertices in
    lexicographic order.  Returns perm with perm[lexicographic_local] = dealii_local."""
    perm = -np.ones(n_loc, dtype=np.int64)
    nxt = 0
    for c in range(lcd.shape[0]):
        for v in range(lcd.shape[1]):
            d = lcd[c, v]
            if perm[d] < 0:
                perm[d] = nxt
                nxt += 1
    return perm


def _select_eigenvectors(w, V, n_eig, mode, v0=None, rtol=1e-9):
    """mode 'lapack': first n_eig columns as returned by the dense solver
    (include/mfmg/cuda/amge_device.templates.cuh:256-310;
     include/mfmg/dealii/amge_host.templates.hpp:446-470).
    mode 'krylov': what a single-vector Krylov method (ARPACK / non-deflated
    Lanczos, amge_host.templates.hpp:407-439, lanczos.templates.hpp:83-140)
    returns: one vector per *distinct* eigenvalue, the normalised projection of
    the start vector onto that eigenspace."""
    if mode == "lapack":
        return w[:n_eig].copy(), V[:, :n_eig].copy()
    assert v0 is not None
    scale = max(abs(w[-1]), 1e-300)
    vals, vecs = [], []
    i = 0
    n = len(w)
    while i < n and len(vals) < n_eig:
        j = i + 1
        while j < n and abs(w[j] - w[i]) <= rtol * scale:
            j += 1
        P = V[:, i:j]
        comp = P @ (P.T @ v0)
        nrm = np.linalg.norm(comp)
        if nrm > 1e-12 * np.linalg.norm(v0):
            vals.append(w[i:j].mean())
            vecs.append(comp / nrm)
        i = j
    return np.array(vals), np.stack(vecs, axis=1)


@dataclass
class Restrictor:
    """R in both layouts: CSR (the reference's TrilinosWrappers::SparseMatrix /
    SparseMatrixDevice) and agglomerate-blocked (dof ids + values per eigenvector)."""
    csr: sp.csr_matrix
    agg_dofs: list  # per agglomerate: int64[nloc]
    agg_vals: list  # per agglomerate: float64[n_eig][nloc]
    eigenvalues: list


def build_restrictor(mesh: StructuredMesh, coef: np.ndarray, global_diag: np.ndarray,
                     agg=(2, 2, 2), n_eig=2, variant="host",
                     eig_mode="lapack", constrained=None,
                     initial_guess: str = "dealii") -> Restrictor:
    """AMGe restriction matrix.

    variant:
      'host'   assembled agglomerate matrix, shifted by the mean diagonal, with
               constrained diagonals := 200 (include/mfmg/dealii/amge_host.templates.hpp:378-394),
               weights diag_loc/diag_glob (include/mfmg/common/amge.templates.hpp:300-321)
      'device' unshifted dense eigenproblem, B = I
               (include/mfmg/cuda/amge_device.templates.cuh:256-310)
      'mf'     matrix-free agglomerate operator [A_ff 0; 0 I], local diagonal with
               constrained entries 1 (amge_host.templates.hpp:278-350,
               tests/test_hierarchy_helpers.hpp:344-361)
    """
    dim = mesh.dim
    con = mesh.constrained_mask() if constrained is None else constrained
    Ae = cell_matrices(mesh, coef)
    aggs, _ = block_agglomerates(mesh, agg[:dim])
    rows, cols, vals = [], [], []
    agg_dofs, agg_vals, eigvals = [], [], []
    row = 0
    for (lo, hi) in aggs:
        gl, lcd, cid, lN = agglomerate_local(mesh, lo, hi)
        nloc = len(gl)
        A = np.zeros((nloc, nloc))
        for c in range(lcd.shape[0]):
            A[np.ix_(lcd[c], lcd[c])] += Ae[cid[c]]
        lcon = con[gl]
        full_diag = A.diagonal().copy()
        # eliminate constrained rows/cols, keep the summed local diagonal
        A[lcon, :] = 0.0
        A[:, lcon] = 0.0
        if variant == "mf":
            A[lcon, lcon] = 1.0
            diag_loc = A.diagonal().copy()
        else:
            A[lcon, lcon] = full_diag[lcon]
            diag_loc = A.diagonal().copy()
        M = A.copy()
        if variant == "host":
            avg = diag_loc.sum() / nloc
            M[np.diag_indices(nloc)] += avg
            M[lcon, lcon] = 200.0
        # start vector of the Krylov eigensolvers
        # (DealIIMeshEvaluator::set_initial_guess, source/dealii/dealii_mesh_evaluator.cc:44-56)
        v0 = None
        if eig_mode == "krylov":
            gen = MinstdRand0()
            if initial_guess == "dealii":
                perm = first_touch_numbering(lcd, nloc)
            else:
                perm = np.arange(nloc)
            inv = np.argsort(perm)  # dealii local id -> lexicographic local id
            v0 = np.zeros(nloc)
            for t in range(nloc):
                li = inv[t]
                v0[li] = 0.0 if lcon[li] else gen.uniform01()
        if variant == "mf":
            free = ~lcon
            w, Vf = sla.eigh(M[np.ix_(free, free)])
            V = np.zeros((nloc, Vf.shape[1]))
            V[free, :] = Vf
        else:
            w, V = sla.eigh(M)
        ne = min(n_eig, V.shape[1])
        wsel, Vsel = _select_eigenvectors(w, V, ne, eig_mode, v0)
        if variant == "host":
            wsel = wsel - avg
        vv = np.empty((Vsel.shape[1], nloc))
        for k in range(Vsel.shape[1]):
            vec = Vsel[:, k]
            wts = diag_loc / global_diag[gl] * vec
            rows.append(np.full(nloc, row))
            cols.append(gl)
            vals.append(wts)
            vv[k] = wts
            row += 1
        agg_dofs.append(gl)
        agg_vals.append(vv)
        eigvals.append(wsel)
    R = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                      shape=(row, mesh.n_dofs)).tocsr()
    R.sort_indices()
    return Restrictor(R, agg_dofs, agg_vals, eigvals)


def restriction_from_eigenvectors(eigenvectors, diag_elements, dof_indices_maps,
                                  n_local_eigenvectors, global_diag, n_cols):
    """AMGe::compute_restriction_sparse_matrix (include/mfmg/common/amge.templates.hpp:271-325)."""
    rows, cols, vals = [], [], []
    pos = 0
    for i, n_local in enumerate(n_local_eigenvectors):
        for _ in range(n_local):
            ev = eigenvectors[pos]
            for j in range(len(ev)):
                g = dof_indices_maps[i][j]
                rows.append(pos)
                cols.append(g)
                vals.append(diag_elements[i][j] / global_diag[g] * ev[j])
            pos += 1
    return sp.coo_matrix((vals, (rows, cols)), shape=(pos, n_cols)).tocsr()


def galerkin_coarse_matrix(apply_A: Callable, R: sp.csr_matrix, A: Optional[sp.csr_matrix] = None):
    """A_c = R (A R^T) (include/mfmg/common/hierarchy.hpp:214-233).  With an
    assembled A this is two SpGEMMs; matrix-free it is the column-by-column
    product of include/mfmg/dealii/dealii_utils.hpp:32-81."""
    if A is not None:
        return (R @ (A @ R.T)).tocsr()
    n_c = R.shape[0]
    Rt = R.T.tocsc()
    cols = []
    for j in range(n_c):
        e = np.asarray(Rt[:, j].todense()).ravel()
        cols.append(apply_A(e))
    AP = np.stack(cols, axis=1)
    return sp.csr_matrix(R @ AP)


# ----------------------------------------------------------------------------
# Coarse solvers
# ----------------------------------------------------------------------------
def direct_coarse_solver(Ac: sp.csr_matrix):
    """lu_dense / Amesos direct (source/cuda/cuda_solver.cu:51-72,
    source/dealii/dealii_solver.cc:43-47)."""
    lu = sla.lu_factor(Ac.toarray())
    return lambda b: sla.lu_solve(lu, b)


def pcg_coarse_solver(Ac: sp.csr_matrix, n_iter: int):
    """Build-defined scalable coarse 'solve': exactly n_iter steps of
    Jacobi-preconditioned CG from a zero guess (replaces ML/AMGx,
    SURVEY.md section 8f rank 3)."""
    dinv = 1.0 / Ac.diagonal()

    def solve(b):
        x = np.zeros_like(b)
        r = b.copy()
        z = dinv * r
        p = z.copy()
        rz = r @ z
        for _ in range(n_iter):
            if rz == 0.0:
                break
            Ap = Ac @ p
            alpha = rz / (p @ Ap)
            x = x + alpha * p
            r = r - alpha * Ap
            z = dinv * r
            rz_new = r @ z
            beta = rz_new / rz
            rz = rz_new
            p = z + beta * p
        return x

    return solve


def amg_coarse_solver(levels, n_cycles: int = 1, pre_smoothing_levels: Optional[int] = None):
    """Build-defined multilevel coarse 'solve' (the role ML / AMGx play upstream,
    source/dealii/dealii_solver.cc:48-66, source/cuda/cuda_solver.cu:204-445): `n_cycles` V-cycles from a
    zero guess over a given aggregation hierarchy.  `levels` = [(A_l, P_l or None, (degree, lmin, lmax)
    or None)], the last level is solved by dense LU.  Every level runs the recursion of
    Hierarchy::apply (hierarchy.hpp:246-309) with restrictor P_l^T and a Chebyshev smoother.
    `pre_smoothing_levels` (None: all): levels from that index on skip the pre-smoother -- from the zero guess
    the residual is -b, so b is restricted and the correction added: a V(0,1) cycle."""
    data = []
    for (A, P, cheb) in levels:
        A = A.tocsr()
        if P is None:
            data.append((A, None, None, sla.lu_factor(A.toarray())))
        else:
            dinv = 1.0 / A.diagonal()
            data.append((A, P.tocsr(), (dinv, ChebyshevParams(cheb[0], cheb[2], cheb[1])), None))

    def cycle(l, b):
        A, P, sm, lu = data[l]
        if P is None:
            return sla.lu_solve(lu, b)
        dinv, p = sm
        if pre_smoothing_levels is not None and l >= pre_smoothing_levels:
            x = P @ cycle(l + 1, P.T @ b)
            return chebyshev_smoother_apply(lambda z: A @ z, dinv, p, b, x)
        x = np.zeros_like(b)
        x = chebyshev_smoother_apply(lambda z: A @ z, dinv, p, b, x)
        res = A @ x - b
        xc = cycle(l + 1, P.T @ res)
        x = x - P @ xc
        return chebyshev_smoother_apply(lambda z: A @ z, dinv, p, b, x)

    def solve(b):
        x = cycle(0, b)
        A0 = data[0][0]
        for _ in range(n_cycles - 1):
            x = x - cycle(0, A0 @ x - b)
        return x

    return solve


# ----------------------------------------------------------------------------
# Hierarchy::apply  (include/mfmg/common/hierarchy.hpp:246-309), two levels
# ----------------------------------------------------------------------------
@dataclass
class TwoLevelHierarchy:
    apply_A: Callable
    smoother: Callable  # (b, x) -> x
    R: sp.csr_matrix
    coarse_solve: Callable
    n_smoothing_steps: int = 1
    is_preconditioner: bool = True

    def apply(self, b, x):
        if self.is_preconditioner:
            x = np.zeros_like(x)
        for _ in range(self.n_smoothing_steps):
            x = self.smoother(b, x)
        res = self.apply_A(x) - b  # negative residual
        b_c = self.R @ res
        x_c = self.coarse_solve(b_c)  # coarse level starts from x_c = 0
        x = x - self.R.T @ x_c
        for _ in range(self.n_smoothing_steps):
            x = self.smoother(b, x)
        return x


def vcycle_history(h: TwoLevelHierarchy, apply_A_monitor, b, x0, n_cycles=20):
    """The 20-cycle harness of tests/test_hierarchy.cc:95-123: returns
    (res[0..n], conv_rate = res[n]/res[n-1])."""
    x = x0.copy()
    r0 = np.linalg.norm(b - apply_A_monitor(x))
    res = [1.0]
    for _ in range(n_cycles):
        x = h.apply(b, x)
        res.append(np.linalg.norm(b - apply_A_monitor(x)) / r0)
    return np.array(res), res[-1] / res[-2], x


def pcg_solve(apply_A, precondition, b, x0, tolerance=1e-6, max_iterations=None):
    """The outer solve of tests/hierarchy_driver.cc:103-116: dealii::SolverCG (third party, restated: standard
    preconditioned CG) with SolverControl(max_iterations, tolerance) on the absolute l2 norm of the residual and
    `precondition(r)` = Hierarchy::vmult.  Returns (x, [||r_0||, ||r_1||, ...])."""
    x = x0.copy()
    r = b - apply_A(x)
    hist = [np.linalg.norm(r)]
    if max_iterations is None:
        max_iterations = b.size
    rz = 0.0
    p = None
    it = 0
    while hist[-1] > tolerance and it < max_iterations:
        z = precondition(r)
        rz_new = r @ z
        p = z.copy() if it == 0 else z + (rz_new / rz) * p
        rz = rz_new
        ap = apply_A(p)
        alpha = rz / (p @ ap)
        x = x + alpha * p
        r = r - alpha * ap
        hist.append(np.linalg.norm(r))
        it += 1
    return x, np.array(hist)


def random_initial_guess(n: int, constrained: Optional[np.ndarray], order=None,
                         zero_constrained=True) -> np.ndarray:
    """x0 of tests/test_hierarchy.cc:76-87 (CPU: constrained entries 0 and the
    generator is not advanced for them) or tests/test_hierarchy_device.cu:304-305
    (all entries random).  `order[t]` = DoF id that receives the t-th draw."""
    gen = MinstdRand0()
    x = np.zeros(n)
    idx = range(n) if order is None else order
    for i in idx:
        if zero_constrained and constrained is not None and constrained[i]:
            x[i] = 0.0
        else:
            x[i] = gen.uniform01()
    return x


def dealii_global_numbering(mesh: StructuredMesh) -> np.ndarray:
    """deal.II DoF ids of GridGenerator::hyper_cube + refine_global (cells in
    Morton order, vertex DoFs numbered at first touch).  Returns
    dealii_id[lexicographic_id]; requires 2^r cells per direction."""
    dim = mesh.dim
    n = mesh.n[0]
    assert all(v == n for v in mesh.n) and (n & (n - 1)) == 0
    levels = int(math.log2(n))
    cd = mesh.cell_dofs().astype(np.int64)

    def morton_cells():
        coords = [tuple([0] * dim)]
        for _ in range(levels):
            nxt = []
            for c in coords:
                for ch in range(2 ** dim):
                    nxt.append(tuple(2 * c[d] + ((ch >> d) & 1) for d in range(dim)))
            coords = nxt
        return coords

    perm = -np.ones(mesh.n_dofs, dtype=np.int64)
    nxt = 0
    for c in morton_cells():
        cid = 0
        stride = 1
        for d in range(dim):
            cid += c[d] * stride
            stride *= mesh.n[d]
        for v in range(2 ** dim):
            g = cd[cid, v]
            if perm[g] < 0:
                perm[g] = nxt
                nxt += 1
    return perm
