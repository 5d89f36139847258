"""Driver-side description of the 3-D / 2-D Laplace model problem on a structured hyper-cube.

What the reference obtains from deal.II in tests/laplace_matrix_free.hpp:243-313
(hyper_cube + refine_global, FE_Q(1), boundary id 1 with homogeneous Dirichlet values,
QGauss(2) coefficient table) is generated here directly as the plain arrays of
``mfmg_hip_mesh_desc``: cell->DoF indices, ``_coefficient(cell, q)`` and the constrained
DoF mask.  Arrays are torch tensors (CPU or the GPU the hierarchy runs on)."""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional, Sequence

import torch

from . import lib as _lib

_G = (0.5 - 0.5 / math.sqrt(3.0), 0.5 + 0.5 / math.sqrt(3.0))


def material_property(kind: str, pts: torch.Tensor) -> torch.Tensor:
    """Coefficient functions of tests/test_hierarchy_helpers.hpp:75-188 at points [..., dim]."""
    dim = pts.shape[-1]
    if kind == "constant":
        return torch.ones(pts.shape[:-1], dtype=torch.float64, device=pts.device)
    if kind == "linear_x":
        return 1.0 + pts[..., 0].abs()
    if kind == "linear":
        val = torch.ones(pts.shape[:-1], dtype=torch.float64, device=pts.device)
        for d in range(dim):
            val = val + (1.0 + d) * pts[..., d].abs()
        return val
    if kind == "discontinuous":
        s = torch.zeros(pts.shape[:-1], dtype=torch.int64, device=pts.device)
        for d in range(dim):
            s += torch.floor(pts[..., d] * 100).to(torch.int64) % 2
        ten = torch.full(pts.shape[:-1], 10.0, dtype=torch.float64, device=pts.device)
        return torch.where(s == dim, 10.0 * ten, ten)
    raise NotImplementedError(f"material property '{kind}'")


class LaplaceProblem:
    """Mesh arrays of the Q1 Laplace problem on [0, length]^dim with n cells per direction."""

    def __init__(self, n_cells: Sequence[int], material: str = "constant", length: float = 1.0,
                 device: str | torch.device = "cpu", dof_numbering: Optional[torch.Tensor] = None,
                 dirichlet: bool = True, cell_size: Optional[Sequence[float]] = None,
                 cell_offset: Optional[Sequence[int]] = None):
        """`cell_size` / `cell_offset` describe a sub-box of a larger mesh (the slab of one rank, first
        cell = global cell `cell_offset`): the coefficient is evaluated at the true coordinates, with
        the same arithmetic as on the global mesh."""
        self.n = tuple(int(v) for v in n_cells)
        self.dim = len(self.n)
        assert self.dim in (2, 3)
        self.N = tuple(v + 1 for v in self.n)
        self.h = tuple(cell_size) if cell_size is not None else tuple(length / v for v in self.n)
        self.cell_offset = tuple(int(v) for v in cell_offset) if cell_offset is not None else (0,) * self.dim
        self.device = torch.device(device)
        self.n_dofs = math.prod(self.N)
        self.n_cells_total = math.prod(self.n)
        dev = self.device
        nc = 2 ** self.dim
        # lexicographic node id of the lowest corner of every cell, cells x-fastest
        idx = [torch.arange(v, device=dev, dtype=torch.int64) for v in self.n]
        if self.dim == 3:
            k, j, i = torch.meshgrid(idx[2], idx[1], idx[0], indexing="ij")
            base = (i + self.N[0] * (j + self.N[1] * k)).reshape(-1)
            org = [i.reshape(-1), j.reshape(-1), k.reshape(-1)]
        else:
            j, i = torch.meshgrid(idx[1], idx[0], indexing="ij")
            base = (i + self.N[0] * j).reshape(-1)
            org = [i.reshape(-1), j.reshape(-1)]
        strides = [1, self.N[0], self.N[0] * self.N[1]]
        cd = torch.empty((self.n_cells_total, nc), dtype=torch.int64, device=dev)
        for m in range(nc):
            off = sum(((m >> d) & 1) * strides[d] for d in range(self.dim))
            cd[:, m] = base + off
        # constrained nodes: the whole boundary (boundary id 1 everywhere)
        con = torch.zeros(self.N[::-1], dtype=torch.bool, device=dev)
        if dirichlet:
            for d in range(self.dim):
                ax = self.dim - 1 - d
                sl = [slice(None)] * self.dim
                sl[ax] = 0
                con[tuple(sl)] = True
                sl[ax] = -1
                con[tuple(sl)] = True
        con = con.reshape(-1)
        # optional renumbering of the DoFs (node -> DoF id), e.g. to mimic DoFRenumbering
        self.node_to_dof = None
        if dof_numbering is not None:
            perm = dof_numbering.to(dev).to(torch.int64)
            assert perm.numel() == self.n_dofs
            cd = perm[cd]
            con_d = torch.zeros_like(con)
            con_d[perm] = con
            con = con_d
            self.node_to_dof = perm
        self.cell_dofs = cd.to(torch.int32).contiguous()
        self.constrained = con.to(torch.uint8).contiguous()
        # coefficient at the Gauss points of every cell
        pts = torch.empty((self.n_cells_total, nc, self.dim), dtype=torch.float64, device=dev)
        for q in range(nc):
            for d in range(self.dim):
                pts[:, q, d] = ((org[d] + self.cell_offset[d]).to(torch.float64) + _G[(q >> d) & 1]) * self.h[d]
        self.coefficient = material_property(material, pts).to(torch.float64).contiguous()
        self.material = material

    def mesh_desc(self) -> _lib.MeshDesc:
        d = _lib.MeshDesc()
        d.dim = self.dim
        for k in range(3):
            d.n_cells[k] = self.n[k] if k < self.dim else 0
            d.cell_size[k] = self.h[k] if k < self.dim else 0.0
        d.n_dofs = self.n_dofs
        d.cell_dofs = self.cell_dofs.data_ptr()
        d.coefficient = self.coefficient.data_ptr()
        d.constrained = self.constrained.data_ptr()
        d.arrays_on_device = 1 if self.device.type == "cuda" else 0
        return d

    def to(self, device) -> "LaplaceProblem":
        import copy
        other = copy.copy(self)
        other.device = torch.device(device)
        other.cell_dofs = self.cell_dofs.to(device)
        other.coefficient = self.coefficient.to(device)
        other.constrained = self.constrained.to(device)
        return other

    def random_initial_guess(self, seed: int = 1, zero_constrained: bool = True) -> torch.Tensor:
        """x0 ~ U(0,1) on the free DoFs in DoF-id order (tests/test_hierarchy.cc:76-87); the
        engine is std::minstd_rand0 so that host drivers of the reference can reproduce it."""
        import numpy as np
        n = self.n_dofs
        con = self.constrained.cpu().numpy().astype(bool)
        out = np.zeros(n)
        state = seed % 2147483647 or 1
        R = 2147483646.0
        for g in range(n):
            if zero_constrained and con[g]:
                continue
            state = (16807 * state) % 2147483647
            s = float(state - 1)
            state = (16807 * state) % 2147483647
            s += float(state - 1) * R
            r = s / (R * R)
            out[g] = r if r < 1.0 else math.nextafter(1.0, 0.0)
        return torch.from_numpy(out)
