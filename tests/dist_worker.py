"""Worker of the distributed tests: run with torch.distributed.run, backend gloo.

  --mode cpu : no GPU.  Every rank rebuilds its part of the operators with the product's HOST setup on its
               local extended slab and applies them with the oracle's numpy kernels + a gloo halo exchange;
               the owned parts must reproduce the global operators (partition correct by construction).
  --mode gpu : the library path on cuda:0 (all ranks share the GPU, layers staged through the host): fine
               operator / smoother / residual / restriction / prolongation / coarse operator / coarse solve against
               the single-process global hierarchy, then the 20-cycle residual history of the distributed V-cycle
               against the single-process cycle AND against the oracle's restatement of it, to 1e-10 relative."""
import argparse
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)

import mfmg_amd as M  # noqa: E402
from mfmg_amd import lib as L  # noqa: E402
import mfmg_oracle as O  # noqa: E402

PRM = {"eigensolver": {"number of eigenvectors": 2}, "agglomeration": {"nx": 2, "ny": 2, "nz": 2}}


def exchange_np(v, part, layer, owned_begin, owned_count):
    """Refresh one ghost layer on each side of a numpy vector (same protocol as HaloTransport)."""
    rank, n = part.rank, part.n_ranks
    ops, bufs = [], {}
    if rank > 0:
        s = torch.from_numpy(v[owned_begin * layer:(owned_begin + 1) * layer].copy())
        r = torch.empty(layer, dtype=torch.float64)
        ops += [dist.P2POp(dist.isend, s, rank - 1), dist.P2POp(dist.irecv, r, rank - 1)]
        bufs["low"] = r
    if rank + 1 < n:
        s = torch.from_numpy(v[(owned_begin + owned_count - 1) * layer:(owned_begin + owned_count) * layer].copy())
        r = torch.empty(layer, dtype=torch.float64)
        ops += [dist.P2POp(dist.isend, s, rank + 1), dist.P2POp(dist.irecv, r, rank + 1)]
        bufs["high"] = r
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    if "low" in bufs:
        v[(owned_begin - 1) * layer: owned_begin * layer] = bufs["low"].numpy()
    if "high" in bufs:
        v[(owned_begin + owned_count) * layer:(owned_begin + owned_count + 1) * layer] = bufs["high"].numpy()


def _all_reduce_cpu(t):
    """Sum of a CPU tensor over the ranks (through the GPU when the process group is nccl)."""
    if dist.get_backend() == "nccl":
        g = t.cuda()
        dist.all_reduce(g)
        t.copy_(g.cpu())
    else:
        dist.all_reduce(t)
    return t


def gather_owned(local, part, n_global, sl_local, sl_global):
    """All ranks end up with the global vector assembled from the owned parts."""
    out = torch.zeros(n_global, dtype=torch.float64)
    out[sl_global] = torch.from_numpy(np.ascontiguousarray(local[sl_local]))
    return _all_reduce_cpu(out).numpy()


def mode_cpu(args):
    rank, world = dist.get_rank(), dist.get_world_size()
    cells = (8, 6, 4 * world)
    material = "linear"
    part = M.SlabPartition(cells, rank, world, length=tuple(c / 8.0 for c in cells))
    # ---- global reference (every rank computes it; tiny)
    mesh = O.StructuredMesh(cells)
    mesh.h = part.h
    gprob = M.LaplaceProblem(cells, material, cell_size=part.h)
    coef_g = gprob.coefficient.numpy()
    mf_g = O.MatrixFreeLaplace(mesh, coef_g)
    Rg = M.host_build_restrictor(gprob, PRM, matrix_free=True)
    Acg = M.host_galerkin(gprob, Rg, "matrix_free")
    rng = np.random.default_rng(0)
    xg = rng.random(mesh.n_dofs)
    # ---- local slab
    lprob = part.local_problem(material)
    lmesh = O.StructuredMesh(part.local_cells)
    lmesh.h = part.h
    flags = lprob.constrained.numpy()
    mf_l = O.MatrixFreeLaplace(lmesh, lprob.coefficient.numpy(), constrained=(flags == 1))
    # coefficient table of the slab == slice of the global table
    nxy = cells[0] * cells[1]
    np.testing.assert_array_equal(lprob.coefficient.numpy(), coef_g[part.z_offset * nxy:(part.z_offset + part.local_cells[2]) * nxy])
    xl = part.local_from_global(torch.from_numpy(xg)).numpy()
    # garbage in the ghosts, then the exchange must restore them
    xl_g = xl.copy()
    lo, cnt = part.owned_plane_begin, part.owned_plane_count
    xl_g[:lo * part.plane] = -7.0
    xl_g[(lo + cnt) * part.plane:] = -7.0
    exchange_np(xl_g, part, part.plane, lo, cnt)
    yl = mf_l.vmult(xl_g)
    y = gather_owned(yl, part, mesh.n_dofs, part.owned_slice(), part.global_slice())
    np.testing.assert_allclose(y, mf_g.vmult(xg), rtol=1e-13, atol=1e-14)
    # ---- restrictor / coarse operator of the local slab against the global ones
    Rl = M.host_build_restrictor(lprob, PRM, matrix_free=True)
    per_layer = (cells[0] // 2) * (cells[1] // 2) * 2
    c_lo, c_cnt = part.ghost_low // 2, (part.z1 - part.z0) // 2
    c_glob0 = part.z0 // 2
    # restriction of owned agglomerates needs the ghost plane above: same rows as the global R
    rl = Rl @ xl                      # xl has correct ghosts
    own_c = slice(c_lo * per_layer, (c_lo + c_cnt) * per_layer)
    glob_c = slice(c_glob0 * per_layer, (c_glob0 + c_cnt) * per_layer)
    np.testing.assert_allclose(rl[own_c], (Rg @ xg)[glob_c], rtol=1e-12, atol=1e-14)
    # prolongation at owned nodes needs the coarse ghost layer below
    xc_g = rng.random(Rg.shape[0])
    lo_c_glob = (c_glob0 - c_lo) * per_layer
    xc_l = xc_g[lo_c_glob: lo_c_glob + Rl.shape[0]].copy()
    pl = Rl.T @ xc_l
    p = gather_owned(pl, part, mesh.n_dofs, part.owned_slice(), part.global_slice())
    np.testing.assert_allclose(p, Rg.T @ xc_g, rtol=1e-12, atol=1e-14)
    # coarse operator: owned rows of the local Galerkin product == global rows (columns shifted)
    Acl = M.host_galerkin(lprob, Rl, "matrix_free")
    al = Acl @ xc_l
    np.testing.assert_allclose(al[own_c], (Acg @ xc_g)[glob_c], rtol=1e-11, atol=1e-14)
    if rank == 0:
        print("cpu distributed checks passed", flush=True)


def parse_grid(text, world):
    if not text:
        return (1, 1, world)
    g = tuple(int(v) for v in text.split("x"))
    assert len(g) == 3 and g[0] * g[1] * g[2] == world, (text, world)
    return g


def exchange_box_np(v, part, dims, own0, own_n, comps=1, width=1, reverse=False):
    """The box exchange of common.hpp (HipHandle::exchange_box) restated in numpy + gloo: one message to and from every
    existing neighbour at an offset o in {-1, 0, 1}^3 (faces, edges, corners), all at once.  Along an axis with o_d != 0 the
    message spans `width` layers -- the owned ones next to that neighbour on the owner's side, the ghost ones beyond them on
    the other -- along an axis with o_d = 0 the owned range."""
    V = v.reshape(dims[2], dims[1], dims[0], comps)
    stride = (1, part.grid[0], part.grid[0] * part.grid[1])
    ops, recv = [], []
    for oz in (-1, 0, 1):
        for oy in (-1, 0, 1):
            for ox in (-1, 0, 1):
                o = (ox, oy, oz)
                if o == (0, 0, 0) or any((o[d] < 0 and part.coord[d] == 0) or (o[d] > 0 and part.coord[d] + 1 == part.grid[d])
                                         for d in range(3)):
                    continue
                own, ghost = [], []
                for d in range(3):
                    o0, o1, w = own0[d], own0[d] + own_n[d], width
                    own.append(slice(o0, o0 + w) if o[d] < 0 else (slice(o1 - w, o1) if o[d] > 0 else slice(o0, o1)))
                    ghost.append(slice(o0 - w, o0) if o[d] < 0 else (slice(o1, o1 + w) if o[d] > 0 else slice(o0, o1)))
                own, ghost = tuple(own[::-1]), tuple(ghost[::-1])
                peer = part.rank + sum(o[d] * stride[d] for d in range(3))
                src = V[ghost] if reverse else V[own]
                t = torch.from_numpy(np.ascontiguousarray(src).reshape(-1).copy())
                r = torch.empty_like(t)
                ops += [dist.P2POp(dist.isend, t, peer), dist.P2POp(dist.irecv, r, peer)]
                recv.append((r, own if reverse else ghost, src.shape))
    if ops:
        for q in dist.batch_isend_irecv(ops):
            q.wait()
    for r, where, shape in recv:
        if reverse:
            V[where] += r.numpy().reshape(shape)
        else:
            V[where] = r.numpy().reshape(shape)


def mode_cpu_box(args):
    """The box partition on CPU ranks: local problems of M.BoxPartition, the product's HOST setup on the local (extended) box,
    the oracle's numpy kernels and the numpy restatement of the axis-by-axis exchange; owned parts must reproduce the global
    operators."""
    rank, world = dist.get_rank(), dist.get_world_size()
    grid = parse_grid(args.grid, world)
    cells = tuple(8 * g if g > 1 else 6 for g in grid)
    material = "linear"
    part = M.BoxPartition(cells, rank, grid, length=tuple(c / 8.0 for c in cells), low_ghost_cells=args.low_ghost)
    mesh = O.StructuredMesh(cells)
    mesh.h = part.h
    gprob = M.LaplaceProblem(cells, material, cell_size=part.h)
    coef_g = gprob.coefficient.numpy()
    mf_g = O.MatrixFreeLaplace(mesh, coef_g)
    Rg = M.host_build_restrictor(gprob, PRM, matrix_free=True)
    Acg = M.host_galerkin(gprob, Rg, "matrix_free")
    rng = np.random.default_rng(0)
    xg = rng.random(mesh.n_dofs)
    lprob = part.local_problem(material)
    lmesh = O.StructuredMesh(part.local_cells)
    lmesh.h = part.h
    flags = lprob.constrained.numpy()
    own_l, own_g, loc_g = part.owned_local_index().numpy(), part.owned_global_index().numpy(), part.local_global_index().numpy()
    # every DoF has exactly one owner; ghost flags exactly off the owned box (global Dirichlet nodes keep 1)
    cover = torch.zeros(mesh.n_dofs, dtype=torch.float64); cover[torch.from_numpy(own_g)] = 1.0
    assert (_all_reduce_cpu(cover).numpy() == 1.0).all()
    ghost = np.ones(part.n_local_dofs, bool); ghost[own_l] = False
    assert ((flags == 2) <= ghost).all() and (flags[ghost] != 0).all() and (flags[own_l] != 2).all()
    mf_l = O.MatrixFreeLaplace(lmesh, lprob.coefficient.numpy(), constrained=(flags == 1))
    # coefficient table of the box == the cells of the global table
    lc, gc = part.local_cells, cells
    k, j, i = np.meshgrid(*(np.arange(lc[d]) + part.offset[d] for d in (2, 1, 0)), indexing="ij")
    np.testing.assert_array_equal(lprob.coefficient.numpy(), coef_g.reshape(-1, coef_g.shape[-1])[((k * gc[1] + j) * gc[0] + i).reshape(-1)])
    # garbage in the ghosts, then the exchange must restore them -- edges and corners included
    xl = xg[loc_g].copy()
    xl_g = xl.copy(); xl_g[ghost] = -7.0
    exchange_box_np(xl_g, part, part.local_nodes, part.own0, part.own_n)
    # (one layer per side travels: the nodes within one layer of the owned box along EVERY axis are valid afterwards)
    ln = part.local_nodes
    near = np.ones(ln[::-1], bool)
    for d in range(3):
        idx = np.arange(ln[d])
        ok = (idx >= part.own0[d] - (part.coord[d] > 0)) & (idx < part.own0[d] + part.own_n[d] + (part.coord[d] + 1 < grid[d]))
        near &= ok.reshape([-1 if a == 2 - d else 1 for a in range(3)])
    near = near.reshape(-1)
    assert near.sum() > len(own_l)
    np.testing.assert_array_equal(xl_g[near], xl[near])
    def gather(v_local, idx_l, idx_g, n):
        out = torch.zeros(n, dtype=torch.float64)
        out[torch.from_numpy(idx_g)] = torch.from_numpy(np.ascontiguousarray(v_local[idx_l]))
        return _all_reduce_cpu(out).numpy()
    np.testing.assert_allclose(gather(mf_l.vmult(xl_g), own_l, own_g, mesh.n_dofs), mf_g.vmult(xg), rtol=1e-13, atol=1e-14)
    # ---- restrictor / coarse operator of the local box against the global ones
    Rl = M.host_build_restrictor(lprob, PRM, matrix_free=True)
    na_l = tuple(c // 2 for c in part.local_cells); na_g = tuple(c // 2 for c in cells)
    c_own0 = tuple(g // 2 for g in part.ghost_lo); c_own_n = tuple(p // 2 for p in part.per)
    c_g0 = tuple(part.c0[d] // 2 - c_own0[d] for d in range(3))
    def cids(lo, n, dims, shift):
        kk = np.arange(lo[2], lo[2] + n[2]).reshape(-1, 1, 1, 1) + shift[2]
        jj = np.arange(lo[1], lo[1] + n[1]).reshape(1, -1, 1, 1) + shift[1]
        ii = np.arange(lo[0], lo[0] + n[0]).reshape(1, 1, -1, 1) + shift[0]
        return ((((kk * dims[1] + jj) * dims[0] + ii) * 2) + np.arange(2).reshape(1, 1, 1, -1)).reshape(-1)
    c_own_l, c_own_g = cids(c_own0, c_own_n, na_l, (0, 0, 0)), cids(c_own0, c_own_n, na_g, c_g0)
    c_loc_g = cids((0, 0, 0), na_l, na_g, c_g0)
    assert Rl.shape[0] == len(c_loc_g)
    ncg = Rg.shape[0]
    np.testing.assert_allclose(gather(Rl @ xl, c_own_l, c_own_g, ncg), Rg @ xg, rtol=1e-12, atol=1e-14)
    xc_g = rng.random(ncg)
    xc_l = xc_g[c_loc_g].copy()
    np.testing.assert_allclose(gather(Rl.T @ xc_l, own_l, own_g, mesh.n_dofs), Rg.T @ xc_g, rtol=1e-12, atol=1e-14)
    Acl = M.host_galerkin(lprob, Rl, "matrix_free")
    np.testing.assert_allclose(gather(Acl @ xc_l, c_own_l, c_own_g, ncg), Acg @ xc_g, rtol=1e-11, atol=1e-14)
    # reverse (adding) exchange on the coarse space: an owned entry ends up with the number of ranks that hold a copy of it
    ones = np.ones(len(c_loc_g))
    exchange_box_np(ones, part, na_l, c_own0, c_own_n, comps=2, reverse=True)
    nd = c_own_l // 2
    pos = (nd % na_l[0], (nd // na_l[0]) % na_l[1], nd // (na_l[0] * na_l[1]))
    expect = np.ones(len(c_own_l))
    for d in range(3):
        expect = expect * (1 + ((pos[d] == c_own0[d]) & (part.coord[d] > 0)) + ((pos[d] == c_own0[d] + c_own_n[d] - 1) & (part.coord[d] + 1 < grid[d])))
    np.testing.assert_array_equal(ones[c_own_l], expect)
    if rank == 0:
        print("cpu box checks passed", flush=True)


def mode_protocol(args):
    """CPU restatement of the distributed setup protocol of amg_device_setup.hip: on a z-slab partition of a node grid
    with two unknowns per node, P = (I - w D^-1 A) P_tent and A_c = P^T A P are READ OFF from operator applications on
    probing vectors defined on global coordinates -- nothing but vectors crosses between the ranks (forward halo
    exchange before A, reverse-add exchange after P^T) -- for two levels in a row (stencil reach 1 -> 2 -> 3), and must
    equal the rows of the globally formed matrices.  numpy / scipy + gloo; no GPU, no library."""
    import scipy.sparse as sp
    rank, world = dist.get_rank(), dist.get_world_size()
    C, nx, ny, per = 2, 6, 4, 12
    nzg = per * world
    rng = np.random.default_rng(7)

    def exchange(v, g, reverse=False):
        """v: local vector [layers][le]; g: level geometry dict."""
        le, w, ob, oc = g["le"], g["w"], g["ob"], g["oc"]
        V = v.reshape(-1, le)
        ops, recv = [], {}
        for side, peer in (("low", rank - 1), ("high", rank + 1)):
            if peer < 0 or peer >= world:
                continue
            if not reverse:
                src = V[ob:ob + w] if side == "low" else V[ob + oc - w:ob + oc]
            else:
                src = V[ob - w:ob] if side == "low" else V[ob + oc:ob + oc + w]
            t = torch.from_numpy(np.ascontiguousarray(src).reshape(-1).copy())
            r = torch.empty_like(t)
            ops += [dist.P2POp(dist.isend, t, peer), dist.P2POp(dist.irecv, r, peer)]
            recv[side] = r
        if ops:
            for q in dist.batch_isend_irecv(ops):
                q.wait()
        for side, r in recv.items():
            blk = r.numpy().reshape(w, le)
            if not reverse:
                if side == "low":
                    V[ob - w:ob] = blk
                else:
                    V[ob + oc:ob + oc + w] = blk
            else:
                if side == "low":
                    V[ob:ob + w] += blk
                else:
                    V[ob + oc - w:ob + oc] += blk

    def geom(dx, dy, oc, reach, gz_total):
        lo, hi = rank > 0, rank + 1 < world
        ob = reach if lo else 0
        nzl = ob + oc + (reach if hi else 0)
        return {"dx": dx, "dy": dy, "nz": nzl, "ob": ob, "oc": oc, "w": reach, "reach": reach, "le": dx * dy * C,
                "g0": rank * oc - ob, "gz": gz_total}

    # ---- a global SPD block-stencil operator of reach 1 on the node grid (the role of A_c)
    def lap1d(n):
        return sp.diags([-np.ones(n - 1), 2.5 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1])
    def m1d(n):
        return sp.diags([np.ones(n - 1), 4 * np.ones(n), np.ones(n - 1)], [-1, 0, 1]) / 6.0
    Kz, Ky, Kx, Mz, My, Mx = lap1d(nzg), lap1d(ny), lap1d(nx), m1d(nzg), m1d(ny), m1d(nx)
    S = sp.kron(Kz, sp.kron(My, Mx)) + sp.kron(Mz, sp.kron(Ky, Mx)) + sp.kron(Mz, sp.kron(My, Kx))   # 27-point
    Ag = sp.kron(S, sp.csr_matrix(np.array([[2.0, 0.5], [0.5, 1.0]]))).tocsr()
    Bg = 0.5 + rng.random(Ag.shape[0])
    omega = 4.0 / 3.0

    def global_level(Ag, Bg, dims):
        dx, dy, dz = dims
        n = Ag.shape[0]
        node = np.arange(n) // C
        comp = np.arange(n) % C
        i, j, k = node % dx, (node // dx) % dy, node // (dx * dy)
        cx, cy, cz = (dx + 1) // 2, (dy + 1) // 2, (dz + 1) // 2
        agg = (((k // 2) * cy + j // 2) * cx + i // 2) * C + comp
        n_c = cx * cy * cz * C
        norm = np.sqrt(np.bincount(agg, weights=Bg * Bg, minlength=n_c))
        Pt = sp.csr_matrix((Bg / norm[agg], (np.arange(n), agg)), shape=(n, n_c))
        d = Ag.diagonal()
        rho = (abs(Ag).sum(axis=1).A1 / abs(d)).max()
        Sm = sp.identity(n) - sp.diags(omega / rho / d) @ Ag
        P = (Sm @ Pt).tocsr()
        return P, (P.T @ Ag @ P).tocsr(), norm, (cx, cy, cz), rho

    def local_of(Mg, gf, gc=None):
        """owned rows of a global matrix in local numbering (rows: level gf, columns: gc or gf)."""
        gc = gc or gf
        r0, r1 = (gf["g0"] + gf["ob"]) * gf["le"], (gf["g0"] + gf["ob"] + gf["oc"]) * gf["le"]
        c0, c1 = gc["g0"] * gc["le"], (gc["g0"] + gc["nz"]) * gc["le"]
        sub = Mg[r0:r1]
        assert sub[:, :c0].nnz == 0 and sub[:, c1:].nnz == 0, "ghost layers too thin for this stencil"
        L = sp.lil_matrix((gf["nz"] * gf["le"], gc["nz"] * gc["le"]))
        L[gf["ob"] * gf["le"]:(gf["ob"] + gf["oc"]) * gf["le"]] = sub[:, c0:c1]
        return L.tocsr()

    def select(g, block, period, phase, comp, values=None):
        n = g["nz"] * g["le"]
        r = np.arange(n)
        c = r % C
        nd = r // C
        i, j, k = nd % g["dx"], (nd // g["dx"]) % g["dy"], nd // (g["dx"] * g["dy"])
        hit = (c == comp) & ((i // block) % period[0] == phase[0]) & ((j // block) % period[1] == phase[1]) & \
              (((k + g["g0"]) // block) % period[2] == phase[2])
        return np.where(hit, 1.0 if values is None else values, 0.0)

    dims = (nx, ny, nzg)
    gf = geom(nx, ny, per, 1, nzg)
    A_loc = local_of(Ag, gf)
    B_loc = Bg[gf["g0"] * gf["le"]:(gf["g0"] + gf["nz"]) * gf["le"]].copy()
    for level in range(2):
        P_ref, Ac_ref, norm_ref, cdims, rho_ref = global_level(Ag, Bg, dims)
        reach_c = (1 + 3 * gf["reach"]) // 2
        gc = geom(cdims[0], cdims[1], gf["oc"] // 2, reach_c, cdims[2])
        own_f = slice(gf["ob"] * gf["le"], (gf["ob"] + gf["oc"]) * gf["le"])
        own_c = slice(gc["ob"] * gc["le"], (gc["ob"] + gc["oc"]) * gc["le"])
        # rho: max over the owned rows of all ranks
        d = np.zeros(A_loc.shape[0]); d[own_f] = A_loc.diagonal()[own_f]
        rho_t = torch.tensor([(abs(A_loc[own_f]).sum(axis=1).A1 / abs(d[own_f])).max()])
        dist.all_reduce(rho_t, op=dist.ReduceOp.MAX)
        rho = float(rho_t)
        assert abs(rho - rho_ref) < 1e-14 * rho_ref
        wgt = omega / rho
        # tentative prolongator: t = B / |B|_aggregate on the owned nodes, exchanged to the ghosts
        r = np.arange(A_loc.shape[0]); nd = r // C
        i, j, k = nd % gf["dx"], (nd // gf["dx"]) % gf["dy"], nd // (gf["dx"] * gf["dy"])
        agg_l = ((((k + gf["g0"]) // 2 - gc["g0"]) * gc["dy"] + j // 2) * gc["dx"] + i // 2) * C + r % C
        own_mask = np.zeros(A_loc.shape[0], bool); own_mask[own_f] = True
        n2 = np.bincount(agg_l[own_mask], weights=(B_loc * B_loc)[own_mask], minlength=gc["nz"] * gc["le"])
        t = np.zeros_like(B_loc); t[own_mask] = B_loc[own_mask] / np.sqrt(n2[agg_l[own_mask]])
        exchange(t, gf)
        # ---- P by probing: aggregates (1 + reach) apart have disjoint columns
        gdc = (gc["dx"], gc["dy"], gc["gz"])
        per_p = [max(1, min(1 + gf["reach"], gdc[dd])) for dd in range(3)]
        P_loc = sp.lil_matrix((A_loc.shape[0], gc["nz"] * gc["le"]))
        for oc in range(per_p[0] * per_p[1] * per_p[2]):
            ph = (oc % per_p[0], (oc // per_p[0]) % per_p[1], oc // (per_p[0] * per_p[1]))
            for comp in range(C):
                y = select(gf, 2, per_p, ph, comp, t)
                z = y - wgt * np.where(own_mask, (A_loc @ y) / np.where(d != 0, d, 1.0), 0.0)
                for row in np.nonzero(own_mask & (z != 0))[0]:
                    x_, y_, zg = i[row], j[row], k[row] + gf["g0"]
                    # the one aggregate of this colour within reach of the row's node
                    cand = [(I, J, K) for K in range(max(0, (zg - gf["reach"]) // 2), min(gdc[2] - 1, (zg + gf["reach"]) // 2) + 1)
                            for J in range(max(0, (y_ - gf["reach"]) // 2), min(gdc[1] - 1, (y_ + gf["reach"]) // 2) + 1)
                            for I in range(max(0, (x_ - gf["reach"]) // 2), min(gdc[0] - 1, (x_ + gf["reach"]) // 2) + 1)
                            if (I % per_p[0], J % per_p[1], K % per_p[2]) == ph]
                    assert len(cand) == 1, (row, cand)
                    I, J, K = cand[0]
                    P_loc[row, (((K - gc["g0"]) * gc["dy"] + J) * gc["dx"] + I) * C + comp] = z[row]
        P_loc = P_loc.tocsr()
        assert abs(P_loc - local_of(P_ref, gf, gc)).max() < 1e-13
        # ---- A_c = P^T A P by probing: coarse nodes (2 reach_c + 1) apart never meet in a row
        per_a = [max(1, min(2 * reach_c + 1, gdc[dd])) for dd in range(3)]
        Ac_loc = sp.lil_matrix((gc["nz"] * gc["le"],) * 2)
        rc = np.arange(gc["nz"] * gc["le"]); ndc = rc // C
        X, Y, Z = ndc % gc["dx"], (ndc // gc["dx"]) % gc["dy"], ndc // (gc["dx"] * gc["dy"]) + gc["g0"]
        own_c_mask = np.zeros(gc["nz"] * gc["le"], bool); own_c_mask[own_c] = True
        for oc in range(per_a[0] * per_a[1] * per_a[2]):
            ph = (oc % per_a[0], (oc // per_a[0]) % per_a[1], oc // (per_a[0] * per_a[1]))
            for comp in range(C):
                u = select(gc, 1, per_a, ph, comp)
                wv = P_loc @ u
                exchange(wv, gf)                      # A reads its ghost layers
                v = np.where(own_mask, A_loc @ wv, 0.0)
                yc = P_loc.T @ v                     # partial sums for the neighbours' aggregates in the ghost layers
                exchange(yc, gc, reverse=True)
                for row in np.nonzero(own_c_mask & (yc != 0))[0]:
                    cand = [(I, J, K) for K in range(max(0, Z[row] - reach_c), min(gdc[2] - 1, Z[row] + reach_c) + 1)
                            for J in range(max(0, Y[row] - reach_c), min(gdc[1] - 1, Y[row] + reach_c) + 1)
                            for I in range(max(0, X[row] - reach_c), min(gdc[0] - 1, X[row] + reach_c) + 1)
                            if (I % per_a[0], J % per_a[1], K % per_a[2]) == ph]
                    assert len(cand) == 1, (row, cand)
                    I, J, K = cand[0]
                    Ac_loc[row, (((K - gc["g0"]) * gc["dy"] + J) * gc["dx"] + I) * C + comp] = yc[row]
        Ac_loc = Ac_loc.tocsr()
        assert abs(Ac_loc - local_of(Ac_ref, gc)).max() < 1e-12 * abs(Ac_ref).max()
        # next level
        Ag, Bg, dims = Ac_ref, norm_ref, cdims
        A_loc, gf = Ac_loc, gc
        B_loc = Bg[gf["g0"] * gf["le"]:(gf["g0"] + gf["nz"]) * gf["le"]].copy()
    if rank == 0:
        print("protocol checks passed", flush=True)


MESHES = {
    # name: (cells per rank along z, (cx, cy) = cells per rank along x and y (slabs: the global counts), material, amg parameters)
    # small: A_c (768 rows at 2 ranks) is gathered right away: replicated hierarchy behind one all-gather
    "small": (8, (16, 12), "linear", {"coarsest_size": 300}),
    # wide: 67 node columns and 65 node rows with a constant material: the operator runs its one-coefficient-per-cell
    # variant with the tail columns as a rotated slab, split into z-tile ranges by the overlapped exchange
    "wide": (8, (66, 64), "constant", {"coarsest_size": 300}),
    # deep: two aggregation levels stay distributed (stencil reach 1 and 2, probing with 16 / 250 and 54 / 686
    # vectors, reverse exchanges two and three layers wide), the third is gathered
    "deep": (24, (16, 16), "linear", {"coarsest_size": 40, "replicate_rows": 40}),
    # deep01: the same with the V(0,1) coarse cycle of the bench (solver.amg.pre_smoothing_levels 0): b is restricted and the
    # correction added on every level of the aggregation hierarchy, distributed and gathered ones alike
    "deep01": (24, (16, 16), "linear", {"coarsest_size": 40, "replicate_rows": 40, "pre_smoothing_levels": 0}),
    # boxes (--grid 2x1x1, 2x2x1, 2x1x2, 1x2x2): 24 cells per rank along every axis keep two aggregation levels distributed along
    # the split axes (12 agglomerates -> 6 -> 3 nodes per rank; reach 1, 2, 3), the third is gathered through the permutation
    # of the rank-ordered blocks
    "cube": (24, (24, 24), "linear", {"coarsest_size": 40, "replicate_rows": 40, "pre_smoothing_levels": 0}),
    "cube11": (24, (24, 24), "constant", {"coarsest_size": 40, "replicate_rows": 40}),
    # boxwide: 133 node columns per rank = two column tiles of the operator + the tail slab, several y- and z-tiles: the exchange
    # of a box overlaps with the tiles that read no ghost plane along any axis, the shell of tiles around them follows
    "boxwide": (16, (130, 32), "constant", {"coarsest_size": 300}),
    # boxnarrow: 80 cells per rank along x -- with two ghost agglomerates below, the local meshes of a 2 x ... grid are 83 and 85
    # node columns wide: one full chunk column + a NARROW last one of 25 resp. 27 columns (the widest the sweep takes is 32 - halo:
    # the 259 / 261 columns of a rank of the 2 x 2 x 2 bench run are 4 full ones + 27 / 29)
    "boxnarrow": (16, (80, 32), "constant", {"coarsest_size": 300}),
}


def mode_gpu(args):
    rank, world = dist.get_rank(), dist.get_world_size()
    # gloo: all ranks share cuda:0 (host transport); nccl: one GPU per rank, the native RCCL transport
    native = dist.get_backend() == "nccl"
    torch.cuda.set_device(rank if native else 0)
    grid = parse_grid(args.grid, world)
    per, (cx, cy), material, amg = MESHES[args.mesh]
    # (cx, cy): cells per rank along x and y, `per` along z
    cells = (cx * grid[0], cy * grid[1], per * grid[2])
    part = M.BoxPartition(cells, rank, grid, length=tuple(c / float(cells[0]) for c in cells), low_ghost_cells=args.low_ghost)   # cubic cells
    box = part.split_xy
    params = dict(PRM)
    params.update({"smoother": {"type": "Chebyshev", "degree": 3, "smoothing_range": 20.0},
                   "solver": {"type": "amg", "amg": dict(amg)}, "is preconditioner": False})
    ctx = M.Context()
    tr = M.HaloTransport(ctx, part, 2)
    assert tr.name() == ("rccl" if native else "host")
    assert tr.selftest(4096) == 0.0          # loop-back, all-gather, sum / max all-reduce through the transport
    h = M.Hierarchy(ctx, "HipMatrixFreeMeshEvaluator", part.local_problem(material, "cuda"), params)
    deg, lmin, lmax = h.smoother_info()       # estimated with dot products summed over the ranks
    assert 1.4 < lmax < 2.2, lmax
    # one coefficient per cell: the first two Chebyshev terms run as ONE sweep (x exchanged two ghost planes deep once, the
    # ghost DoFs computed redundantly), the third as a launch of its own; eight coefficients per cell: a launch per term
    # (with two agglomerates of every lower neighbour in the local mesh, --low-ghost 4, the whole Chebyshev(3) smoother of the
    # cycle is one sweep: x three planes deep, b two; an in-place call keeps the last term as a launch of its own)
    assert h.smoother_sweep_terms() == ((2, 3 if args.low_ghost == 4 else 0) if material == "constant" else (0, 0)), h.smoother_sweep_terms()
    # the halo spaces of the levels belong to this hierarchy: a second one on the same communicator context is refused
    # while it lives (every rank raises before any collective of the second setup)
    try:
        M.Hierarchy(ctx, "HipMatrixFreeMeshEvaluator", part.local_problem(material, "cuda"), params)
        raise AssertionError("a second hierarchy on a distributed context must be refused")
    except L.MfmgError as e:
        assert "already carries a hierarchy" in str(e)
    # global single-process reference on the same GPU (context without communicator): the SAME parameters -- the
    # distributed hierarchy must be the same preconditioner, eigenvalue estimates included
    gctx = M.Context()
    gprob = M.LaplaceProblem(cells, material, device="cuda", cell_size=part.h)
    hg = M.Hierarchy(gctx, "HipMatrixFreeMeshEvaluator", gprob, params)
    _, glmin, glmax = hg.smoother_info()
    assert abs(glmax - lmax) < 1e-9 * lmax and abs(glmin - lmin) < 1e-9 * lmax, (lmin, lmax, glmin, glmax)
    ng, nl = gprob.n_dofs, part.n_local_dofs
    own_l, own_g, loc_g = part.owned_local_index().numpy(), part.owned_global_index().numpy(), part.local_global_index().numpy()
    f1 = tr.space_index(1)                     # the library's own description of the fine space must say the same
    assert np.array_equal(f1[0].numpy(), own_l) and np.array_equal(f1[1].numpy(), own_g) and np.array_equal(f1[2].numpy(), loc_g)
    ghost_l = np.ones(nl, bool); ghost_l[own_l] = False
    rng = np.random.default_rng(0)
    xg, bg = rng.random(ng), rng.random(ng)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    def local(vg, poison=True):
        v = vg[loc_g].copy()
        if poison:   # ghosts must come from the exchange, not from the test
            v[ghost_l] = 1e30
        return v
    def gather(v_local, idx_l, idx_g, n):
        out = torch.zeros(n, dtype=torch.float64)
        out[torch.from_numpy(idx_g)] = torch.from_numpy(np.ascontiguousarray(v_local[idx_l]))
        return _all_reduce_cpu(out).numpy()
    def check(local_out, global_out, what, tol=1e-12):
        got = gather(local_out.cpu().numpy(), own_l, own_g, ng)
        ref = global_out.cpu().numpy()
        err = np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-300)
        assert err < tol, f"{what}: {err}"
    # fine operator
    yl = torch.zeros(nl, dtype=torch.float64, device="cuda"); yg = torch.empty(ng, dtype=torch.float64, device="cuda")
    v0 = tr.exchange_volume()
    h.operator_apply(0, dev(local(xg)), yl); hg.operator_apply(0, dev(xg), yg)
    sent = tr.exchange_volume() - v0
    check(yl, yg, "fine operator")
    if args.mesh in ("wide", "boxwide"):
        # these meshes have tiles that read no ghost plane: the exchange ran beside them on the second stream (on the ranks
        # whose interior is not empty: rank 0 always)
        over = torch.tensor([float(tr.n_overlapped())])
        assert (rank != 0 or over.item() > 0) and _all_reduce_cpu(over).item() > 0
    # what one fine exchange moves: the faces, edges and corners of the owned box towards the neighbours that exist -- a box
    # sends faces of (N / 2)^2 where a slab sends planes of N^2
    assert sent == part.exchange_doubles(), (sent, part.exchange_doubles())
    # smoother (3 fused steps, 3 exchanges)
    xl = dev(local(xg)); xs = dev(xg)
    h.smoother_apply(0, dev(local(bg, False)), xl); hg.smoother_apply(0, dev(bg), xs)
    check(xl, xs, "smoother", 1e-11)
    # the same two with the exchange NOT overlapped with the interior tiles: identical bits
    ctx.set_overlap_exchange(False)
    yl2 = torch.zeros(nl, dtype=torch.float64, device="cuda")
    h.operator_apply(0, dev(local(xg)), yl2)
    xl2 = dev(local(xg))
    h.smoother_apply(0, dev(local(bg, False)), xl2)
    sl = torch.from_numpy(own_l).cuda()
    assert torch.equal(yl2[sl], yl[sl]) and torch.equal(xl2[sl], xl[sl]), "overlapped exchange changed the result"
    ctx.set_overlap_exchange(True)
    # restriction / prolongation / coarse operator
    c_own_l, c_own_g, c_loc_g = (t.numpy() for t in tr.space_index(2))
    cb = tr.box(2)
    assert cb["comps"] == 2 and cb["gn"] == tuple(c // 2 for c in cells)
    for d in range(3):
        assert cb["own_n"][d] == part.per[d] // 2 and cb["g0"][d] + cb["own0"][d] == part.c0[d] // 2
    if not box:
        lay, nlay, cb0, cc0 = tr.layout(2)
        sp = tr.space(2)
        assert sp["global_begin"] == part.z0 // 2 - cb0 and sp["global_layers"] == cells[2] // 2
    ncl, ncg = len(c_loc_g), hg.level_size(1)
    assert h.level_size(1) == ncl
    c_ghost = np.ones(ncl, bool); c_ghost[c_own_l] = False
    gather_c = lambda v: gather(v.cpu().numpy(), c_own_l, c_own_g, ncg)
    rl = torch.zeros(ncl, dtype=torch.float64, device="cuda"); rg = torch.empty(ncg, dtype=torch.float64, device="cuda")
    h.restrictor_apply(1, dev(local(xg)), rl); hg.restrictor_apply(1, dev(xg), rg)
    np.testing.assert_allclose(gather_c(rl), rg.cpu().numpy(), rtol=1e-12, atol=1e-13)
    # residual + restriction as the cycle computes them (one pass over x and b where the rows of R A repeat themselves:
    # x is then needed two ghost layers deep, b one -- both poisoned here, so they must come from the exchanges)
    if args.mesh in ("wide", "boxwide"):
        assert h.residual_restriction_classes() > 0 and hg.residual_restriction_classes() > 0
    r1 = torch.zeros(ncl, dtype=torch.float64, device="cuda"); r1g = torch.empty(ncg, dtype=torch.float64, device="cuda")
    h.restrict_residual(dev(local(xg)), dev(local(bg)), r1); hg.restrict_residual(dev(xg), dev(bg), r1g)
    np.testing.assert_allclose(gather_c(r1), r1g.cpu().numpy(), rtol=1e-11, atol=1e-12 * np.abs(r1g.cpu().numpy()).max())
    xcg = rng.random(ncg)
    xcl = xcg[c_loc_g].copy()
    xcl[c_ghost] = 1e30
    pl = torch.zeros(nl, dtype=torch.float64, device="cuda"); pg = torch.empty(ng, dtype=torch.float64, device="cuda")
    h.restrictor_apply(1, dev(xcl), pl, 1); hg.restrictor_apply(1, dev(xcg), pg, 1)
    check(pl, pg, "prolongation")
    al = torch.zeros(ncl, dtype=torch.float64, device="cuda"); ag = torch.empty(ncg, dtype=torch.float64, device="cuda")
    h.operator_apply(1, dev(xcl), al); hg.operator_apply(1, dev(xcg), ag)
    np.testing.assert_allclose(gather_c(al), ag.cpu().numpy(), rtol=1e-11, atol=1e-13)
    # reverse (adding) exchange: every ghost entry returns to its owner exactly once -- an owned entry ends up with the number
    # of ranks that hold a copy of it (its boundary layers along every axis with a neighbour)
    ones = torch.ones(ncl, dtype=torch.float64, device="cuda")
    tr.exchange(2, ones, reverse=True)
    nd = c_own_l // cb["comps"]
    pos = (nd % cb["dims"][0], (nd // cb["dims"][0]) % cb["dims"][1], nd // (cb["dims"][0] * cb["dims"][1]))
    expect = np.ones(len(c_own_l))
    for d in range(3):
        copies = 1 + ((pos[d] == cb["own0"][d]) & (part.coord[d] > 0)) + ((pos[d] == cb["own0"][d] + cb["own_n"][d] - 1) & (part.coord[d] + 1 < grid[d]))
        expect = expect * copies
    np.testing.assert_array_equal(ones.cpu().numpy()[c_own_l], expect)
    # the coarse solve: the aggregation hierarchy coupled across the ranks against the single-process one
    bcg = rng.random(ncg)
    bcl = bcg[c_loc_g].copy()
    scl = torch.zeros(ncl, dtype=torch.float64, device="cuda"); scg = torch.empty(ncg, dtype=torch.float64, device="cuda")
    h.coarse_apply(dev(bcl), scl); hg.coarse_apply(dev(bcg), scg)
    np.testing.assert_allclose(gather_c(scl), scg.cpu().numpy(), rtol=1e-9, atol=1e-11 * np.abs(scg.cpu().numpy()).max())
    # ---- V-cycles: 20-cycle residual history of the distributed cycle == the single-process cycle == the oracle's
    con_g = (gprob.constrained == 1).cpu().numpy()
    x0g = np.where(con_g, 0.0, xg)
    b0 = torch.zeros(nl, dtype=torch.float64, device="cuda")
    x = dev(local(x0g, False))
    r = torch.empty_like(x)
    n_cycles = 20
    hist = []
    for _ in range(n_cycles + 1):
        h.operator_apply(0, x, r)
        hist.append(tr.owned_norm(r))
        h.apply(b0, x)
    xs = dev(x0g); bs = torch.zeros(ng, dtype=torch.float64, device="cuda"); rs = torch.empty_like(xs)
    hist_g = []
    for _ in range(n_cycles + 1):
        hg.operator_apply(0, xs, rs)
        hist_g.append(gctx.l2_norm(rs))
        hg.apply(bs, xs)
    hist, hist_g = np.array(hist), np.array(hist_g)
    floor = 1e-12 * hist_g[0]          # where b - A x sits on FP64 rounding
    np.testing.assert_allclose(hist, hist_g, rtol=1e-10, atol=floor)
    # the oracle's restatement of the cycle (numpy / scipy), built from the single-process level matrices
    mesh = O.StructuredMesh(cells)
    mesh.h = part.h
    mf = O.MatrixFreeLaplace(mesh, gprob.coefficient.cpu().numpy())
    p = O.ChebyshevParams(deg, glmax, glmin)
    smoother = lambda b, xx: O.chebyshev_smoother_apply(mf.vmult, mf.diagonal_inverse(), p, b, xx)
    ho = O.TwoLevelHierarchy(mf.vmult, smoother, hg.restrictor().to_scipy(), O.amg_coarse_solver(hg.coarse_amg_levels(), 1, pre_smoothing_levels=amg.get("pre_smoothing_levels")), 1, False)
    res_o, rate, _ = O.vcycle_history(ho, mf.vmult, np.zeros(ng), x0g, n_cycles=n_cycles)
    res_o = np.array(res_o)
    np.testing.assert_allclose(hist / hist[0], res_o[:n_cycles + 1] / res_o[0], rtol=1e-9, atol=1e-12)
    assert rate < 0.6
    if material == "constant":
        # a preconditioner application (Hierarchy::vmult with "is preconditioner" true, include/mfmg/common/hierarchy.hpp:253-259): with
        # the one-sweep smoother the first pre-smoothing step starts from zero inside the sweep -- x is not exchanged (zero on every
        # rank), b travels as deep as the sweep reads it; garbage in x must not matter.  Against the single-process hierarchy.
        del h
        pp = dict(params); pp["is preconditioner"] = True
        hp = M.Hierarchy(ctx, "HipMatrixFreeMeshEvaluator", part.local_problem(material, "cuda"), pp)
        hgp = M.Hierarchy(gctx, "HipMatrixFreeMeshEvaluator", gprob, pp)
        zl = dev(1e6 * rng.random(nl)); zg = dev(1e6 * rng.random(ng))
        hp.vmult(zl, dev(local(bg, False))); hgp.vmult(zg, dev(bg))
        check(zl, zg, "preconditioner application", 1e-10)
        del hp, hgp
    if rank == 0:
        print("gpu distributed checks passed; grid", "x".join(map(str, grid)), "transport", tr.name(), "exchanges", tr.n_exchanges(),
              "spaces", tr.space(1)["n_spaces"], "residuals", ["%.3e" % v for v in hist[:6]], flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="cpu")
    ap.add_argument("--mesh", default="small")
    ap.add_argument("--backend", default="gloo")
    ap.add_argument("--grid", default="", help="ranks along x, y, z as 2x2x1 (default: slabs, 1x1xworld)")
    ap.add_argument("--low-ghost", type=int, default=2, help="ghost cell layers towards a lower neighbour: 2 or 4 (BoxPartition)")
    a = ap.parse_args()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if a.backend == "nccl":
        # before any HIP call of this process: pick the rank's GPU
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl", device_id=torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))))
    else:
        dist.init_process_group("gloo")
    try:
        {"cpu": mode_cpu, "cpu_box": mode_cpu_box, "gpu": mode_gpu, "protocol": mode_protocol}[a.mode](a)
    finally:
        dist.destroy_process_group()
