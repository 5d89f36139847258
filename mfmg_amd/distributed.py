"""One process per GPU: decomposition of the mesh into slabs along z or into boxes, and the halo transport.

The reference partitions the mesh with p4est and exchanges ghost DoFs inside deal.II
(tests/laplace_matrix_free.hpp:222, MatrixFree::cell_loop); its CUDA path all-gathers the whole
vector in front of every SpMV (source/cuda/utils.cu:363-482).  Here every rank owns a slab of cell
layers; its local mesh is that slab plus one agglomerate (2 cell layers) of each neighbour, numbered
lexicographically, so that ghost planes are contiguous and a halo exchange is two sends and two
receives of a few layers -- RCCL point-to-point over xGMI, issued by the library itself on its HIP stream
(include/mfmg_hip.h: mfmg_hip_context_use_rccl).  In the tests (several ranks sharing one GPU) the library
stages the layers through the host and torch.distributed / gloo carries them."""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import torch
import torch.distributed as dist

from . import lib as _lib
from .laplace import LaplaceProblem
from .lib import check


class BoxPartition:
    """Owned cell box of rank `rank` in a grid (gx, gy, gz) of equal boxes over a global mesh of `cells`
    (rank = cx + gx (cy + gy cz); SURVEY.md 8e: 2 x 1 x 1, 2 x 2 x 1, 2 x 2 x 2).  The local mesh is the box plus one
    agglomerate (2 cell layers) of each face neighbour, numbered lexicographically; interface planes belong to the upper
    box, the last box of an axis also owns the top plane.  (1, 1, n) are the slabs along z."""

    def __init__(self, cells: Sequence[int], rank: int, grid: Sequence[int], length: Sequence[float] | float = 1.0,
                 low_ghost_cells: int = 2):
        """low_ghost_cells = 4: TWO agglomerates of every lower neighbour.  The interface plane belongs to the upper box, so a
        box holds three ghost node planes above it but only low_ghost_cells below; with four the whole Chebyshev(3) smoother of
        a rank is one sweep (mf_cheb_fused.hip: K terms need K ghost planes on every side with a neighbour) and x travels once
        per smoother, three planes deep."""
        assert low_ghost_cells in (2, 4)
        self.low_ghost_cells = int(low_ghost_cells)
        self.cells = tuple(int(c) for c in cells)
        self.grid = tuple(int(g) for g in grid)
        assert len(self.cells) == 3 and len(self.grid) == 3
        self.n_ranks = self.grid[0] * self.grid[1] * self.grid[2]
        self.rank = int(rank)
        assert 0 <= self.rank < self.n_ranks
        self.coord = (self.rank % self.grid[0], (self.rank // self.grid[0]) % self.grid[1], self.rank // (self.grid[0] * self.grid[1]))
        for d in range(3):
            if self.grid[d] > 1 and self.cells[d] % (2 * self.grid[d]) != 0:
                raise ValueError("the cell layers must split into whole agglomerate layers per rank")
        self.per = tuple(self.cells[d] // self.grid[d] for d in range(3))
        self.c0 = tuple(self.coord[d] * self.per[d] for d in range(3))
        self.ghost_lo = tuple(self.low_ghost_cells if self.coord[d] > 0 else 0 for d in range(3))
        self.ghost_hi = tuple(2 if self.coord[d] + 1 < self.grid[d] else 0 for d in range(3))
        self.local_cells = tuple(self.per[d] + self.ghost_lo[d] + self.ghost_hi[d] for d in range(3))
        self.offset = tuple(self.c0[d] - self.ghost_lo[d] for d in range(3))     # global index of local cell / node 0
        ln = (length,) * 3 if isinstance(length, (int, float)) else tuple(length)
        self.h = tuple(ln[d] / self.cells[d] for d in range(3))
        self.local_nodes = tuple(c + 1 for c in self.local_cells)
        self.global_nodes = tuple(c + 1 for c in self.cells)
        self.own0 = self.ghost_lo
        self.own_n = tuple(self.per[d] + (1 if self.coord[d] + 1 == self.grid[d] else 0) for d in range(3))
        self.n_global_dofs = self.global_nodes[0] * self.global_nodes[1] * self.global_nodes[2]
        self.n_local_dofs = self.local_nodes[0] * self.local_nodes[1] * self.local_nodes[2]
        # the names of the slab code (z axis)
        self.z0, self.z1 = self.c0[2], self.c0[2] + self.per[2]
        self.ghost_low, self.ghost_high = self.ghost_lo[2], self.ghost_hi[2]
        self.z_offset = self.offset[2]
        self.plane = self.local_nodes[0] * self.local_nodes[1]
        self.n_local_planes = self.local_nodes[2]
        self.owned_plane_begin, self.owned_plane_count = self.own0[2], self.own_n[2]

    @property
    def split_xy(self) -> bool:
        return self.grid[0] > 1 or self.grid[1] > 1

    def exchange_doubles(self, width: int = 1) -> int:
        """Doubles this rank sends in one exchange of a fine-level vector: to every neighbour (faces, edges, corners) `width`
        layers along the axes in which it is offset, the owned range along the others.  Slabs send whole local planes."""
        if not self.split_xy:
            return width * self.plane * ((self.coord[2] > 0) + (self.coord[2] + 1 < self.grid[2]))
        total = 0
        for oz in (-1, 0, 1):
            for oy in (-1, 0, 1):
                for ox in (-1, 0, 1):
                    o = (ox, oy, oz)
                    if o == (0, 0, 0) or any((o[d] < 0 and self.coord[d] == 0) or (o[d] > 0 and self.coord[d] + 1 == self.grid[d])
                                             for d in range(3)):
                        continue
                    n = 1
                    for d in range(3):
                        n *= width if o[d] else self.own_n[d]
                    total += n
        return total

    def local_problem(self, material: str = "constant", device="cpu") -> LaplaceProblem:
        """Mesh arrays of the local (extended) box: global Dirichlet nodes carry 1, ghost nodes 2."""
        prob = LaplaceProblem(self.local_cells, material, device=device, dirichlet=False, cell_size=self.h,
                              cell_offset=self.offset)
        dv = prob.device
        idx = [torch.arange(self.local_nodes[d], device=dv) for d in range(3)]
        shape = [(1, 1, -1), (1, -1, 1), (-1, 1, 1)]
        boundary = None
        ghost = None
        for d in range(3):
            g = (idx[d] + self.offset[d]).view(shape[d])
            bd = (g == 0) | (g == self.cells[d])
            gh = ((idx[d] < self.own0[d]) | (idx[d] >= self.own0[d] + self.own_n[d])).view(shape[d])
            boundary = bd if boundary is None else (boundary | bd)
            ghost = gh if ghost is None else (ghost | gh)
        full = (self.local_nodes[2], self.local_nodes[1], self.local_nodes[0])
        boundary, ghost = boundary.expand(full), ghost.expand(full)
        flags = torch.where(boundary, 1, torch.where(ghost, 2, 0)).to(torch.uint8)
        prob.constrained = flags.reshape(-1).contiguous()
        return prob

    # -- index maps between the local (extended) vector, its owned part and the global lexicographic vector --------
    def _ids(self, lo, n, dims, shift):
        k = torch.arange(lo[2], lo[2] + n[2]).view(-1, 1, 1) + shift[2]
        j = torch.arange(lo[1], lo[1] + n[1]).view(1, -1, 1) + shift[1]
        i = torch.arange(lo[0], lo[0] + n[0]).view(1, 1, -1) + shift[0]
        return ((k * dims[1] + j) * dims[0] + i).reshape(-1)

    def owned_local_index(self) -> torch.Tensor:
        """Positions of the owned DoFs in the local vector (lexicographic order of the owned box)."""
        return self._ids(self.own0, self.own_n, self.local_nodes, (0, 0, 0))

    def owned_global_index(self) -> torch.Tensor:
        """... and where they live in the global lexicographic vector."""
        return self._ids(self.own0, self.own_n, self.global_nodes, self.offset)

    def local_global_index(self) -> torch.Tensor:
        """Global position of every entry of the local vector (ghosts included)."""
        return self._ids((0, 0, 0), self.local_nodes, self.global_nodes, self.offset)

    def owned_slice(self) -> slice:
        assert not self.split_xy, "the owned DoFs of a box are not one run: owned_local_index()"
        return slice(self.owned_plane_begin * self.plane, (self.owned_plane_begin + self.owned_plane_count) * self.plane)

    def global_slice(self) -> slice:
        """Where the owned entries live in the global lexicographic vector (slabs)."""
        assert not self.split_xy, "the owned DoFs of a box are not one run: owned_global_index()"
        return slice(self.z0 * self.plane, (self.z0 + self.owned_plane_count) * self.plane)

    def local_from_global(self, xg: torch.Tensor) -> torch.Tensor:
        """Local (extended) vector cut out of a global lexicographic vector (ghosts filled as well)."""
        if not self.split_xy:
            lo = self.z_offset * self.plane
            return xg[lo: lo + self.n_local_planes * self.plane].clone()
        return xg[self.local_global_index().to(xg.device)].clone()


class SlabPartition(BoxPartition):
    """Owned cell layers [z0, z1) of rank `rank` out of `n_ranks` for a global mesh of `cells`: the grid 1 x 1 x n_ranks."""

    def __init__(self, cells: Sequence[int], rank: int, n_ranks: int, length: Sequence[float] | float = 1.0, low_ghost_cells: int = 2):
        super().__init__(cells, rank, (1, 1, int(n_ranks)), length, low_ghost_cells)


def box_grid(n_ranks: int) -> tuple:
    """The grid (ranks along x, y, z) of SURVEY.md 8e for a rank count, split along z FIRST -- 1 x 1 x 2, 1 x 2 x 2, 2 x 2 x 2: the
    grid the weak-scaling mesh of bench.py grows by (z, then y, then x; the boundary layers of a z-split are contiguous runs of the
    vector and travel without packing); otherwise as cubic as the factors allow, slabs along z for a prime count."""
    grid = [1, 1, 1]
    n, d = int(n_ranks), 0
    f = 2
    while n > 1:
        while n % f:
            f += 1
        grid[2 - d % 3] *= f
        n //= f
        d += 1
    return tuple(grid)


_EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.POINTER(C.c_double)),
                           C.POINTER(C.POINTER(C.c_double)), C.POINTER(C.c_int64))
_ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_int)
_ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int64, C.POINTER(C.c_double))


def _host_tensor(ptr, n):
    """torch view (no copy) of n doubles of host memory owned by the library."""
    import numpy as np
    return torch.from_numpy(np.ctypeslib.as_array(ptr, shape=(int(n),)))


class HaloTransport:
    """Registers the communicator of a Context and its transport.

    transport "rccl": the library exchanges the boundary layers itself with ncclSend / ncclRecv on its HIP stream
    (one process per GPU; this object only carries the 128-byte RCCL id from rank 0 to the other ranks through
    torch.distributed).  transport "host": the library stages the layers through pinned host memory and calls back
    into torch.distributed on a gloo group (tests: several ranks on one card).  Default: "rccl" when the default
    process group is nccl, else "host"."""

    def __init__(self, ctx, part: BoxPartition, n_eigenvectors: int = 2, group=None, transport: str | None = None,
                 callbacks=None):
        """callbacks = (exchange, allreduce, allgather): a host transport of the caller's own instead of torch.distributed --
        exchange(peers, send, recv) with lists of numpy views (send[i] goes to rank peers[i], recv[i] is filled from it, all
        at once), allreduce(values, op) in place on a numpy view (op 0 sum, 1 max), allgather(src, out) -- e.g. mailboxes
        between threads of one process that each drive a rank (tests/test_box_threads.py)."""
        self._lib = _lib.load()
        self.ctx, self.part, self.group = ctx, part, group
        self._user_callbacks = callbacks
        self.backend = "custom" if callbacks else (dist.get_backend(group) if dist.is_initialized() else "none")
        self.rank, self.n_ranks = part.rank, part.n_ranks
        if callbacks:
            transport = "host"
        if transport is None:
            transport = "rccl" if self.backend == "nccl" else "host"
        assert transport in ("rccl", "host")
        self.transport = transport
        if part.split_xy:
            i3 = C.c_int32 * 3
            check(self._lib.mfmg_hip_context_set_communicator_box(ctx.handle, self.rank, i3(*part.grid), i3(*part.ghost_lo),
                                                                  i3(*part.ghost_hi)))
        else:
            check(self._lib.mfmg_hip_context_set_communicator(ctx.handle, self.rank, self.n_ranks, part.ghost_low,
                                                              part.ghost_high))
        if part.low_ghost_cells != 2:
            check(self._lib.mfmg_hip_context_set_low_ghost_cells(ctx.handle, part.low_ghost_cells))
        if transport == "rccl":
            # can EVERY rank reach RCCL?  The probe resolves the library and its entry points only (no RCCL call, no
            # bootstrap thread); the decision is a MIN all-reduce over the ranks, so that no rank enters the gloo branch
            # while the others wait in the broadcast of the unique id
            ok = 1 if self._lib.mfmg_hip_rccl_available() == 0 else 0
            why = "" if ok else self._lib.mfmg_hip_last_error().decode(errors="replace")
            if self.n_ranks > 1 and dist.is_initialized():
                flag = torch.tensor([ok], dtype=torch.int32)
                if self.backend == "nccl":
                    flag = flag.cuda()
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
                ok = int(flag.item())
            if not ok:
                print(f"[mfmg_amd] RCCL transport unavailable on at least one rank ({why or 'another rank'}); "
                      f"all ranks fall back to the host transport", flush=True)
                transport = self.transport = "host"
        if transport == "rccl":
            uid = (C.c_ubyte * 128)()
            if self.rank == 0:
                check(self._lib.mfmg_hip_rccl_unique_id(uid))
            if self.n_ranks > 1:
                t = torch.tensor(list(uid), dtype=torch.uint8)
                if self.backend == "nccl":
                    t = t.cuda()
                dist.broadcast(t, src=0, group=group)
                uid = (C.c_ubyte * 128)(*[int(v) for v in t.cpu().tolist()])
            check(self._lib.mfmg_hip_context_use_rccl(ctx.handle, uid))
            self._host_group = None
        else:
            # CPU tensors travel over gloo (a second group when the default one is nccl)
            self._host_group = None if callbacks else (group if self.backend == "gloo" else dist.new_group(backend="gloo"))
            self._sendrecv_cb = _EXCHANGE_FN(self._exchange)
            self._allreduce_cb = _ALLREDUCE_FN(self._allreduce)
            self._allgather_cb = _ALLGATHER_FN(self._allgather)
            check(self._lib.mfmg_hip_context_use_host_transport(
                ctx.handle, C.cast(self._sendrecv_cb, C.c_void_p), C.cast(self._allreduce_cb, C.c_void_p),
                C.cast(self._allgather_cb, C.c_void_p), None))
        ctx._transport = self  # keep the callbacks alive as long as the context

    # -- callbacks of the host transport (invoked from inside the library, on the calling Python thread) --------
    def _exchange(self, user, n, peers, send, recv, count):
        try:
            if self._user_callbacks:
                import numpy as np
                view = lambda p, m: np.ctypeslib.as_array(p, shape=(int(m),))
                self._user_callbacks[0]([int(peers[i]) for i in range(n)], [view(send[i], count[i]) for i in range(n)],
                                        [view(recv[i], count[i]) for i in range(n)])
                return 0
            ops = []
            for i in range(n):
                if count[i] > 0:
                    ops += [dist.P2POp(dist.isend, _host_tensor(send[i], count[i]), int(peers[i]), self._host_group),
                            dist.P2POp(dist.irecv, _host_tensor(recv[i], count[i]), int(peers[i]), self._host_group)]
            if ops:
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
            return 0
        except Exception as e:  # noqa: BLE001 - must not propagate through the C frame
            print(f"[mfmg_amd] halo exchange failed on rank {self.rank}: {e!r}", flush=True)
            return 1

    def _allreduce(self, user, values, n, op):
        try:
            if self._user_callbacks:
                import numpy as np
                self._user_callbacks[1](np.ctypeslib.as_array(values, shape=(int(n),)), int(op))
                return 0
            t = _host_tensor(values, n)
            dist.all_reduce(t, op=dist.ReduceOp.MAX if op == 1 else dist.ReduceOp.SUM, group=self._host_group)
            return 0
        except Exception as e:  # noqa: BLE001
            print(f"[mfmg_amd] all-reduce failed on rank {self.rank}: {e!r}", flush=True)
            return 1

    def _allgather(self, user, src, n, out):
        try:
            if self._user_callbacks:
                import numpy as np
                self._user_callbacks[2](np.ctypeslib.as_array(src, shape=(int(n),)), np.ctypeslib.as_array(out, shape=(int(n) * self.n_ranks,)))
                return 0
            o = _host_tensor(out, n * self.n_ranks)
            dist.all_gather_into_tensor(o, _host_tensor(src, n).clone(), group=self._host_group)
            return 0
        except Exception as e:  # noqa: BLE001
            print(f"[mfmg_amd] all-gather failed on rank {self.rank}: {e!r}", flush=True)
            return 1

    # -- helpers for drivers -----------------------------------------------------------------------
    def layout(self, space: int):
        v = [C.c_int64() for _ in range(4)]
        check(self._lib.mfmg_hip_context_halo_layout(self.ctx.handle, space, *[C.byref(x) for x in v]))
        return tuple(x.value for x in v)

    def space(self, space: int) -> dict:
        out = (C.c_int64 * 8)()
        check(self._lib.mfmg_hip_context_halo_space(self.ctx.handle, space, out))
        keys = ("layer_elems", "n_layers", "owned_begin", "owned_count", "global_begin", "global_layers", "width", "n_spaces")
        return dict(zip(keys, [int(v) for v in out]))

    def box(self, space: int) -> dict:
        """The space per axis (x, y, z): local nodes, owned range, global position and size; `comps` entries per node."""
        out = (C.c_int64 * 16)()
        check(self._lib.mfmg_hip_context_halo_box(self.ctx.handle, space, out))
        v = [int(x) for x in out]
        return {"comps": v[0], "dims": tuple(v[1:4]), "own0": tuple(v[4:7]), "own_n": tuple(v[7:10]), "g0": tuple(v[10:13]),
                "gn": tuple(v[13:16])}

    def space_index(self, space: int):
        """(owned local positions, their global positions, global position of every local entry) of a vector of `space`."""
        b = self.box(space)
        c = b["comps"]

        def ids(lo, n, dims, shift):
            k = torch.arange(lo[2], lo[2] + n[2]).view(-1, 1, 1, 1) + shift[2]
            j = torch.arange(lo[1], lo[1] + n[1]).view(1, -1, 1, 1) + shift[1]
            i = torch.arange(lo[0], lo[0] + n[0]).view(1, 1, -1, 1) + shift[0]
            return ((((k * dims[1] + j) * dims[0] + i) * c) + torch.arange(c).view(1, 1, 1, -1)).reshape(-1)
        return (ids(b["own0"], b["own_n"], b["dims"], (0, 0, 0)), ids(b["own0"], b["own_n"], b["gn"], b["g0"]),
                ids((0, 0, 0), b["dims"], b["gn"], b["g0"]))

    def exchange_volume(self) -> int:
        """Doubles this rank has sent in halo exchanges so far."""
        n = C.c_int64()
        check(self._lib.mfmg_hip_context_exchange_volume(self.ctx.handle, C.byref(n), None))
        return n.value

    def n_overlapped(self) -> int:
        """Exchanges so far that ran on the second stream beside the operator tiles that read no ghost plane."""
        n, m = C.c_int64(), C.c_int64()
        check(self._lib.mfmg_hip_context_exchange_volume(self.ctx.handle, C.byref(n), C.byref(m)))
        return m.value

    def exchange(self, space: int, v: torch.Tensor, reverse: bool = False):
        """One halo exchange of a vector of `space` (the cycle does this by itself; for tests)."""
        check(self._lib.mfmg_hip_context_exchange(self.ctx.handle, space, v.data_ptr(), 1 if reverse else 0))

    def reflect(self, delay_us: float = 0.0):
        """MEASUREMENT: from here on the messages of this rank are mirrored on the device (no partner is involved any more):
        its share of a distributed cycle with a wire that costs nothing -- or `delay_us` of stream time per grouped send/recv and
        collective (the latency of a real group).  For a hierarchy that was set up with the real transport."""
        if delay_us > 0.0:
            check(self._lib.mfmg_hip_context_use_reflecting_transport_delay(self.ctx.handle, float(delay_us)))
        else:
            check(self._lib.mfmg_hip_context_use_reflecting_transport(self.ctx.handle))

    def loopback_time(self, n: int = 32768, reps: int = 50) -> float:
        """Stream time (us) of one loop-back group of the registered transport: send to self + receive from self, n doubles."""
        us = C.c_double()
        check(self._lib.mfmg_hip_context_transport_loopback_time(self.ctx.handle, int(n), int(reps), C.byref(us)))
        return us.value

    def n_exchanges(self) -> int:
        n = C.c_int64()
        check(self._lib.mfmg_hip_context_exchange_count(self.ctx.handle, C.byref(n)))
        return n.value

    def selftest(self, n: int = 1 << 16) -> float:
        """Loop-back send/recv, all-gather and all-reduces through the registered transport; largest deviation."""
        e = C.c_double()
        check(self._lib.mfmg_hip_context_transport_selftest(self.ctx.handle, n, C.byref(e)))
        return e.value

    def comm_ranks(self) -> int:
        """Ranks the transport's own communicator reports (RCCL: ncclCommCount)."""
        n = C.c_int()
        check(self._lib.mfmg_hip_context_transport_ranks(self.ctx.handle, C.byref(n)))
        return n.value

    def name(self) -> str:
        buf = C.create_string_buffer(32)
        check(self._lib.mfmg_hip_context_transport_name(self.ctx.handle, buf, 32))
        return buf.value.decode()

    def owned_dot(self, x: torch.Tensor, y: torch.Tensor, space: int = 1) -> float:
        r = C.c_double()
        check(self._lib.mfmg_hip_context_owned_dot(self.ctx.handle, space, x.data_ptr(), y.data_ptr(), C.byref(r)))
        return r.value

    def owned_norm(self, x: torch.Tensor, space: int = 1) -> float:
        return self.owned_dot(x, x, space) ** 0.5
