"""One process per GPU: slab decomposition of the mesh along z and the halo transport.

The reference partitions the mesh with p4est and exchanges ghost DoFs inside deal.II
(tests/laplace_matrix_free.hpp:222, MatrixFree::cell_loop); its CUDA path all-gathers the whole
vector in front of every SpMV (source/cuda/utils.cu:363-482).  Here every rank owns a slab of cell
layers; its local mesh is that slab plus one agglomerate (2 cell layers) of each neighbour, numbered
lexicographically, so that ghost planes are contiguous and a halo exchange is two sends and two
receives of one layer -- RCCL point-to-point over xGMI through torch.distributed ("nccl"), issued on
the library's stream.  With the "gloo" backend (CPU tests, several ranks sharing one GPU) the layers
are staged through the host."""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import torch
import torch.distributed as dist

from . import lib as _lib
from .laplace import LaplaceProblem
from .lib import check


class SlabPartition:
    """Owned cell layers [z0, z1) of rank `rank` out of `n_ranks` for a global mesh of `cells`."""

    def __init__(self, cells: Sequence[int], rank: int, n_ranks: int, length: Sequence[float] | float = 1.0):
        self.cells = tuple(int(c) for c in cells)
        assert len(self.cells) == 3
        self.rank, self.n_ranks = int(rank), int(n_ranks)
        cz = self.cells[2]
        if cz % (2 * n_ranks) != 0:
            raise ValueError("the cell layers must split into whole agglomerate layers per rank")
        per = cz // n_ranks
        self.z0, self.z1 = rank * per, (rank + 1) * per
        self.ghost_low = 2 if rank > 0 else 0
        self.ghost_high = 2 if rank + 1 < n_ranks else 0
        self.local_cells = (self.cells[0], self.cells[1], per + self.ghost_low + self.ghost_high)
        self.z_offset = self.z0 - self.ghost_low          # global index of local cell layer 0
        ln = (length,) * 3 if isinstance(length, (int, float)) else tuple(length)
        self.h = tuple(ln[d] / self.cells[d] for d in range(3))
        self.plane = (self.cells[0] + 1) * (self.cells[1] + 1)
        self.n_local_planes = self.local_cells[2] + 1
        # owned node planes: [z0, z1), plus the top plane on the last rank
        self.owned_plane_begin = self.ghost_low
        self.owned_plane_count = per + (1 if rank + 1 == n_ranks else 0)
        self.n_global_dofs = self.plane * (cz + 1)

    def local_problem(self, material: str = "constant", device="cpu") -> LaplaceProblem:
        """Mesh arrays of the local (extended) slab: global Dirichlet nodes carry 1, ghost nodes 2."""
        prob = LaplaceProblem(self.local_cells, material, device=device, dirichlet=False, cell_size=self.h,
                              cell_offset=(0, 0, self.z_offset))
        Nx, Ny = self.cells[0] + 1, self.cells[1] + 1
        k = torch.arange(self.n_local_planes, device=prob.device).view(-1, 1, 1) + self.z_offset
        j = torch.arange(Ny, device=prob.device).view(1, -1, 1)
        i = torch.arange(Nx, device=prob.device).view(1, 1, -1)
        boundary = (i == 0) | (i == Nx - 1) | (j == 0) | (j == Ny - 1) | (k == 0) | (k == self.cells[2])
        lk = torch.arange(self.n_local_planes, device=prob.device).view(-1, 1, 1)
        ghost = (lk < self.owned_plane_begin) | (lk >= self.owned_plane_begin + self.owned_plane_count)
        flags = torch.where(boundary, 1, torch.where(ghost.expand_as(boundary), 2, 0)).to(torch.uint8)
        prob.constrained = flags.reshape(-1).contiguous()
        return prob

    def owned_slice(self) -> slice:
        return slice(self.owned_plane_begin * self.plane, (self.owned_plane_begin + self.owned_plane_count) * self.plane)

    def global_slice(self) -> slice:
        """Where the owned entries live in the global lexicographic vector."""
        return slice(self.z0 * self.plane, (self.z0 + self.owned_plane_count) * self.plane)

    def local_from_global(self, xg: torch.Tensor) -> torch.Tensor:
        """Local (extended) vector cut out of a global lexicographic vector (ghosts filled as well)."""
        lo = self.z_offset * self.plane
        return xg[lo: lo + self.n_local_planes * self.plane].clone()


_EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p)
_ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int)


class HaloTransport:
    """Registers the communicator of a Context: staging buffers + exchange / all-reduce callbacks."""

    def __init__(self, ctx, part: SlabPartition, n_eigenvectors: int, group=None):
        self._lib = _lib.load()
        self.ctx, self.part, self.group = ctx, part, group
        self.backend = dist.get_backend(group) if dist.is_initialized() else "none"
        self.rank, self.n_ranks = part.rank, part.n_ranks
        coarse_layer = (part.cells[0] // 2) * (part.cells[1] // 2) * n_eigenvectors
        self.sizes = {1: part.plane, 2: coarse_layer}
        self.bufs = {}
        for space, n in self.sizes.items():
            b = [torch.zeros(n, dtype=torch.float64, device="cuda") for _ in range(4)]  # send_low/high, recv_low/high
            self.bufs[space] = b
        self._host = {s: [t.cpu().pin_memory() for t in b] for s, b in self.bufs.items()} if self.backend != "nccl" else None
        self._ops, self._streams = {}, {}
        self._exchange_cb = _EXCHANGE_FN(self._exchange)
        self._allreduce_cb = _ALLREDUCE_FN(self._allreduce)
        check(self._lib.mfmg_hip_context_set_communicator(
            ctx.handle, self.rank, self.n_ranks, part.ghost_low, part.ghost_high,
            C.cast(self._exchange_cb, C.c_void_p), C.cast(self._allreduce_cb, C.c_void_p), None))
        for space, b in self.bufs.items():
            check(self._lib.mfmg_hip_context_set_halo_buffers(ctx.handle, space, self.sizes[space],
                                                              b[0].data_ptr(), b[1].data_ptr(), b[2].data_ptr(),
                                                              b[3].data_ptr()))
        ctx._transport = self  # keep the callbacks alive as long as the context
        if self.backend == "nccl":
            # RCCL builds its communicators at the first collective: do it here, with every rank present
            warm = torch.zeros(1, dtype=torch.float64, device="cuda")
            dist.all_reduce(warm, group=group)
            torch.cuda.synchronize()

    # -- callbacks (invoked from inside the library, on the calling Python thread) ------------------
    def _exchange(self, user, space, stream_ptr):
        try:
            send_low, send_high, recv_low, recv_high = self.bufs[space]
            lo, hi = self.rank - 1, self.rank + 1
            if self.backend == "nccl":
                stream = self._streams.get(stream_ptr)
                if stream is None:
                    stream = torch.cuda.ExternalStream(stream_ptr) if stream_ptr else torch.cuda.default_stream()
                    self._streams[stream_ptr] = stream
                ops = self._ops.get(space)
                if ops is None:      # the staging buffers never move: build the operation list once
                    ops = []
                    if lo >= 0:
                        ops += [dist.P2POp(dist.isend, send_low, lo, self.group), dist.P2POp(dist.irecv, recv_low, lo, self.group)]
                    if hi < self.n_ranks:
                        ops += [dist.P2POp(dist.isend, send_high, hi, self.group), dist.P2POp(dist.irecv, recv_high, hi, self.group)]
                    self._ops[space] = ops
                if ops:
                    with torch.cuda.stream(stream):
                        for req in dist.batch_isend_irecv(ops):
                            req.wait()      # orders the library's stream behind the transfers (no host wait)
            else:
                # gloo: stage through the host
                self.ctx.synchronize()
                h = self._host[space]
                ops = []
                if lo >= 0:
                    h[0].copy_(send_low)
                    ops += [dist.P2POp(dist.isend, h[0], lo, self.group), dist.P2POp(dist.irecv, h[2], lo, self.group)]
                if hi < self.n_ranks:
                    h[1].copy_(send_high)
                    ops += [dist.P2POp(dist.isend, h[1], hi, self.group), dist.P2POp(dist.irecv, h[3], hi, self.group)]
                if ops:
                    for req in dist.batch_isend_irecv(ops):
                        req.wait()
                if lo >= 0:
                    recv_low.copy_(h[2])
                if hi < self.n_ranks:
                    recv_high.copy_(h[3])
                torch.cuda.synchronize()
            return 0
        except Exception as e:  # noqa: BLE001 - must not propagate through the C frame
            print(f"[mfmg_amd] halo exchange failed on rank {self.rank}: {e!r}", flush=True)
            return 1

    def _allreduce(self, user, values, n):
        try:
            t = torch.tensor([values[i] for i in range(n)], dtype=torch.float64)
            if self.backend == "nccl":
                t = t.cuda()
            dist.all_reduce(t, group=self.group)
            t = t.cpu()
            for i in range(n):
                values[i] = float(t[i])
            return 0
        except Exception as e:  # noqa: BLE001
            print(f"[mfmg_amd] all-reduce failed on rank {self.rank}: {e!r}", flush=True)
            return 1

    # -- helpers for drivers -----------------------------------------------------------------------
    def layout(self, space: int):
        v = [C.c_int64() for _ in range(4)]
        check(self._lib.mfmg_hip_context_halo_layout(self.ctx.handle, space, *[C.byref(x) for x in v]))
        return tuple(x.value for x in v)

    def owned_dot(self, x: torch.Tensor, y: torch.Tensor) -> float:
        sl = self.part.owned_slice()
        local = self.ctx.dot(x[sl], y[sl])
        t = torch.tensor([local], dtype=torch.float64)
        if self.backend == "nccl":
            t = t.cuda()
        dist.all_reduce(t, group=self.group)
        return float(t.item())

    def owned_norm(self, x: torch.Tensor) -> float:
        return self.owned_dot(x, x) ** 0.5
