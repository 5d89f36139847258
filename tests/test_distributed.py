"""N > 1 path: world_size-2/3 gloo runs of tests/dist_worker.py (CPU: partition + host setup + exchange
protocol; GPU: the library path with all ranks sharing one card)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

import mfmg_amd as M

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(mode, world, timeout=600, mesh="small", backend="gloo", grid="", low_ghost=2):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "dist_worker.py"), "--mode", mode, "--mesh", mesh, "--backend", backend, "--grid", grid,
           "--low-ghost", str(low_ghost)]
    env = dict(os.environ, OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    if res.returncode != 0:
        # the first traceback of a rank (the tail of stderr is the launcher's summary)
        at = res.stderr.find("Traceback")
        raise AssertionError(res.stdout[-1500:] + (res.stderr[at:at + 3000] if at >= 0 else res.stderr[-3000:]))
    return res.stdout


def test_slab_partition_geometry(mfmg_lib):
    cells = (8, 6, 12)
    parts = [M.SlabPartition(cells, r, 3) for r in range(3)]
    plane = 9 * 7
    covered = np.zeros(plane * 13, dtype=int)
    for p in parts:
        covered[p.global_slice()] += 1
        assert p.local_cells[2] == 4 + p.ghost_low + p.ghost_high
        prob = p.local_problem("constant")
        flags = prob.constrained.numpy().reshape(p.n_local_planes, 7, 9)
        lo, cnt = p.owned_plane_begin, p.owned_plane_count
        assert (flags[lo:lo + cnt, 1:-1, 1:-1] != 2).all()                 # owned nodes are never ghosts
        assert (flags[:lo, 1:-1, 1:-1] == 2).all() and (flags[lo + cnt:, 1:-1, 1:-1] == 2).all()
        assert (flags[:, 0, :] == 1).all() and (flags[:, :, -1] == 1).all()  # global Dirichlet faces
    assert (covered == 1).all()                                           # every DoF has exactly one owner
    assert parts[0].ghost_low == 0 and parts[2].ghost_high == 0 and parts[1].ghost_low == parts[1].ghost_high == 2
    with pytest.raises(ValueError):
        M.SlabPartition((8, 8, 10), 0, 2)


def test_box_partition_geometry(mfmg_lib):
    """2 x 2 x 2 boxes: one owner per DoF, ghost layers exactly towards the neighbours, and the volume of a fine exchange at
    the shape of BASELINE configs[3] (8 ranks, 256^3 cells each) against slabs of the same mesh."""
    cells = (8, 12, 16)
    grid = (2, 2, 2)
    covered = np.zeros(9 * 13 * 17, dtype=int)
    for r in range(8):
        p = M.BoxPartition(cells, r, grid)
        assert p.coord == (r % 2, (r // 2) % 2, r // 4)
        covered[p.owned_global_index().numpy()] += 1
        for d in range(3):
            assert p.ghost_lo[d] == (2 if p.coord[d] > 0 else 0) and p.ghost_hi[d] == (2 if p.coord[d] == 0 else 0)
            assert p.local_cells[d] == cells[d] // 2 + 2
        flags = p.local_problem("constant").constrained.numpy()
        own = np.zeros(p.n_local_dofs, bool); own[p.owned_local_index().numpy()] = True
        assert (flags[own] != 2).all() and (flags[~own] != 0).all()
        assert np.array_equal(p.local_global_index().numpy()[p.owned_local_index().numpy()], p.owned_global_index().numpy())
    assert (covered == 1).all()
    # two agglomerates of every lower neighbour (low_ghost_cells=4: three ghost node planes on either side of a box, what a
    # sweep of three smoother terms needs): the same owners, deeper ghosts below
    covered[:] = 0
    for r in range(8):
        p = M.BoxPartition(cells, r, grid, low_ghost_cells=4)
        covered[p.owned_global_index().numpy()] += 1
        for d in range(3):
            assert p.ghost_lo[d] == (4 if p.coord[d] > 0 else 0) and p.ghost_hi[d] == (2 if p.coord[d] == 0 else 0)
            assert p.own0[d] == p.ghost_lo[d] and p.local_nodes[d] - p.own0[d] - p.own_n[d] == (3 if p.coord[d] == 0 else 0)
        flags = p.local_problem("constant").constrained.numpy()
        own = np.zeros(p.n_local_dofs, bool); own[p.owned_local_index().numpy()] = True
        assert (flags[own] != 2).all() and (flags[~own] != 0).all()
    assert (covered == 1).all()
    assert M.box_grid(2) == (1, 1, 2) and M.box_grid(4) == (1, 2, 2) and M.box_grid(8) == (2, 2, 2) and M.box_grid(3) == (1, 1, 3)
    with pytest.raises(ValueError):
        M.BoxPartition((10, 8, 8), 0, (2, 1, 1))
    # doubles one rank sends per fine exchange, 512^3 cells on 8 ranks: three faces of 256^2 (+ three edges and a corner)
    # against two planes of 513^2
    box = M.BoxPartition((512,) * 3, 0, grid)
    slab = M.SlabPartition((512,) * 3, 3, 8)
    assert box.exchange_doubles() == 3 * 256 * 256 + 3 * 256 + 1 and slab.exchange_doubles() == 2 * 513 * 513
    assert slab.exchange_doubles() / box.exchange_doubles() > 2.6


@pytest.mark.parametrize("world,grid,low_ghost", [(2, "2x1x1", 2), (4, "2x2x1", 2), (4, "1x2x2", 2), (8, "2x2x2", 2), (2, "1x1x2", 4), (8, "2x2x2", 4)])
def test_box_construction_cpu_gloo(mfmg_lib, world, grid, low_ghost):
    """Box partition: host setup on the local boxes + the all-neighbours exchange restated in numpy over gloo;
    with one and with two agglomerates of the lower neighbours in the local mesh."""
    assert "cpu box checks passed" in _run("cpu_box", world, grid=grid, low_ghost=low_ghost)


@pytest.mark.parametrize("world", [2, 3])
def test_distributed_construction_cpu_gloo(mfmg_lib, world):
    assert "cpu distributed checks passed" in _run("cpu", world)


@pytest.mark.parametrize("world", [2, 3])
def test_distributed_setup_protocol_cpu_gloo(mfmg_lib, world):
    """The probing protocol that couples the aggregation levels across ranks (amg_device_setup.hip), restated in numpy on
    CPU ranks: prolongator and Galerkin operator of two consecutive levels from operator applications and halo exchanges
    alone, equal to the rows of the globally formed matrices."""
    assert "protocol checks passed" in _run("protocol", world)


@pytest.mark.gpu
@pytest.mark.parametrize("world,mesh", [(2, "small"), (3, "small"), (2, "wide"), (2, "deep"), (4, "deep"), (2, "deep01")])
def test_distributed_library_path_shared_gpu(mfmg_lib, world, mesh):
    assert "gpu distributed checks passed" in _run("gpu", world, mesh=mesh)


@pytest.mark.gpu
@pytest.mark.parametrize("world,grid,mesh", [(2, "1x1x2", "cube11"), (4, "2x1x2", "cube11"), (4, "2x2x1", "boxwide"), (4, "1x2x2", "cube"),
                                             (2, "2x1x1", "boxnarrow")])
def test_box_decomposition_with_two_ghost_agglomerates_below_shared_gpu(mfmg_lib, world, grid, mesh):
    """BoxPartition(low_ghost_cells=4): the local mesh holds two agglomerates of every lower neighbour, so that the whole
    Chebyshev(3) smoother of a rank is ONE sweep (x exchanged once, three ghost planes deep; asserted in the worker for the
    constant material) -- every operator of the cycle and the 20-cycle history as in the test below."""
    assert "gpu distributed checks passed; grid " + grid in _run("gpu", world, mesh=mesh, grid=grid, timeout=900, low_ghost=4)


@pytest.mark.gpu
@pytest.mark.parametrize("world,grid,mesh", [(2, "2x1x1", "small"), (4, "2x2x1", "small"), (4, "2x2x1", "cube"), (4, "2x1x2", "cube11"),
                                             (4, "1x2x2", "cube"), (4, "2x2x1", "boxwide"), (4, "2x1x2", "boxwide")])
def test_box_decomposition_library_path_shared_gpu(mfmg_lib, world, grid, mesh):
    """Boxes instead of slabs: every pair of split axes, with the aggregation levels distributed along them --
    each operator against the single-process hierarchy, 20-cycle history == single process == oracle to 1e-10, and the
    doubles a fine exchange moves."""
    assert "gpu distributed checks passed; grid " + grid in _run("gpu", world, mesh=mesh, grid=grid, timeout=900)


@pytest.mark.gpu
def test_rccl_transport_on_one_gpu(mfmg_lib):
    """The native transport (ncclSend / ncclRecv / ncclAllGather / ncclAllReduce resolved from librccl at run time) on a
    one-rank communicator: a loop-back send/recv of 512 KiB, the collectives, and a hierarchy built and applied with
    the communicator registered (one rank: no neighbours, every exchange a no-op)."""
    ctx = M.Context()
    part = M.SlabPartition((8, 8, 8), 0, 1)
    tr = M.HaloTransport(ctx, part, transport="rccl")
    assert tr.name() == "rccl"
    assert tr.selftest(1 << 16) == 0.0
    params = {"eigensolver": {"number of eigenvectors": 2}, "agglomeration": {"nx": 2, "ny": 2, "nz": 2},
              "smoother": {"type": "Chebyshev", "degree": 2, "smoothing_range": 20.0}, "solver": {"type": "amg"}}
    h = M.Hierarchy(ctx, "HipMatrixFreeMeshEvaluator", part.local_problem("linear", "cuda"), params)
    x = torch.rand(h.level_size(0), dtype=torch.float64, device="cuda")
    h.apply(torch.zeros_like(x), x)
    ctx.synchronize()
    assert torch.isfinite(x).all()


@pytest.mark.gpu
@pytest.mark.parametrize("world,mesh,grid", [(2, "deep", ""), (4, "deep", ""), (4, "cube", "2x2x1"), (8, "cube", "2x2x2")])
def test_distributed_rccl_one_gpu_per_rank(mfmg_lib, world, mesh, grid):
    """The same checks with one GPU per rank and the native transport (ncclSend / ncclRecv over xGMI on the library's
    stream, ncclAllReduce / ncclAllGather for the setup): needs `world` GPUs in the box, skipped otherwise (the pool's
    test boxes have one; a multi-GPU node runs it).  The worker never counts devices after HIP is initialised: the
    count is taken here, in the parent, without touching the GPU."""
    if torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs, this box has {torch.cuda.device_count()}")
    assert "gpu distributed checks passed" in _run("gpu", world, mesh=mesh, backend="nccl", grid=grid)
