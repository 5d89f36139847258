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


def _run(mode, world, timeout=600, mesh="small", backend="gloo"):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "dist_worker.py"), "--mode", mode, "--mesh", mesh, "--backend", backend]
    env = dict(os.environ, OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
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
@pytest.mark.parametrize("world,mesh", [(2, "deep"), (4, "deep")])
def test_distributed_rccl_one_gpu_per_rank(mfmg_lib, world, mesh):
    """The same checks with one GPU per rank and the native transport (ncclSend / ncclRecv over xGMI on the library's
    stream, ncclAllReduce / ncclAllGather for the setup): needs `world` GPUs in the box, skipped otherwise (the pool's
    test boxes have one; a multi-GPU node runs it).  The worker never counts devices after HIP is initialised: the
    count is taken here, in the parent, without touching the GPU."""
    if torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs, this box has {torch.cuda.device_count()}")
    assert "gpu distributed checks passed" in _run("gpu", world, mesh=mesh, backend="nccl")
