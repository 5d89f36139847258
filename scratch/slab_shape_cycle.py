"""Cycle time of the local problem of a middle rank of an N-rank z-slab run, without transport (ghost planes keep
their start values): what the kernels cost on that slab shape.  usage: slab_shape_cycle.py gx gy gz rank n_ranks"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mfmg_amd as M
gx, gy, gz, rank, nr = (int(v) for v in sys.argv[1:6])
ctx = M.Context()
part = M.SlabPartition((gx, gy, gz), rank, nr)
prob = part.local_problem("constant", device="cuda")
params = {"eigensolver": {"number of eigenvectors": 2}, "agglomeration": {"partitioner": "block", "nx": 2, "ny": 2, "nz": 2},
          "smoother": {"type": "Chebyshev", "degree": 3, "smoothing_range": 20.0, "n_smoothing_steps": 1},
          "solver": {"type": "amg", "amg": {"smoother_degree": 1, "smoothing_range": 4.0, "n_cycles": 1, "aggregate_block": 2}},
          "is preconditioner": False, "max levels": 2}
t = time.perf_counter()
h = M.Hierarchy(ctx, "HipMatrixFreeMeshEvaluator", prob, params)
ctx.synchronize()
print("setup s", time.perf_counter() - t, "tile", h.operator_tile(), "local dofs", h.level_size(0), h.level_size(1))
x = torch.rand(h.level_size(0), dtype=torch.float64, device="cuda"); b = torch.zeros_like(x)
for tz in [int(v) for v in sys.argv[6:]]:       # optional: layers per operator tile to compare with the default choice
    h.set_operator_tile(4, 3, tz)
    for _ in range(3): h.apply(b, x)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): h.apply(b, x)
    torch.cuda.synchronize(); print(f"tz {tz}: ms/cycle {(time.perf_counter() - t) / 10 * 1e3:.3f}")
h.set_operator_tile(0, 0, 0)
for _ in range(3): h.apply(b, x)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(10): h.apply(b, x)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
owned = part.plane * part.owned_plane_count
print(f"ms/cycle {dt*1e3:.3f}  owned DoF/s {owned/dt:.4e}")
ctx.profile_enable(True)
for _ in range(3): h.apply(b, x)
torch.cuda.synchronize()
for k in ("mf_laplace_kernel", "csr_spmv_kernel"):
    l, ms, by = ctx.profile_query(k); print(k, l // 3, "launches/cycle", ms / 3, "ms/cycle")
