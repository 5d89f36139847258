# debug: compare the aggregation hierarchy of a distributed run (gathered levels) with the single-process one
import os, sys
import numpy as np, torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import mfmg_amd as M
from dist_worker import MESHES, PRM
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
per, (cx, cy), material, amg = MESHES[sys.argv[1]]
cells = (cx, cy, per * world)
part = M.SlabPartition(cells, rank, world, length=tuple(c / float(cells[0]) for c in cells))
params = dict(PRM)
params.update({"smoother": {"type": "Chebyshev", "degree": 3, "smoothing_range": 20.0},
               "solver": {"type": "amg", "amg": dict(amg)}, "is preconditioner": False})
ctx = M.Context(); tr = M.HaloTransport(ctx, part, 2)
h = M.Hierarchy(ctx, "HipMatrixFreeMeshEvaluator", part.local_problem(material, "cuda"), params)
gctx = M.Context()
hg = M.Hierarchy(gctx, "HipMatrixFreeMeshEvaluator", M.LaplaceProblem(cells, material, device="cuda", cell_size=part.h), params)
L, Lg = h.coarse_amg_levels(), hg.coarse_amg_levels()
if rank == 0:
    print("levels", [a.shape for a, _, _ in L], [a.shape for a, _, _ in Lg])
    for l, ((A, P, sm), (Ag, Pg, smg)) in enumerate(zip(L, Lg)):
        print(l, "smoother", sm, smg)
        if A.shape == Ag.shape:
            print(l, "A diff", abs(A - Ag).max(), abs(Ag).max(), "nnz", A.nnz, Ag.nnz)
        if P is not None and Pg is not None and P.shape == Pg.shape:
            print(l, "P diff", abs(P - Pg).max(), abs(Pg).max())
dist.destroy_process_group()
