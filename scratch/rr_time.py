"""b_c = R (A x - b) in one pass at the bench size: ms per launch (HIP events), the tile form against the row-wise kernel
(MFMG_RR_KERNEL=rows), and the two against each other bit for bit via a file.  usage: rr_time.py [cells] [out.npy]"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mfmg_amd as M
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx = M.Context()
prob = M.LaplaceProblem((cells,) * 3, "constant", device="cuda")
params = {"eigensolver": {"number of eigenvectors": 2}, "agglomeration": {"partitioner": "block", "nx": 2, "ny": 2, "nz": 2},
          "smoother": {"type": "Chebyshev", "degree": 3, "smoothing_range": 20.0},
          "solver": {"type": "pcg", "n_iterations": 1}, "is preconditioner": False, "max levels": 2}
h = M.Hierarchy(ctx, "HipMatrixFreeMeshEvaluator", prob, params)
nf, nc = h.level_size(0), h.level_size(1)
g = torch.Generator(device="cuda"); g.manual_seed(3)
x = torch.rand(nf, dtype=torch.float64, device="cuda", generator=g); b = torch.rand(nf, dtype=torch.float64, device="cuda", generator=g)
y = torch.empty(nc, dtype=torch.float64, device="cuda")
for _ in range(5): h.restrict_residual(x, b, y)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(5):
    e0.record()
    for _ in range(20): h.restrict_residual(x, b, y)
    e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 20)
print(f"{os.environ.get('MFMG_RR_KERNEL', 'tile')} layers {os.environ.get('MFMG_RR_TILE_LAYERS', 'auto')}: {cells}^3 cells, {min(ts)*1e3:.1f} us per launch (min of 5 x 20), "
      f"{(16. * nf + 8. * nc) / min(ts) / 1e9 * 1e3:.0f} GB/s on x, b and b_c")
if len(sys.argv) > 2:
    out = sys.argv[2]
    if os.path.exists(out):
        ref = np.load(out); print("bitwise equal to", out, ":", bool(np.array_equal(ref, y.cpu().numpy())), "max diff", float(np.abs(ref - y.cpu().numpy()).max()))
    else:
        np.save(out, y.cpu().numpy())
