import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle')
import mfmg_amd as M, mfmg_oracle as O
n=(4,4,4)
mesh=O.StructuredMesh(n); coef=O.coefficient_table(mesh); con=mesh.constrained_mask()
mf=O.MatrixFreeLaplace(mesh, coef)
ctx=M.Context()
prob=M.LaplaceProblem(n, device='cuda')
P={"eigensolver": {"number of eigenvectors": 2}, "agglomeration": {"nx": 2, "ny": 2, "nz": 2}, "is preconditioner": False, "max levels": 2,
   "smoother": {"type":"Chebyshev","degree":1,"smoothing_range":20.0}}
h=M.Hierarchy(ctx,"HipMatrixFreeMeshEvaluator",prob,P)
deg,lmin,lmax=h.smoother_info(); print(deg,lmin,lmax)
p=O.ChebyshevParams(deg,lmax,lmin)
x0=O.random_initial_guess(mesh.n_dofs, con); b=np.zeros(mesh.n_dofs)
dev=lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
for trial in range(3):
    xs=dev(x0); h.smoother_apply(0, dev(b), xs); ctx.synchronize()
    ref=O.chebyshev_smoother_apply(mf.vmult, mf.diagonal_inverse(), p, b, x0)
    print('smoother err', abs(xs.cpu().numpy()-ref).max())
    y=torch.empty_like(xs); h.operator_apply(0, dev(x0), y); ctx.synchronize()
    print('op err', abs(y.cpu().numpy()-mf.vmult(x0)).max())
R=h.restrictor().to_scipy(); Ac=h.coarse_operator().to_scipy()
nc=R.shape[0]
rc=np.random.rand(nc); out=torch.empty(nc,dtype=torch.float64,device='cuda')
h.coarse_apply(dev(rc), out); ctx.synchronize(); print('coarse err', abs(out.cpu().numpy()-np.linalg.solve(Ac.toarray(), rc)).max())
rf=np.random.rand(125)
h.restrictor_apply(1, dev(rf), out); ctx.synchronize(); print('R err', abs(out.cpu().numpy()-R@rf).max())
of=torch.empty(125,dtype=torch.float64,device='cuda'); h.restrictor_apply(1, dev(rc), of, 1); ctx.synchronize(); print('Rt err', abs(of.cpu().numpy()-R.T@rc).max())
smoother=lambda b,x: O.chebyshev_smoother_apply(mf.vmult, mf.diagonal_inverse(), p, b, x)
ho=O.TwoLevelHierarchy(mf.vmult, smoother, R, O.direct_coarse_solver(Ac), 1, False)
for trial in range(3):
    xg=dev(x0); h.apply(dev(b), xg); ctx.synchronize()
    print('vcycle err', abs(xg.cpu().numpy()-ho.apply(b,x0)).max())
