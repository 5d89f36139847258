import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0,'.'); sys.path.insert(0,'oracle')
import mfmg_amd as M
os.environ.setdefault("MASTER_ADDR","127.0.0.1")
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
cells=(16,12,8*world)
part=M.SlabPartition(cells, rank, world)
deg=int(os.environ.get("AMGDEG","1")); rng_=float(os.environ.get("AMGRANGE","4"))
params={"eigensolver": {"number of eigenvectors": 2}, "agglomeration": {"nx": 2, "ny": 2, "nz": 2},
  "smoother": {"type": "Chebyshev", "degree": 3, "lambda_max": 1.75, "lambda_min": 0.0875},
  "solver": {"type": "amg", "amg": {"coarsest_size": 300, "smoother_degree": deg, "smoothing_range": rng_}}, "is preconditioner": False}
ctx=M.Context(); tr=M.HaloTransport(ctx, part, 2)
h=M.Hierarchy(ctx,"HipMatrixFreeMeshEvaluator", part.local_problem("linear","cuda"), params)
lay,nlay,cb,cc=tr.layout(2)
ncl=lay*nlay
lv=h.coarse_amg_levels()
if rank==0: print('amg levels', [(A.shape, None if P is None else P.shape, ch) for A,P,ch in lv], flush=True)
def cnorm(v):
    t=torch.tensor([float((v[cb*lay:(cb+cc)*lay]**2).sum())],dtype=torch.float64); dist.all_reduce(t); return float(t.sqrt())
g=torch.Generator(device='cuda').manual_seed(5)
bc=torch.rand(ncl,dtype=torch.float64,device='cuda',generator=g)
xc=torch.zeros_like(bc); r=torch.zeros_like(bc)
h.coarse_apply(bc, xc)
h.operator_apply(1, xc, r)
res=(bc-r)
print(rank,'coarse solve: |b|',cnorm(bc),'|b-Ax|',cnorm(res), flush=True)

# manual V-cycle with stage monitoring
prob=part.local_problem("linear")
con=(prob.constrained==1).numpy()
nl=part.plane*part.n_local_planes
rng=np.random.default_rng(0)
xg=rng.random(part.n_global_dofs)
xl=part.local_from_global(torch.from_numpy(xg)).numpy()
x=torch.from_numpy(np.where(con,0.0,xl)).cuda()
b=torch.zeros(nl,dtype=torch.float64,device='cuda')
rr=torch.zeros_like(x)
def fnorm(v): return tr.owned_norm(v)
def resn():
    h.operator_apply(0,x,rr); return fnorm(rr)
for cyc in range(3):
    n0=resn()
    h.smoother_apply(0,b,x); n1=resn()
    res=torch.zeros_like(x); h.operator_apply(0,x,res)  # res = A x - 0
    bc2=torch.zeros(ncl,dtype=torch.float64,device='cuda'); h.restrictor_apply(1,res,bc2)
    xc2=torch.zeros_like(bc2); h.coarse_apply(bc2,xc2)
    # coarse residual
    rc=torch.zeros_like(bc2); h.operator_apply(1,xc2,rc)
    corr=torch.zeros_like(x); h.restrictor_apply(1,xc2,corr,1)
    ctx.synchronize()
    x-=corr; n2=resn()
    h.smoother_apply(0,b,x); n3=resn()
    c1=cnorm(bc2); c2=cnorm(bc2-rc)
    if rank==0: print(f'cycle {cyc}: start {n0:.3e} presmooth {n1:.3e} |bc| {c1:.3e} |bc-Ac xc| {c2:.3e} after corr {n2:.3e} post {n3:.3e}', flush=True)
dist.destroy_process_group()
