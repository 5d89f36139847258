import sys, numpy as np, time
sys.path.insert(0,'oracle'); sys.path.insert(0,'.')
import mfmg_oracle as O
import mfmg_amd as M
import scipy.sparse as sp, scipy.sparse.linalg as spla, scipy.linalg as sla

def cheb_coefs(deg, lmin, lmax):
    return O.ChebyshevParams(deg, lmax, lmin).step_coefficients()

def est_lmax(A, dinv, n_iter=10):
    n=A.shape[0]; v=O.hashed_initial_guess(n)
    mn,mx=O.dealii_chebyshev_eigen_estimate(lambda z: A@z, dinv, n, n_iter=n_iter, start='hashed')
    return 1.2*mx

def make_amg(levels, degree=2, srange=10.0):
    data=[]
    for (A,P) in levels:
        if P is None:
            data.append((A, None, None, sla.lu_factor(A.toarray())))
        else:
            dinv=1.0/A.diagonal(); lmax=est_lmax(A,dinv); p=O.ChebyshevParams(degree, lmax, lmax/srange)
            data.append((A,P,(dinv,p),None))
    def cycle(l, b):
        A,P,sm,lu=data[l]
        if P is None: return sla.lu_solve(lu, b)
        dinv,p=sm
        x=np.zeros_like(b)
        x=O.chebyshev_smoother_apply(lambda z:A@z, dinv, p, b, x)
        res=A@x-b
        xc=cycle(l+1, P.T@res)
        x=x-P@xc
        x=O.chebyshev_smoother_apply(lambda z:A@z, dinv, p, b, x)
        return x
    return lambda b: cycle(0,b)

for n in (33, 65):
    nn=(n-1,)*3
    mesh=O.StructuredMesh(nn); coef=O.coefficient_table(mesh); con=mesh.constrained_mask()
    p=M.LaplaceProblem(nn)
    R=M.host_build_restrictor(p, {'eigensolver': {'number of eigenvectors': 2}}, True)
    Ac=M.host_galerkin(p, R, 'matrix_free')
    B=np.asarray(R.sum(axis=1)).ravel()
    t=time.perf_counter()
    for smooth in (True,):
        na=(n-1)//2
        rows=np.arange(Ac.shape[0]); e=rows%2
        B2=np.where(e==0, B, 1.0)
        levels=M.host_amg_build(Ac, B2, {'solver': {'amg': {'smooth_prolongator': smooth, 'coarsest_size': 3000}}}, grid_dims=[na,na,na], node_of_row=rows//2, component_of_row=e)
        print(n, 'amg setup', time.perf_counter()-t, [(A.shape[0], A.nnz/A.shape[0]) for A,_ in levels])
        mf=O.MatrixFreeLaplace(mesh, coef); dinv=mf.diagonal_inverse()
        lmax=1.73; lmin=lmax/20
        pch=O.ChebyshevParams(3,lmax,lmin)
        sm=lambda b,x: O.chebyshev_smoother_apply(mf.vmult, dinv, pch, b, x)
        x0=O.random_initial_guess(mesh.n_dofs, con); b=np.zeros(mesh.n_dofs)
        for (deg,sr) in ((2,10.0),(3,10.0),(2,4.0),(1,2.0)):
            amg=make_amg(levels, deg, sr)
            # quality of amg as stand-alone solver for Ac
            rc=np.random.default_rng(0).random(Ac.shape[0]); xc=amg(rc); 
            q=np.linalg.norm(rc-Ac@xc)/np.linalg.norm(rc)
            ho=O.TwoLevelHierarchy(mf.vmult, sm, R, amg, 1, False)
            res,rate,_=O.vcycle_history(ho, mf.vmult, b, x0, n_cycles=10)
            print('   smooth',smooth,'deg',deg,'range',sr,'amg one-cycle residual reduction',q,'outer rate',rate, res[-1])
