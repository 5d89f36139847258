import sys, numpy as np, time
sys.path.insert(0,'oracle'); sys.path.insert(0,'.')
import mfmg_oracle as O, oracle_native as N, mfmg_amd as M
for cells in (64, 128):
    nn=(cells,)*3
    mesh=O.StructuredMesh(nn); coef=O.coefficient_table(mesh); con=mesh.constrained_mask()
    p=M.LaplaceProblem(nn)
    R=M.host_build_restrictor(p, {'eigensolver': {'number of eigenvectors': 2}}, True)
    Ac=M.host_galerkin(p, R, 'matrix_free')
    rows=np.arange(Ac.shape[0]); B=np.where(rows%2==0, np.asarray(R.sum(axis=1)).ravel(), 1.0)
    na=cells//2
    levels=M.host_amg_build(Ac, B, {}, grid_dims=[na,na,na], node_of_row=rows//2, component_of_row=rows%2)
    print(cells, [(A.shape[0], round(A.nnz/A.shape[0],1)) for A,_ in levels], flush=True)
    ests=[]
    for (A,P) in levels:
        if P is None: ests.append(None)
        else:
            mn,mx=O.dealii_chebyshev_eigen_estimate(lambda z:A@z, 1/A.diagonal(), A.shape[0], n_iter=10, start='hashed'); ests.append(1.2*mx)
    cd=mesh.cell_dofs(); mf=O.MatrixFreeLaplace(mesh,coef); dinv=mf.diagonal_inverse()
    x0=O.random_initial_guess(mesh.n_dofs, con); b=np.zeros(mesh.n_dofs)
    lmax=1.70; lmin=lmax/20
    for (deg,sr) in ((2,10.0),(1,2.0),(1,4.0),(2,4.0),(3,10.0)):
        lv=[(A,P,None if P is None else (deg, e/sr, e)) for (A,P),e in zip(levels,ests)]
        x,h=N.vcycles(nn, mesh.h, cd, coef, con, dinv, 3, lmin, lmax, R, Ac, 0, b, x0, 10, amg_levels=lv)
        print('   amg smoother deg',deg,'range',sr,'outer rate', h[-1]/h[-2], 'res10', h[-1], flush=True)
    x,h=N.vcycles(nn, mesh.h, cd, coef, con, dinv, 3, lmin, lmax, R, Ac, 300, b, x0, 10)
    print('   exact-ish (PCG 300) rate', h[-1]/h[-2], h[-1])
