import sys, numpy as np, time
sys.path.insert(0,'oracle'); sys.path.insert(0,'.')
import mfmg_oracle as O, oracle_native as N
import mfmg_amd as M
import scipy.sparse.linalg as spla
for n in (32, 64):
    nn=(n-1,)*3
    mesh=O.StructuredMesh(nn); coef=O.coefficient_table(mesh); con=mesh.constrained_mask()
    p=M.LaplaceProblem(nn)
    R=M.host_build_restrictor(p, {'eigensolver': {'number of eigenvectors': 2}}, True)
    Ac=M.host_galerkin(p, R, 'matrix_free')
    mf=O.MatrixFreeLaplace(mesh, coef); dinv=mf.diagonal_inverse()
    lmax=1.73; lmin=lmax/20
    x0=O.random_initial_guess(mesh.n_dofs, con); b=np.zeros(mesh.n_dofs)
    for K in (2,5,10,20,40,80,200):
        x,h=N.vcycles(nn, mesh.h, mesh.cell_dofs(), coef, con, dinv, 3, lmin, lmax, R, Ac, K, b, x0, 12)
        print(n, 'PCG', K, 'rate', h[-1]/h[-2], 'res12', h[-1], flush=True)
