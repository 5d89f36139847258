import sys, numpy as np, time
sys.path.insert(0,'oracle'); sys.path.insert(0,'.')
import mfmg_oracle as O
import mfmg_amd as M
import scipy.sparse as sp, scipy.sparse.linalg as spla, scipy.linalg as sla
n=32
nn=(n-1,)*3
mesh=O.StructuredMesh(nn); coef=O.coefficient_table(mesh)
p=M.LaplaceProblem(nn)
R=M.host_build_restrictor(p, {'eigensolver': {'number of eigenvectors': 2}}, True)
Ac=M.host_galerkin(p, R, 'matrix_free').tocsr()
d=Ac.diagonal(); print('diag e0/e1 sample', d[:8])
Dm=sp.diags(1/np.sqrt(d)); S=Dm@Ac@Dm
lam=spla.eigsh(S,k=1,which='LA',return_eigenvectors=False)[0]; lmin=spla.eigsh(S,k=1,sigma=0,which='LM',return_eigenvectors=False)[0]
print('true lmax(D^-1 Ac)', lam, 'lmin', lmin, 'gershgorin', (abs(Ac).sum(axis=1).A1/d).max())
for it in (8,10,20):
    mn,mx=O.dealii_chebyshev_eigen_estimate(lambda z: Ac@z, 1/d, Ac.shape[0], n_iter=it, start='hashed'); print('cg est', it, mx, mx*1.2)
# block structure
row=Ac[2*1000].toarray().ravel(); nz=np.nonzero(row)[0]; print('row e0: cols parity counts', (nz%2==0).sum(), (nz%2==1).sum(), 'vals e0', row[nz[nz%2==0]][:6], 'vals e1', row[nz[nz%2==1]][:6])
row=Ac[2*1000+1].toarray().ravel(); nz=np.nonzero(row)[0]; print('row e1: vals e0', row[nz[nz%2==0]][:6], 'vals e1', row[nz[nz%2==1]][:6], 'diag', row[2001])
