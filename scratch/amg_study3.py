import sys, numpy as np
sys.path.insert(0,'oracle'); sys.path.insert(0,'.')
import mfmg_oracle as O
import mfmg_amd as M
import scipy.sparse as sp
for n in (16,17):
    nn=(n-1,)*3
    p=M.LaplaceProblem(nn)
    R=M.host_build_restrictor(p, {'eigensolver': {'number of eigenvectors': 2}}, True)
    Ac=M.host_galerkin(p, R, 'matrix_free').toarray()
    d=np.diag(Ac); S=Ac/np.sqrt(np.outer(d,d))
    w=np.linalg.eigvalsh(S)
    sv=np.linalg.svd(R.toarray(), compute_uv=False)
    print(n, 'n_c', Ac.shape[0], 'eig(D^-1Ac) min', w[:4], 'max', w[-3:], 'R sv min', sv[-4:], 'n tiny sv', (sv<1e-10*sv[0]).sum())
    # where is the null vector?
    ww,V=np.linalg.eigh(S); v=V[:,0]/np.sqrt(d); v/=abs(v).max()
    big=np.nonzero(abs(v)>0.1)[0]; print('   null vector support size', len(big), 'rows', big[:20], 'parities', (big%2).sum())
