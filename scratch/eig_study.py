import sys, numpy as np
sys.path.insert(0,'oracle'); sys.path.insert(0,'.')
import mfmg_oracle as O
import scipy.sparse as sp, scipy.sparse.linalg as spla
for n in (16, 32):
    nn=(n-1,)*3
    mesh=O.StructuredMesh(nn); coef=O.coefficient_table(mesh); con=mesh.constrained_mask()
    A=O.assemble_csr(mesh, coef); d=A.diagonal().copy(); 
    mf=O.MatrixFreeLaplace(mesh, coef); dinv=mf.diagonal_inverse()
    # true lambda_max of D^-1 A (mf semantics)
    Amf=sp.csr_matrix(A); 
    Dm=sp.diags(np.sqrt(dinv))
    S=Dm@A@Dm
    lam=spla.eigsh(S, k=1, which='LA', return_eigenvectors=False)[0]
    e_de=O.dealii_chebyshev_eigen_estimate(mf.vmult, dinv, mesh.n_dofs)
    # random start
    def est(rhs, n_iter=8):
        x=np.zeros_like(rhs); r=rhs.copy(); z=dinv*r; p=z.copy(); rz=r@z; al=[];be=[]
        for _ in range(n_iter):
            Ap=mf.vmult(p); a=rz/(p@Ap); x+=a*p; r-=a*Ap; al.append(a); z=dinv*r; rzn=r@z; b=rzn/rz; be.append(b); rz=rzn; p=z+b*p
        m=len(al); T=np.zeros((m,m))
        for i in range(m):
            T[i,i]=1/al[i]+(be[i-1]/al[i-1] if i>0 else 0)
            if i+1<m: T[i,i+1]=T[i+1,i]=np.sqrt(be[i])/al[i]
        ev=np.linalg.eigvalsh(T); return ev[0],ev[-1]
    rng=np.random.default_rng(0); v=rng.random(mesh.n_dofs)-0.5
    h=((np.arange(mesh.n_dofs)*2654435761)%4294967296)/4294967296.0-0.5
    print(n,'true lmax',lam,'dealii-lex est',e_de,'random est', est(v), 'hash est', est(h), 'random16', est(v,16))
