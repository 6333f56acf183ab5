import numpy as np, sys
sys.path.insert(0,'/root/repo')
import interiorpointmethod_amd as ipm
for m in (1,2,3,4,5,8,16,17,32):
    rng=np.random.default_rng(m)
    M=rng.standard_normal((m,m+10)); B=M@M.T+0.1*np.eye(m); rhs=rng.standard_normal(m)
    with ipm.IpmSolver(np.eye(m,1),np.zeros(m),np.zeros(1)) as sv:
        z,nfix=sv.solve_linear(B,rhs); L=sv.get_factor()
    Lr=np.linalg.cholesky(B)
    err=np.abs(L-Lr)
    print(m,'nfix',nfix,'Lerr',err.max()/np.abs(Lr).max(), 'bad cols', np.where(err.max(axis=0)>1e-8)[0][:20])
