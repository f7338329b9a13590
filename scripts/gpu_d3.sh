#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python - <<'PY' 2>&1 | tee gpurun_out/dual_dbg3.log
import numpy as np, torch, sys, os
sys.path.insert(0,'.')
dev=torch.device('cuda:0')
dbg=torch.zeros(64*210,dtype=torch.float64,device=dev)
os.environ['LSSVR_DUAL_DEBUG']=str(dbg.data_ptr())
from hybrid_fem_lssvr_amd import ops
from oracle import lssvr_oracle as orc
t=lambda a: torch.as_tensor(np.ascontiguousarray(a),device=dev)
M,n=9,33
nodes=np.linspace(-1,1,6); values=np.sin(np.pi*nodes)+0.1
W,st=ops.enhance(t(nodes),t(values),M,1e4,n,global_domain=(-1.0,1.0),solver=ops.SOLVER_DUAL)
torch.cuda.synchronize()
d=dbg.cpu().numpy()
K=d[:64*64].reshape(64,64); dr=d[64*64:64*65]; fp=d[64*65:64*66]; pvec=d[64*66:64*67].astype(int); pivstep=d[64*67:64*68].astype(int); y=d[64*68:64*69]; lam=d[64*69:64*70]; F=d[64*70:64*70+64*64].reshape(64,64)
np.save('gpurun_out/dual_dbg3.npy', d)
print('K sym err', np.abs(K[:n,:n]-K[:n,:n].T).max(), 'diag', K[:4,:4])
print('pvec', pvec[:n]); print('pivstep', pivstep[:40])
Ks=K[:n,:n]; rhs=dr[:n]*fp[:n]
x=np.linalg.solve(Ks,rhs); lam_ref=dr[:n]*x
print('lam gpu', lam[:6], 'ref', lam_ref[:6], 'relerr', np.abs(lam[:n]-lam_ref).max()/np.abs(lam_ref).max())
A=Ks.copy(); yy=rhs.copy(); act=np.ones(n,bool)
for j in range(n):
    P=pvec[j]
    best=np.argmax(np.where(act,np.abs(A[:,j]),-1))
    if best!=P: print('step',j,'gpu pivot',P,'|a|',abs(A[P,j]),'best',best,abs(A[best,j]), 'active?',act[P])
    act[P]=False
    m=np.where(act,A[:,j]/A[P,j],0.0)
    A-=np.outer(m,A[P]); yy-=m*yy[P]
print('done')
wv0=d[64*134:64*135]; wv1=d[64*135:64*136]; dl1=d[64*136:64*137]; wv2=d[64*137:64*138]; dl2=d[64*138:64*139]
AR=d[64*140:64*140+64*64].reshape(64,64)[:, :36]
print('wv0 gpu', wv0[:10]); print('A^T lam host', (AR[:n].T@lam[:n])[:10])
print('dl1 max', np.abs(dl1).max(), 'lam max', np.abs(lam).max(), 'dl2 max', np.abs(dl2).max())
print('wv1', wv1[:10]); print('wv2', wv2[:10])
eps_=None
print('W gpu row0', W.cpu().numpy()[0])
Wo=orc.enhance_all_vec(nodes,values,M,1e4,n,global_domain=(-1.0,1.0)); print('W ref row0', Wo[0])

PY
