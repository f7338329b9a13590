#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_enhance_dual.py -m gpu -q -x -k "G5 or 33-33 or 33-38 or size_class" 2>&1 | grep -E "^E |passed|failed" | head -40
timeout -k 10 300 python - <<'PY' 2>&1 | tail -30
import numpy as np, torch, sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from hybrid_fem_lssvr_amd import ops
from oracle import lssvr_oracle as orc
from oracle import closed_form_mp as cf
dev=torch.device('cuda:0')
t=lambda a: torch.as_tensor(np.ascontiguousarray(a),device=dev)
g=dict(np.load('tests/golden/G5_ne24_M33_n64.npz'))
ne,M,n,gamma=int(g['ne']),int(g['M']),int(g['n']),float(g['gamma'])
nodes=np.linspace(-1,1,ne+1); values=np.sin(np.pi*nodes); values[g['elements']]=g['values_sel'][:,0]; values[g['elements']+1]=g['values_sel'][:,1]
W,st=ops.enhance(t(nodes),t(values),M,gamma,n,global_domain=(-1.0,1.0),solver=ops.SOLVER_DUAL)
W=W.cpu().numpy()
print('G5 truth', orc.rel_l2_coef(W[g['elements']],g['coef_truth']), 'ref', orc.rel_l2_coef(W[g['elements']],g['coef_ref']))
Wp,_=ops.enhance(t(nodes),t(values),M,gamma,n,global_domain=(-1.0,1.0)); print('vs primal max', orc.rel_l2_coef(W,Wp.cpu().numpy()).max())
nodes=np.linspace(-1,1,38); values=orc.fem_p1_solve(nodes)
for M,n in [(33,31),(33,33),(33,38),(32,29)]:
    W,st=ops.enhance(t(nodes),t(values),M,1e4,n,global_domain=(-1.0,1.0),solver=ops.SOLVER_DUAL)
    sel=[0,1,18,36]; tr=cf.truth_all(nodes,values,M,1e4,n,orc.poisson_rhs,(-1.0,1.0),sel)
    print(M,n,'dual vs truth',orc.rel_l2_coef(W.cpu().numpy()[sel],tr))
PY
