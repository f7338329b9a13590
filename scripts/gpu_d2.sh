#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python - <<'PY' 2>&1 | tee gpurun_out/dual_dbg.log
import numpy as np, torch, sys
sys.path.insert(0,'.')
from hybrid_fem_lssvr_amd import ops
from oracle import lssvr_oracle as orc
dev=torch.device('cuda:0')
t=lambda a: torch.as_tensor(np.ascontiguousarray(a),device=dev)
for M,n in [(9,17),(9,32),(17,24),(24,20),(25,20),(9,33)]:
    ne=5
    nodes=np.linspace(-1,1,ne+1); values=np.sin(np.pi*nodes)+0.1
    W,st=ops.enhance(t(nodes),t(values),M,1e4,n,global_domain=(-1.0,1.0),solver=ops.SOLVER_DUAL)
    W=W.cpu().numpy(); st=st.cpu().numpy()
    Wo=orc.enhance_all_vec(nodes,values,M,1e4,n,global_domain=(-1.0,1.0)) if n>=M-2 else None
    err=orc.rel_l2_coef(W,Wo).max() if Wo is not None else float('nan')
    print(M,n,'status',st.tolist(),'err',err,'maxabs',np.abs(W).max())
PY
