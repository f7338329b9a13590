#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python - <<'PY' 2>&1 | tee gpurun_out/dual_dbg4.log
import numpy as np, torch, sys, os
sys.path.insert(0,'.')
dev=torch.device('cuda:0')
from hybrid_fem_lssvr_amd import ops
from oracle import lssvr_oracle as orc
t=lambda a: torch.as_tensor(np.ascontiguousarray(a),device=dev)
for M,n,ne in [(9,33,5),(9,33,1),(9,33,2),(9,40,3)]:
    nodes=np.linspace(-1,1,ne+1); values=np.sin(np.pi*nodes)+0.1
    for rep in range(2):
        W,st=ops.enhance(t(nodes),t(values),M,1e4,n,global_domain=(-1.0,1.0),solver=ops.SOLVER_DUAL)
        torch.cuda.synchronize()
        Wo=orc.enhance_all_vec(nodes,values,M,1e4,n,global_domain=(-1.0,1.0))
        print(M,n,ne,'per-element err',orc.rel_l2_coef(W.cpu().numpy(),Wo))
    # tabulated rhs instead of in-kernel sin
    f=t(orc.poisson_rhs(ops.colloc_points(t(nodes),n).cpu().numpy()))
    W,st=ops.enhance(t(nodes),t(values),M,1e4,n,global_domain=(-1.0,1.0),solver=ops.SOLVER_DUAL,rhs_values=f)
    print('   rhs array: ',orc.rel_l2_coef(W.cpu().numpy(),Wo))
PY
