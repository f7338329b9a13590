#!/bin/bash
for R in 1 2 3 5; do
LSSVR_DUAL_REFINE=$R timeout -k 10 300 python - <<'PY' 2>&1 | tail -3
import numpy as np, torch, sys, os
sys.path.insert(0,'.')
from hybrid_fem_lssvr_amd import ops
from oracle import lssvr_oracle as orc
dev=torch.device('cuda:0')
t=lambda a: torch.as_tensor(np.ascontiguousarray(a),device=dev)
out=[]
for (ne,M,n,lo,hi) in [(24,33,64,-1,1),(2000,33,64,-1,1),(100000,33,64,-1,1),(37,33,33,-1,1),(37,33,38,-1,1),(8,5,5,-1,1),(4096,9,16,-1,1)]:
    nodes=np.linspace(lo,hi,ne+1); values=np.sin(np.pi*nodes)
    W,st=ops.enhance(t(nodes),t(values),M,1e4,n,global_domain=(float(lo),float(hi)),solver=ops.SOLVER_DUAL)
    Wp,_=ops.enhance(t(nodes),t(values),M,1e4,n,global_domain=(float(lo),float(hi)))
    e=orc.rel_l2_coef(W.cpu().numpy(),Wp.cpu().numpy())
    out.append("(%d,%d,%d) max %.1e med %.1e nfail %d"%(ne,M,n,e.max(),np.median(e),int(st.sum())))
print('refine',' | '.join(out))
PY
done
