"""Run only the shared-operator kernel a few times (target for rocprofv3 --pmc / --kernel-trace).
usage: prof_shared.py ne,M,n [reps] [wide|narrow]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hybrid_fem_lssvr_amd import ops
ne, M, n = (int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else (10000000, 9, 16)))
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
domain = sys.argv[3] if len(sys.argv) > 3 else "narrow"
dev = torch.device("cuda:0")
lo, hi = (-ne / 24.0, ne / 24.0) if domain == "wide" else (-1.0, 1.0)
nodes = np.arange(ne + 1, dtype=np.float64) * ((hi - lo) / ne) + lo
nodes[-1] = hi
x = torch.as_tensor(nodes, device=dev)
u = torch.sin(np.pi * x)
W = torch.empty((ne, M), dtype=torch.float64, device=dev)
st = torch.empty(ne, dtype=torch.int32, device=dev)
op = ops.build_shared_operator((hi - lo) / ne, M, 1e4, n, device=dev)
for _ in range(reps):
    ops.enhance_shared(x, u, op, M, n, global_domain=(lo, hi), out=W, status=st)
torch.cuda.synchronize()
print("done", ne, M, n, "fallback", int(st.sum()))
