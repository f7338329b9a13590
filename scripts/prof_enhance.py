"""Run only the enhancement kernel a few times (target for rocprofv3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hybrid_fem_lssvr_amd import ops

ne, M, n = (int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else (100000, 9, 16)))
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
solver = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dev = torch.device("cuda:0")
x = torch.linspace(-1, 1, ne + 1, dtype=torch.float64, device=dev)
u = torch.sin(np.pi * x)
W = torch.empty((ne, M), dtype=torch.float64, device=dev)
st = torch.empty(ne, dtype=torch.int32, device=dev)
for _ in range(reps):
    ops.enhance(x, u, M, 1e4, n, out=W, status=st, global_domain=(-1.0, 1.0), solver=solver)
torch.cuda.synchronize()
print("done", ne, M, n, int(st.sum()))
