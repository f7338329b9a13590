"""Mid degrees: lane kernel (default below M = 23) against the wave / MFMA kernel, 1e5 elements."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hybrid_fem_lssvr_amd import ops
dev = torch.device("cuda:0")
ne = 100000
lo, hi = -1.0, 1.0
nodes = np.arange(ne + 1, dtype=np.float64) * ((hi - lo) / ne) + lo
nodes[-1] = hi
x = torch.as_tensor(nodes, device=dev)
u = torch.sin(np.pi * x)
for M, n in ((9, 16), (13, 24), (15, 28), (17, 32), (19, 36), (22, 40), (23, 42), (27, 50), (33, 64)):
    W = torch.empty((ne, M), dtype=torch.float64, device=dev)
    st = torch.empty(ne, dtype=torch.int32, device=dev)
    out = []
    for name, kw in (("default", {}), ("mfma-wave", dict(solver=ops.SOLVER_PRIMAL_WAVE, work=False))):
        ts = [ops.enhance_profiled(x, u, M, 1e4, n, global_domain=(lo, hi), out=W, status=st, **kw) for _ in range(6)]
        out.append("%s %.1f us" % (name, np.median(ts[2:]) * 1e6))
    print("M=%d n=%d:" % (M, n), ", ".join(out), flush=True)
