"""Dual solver timing (hipExt-stamped launches) and agreement with the primal kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hybrid_fem_lssvr_amd import ops
dev = "cuda:0"
for ne, M, n in ((100000, 33, 64), (100000, 33, 40), (100000, 20, 50), (100000, 9, 16), (100000, 20, 30), (100000, 17, 32), (100000, 12, 12)):
    x = torch.linspace(-1, 1, ne + 1, dtype=torch.float64, device=dev); u = torch.sin(np.pi * x)
    W = torch.empty((ne, M), dtype=torch.float64, device=dev)
    ts = sorted(ops.enhance_profiled(x, u, M, 1e4, n, global_domain=(-1.0, 1.0), out=W, solver=ops.SOLVER_DUAL) for _ in range(5))
    Wp, _ = ops.enhance(x, u, M, 1e4, n, global_domain=(-1.0, 1.0))
    d = ((W - Wp).norm(dim=1) / Wp.norm(dim=1)).max().item()
    print(f"dual M={M} n={n}: median {ts[2]*1e3:.3f} ms = {ne/ts[2]:.3e} el/s; max rel diff to primal {d:.2e}", flush=True)
