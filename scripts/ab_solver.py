"""Lane mapping (solver 0) vs wave mapping (solver 2) of the same algorithm, per M."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hybrid_fem_lssvr_amd import ops
for arg in sys.argv[1:]:
    ne, M, n = (int(v) for v in arg.split(","))
    x = torch.linspace(-1, 1, ne + 1, dtype=torch.float64, device="cuda:0")
    u = torch.sin(np.pi * x)
    W = torch.empty((ne, M), dtype=torch.float64, device="cuda:0")
    out = []
    for solver in (0, 2):
        ts = sorted(ops.enhance_profiled(x, u, M, 1e4, n, global_domain=(-1.0, 1.0), out=W, solver=solver)
                    for _ in range(15))
        out.append(ts[len(ts) // 2] * 1e6)
    print(f"ne={ne} M={M} n={n}: lane {out[0]:9.1f} us   wave {out[1]:9.1f} us   ratio {out[1]/out[0]:.2f}")
