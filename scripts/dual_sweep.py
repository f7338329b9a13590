"""Sweep of the dual solver over every (M, n) it accepts with more than 32 rows (the wave-per-element kernel of
round 3) and a sample below, against the primal kernels on the same non-uniform mesh: worst relative L2
difference per regime.  usage: dual_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hybrid_fem_lssvr_amd import ops
dev = "cuda:0"
rng = np.random.default_rng(5)
ne = 96
nodes = np.cumsum(np.concatenate([[-0.7], rng.uniform(0.02, 0.09, ne)]))
values = np.sin(np.pi * nodes) + 0.01 * rng.standard_normal(ne + 1)
x, u = torch.as_tensor(nodes, device=dev), torch.as_tensor(values, device=dev)
gd = (nodes[0], nodes[-1])
worst = {}
bad = []
count = 0
for M in range(3, 34):
    for n in list(range(max(M - 2, 2), 65)):
        if max(M, n) <= 32 and (n % 5):          # below 33 rows: a sample
            continue
        Wd, sd = ops.enhance(x, u, M, 1e4, n, global_domain=gd, solver=ops.SOLVER_DUAL)
        Wp, sp = ops.enhance(x, u, M, 1e4, n, global_domain=gd)
        d = ((Wd - Wp).norm(dim=1) / Wp.norm(dim=1)).max().item()
        count += 1
        if int(sd.sum()) or int(sp.sum()) or not np.isfinite(d):
            bad.append((M, n, "status/nan", d))
            continue
        excess = n - (M - 2)
        key = "near-square (n - (M-2) <= 6)" if excess <= 6 else ("n < 2(M-2)" if n < 2 * (M - 2) else "n >= 2(M-2)")
        key = ("rows > 32: " if max(M, n) > 32 else "rows <= 32: ") + key
        if d > worst.get(key, (0,))[0]:
            worst[key] = (d, M, n)
print("launch pairs:", count, "problems:", bad[:10])
for k in sorted(worst):
    print(f"{k:45s} worst {worst[k][0]:.2e} at M={worst[k][1]} n={worst[k][2]}")
