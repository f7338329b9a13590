"""Timing of the shared-operator path vs the general kernel (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hybrid_fem_lssvr_amd import ops
dev = "cuda:0"
M, n = 9, 16
for ne, wide in ((100008, True), (1000008, True), (10000008, True), (100000, False), (10000000, False)):
    lo, hi = (-ne / 24.0, ne / 24.0) if wide else (-1.0, 1.0)
    nodes = np.arange(ne + 1, dtype=np.float64) * ((hi - lo) / ne) + lo
    nodes[-1] = hi
    x = torch.as_tensor(nodes, device=dev)
    u = torch.sin(np.pi * x)
    W = torch.empty((ne, M), dtype=torch.float64, device=dev)
    op = ops.build_shared_operator((hi - lo) / ne, M, 1e4, n, device=dev)
    ts = sorted(ops.enhance_shared(x, u, op, M, n, global_domain=(lo, hi), out=W, profiled=True) for _ in range(30))
    tg = sorted(ops.enhance_profiled(x, u, M, 1e4, n, global_domain=(lo, hi), out=W) for _ in range(30))
    print(f"ne={ne} [{lo:g},{hi:g}]: shared med {ts[15]*1e6:8.2f} us -> {ne/ts[15]:.3e} el/s, {88*ne/ts[15]/1e9:7.1f} GB/s algorithmic;"
          f"  general med {tg[15]*1e6:8.2f} us -> {ne/tg[15]:.3e} el/s", flush=True)
