"""Timing of the Dirichlet solves (bands / flux) at a few sizes (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hybrid_fem_lssvr_amd import ops
dev = "cuda:0"
def med(fn, reps=30):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a0 = torch.cuda.Event(enable_timing=True); a1 = torch.cuda.Event(enable_timing=True)
        a0.record(); fn(); a1.record(); a1.synchronize()
        ts.append(a0.elapsed_time(a1) * 1e3)
    return sorted(ts)[len(ts) // 2]
for ne in (1000, 100008, 1000000, 4000000, 10000008):
    x = torch.linspace(-1, 1, ne + 1, dtype=torch.float64, device=dev)
    b = ops.p1_assemble(x, 2, want_local=True)
    tb = med(lambda: ops.tridiag_dirichlet_solve(b["diag"], b["off"], b["load"]))
    tf = med(lambda: ops.p1_flux_solve(b["kloc"], b["load"]))
    print(f"ne={ne}: bands {tb:8.1f} us   flux {tf:8.1f} us", flush=True)
