import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hybrid_fem_lssvr_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps
for ne, P, M in ((100000, 1000000, 9), (1000000, 10000000, 9), (10000000, 10000001, 9), (100000, 1000000, 33)):
    x = torch.linspace(-1, 1, ne + 1, dtype=torch.float64, device=dev)
    W = torch.randn((ne, M), dtype=torch.float64, device=dev)
    xq = torch.linspace(-1, 1, P, dtype=torch.float64, device=dev)
    t = timeit(lambda: ops.evaluate(x, W, xq))
    xr = xq[torch.randperm(P, device=dev)]
    t2 = timeit(lambda: ops.evaluate(x, W, xr))
    print(f"eval ne={ne} P={P} M={M}: sorted {t*1e6:.0f} us ({P/t:.3e} pts/s)  random {t2*1e6:.0f} us ({P/t2:.3e} pts/s)")
for ne in (100000, 10000000):
    x = torch.linspace(-1, 1, ne + 1, dtype=torch.float64, device=dev)
    b = ops.p1_assemble(x, 2, want_local=True)
    t = timeit(lambda: ops.p1_assemble(x, 2, out=b))
    t1 = timeit(lambda: ops.tridiag_dirichlet_solve(b["diag"], b["off"], b["load"]))
    t2 = timeit(lambda: ops.p1_flux_solve(b["kloc"], b["load"]))
    print(f"ne={ne}: assemble {t*1e6:.0f} us, tridiag {t1*1e6:.0f} us, flux {t2*1e6:.0f} us")
