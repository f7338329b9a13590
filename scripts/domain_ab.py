"""Does the kernel time depend on the domain?  Same element count, narrow [-1,1] vs wide (h = 1/12)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hybrid_fem_lssvr_amd import ops
ne, M, n = 100008, 9, 16
if len(sys.argv) > 1:
    ne, M, n = (int(v) for v in sys.argv[1].split(","))
dev = "cuda:0"
def mk(lo, hi):
    nodes = np.arange(ne + 1, dtype=np.float64) * ((hi - lo) / ne) + lo
    nodes[-1] = hi
    x = torch.as_tensor(nodes, device=dev)
    return x, torch.sin(np.pi * x), (lo, hi)
cases = {"narrow": mk(-1.0, 1.0), "wide": mk(-ne / 24.0, ne / 24.0), "wide_small_u": None, "mid[-100,100]": mk(-100.0, 100.0)}
xw, uw, gdw = cases["wide"]
cases["wide_small_u"] = (xw, torch.zeros_like(uw), gdw)
W = torch.empty((ne, M), dtype=torch.float64, device=dev)
for rnd in range(3):
    for name, (x, u, gd) in cases.items():
        ts = sorted(ops.enhance_profiled(x, u, M, 1e4, n, global_domain=gd, out=W) for _ in range(60))
        print(f"round {rnd} {name:16s} med {ts[30]*1e6:7.2f} us  min {ts[0]*1e6:7.2f} us", flush=True)
