"""BASELINE config 5 timing: variable-coefficient rows, 1e6 elements, degree 8, 16 points
(tabulated a, a', f: 16 + 72 + 3*128 = 472 B of HBM traffic per element)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hybrid_fem_lssvr_amd import ops
dev = "cuda:0"
ne, M, n = 1000000, 9, 16
x = torch.linspace(-1, 1, ne + 1, dtype=torch.float64, device=dev)
u = torch.sin(np.pi * x)
xc = ops.colloc_points(x, n)
a = 1.0 + 0.3 * torch.sin(3.0 * xc)
da = 0.9 * torch.cos(3.0 * xc)
f = (np.pi ** 2) * torch.sin(np.pi * xc)
W = torch.empty((ne, M), dtype=torch.float64, device=dev)
st = torch.empty(ne, dtype=torch.int32, device=dev)
def run():
    ops.enhance_varcoef(x, u, M, 1e4, n, a, da, f, global_domain=(-1.0, 1.0), out=W, status=st)
run(); torch.cuda.synchronize()
ts = []
for _ in range(30):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); e1.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e-3)
t = sorted(ts)[len(ts) // 2]
print(f"config 5: {t*1e6:.1f} us -> {ne/t:.3e} el/s, {472*ne/t/1e9:.0f} GB/s of tabulated + output traffic, fallback {int(st.sum())}")
