"""BASELINE config 5 timing: variable-coefficient rows, 1e6 elements, degree 8, 16 points, tabulated
a, a', f (16 + 72 + 3*128 = 472 B per element), both table layouts; hipExt-stamped launches.
usage: c5_quick.py [ne]   (LSSVR_VC_MINW=1: the two-waves-per-SIMD build of the point-major kernel)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hybrid_fem_lssvr_amd import ops
import bench
dev = "cuda:0"
ne = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
M, n = 9, 16
x = torch.linspace(-1, 1, ne + 1, dtype=torch.float64, device=dev)
u = torch.sin(np.pi * x)
W = torch.empty((ne, M), dtype=torch.float64, device=dev)
st = torch.empty(ne, dtype=torch.int32, device=dev)
res = {}
for pm in (True, False):
    a, da, f = bench._varcoef_device_tables(ops.colloc_points(x, n, point_major=pm))
    run = lambda: ops.enhance_varcoef(x, u, M, 1e4, n, a, da, f, global_domain=(-1.0, 1.0), out=W, status=st,
                                      point_major=pm, profiled=True)
    run()
    ts = sorted(run() for _ in range(40))
    res[pm] = W.clone()
    print(f"config 5 {'point' if pm else 'element'}-major: median {ts[20]*1e6:.1f} us  min {ts[0]*1e6:.1f} us -> "
          f"{ne/ts[20]:.3e} el/s, {472*ne/ts[20]/1e9:.0f} GB/s, fallback {int(st.sum())}")
print("bit-equal:", bool(torch.equal(res[True], res[False])))
