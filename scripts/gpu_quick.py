"""Quick on-GPU timing of the enhancement kernels (development aid, not the bench)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hybrid_fem_lssvr_amd import ops

dev = torch.device("cuda:0")
print("device:", torch.cuda.get_device_name(0))
print("fp64 FMA probe  TFLOP/s:", round(ops.fp64_probe(8192, 4096, False), 2))
print("fp64 MFMA probe TFLOP/s:", round(ops.fp64_probe(8192, 2048, True), 2))

def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps

cfgs = [(100000, 9, 16), (1000000, 9, 16), (10000000, 9, 16), (1000000, 5, 5), (1000000, 12, 12), (1000000, 14, 16)]
if len(sys.argv) > 1:
    cfgs = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for ne, M, n in cfgs:
    x = torch.linspace(-1, 1, ne + 1, dtype=torch.float64, device=dev)
    u = torch.sin(np.pi * x)
    W = torch.empty((ne, M), dtype=torch.float64, device=dev)
    st = torch.empty(ne, dtype=torch.int32, device=dev)
    t = timeit(lambda: ops.enhance(x, u, M, 1e4, n, out=W, status=st, global_domain=(-1.0, 1.0)))
    print(f"ne={ne} M={M} n={n}: {t*1e6:.1f} us  {ne/t:.3e} el/s  fallback={int(st.sum())}")
    if M == 9:
        f = torch.zeros((ne, n), dtype=torch.float64, device=dev)
        t = timeit(lambda: ops.enhance(x, u, M, 1e4, n, rhs_values=f, out=W, status=st, global_domain=(-1.0, 1.0)))
        print(f"   rhs-array path: {t*1e6:.1f} us  {ne/t:.3e} el/s")
