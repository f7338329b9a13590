import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from hybrid_fem_lssvr_amd import ops
dev = "cuda:0"
for ne, M, n in ((1000000, 9, 16), (1000000, 9, 12), (1000000, 14, 28), (100000, 9, 16)):
    x = torch.linspace(-1, 1, ne + 1, dtype=torch.float64, device=dev); u = torch.sin(np.pi * x)
    f = (np.pi ** 2) * torch.sin(np.pi * ops.colloc_points(x, n))
    W = torch.empty((ne, M), dtype=torch.float64, device=dev); st = torch.empty(ne, dtype=torch.int32, device=dev)
    def t(fn):
        fn(); torch.cuda.synchronize(); ts = []
        for _ in range(30):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
        return sorted(ts)[15]
    a = t(lambda: ops.enhance(x, u, M, 1e4, n, rhs_values=f, out=W, status=st, global_domain=(-1.0, 1.0)))
    fp = f.t().contiguous()
    b = t(lambda: ops.enhance(x, u, M, 1e4, n, rhs_values=fp, point_major=True, out=W, status=st, global_domain=(-1.0, 1.0)))
    c = t(lambda: ops.enhance(x, u, M, 1e4, n, out=W, status=st, global_domain=(-1.0, 1.0)))
    print(f"ne={ne} M={M} n={n}: element-major {a:.1f} us, point-major {b:.1f} us, in-kernel sin {c:.1f} us")
