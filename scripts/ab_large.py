"""A/B of the large-degree paths (M = 33, 64 points): two-kernel default vs the single f64-MFMA kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hybrid_fem_lssvr_amd import ops
dev = torch.device("cuda:0")
M, n = (int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else (33, 64)))
for ne in (100000, 1000000):
    for dom, (lo, hi) in (("narrow", (-1.0, 1.0)), ("wide", (-ne / 24.0, ne / 24.0))):
        nodes = np.arange(ne + 1, dtype=np.float64) * ((hi - lo) / ne) + lo
        nodes[-1] = hi
        x = torch.as_tensor(nodes, device=dev)
        u = torch.sin(np.pi * x)
        res = {}
        for name, kw in (("mfma", dict(work=False)), ("split", dict())):
            W = torch.empty((ne, M), dtype=torch.float64, device=dev)
            st = torch.empty(ne, dtype=torch.int32, device=dev)
            ts = [ops.enhance_profiled(x, u, M, 1e4, n, global_domain=(lo, hi), out=W, status=st, **kw)
                  for _ in range(7)]
            res[name] = (np.median(ts[2:]) * 1e6, W)
            print(ne, dom, name, "median %.1f us" % res[name][0], "fallback", int(st.sum()), flush=True)
        d = (res["split"][1] - res["mfma"][1])
        rel = (d.norm(dim=1) / res["mfma"][1].norm(dim=1)).max().item()
        print(ne, dom, "max rel diff split vs mfma %.2e" % rel, flush=True)
