import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from hybrid_fem_lssvr_amd import ops
from oracle import lssvr_oracle as orc, closed_form_mp as cf
dev = torch.device("cuda:0")
for h in (1.0 / 12, 0.5):
    for M, n in ((22, 20), (22, 21), (22, 24), (22, 27), (20, 18), (18, 16), (16, 14), (14, 12), (14, 13), (22, 44)):
        ne = 70
        nodes = -1.0 + h * np.arange(ne + 1)
        values = np.sin(np.pi * nodes) + 0.01 * np.random.default_rng(M * n).standard_normal(ne + 1)
        gd = (nodes[0], nodes[-1])
        x, u = torch.as_tensor(nodes, device=dev), torch.as_tensor(values, device=dev)
        W, st = ops.enhance(x, u, M, 1e4, n, global_domain=gd)
        xc = ops.colloc_points(x, n).cpu().numpy()
        W2, _ = ops.enhance(x, u, M, 1e4, n, global_domain=gd, rhs_values=torch.as_tensor(orc.poisson_rhs(xc), device=dev))
        sel = [0, 1, 35, 64, ne - 1]
        tr = cf.truth_all(nodes, values, M, 1e4, n, orc.poisson_rhs, gd, sel)
        print("h=%.3f M=%d n=%d  vs truth: in-kernel rhs %.2e  tabulated %.2e  fallback %d" % (
            h, M, n, orc.rel_l2_coef(W.cpu().numpy()[sel], tr).max(), orc.rel_l2_coef(W2.cpu().numpy()[sel], tr).max(), int(st.sum())), flush=True)
