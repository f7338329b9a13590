"""Shared-operator path at degree 32 / 64 points (BASELINE config 4 on a uniform mesh): accuracy against the
general kernel and the float64 oracle, and timing (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hybrid_fem_lssvr_amd import ops
from oracle import lssvr_oracle as orc
dev = "cuda:0"
M, n = 33, 64
for ne, lo, hi in ((24, -1.0, 1.0), (4096, -1.0, 1.0), (100000, -1.0, 1.0), (100008, -4167.0, 4167.0)):
    nodes = np.arange(ne + 1, dtype=np.float64) * ((hi - lo) / ne) + lo
    nodes[-1] = hi
    values = np.sin(np.pi * nodes); values[0] = values[-1] = 0.0
    x = torch.as_tensor(nodes, device=dev); u = torch.as_tensor(values, device=dev)
    op = ops.build_shared_operator((hi - lo) / ne, M, 1e4, n, device=dev)
    Ws, st = ops.enhance_shared(x, u, op, M, n, global_domain=(lo, hi))
    Wg, sg = ops.enhance(x, u, M, 1e4, n, global_domain=(lo, hi))
    torch.cuda.synchronize()
    Wsh, Wgh = Ws.cpu().numpy(), Wg.cpu().numpy()
    sel = np.unique(np.linspace(0, ne - 1, 12).astype(np.int64))
    Wo = np.array([orc.solve_primal_kkt(orc.element_system(nodes[i], nodes[i + 1], *orc.boundary_values(
        int(i), ne, nodes[i], nodes[i + 1], values[i], values[i + 1], (lo, hi)), M, 1e4, n)) for i in sel])
    ts = sorted(ops.enhance_shared(x, u, op, M, n, global_domain=(lo, hi), out=Ws, profiled=True) for _ in range(20))
    tg = sorted(ops.enhance_profiled(x, u, M, 1e4, n, global_domain=(lo, hi), out=Wg) for _ in range(10))
    print(f"ne={ne} [{lo:g},{hi:g}]: shared-vs-general {orc.rel_l2_coef(Wsh, Wgh).max():.2e}  shared-vs-oracle {orc.rel_l2_coef(Wsh[sel], Wo).max():.2e}"
          f"  general-vs-oracle {orc.rel_l2_coef(Wgh[sel], Wo).max():.2e}  fallback {int(st.sum())}"
          f"  | shared {ts[10]*1e6:.1f} us ({ne/ts[10]:.2e} el/s, {280*ne/ts[10]/1e9:.0f} GB/s)  general {tg[5]*1e6:.1f} us", flush=True)
