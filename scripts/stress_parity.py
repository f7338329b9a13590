"""Randomised parity sweep: GPU kernels vs the 60-digit minimiser over random (M, n, gamma, h, x0)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hybrid_fem_lssvr_amd import ops
from oracle import lssvr_oracle as orc, closed_form_mp as cf

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
solver = int(sys.argv[3]) if len(sys.argv) > 3 else ops.SOLVER_PRIMAL      # 2 = force the wave / MFMA mapping
m_lo = int(sys.argv[4]) if len(sys.argv) > 4 else 2                        # e.g. 23: large-degree kernels only
dev = torch.device("cuda:0")
worst = []
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 120):
    M = int(rng.integers(m_lo, 34))
    n = int(rng.integers(2, 81))
    gamma = 10.0 ** rng.uniform(-2, 8)
    h = 10.0 ** rng.uniform(-7, 1)
    x0 = rng.choice([-1, 1]) * 10.0 ** rng.uniform(-1, 6) * rng.choice([0, 1, 1])
    ne = 12
    nodes = x0 + h * np.cumsum(np.concatenate([[0.0], rng.uniform(0.5, 1.5, ne)]))
    if not np.all(np.diff(nodes) > 0):
        continue
    values = np.sin(np.pi * nodes) + 0.1 * rng.standard_normal(ne + 1)
    gd = (nodes[0], nodes[-1])
    x, u = torch.as_tensor(nodes, device=dev), torch.as_tensor(values, device=dev)
    try:
        W, st = ops.enhance(x, u, M, gamma, n, global_domain=gd,
                            solver=solver if n >= M - 2 else ops.SOLVER_PRIMAL)
    except Exception as exc:
        print(f"M={M} n={n}: {str(exc)[:90]}")
        continue
    W, st = W.cpu().numpy(), st.cpu().numpy()
    sel = [0, 5, 11]
    tr = cf.truth_all(nodes, values, M, gamma, n, orc.poisson_rhs, gd, sel)
    err = orc.rel_l2_coef(W[sel], tr)
    if n < M - 2:
        route = "dual"
    elif M <= 22 and solver == ops.SOLVER_PRIMAL:
        route = "lane"
    elif solver != ops.SOLVER_PRIMAL:
        route = "mfma"
    else:
        route = "parity" if n >= 2 * (M - 2) else ("refined" if n - (M - 2) <= 14 else "solve4")
    e = float(np.nanmax(err)) if np.all(st[sel] == 0) else float("nan")
    worst.append((e, M, n, gamma, h, x0, route, int(st.sum())))
worst.sort(key=lambda t: (-(t[0] if t[0] == t[0] else 1e9)))
for w in worst[:25]:
    print("err %.2e  M=%2d n=%2d gamma=%.1e h=%.1e x0=%.1e %s fallback=%d" % w)
print("n cases", len(worst), "median err", np.nanmedian([w[0] for w in worst]))
