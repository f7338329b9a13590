"""Prototype (numpy float64) of the dual Gram solver to spec: boundary rows eliminated as a 2x2 block
pivot (projected feature map), Jacobi-equilibrated n x n system (K + eps I) lam = f', partial-pivot
LU, optional iterative refinement with the operator-form residual."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import scipy.linalg as sla
from oracle import lssvr_oracle as orc
from oracle import closed_form_mp as cf

def solve_dual_projected(s, refine=0, equil=True):
    n, M = s.n, s.M
    A = s.Ahat                      # n x M
    B = s.B                         # 2 x M
    eps = 1.0 / s.gamma_t
    Q = B @ B.T
    Qi = np.linalg.inv(Q)
    Cc = (A @ B.T) @ Qi             # n x 2: c_k
    Ap = A - Cc @ B                 # projected rows
    wbc = B.T @ (Qi @ s.g)          # minimum-norm solution of B w = g
    fp = s.ftil - A @ wbc
    K = Ap @ Ap.T
    Kd = K + eps * np.eye(n)
    d = 1.0 / np.sqrt(np.diag(Kd)) if equil else np.ones(n)
    Ks = Kd * d[:, None] * d[None, :]
    lu = sla.lu_factor(Ks)
    lam = d * sla.lu_solve(lu, d * fp)
    for _ in range(refine):
        w1 = Ap.T @ lam
        r = fp - Ap @ w1 - eps * lam           # operator-form residual
        lam = lam + d * sla.lu_solve(lu, d * r)
    return wbc + Ap.T @ lam

def solve_dual_full_lu(s):
    return orc.solve_dual_gram(s)

if __name__ == "__main__":
    rng = np.random.default_rng(0)
    cases = [("C1 ne8 M5 n5", np.linspace(-1, 1, 9), 5, 1e4, 5),
             ("default ne24 M8 n12", np.linspace(-1, 1, 25), 8, 1e4, 12),
             ("G3 ne24 M9 n16", np.linspace(-1, 1, 25), 9, 1e4, 16),
             ("G4 ne4096 M9 n16", np.linspace(-1, 1, 4097), 9, 1e4, 16),
             ("G5 ne24 M33 n64", np.linspace(-1, 1, 25), 33, 1e4, 64),
             ("c4 ne1e5 M33 n64", np.linspace(-1, 1, 100001), 33, 1e4, 64),
             ("G8 classdef ne4 M12 n12 g1e6", np.linspace(-1, 1, 5), 12, 1e6, 12),
             ("c2 narrow ne1e5 M9 n16", np.linspace(-1, 1, 100001), 9, 1e4, 16),
             ("M17 n12 (n<M-2)", np.linspace(-1, 1, 25), 17, 1e4, 12),
             ("M22 n12", np.linspace(-1, 1, 25), 22, 1e4, 12),
             ("M33 n31", np.linspace(-1, 1, 25), 33, 1e4, 31),
             ("M33 n33", np.linspace(-1, 1, 25), 33, 1e4, 33),
             ("M33 n38", np.linspace(-1, 1, 25), 33, 1e4, 38),
             ]
    for name, nodes, M, gamma, n in cases:
        values = np.sin(np.pi * nodes); values[0] = values[-1] = 0
        ne = len(nodes) - 1
        sel = sorted(set([0, 1, ne // 3, ne // 2, ne - 1]) & set(range(ne)))
        e = {k: 0 for k in ("full", "proj", "proj+1", "proj+2", "noeq+1", "primal")}
        for i in sel:
            gl, gr = orc.boundary_values(i, ne, nodes[i], nodes[i + 1], values[i], values[i + 1], (nodes[0], nodes[-1]))
            s = orc.element_system(nodes[i], nodes[i + 1], gl, gr, M, gamma, n)
            tr = cf.solve_truth(s)
            f = lambda w: orc.rel_l2_coef(w[None], tr[None])[0]
            e["full"] = max(e["full"], f(solve_dual_full_lu(s)))
            e["proj"] = max(e["proj"], f(solve_dual_projected(s)))
            e["proj+1"] = max(e["proj+1"], f(solve_dual_projected(s, 1)))
            e["proj+2"] = max(e["proj+2"], f(solve_dual_projected(s, 2)))
            e["noeq+1"] = max(e["noeq+1"], f(solve_dual_projected(s, 1, equil=False)))
            e["primal"] = max(e["primal"], f(orc.solve_bc_eliminated(s)))
        print(f"{name:30s} " + "  ".join(f"{k} {v:.1e}" for k, v in e.items()))
    # wide
    for ne, half in ((100008, 4167.0), (10000008, 416667.0)):
        sel = [0, 1, ne // 3, ne - 1]; step = 2 * half / ne
        e = {k: 0 for k in ("full", "proj", "proj+1", "proj+2")}
        for i in sel:
            a = i * step - half; b = (i + 1) * step - half if i + 1 < ne else half
            gl = 0.0 if i == 0 else np.sin(np.pi * a); gr = 0.0 if i == ne - 1 else np.sin(np.pi * b)
            s = orc.element_system(a, b, gl, gr, 9, 1e4, 16)
            tr = cf.solve_truth(s); f = lambda w: orc.rel_l2_coef(w[None], tr[None])[0]
            e["full"] = max(e["full"], f(solve_dual_full_lu(s))); e["proj"] = max(e["proj"], f(solve_dual_projected(s)))
            e["proj+1"] = max(e["proj+1"], f(solve_dual_projected(s, 1))); e["proj+2"] = max(e["proj+2"], f(solve_dual_projected(s, 2)))
        print(f"wide ne={ne:9d} M9 n16          " + "  ".join(f"{k} {v:.1e}" for k, v in e.items()))
