"""Prototype: the BC-eliminated problem as a stacked least-squares problem
    min | [Abar; sqrt(eps) R] v - [fbar; sqrt(eps) q] |^2
solved by column-scaled Householder QR, in the n ~ M-2 regime where the normal equations lose digits.
(eps (I + C^T C) = eps R^T R with R the Cholesky factor, or stacked rows [I; C].)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import lssvr_oracle as orc
from oracle import closed_form_mp as mp


def solve_qr(s, refine=0):
    M = s.M
    B1 = s.B[:, :2]; B2 = s.B[:, 2:]
    det = B1[0, 0] * B1[1, 1] - B1[0, 1] * B1[1, 0]
    B1i = np.array([[B1[1, 1], -B1[0, 1]], [-B1[1, 0], B1[0, 0]]]) / det
    d = B1i @ s.g
    C = B1i @ B2
    eps = 1.0 / s.gamma_t
    se = np.sqrt(eps)
    Abar = s.Ahat[:, 2:] - s.Ahat[:, :2] @ C
    fbar = s.ftil - s.Ahat[:, :2] @ d
    # |d - C v|^2 + |v|^2 = |[C; I] v - [d; 0]|^2
    St = np.vstack([Abar, se * C, se * np.eye(M - 2)])
    rt = np.concatenate([fbar, se * d, np.zeros(M - 2)])
    cs = 1.0 / np.sqrt((St * St).sum(0))
    Q, R = np.linalg.qr(St * cs[None, :])
    v = cs * np.linalg.solve(R, Q.T @ rt)
    w = np.zeros(M)
    w[2:] = v
    w[:2] = d - C @ v
    return w


def solve_semi(s):
    """Corrected semi-normal equations: Cholesky of the scaled normal matrix + one refinement step
    with the residual formed through the STACKED operator (r = St^T (rt - St v))."""
    M = s.M
    B1 = s.B[:, :2]; B2 = s.B[:, 2:]
    det = B1[0, 0] * B1[1, 1] - B1[0, 1] * B1[1, 0]
    B1i = np.array([[B1[1, 1], -B1[0, 1]], [-B1[1, 0], B1[0, 0]]]) / det
    d = B1i @ s.g
    C = B1i @ B2
    eps = 1.0 / s.gamma_t
    se = np.sqrt(eps)
    Abar = s.Ahat[:, 2:] - s.Ahat[:, :2] @ C
    fbar = s.ftil - s.Ahat[:, :2] @ d
    St = np.vstack([Abar, se * C, se * np.eye(M - 2)])
    rt = np.concatenate([fbar, se * d, np.zeros(M - 2)])
    cs = 1.0 / np.sqrt((St * St).sum(0))
    Ss = St * cs[None, :]
    N = Ss.T @ Ss
    L = np.linalg.cholesky(N)
    def solve(b):
        return np.linalg.solve(L.T, np.linalg.solve(L, b))
    y = solve(Ss.T @ rt)
    for _ in range(3):
        res = rt - Ss @ y
        y = y + solve(Ss.T @ res)
    v = cs * y
    w = np.zeros(M); w[2:] = v; w[:2] = d - C @ v
    return w


if __name__ == "__main__":
    cases = [(33, n) for n in (31, 32, 33, 35, 38, 42)] + [(24, n) for n in (22, 23, 24, 26)] + \
            [(17, 15), (17, 16), (9, 7), (9, 8)]
    h = 1.0 / 12
    for M, n in cases:
        worst = dict(primal=0, qr=0, semi=0, dual=0)
        for a in (-1.0, -0.25, 0.5, 1.0 - h):
            b = a + h
            gl, gr = np.sin(np.pi * a), np.sin(np.pi * b)
            s = orc.element_system(a, b, gl, gr, M, 1e4, n)
            wt = mp.solve_truth(s)
            for name, fn in (("primal", orc.solve_bc_eliminated), ("qr", solve_qr), ("semi", solve_semi),
                             ("dual", orc.solve_dual_gram)):
                try:
                    w = fn(s)
                    worst[name] = max(worst[name], orc.rel_l2_coef(w, wt))
                except Exception as exc:
                    worst[name] = float("nan")
        print(M, n, " ".join("%s %.1e" % kv for kv in worst.items()))
