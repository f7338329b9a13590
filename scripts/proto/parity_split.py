"""Prototype (numpy float64) of the parity-split solve of the Chebyshev-moment system (enhance_large_parity.hip):
S = P + E with P the same-parity entries; z <- P^-1 (b - E z) from z = P^-1 b; prints cond(S), the contraction
rate rho = |P^-1 E| and the error against the 60-digit minimiser after 0..3 corrections."""
import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts", "proto"))
import numpy as np
import cheb_moment as cm
from oracle import lssvr_oracle as orc
from oracle import closed_form_mp as mp

def build(s):
    Y, a, al, bb, N = cm.gen_tables()
    M, n = s.M, s.n; MR = M - 2; t = s.t
    hh = 0.5 * (s.b - s.a); inv_scl2 = hh * hh
    eps2 = 2.0 * inv_scl2 * inv_scl2 * (1.0 / s.gamma)
    ta = s.off + s.scl * s.a; tb = s.off + s.scl * s.b
    ea = 1.0 + ta; eb = 1.0 - tb
    sig = 0.5 * (ea + eb); dl = 0.5 * (ea - eb)
    idet = 0.5 * (1.0 + sig * (1.0 + sig))
    gl, gr = s.g
    d0 = (tb * gl - ta * gr) * idet; d1 = (gr - gl) * idet
    phi2 = -2.0 * (s.f * inv_scl2)
    D = 2 * MR - 1
    T = np.zeros((n, D)); T[:, 0] = 1.0; T[:, 1] = t
    for d in range(2, D): T[:, d] = (t + t) * T[:, d - 1] - T[:, d - 2]
    Mom = T.sum(0); Mom[0] = n
    r = (T[:, :MR] * phi2[:, None]).sum(0)
    S = np.zeros((MR, MR)); rhs = np.zeros(MR)
    es = eps2 * sig; ed = eps2 * dl; e_d0 = eps2 * d0; e_d1 = eps2 * d1
    q_ev = eps2 * (dl * d1 - sig * d0); q_od = eps2 * (dl * d0 - sig * d1)
    for i in range(MR):
        for k in range(i + 1):
            Q = al[i] * bb[k] + bb[i] * al[k]
            if (i + k) % 2 == 0: rg = eps2 * (N[i, k] + al[i] * al[k]) - es * Q
            else: rg = ed * Q
            S[i, k] = S[k, i] = Mom[i + k] + Mom[i - k] + rg
        rr = (al[i] * e_d0 + bb[i] * q_ev) if i % 2 == 0 else (al[i] * e_d1 + bb[i] * q_od)
        rhs[i] = r[i] + rr
    def finish(z):
        v = Y[:MR, :MR] @ z
        ev = (np.arange(MR) % 2 == 0)
        C0 = np.where(ev, 1.0 - a[:MR] * sig, (a[:MR] - 1.0) * dl)
        C1 = np.where(ev, a[:MR] * dl, 1.0 - (a[:MR] - 1.0) * sig)
        w = np.zeros(M); w[2:] = v; w[0] = d0 - C0 @ v; w[1] = d1 - C1 @ v
        return w
    return S, rhs, finish

def ldl_solve(A, b):
    A = A.copy(); n = len(b); L = np.eye(n); dd = np.zeros(n)
    for j in range(n):
        dd[j] = A[j, j]; L[j + 1:, j] = A[j + 1:, j] / dd[j]
        A[j + 1:, j + 1:] -= np.outer(L[j + 1:, j], A[j, j + 1:])
    y = np.linalg.solve(L, b); return np.linalg.solve(L.T, y / dd)

def run(a, h, M, n, gamma=1e4):
    b = a + h
    s = orc.element_system(a, b, np.sin(np.pi * a), np.sin(np.pi * b), M, gamma, n)
    wt = mp.solve_truth(s)
    S, rhs, fin = build(s)
    MR = M - 2
    par = (np.add.outer(np.arange(MR), np.arange(MR)) % 2 == 0)
    P = np.where(par, S, 0.0); E = S - P
    ev = np.arange(MR) % 2 == 0
    def psolve(b):
        z = np.zeros(MR)
        z[ev] = ldl_solve(P[np.ix_(ev, ev)], b[ev]); z[~ev] = ldl_solve(P[np.ix_(~ev, ~ev)], b[~ev])
        return z
    rho = np.linalg.norm(np.linalg.solve(P, E), 2)
    full = orc.rel_l2_coef(fin(ldl_solve(S, rhs)), wt)
    z = psolve(rhs); errs = [orc.rel_l2_coef(fin(z), wt)]
    for _ in range(3):
        z = z + psolve(rhs - S @ z)   # residual: P part cancels to rounding, E part is the coupling
        errs.append(orc.rel_l2_coef(fin(z), wt))
    # cheaper: residual = -E z only
    z2 = psolve(rhs); z2 = z2 - psolve(E @ z2)
    e2 = orc.rel_l2_coef(fin(z2), wt)
    print("a=%-12g h=%-8.3g M=%d n=%d condS=%.1e rho=%.1e full %.1e | block %s | E-only 1 step %.1e" % (
        a, h, M, n, np.linalg.cond(S), rho, full, " ".join("%.1e" % e for e in errs), e2))

for M, n in [(33, 64), (9, 16), (24, 48)]:
    run(-1.0, 1/12, M, n); run(0.25, 1/12, M, n); run(-0.5, 0.5, M, n)
    run(4000.0, 1/12, M, n); run(-4166.0, 1/12, M, n); run(416666.0, 1/12, M, n); run(0.3, 2e-5, M, n); run(0.9999, 2e-7, M, n)
print("near the refinement boundary")
for M, n in [(33, 46), (33, 50), (28, 41), (23, 36), (23, 40), (30, 96)]:
    run(-4166.0, 1/12, M, n); run(0.9999, 2e-7, M, n); run(-0.5, 0.5, M, n)
