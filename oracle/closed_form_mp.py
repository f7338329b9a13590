"""Extended-precision minimiser of the reference's per-element QP ("truth").

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

The QP is the one ``lssvr_primal`` hands to SLSQP (Dual.py:46-78).  Its data are
taken exactly as the reference's float64 arithmetic produces them -- collocation
abscissae ``t_k = off + scl*x_k`` (float64), right-hand side ``f(x_k)`` (float64),
``scl`` (float64) -- and from there on everything is done with ``mpmath`` at
``dps`` significant digits: Legendre values/derivatives, the KKT matrix

    [[I + gamma A^T A, B^T], [B, 0]] [w; mu] = [gamma A^T f; g]      

and its LU solve.  ``mpmath`` ships with sympy in this image; when it is missing
the callers fall back to ``oracle.lssvr_oracle.solve_primal_kkt`` (float64).
"""
from __future__ import annotations

import numpy as np

try:  # pragma: no cover - availability probe
    import mpmath as mp
    HAVE_MP = True
except Exception:  # pragma: no cover
    mp = None
    HAVE_MP = False


def _legendre_tables_mp(t, M):
    L = [mp.mpf(1)] + [mp.mpf(0)] * (M - 1)
    D1 = [mp.mpf(0)] * M
    D2 = [mp.mpf(0)] * M
    if M > 1:
        L[1] = t
        D1[1] = mp.mpf(1)
    for p in range(1, M - 1):
        L[p + 1] = ((2 * p + 1) * t * L[p] - p * L[p - 1]) / (p + 1)
        D1[p + 1] = D1[p - 1] + (2 * p + 1) * L[p]
        D2[p + 1] = D2[p - 1] + (2 * p + 1) * D1[p]
    return L, D1, D2


def solve_truth(s, dps=60):
    """``s`` = ``oracle.lssvr_oracle.ElementSystem``.  Rows are rebuilt in
    extended precision from the float64 (t, x, scl, f, g) it carries; Ahat for the
    variable-coefficient case is taken from ``s`` as float64 data scaled exactly."""
    if not HAVE_MP:
        raise RuntimeError("mpmath not available")
    mp.mp.dps = dps
    M, n = s.M, s.n
    scl = mp.mpf(float(s.scl))
    gam = mp.mpf(float(s.gamma))
    poisson = bool(np.all(s.Ahat[:, :2] == 0.0))
    A = mp.zeros(n, M)
    for k in range(n):
        tk = mp.mpf(float(s.t[k]))
        _, D1, D2 = _legendre_tables_mp(tk, M)
        for p in range(M):
            if poisson:
                A[k, p] = -scl * scl * D2[p]
            else:
                # variable coefficient: rows are *defined* by the float64 a(x_k), a'(x_k)
                # carried in s (ak, dak recovered from columns 1 and 2 of Ahat)
                dak_over_scl = -mp.mpf(float(s.Ahat[k, 1]))        # L_1'=1, L_1''=0
                ak = (-mp.mpf(float(s.Ahat[k, 2])) - dak_over_scl * 3 * tk) / 3  # L_2''=3, L_2'=3t
                A[k, p] = -scl * scl * (ak * D2[p] + dak_over_scl * D1[p])
    B = mp.zeros(2, M)
    # the reference evaluates u(xmin), u(xmax) through the same float64 mapdomain
    ta = mp.mpf(float(np.float64(s.off) + np.float64(s.scl) * np.float64(s.a)))
    tb = mp.mpf(float(np.float64(s.off) + np.float64(s.scl) * np.float64(s.b)))
    La, _, _ = _legendre_tables_mp(ta, M)
    Lb, _, _ = _legendre_tables_mp(tb, M)
    for p in range(M):
        B[0, p] = La[p]
        B[1, p] = Lb[p]
    f = mp.matrix([mp.mpf(float(v)) for v in s.f])
    g = mp.matrix([mp.mpf(float(v)) for v in s.g])
    K = mp.zeros(M + 2, M + 2)
    AtA = A.T * A
    for i in range(M):
        for j in range(M):
            K[i, j] = gam * AtA[i, j] + (1 if i == j else 0)
        K[i, M] = B[0, i]
        K[i, M + 1] = B[1, i]
        K[M, i] = B[0, i]
        K[M + 1, i] = B[1, i]
    Atf = A.T * f
    rhs = mp.matrix(M + 2, 1)
    for i in range(M):
        rhs[i] = gam * Atf[i]
    rhs[M] = g[0]
    rhs[M + 1] = g[1]
    sol = mp.lu_solve(K, rhs)
    return np.array([float(sol[i]) for i in range(M)])


def truth_all(nodes, values, M, gamma, n, rhs, global_domain=None, elements=None, dps=60,
              coef_a=None, coef_da=None, bc_left=0.0, bc_right=0.0):
    """Truth coefficients for ``elements`` (default all) of a mesh."""
    from . import lssvr_oracle as orc
    nodes = np.asarray(nodes, dtype=np.float64)
    ne = len(nodes) - 1
    if global_domain is None:
        global_domain = (nodes[0], nodes[-1])
    if elements is None:
        elements = range(ne)
    out = []
    for i in elements:
        a, b = nodes[i], nodes[i + 1]
        g_l, g_r = orc.boundary_values(i, ne, a, b, values[i], values[i + 1], global_domain,
                                       bc_left, bc_right)
        s = orc.element_system(a, b, g_l, g_r, M, gamma, n, rhs, coef_a, coef_da)
        out.append(solve_truth(s, dps))
    return np.array(out)
