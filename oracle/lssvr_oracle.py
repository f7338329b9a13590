"""Float64 numpy restatement of the reference's per-element LSSVR path.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``): the checker for the HIP
path and the reported CPU baseline, never the product.

Reference = ``/root/reference/1D-Possion/Hybrid-FEM-LSSVR-Dual.py`` (``Dual.py``).
The reference poses, per element, the equality-constrained QP (``Dual.py:46-78``)

    minimise  1/2 |w|^2 + gamma/2 |e|^2
    s.t.      A w + e = f      (n PDE-residual rows, ``Dual.py:43-44,59``)
              B w     = g      (2 boundary rows,     ``Dual.py:61-76``)

and hands it to SLSQP (``Dual.py:84-88``).  This module restates

* the arithmetic that builds (A, B, f, g): ``numpy.linspace`` (``Dual.py:40``),
  ``polyutils.mapparms/mapdomain``, ``legder`` and the Clenshaw ``legval``
  (``Dual.py:44,56,66-75`` through numpy's Legendre class);
* the closed-form minimiser of that QP in three float64 variants (primal KKT,
  dual Gram, BC-eliminated SPD = the algorithm the HIP kernels implement);
* the reference's own SLSQP loop (``Dual.py:20-98,139-169``), kept as the CPU
  baseline that ``bench.py`` times;
* ``evaluate_solution`` (``Dual.py:176-203``) and the P1 FEM step the reference
  delegates to scikit-fem (``Dual.py:110-137``; scikit-fem is absent in this
  image, so that part follows SURVEY.md Appendix C and is pinned by the analytic
  tridiagonal system, not by reference output).

The extended-precision "truth" lives in ``oracle/closed_form_mp.py``.
"""
from __future__ import annotations

import numpy as np
from numpy.polynomial import legendre as _leg

RHS_ARRAY = 0        # f supplied as values at the collocation points
RHS_SIN = 1          # f(x) = amp * sin(omega * x), numpy arithmetic order
RHS_VARCOEF = 2      # manufactured f for -(a u')' = f with u = sin(pi x)

PI = float(np.pi)
PI_SQ = float(np.pi ** 2)          # Dual.py:12 -> Python float pow


# --------------------------------------------------------------------------
# problem definition, Dual.py:8-18
# --------------------------------------------------------------------------
def true_solution(x):
    """Dual.py:8-9."""
    return np.sin(np.pi * x)


def poisson_rhs(x):
    """Dual.py:11-12 (same association: (pi**2) * sin(pi * x))."""
    return np.pi ** 2 * np.sin(np.pi * x)


# --------------------------------------------------------------------------
# numpy arithmetic the reference inherits
# --------------------------------------------------------------------------
def np_linspace(a, b, n):
    """numpy/_core/function_base.py:140-175 for scalar float64 end points.

    y_k = fl(fl(k*step) + a), step = fl((b-a)/(n-1)), last sample forced to b.
    """
    a = np.float64(a)
    b = np.float64(b)
    n = int(n)
    if n <= 0:
        return np.zeros(0)
    if n == 1:
        return np.array([a], dtype=np.float64)
    div = n - 1
    delta = b - a
    k = np.arange(0, n, dtype=np.float64)
    step = delta / div
    if step == 0:
        y = (k / div) * delta
    else:
        y = k * step
    y = y + a
    y[-1] = b
    return y


def mapparms(a, b):
    """polyutils.py:282-286 with old=[a,b], new=[-1,1] -> (off, scl)."""
    a = np.float64(a)
    b = np.float64(b)
    oldlen = b - a
    off = (b * -1.0 - a * 1.0) / oldlen
    scl = 2.0 / oldlen
    return off, scl


def clenshaw(t, c):
    """legendre.py:895-911 (``legval``) restated; t scalar or array."""
    c = np.asarray(c, dtype=np.float64)
    t = np.asarray(t, dtype=np.float64)
    if len(c) == 1:
        c0 = c[0] + 0 * t
        c1 = 0 * t
    elif len(c) == 2:
        c0 = c[0] + 0 * t
        c1 = c[1] + 0 * t
    else:
        nd = len(c)
        c0 = c[-2] + 0 * t
        c1 = c[-1] + 0 * t
        for i in range(3, len(c) + 1):
            tmp = c0
            nd = nd - 1
            c0 = c[-i] - (c1 * (nd - 1)) / nd
            c1 = tmp + (c1 * t * (2 * nd - 1)) / nd
    return c0 + c1 * t


def legendre_rows_reference(a, b, M, x):
    """PDE-row matrix exactly as the reference's arithmetic produces it.

    Row k, column p = -[Legendre(e_p,[a,b]).deriv(2)](x_k)  (Dual.py:43-44,56):
    ``legder(e_p, 2, scl)`` then Clenshaw at ``off + scl*x``.
    """
    off, scl = mapparms(a, b)
    t = off + scl * np.asarray(x, dtype=np.float64)
    A = np.zeros((len(t), M))
    for p in range(M):
        e = np.zeros(M)
        e[p] = 1.0
        d2 = _leg.legder(e, 2, scl)
        A[:, p] = -clenshaw(t, d2)
    return A


def legendre_tables(t, M):
    """L_p(t), L_p'(t), L_p''(t) for p < M by the stable upward recurrences

        (p+1) L_{p+1} = (2p+1) t L_p - p L_{p-1}
        L'_{p+1} = L'_{p-1} + (2p+1) L_p ,  L''_{p+1} = L''_{p-1} + (2p+1) L'_p
    """
    t = np.atleast_1d(np.asarray(t, dtype=np.float64))
    L = np.zeros((len(t), M))
    D1 = np.zeros((len(t), M))
    D2 = np.zeros((len(t), M))
    L[:, 0] = 1.0
    if M > 1:
        L[:, 1] = t
        D1[:, 1] = 1.0
    for p in range(1, M - 1):
        L[:, p + 1] = ((2 * p + 1) * t * L[:, p] - p * L[:, p - 1]) / (p + 1)
        D1[:, p + 1] = D1[:, p - 1] + (2 * p + 1) * L[:, p]
        D2[:, p + 1] = D2[:, p - 1] + (2 * p + 1) * D1[:, p]
    return L, D1, D2


def gegenbauer_d2(t, M):
    """L_p''(t) for p < M through the ultraspherical three-term recurrence the
    HIP kernels use:  q_m = L''_{m+2} = 3 C^{(5/2)}_m,
        q_0 = 3, q_1 = 15 t, m q_m = (2m+3) t q_{m-1} - (m+3) q_{m-2}.
    """
    t = np.atleast_1d(np.asarray(t, dtype=np.float64))
    D2 = np.zeros((len(t), M))
    if M > 2:
        D2[:, 2] = 3.0
    if M > 3:
        D2[:, 3] = 15.0 * t
    for m in range(2, M - 2):
        D2[:, m + 2] = ((2 * m + 3) * t * D2[:, m + 1] - (m + 3) * D2[:, m]) / m
    return D2


def gegenbauer_d1(t, M):
    """L_p'(t) for p < M:  r_m = L'_{m+1} = C^{(3/2)}_m,
        r_0 = 1, r_1 = 3 t, m r_m = (2m+1) t r_{m-1} - (m+1) r_{m-2}.
    """
    t = np.atleast_1d(np.asarray(t, dtype=np.float64))
    D1 = np.zeros((len(t), M))
    if M > 1:
        D1[:, 1] = 1.0
    if M > 2:
        D1[:, 2] = 3.0 * t
    for m in range(2, M - 1):
        D1[:, m + 1] = ((2 * m + 1) * t * D1[:, m] - (m + 1) * D1[:, m - 1]) / m
    return D1


# --------------------------------------------------------------------------
# per-element system
# --------------------------------------------------------------------------
class ElementSystem:
    """h-free scaled rows of one element's QP.

    A = scl^2 * Ahat,  f = scl^2 * ftil,  gamma_t = gamma * scl^4, so that
    gamma |f - A w|^2 = gamma_t |ftil - Ahat w|^2   (SURVEY.md Appendix A.3).
    """

    __slots__ = ("a", "b", "off", "scl", "x", "t", "Ahat", "B", "f", "ftil",
                 "g", "gamma", "gamma_t", "M", "n")


def element_system(a, b, g_l, g_r, M, gamma, n, rhs=poisson_rhs, coef_a=None,
                   coef_da=None, f_values=None):
    """Build (Ahat, B, ftil, g) for element [a,b] the way the reference's
    arithmetic defines them (float64 t_k = off + scl*x_k, x_k = linspace).

    ``coef_a``/``coef_da`` (callables) switch on the variable-coefficient rows of
    BASELINE config 5 (no reference counterpart: Dual.py:44 hard-codes -u'').
    """
    s = ElementSystem()
    s.a, s.b, s.M, s.n = np.float64(a), np.float64(b), int(M), int(n)
    s.off, s.scl = mapparms(a, b)
    s.x = np_linspace(a, b, n)
    s.t = s.off + s.scl * s.x
    L, D1, D2 = legendre_tables(s.t, M)
    if coef_a is None:
        s.Ahat = -D2
    else:
        ak = np.asarray(coef_a(s.x), dtype=np.float64)
        dak = np.asarray(coef_da(s.x), dtype=np.float64)
        s.Ahat = -(ak[:, None] * D2) - (dak / s.scl)[:, None] * D1
    ta = s.off + s.scl * s.a
    tb = s.off + s.scl * s.b
    La, _, _ = legendre_tables(np.array([ta, tb]), M)
    s.B = La
    s.f = np.asarray(rhs(s.x) if f_values is None else f_values, dtype=np.float64)
    s.ftil = s.f / (s.scl * s.scl)
    s.g = np.array([g_l, g_r], dtype=np.float64)
    s.gamma = float(gamma)
    s.gamma_t = float(gamma) * float(s.scl) ** 4
    return s


def solve_primal_kkt(s):
    """[[I + gt Ahat^T Ahat, B^T],[B, 0]] [w; mu] = [gt Ahat^T ftil; g]
    (SURVEY.md A.3 primal form), numpy LU with partial pivoting, rows/cols
    equilibrated by 1/gt so no 1e24 entries reach LAPACK."""
    M = s.M
    eps = 1.0 / s.gamma_t
    K = np.zeros((M + 2, M + 2))
    K[:M, :M] = eps * np.eye(M) + s.Ahat.T @ s.Ahat
    K[:M, M:] = s.B.T            # multiplier rescaled by eps: mu' = eps*mu
    K[M:, :M] = s.B
    rhs = np.concatenate([s.Ahat.T @ s.ftil, s.g])
    sol = np.linalg.solve(K, rhs)
    return sol[:M]


def solve_dual_gram(s):
    """north_star's Gram form (SURVEY.md A.3 dual form):
    [[A A^T + I/gamma, A B^T],[B A^T, B B^T]] [lam; mu] = [f; g], w = A^T lam + B^T mu,
    solved in h-free scaling with symmetric Jacobi equilibration + LU."""
    n = s.n
    eps = 1.0 / s.gamma_t
    K = np.zeros((n + 2, n + 2))
    K[:n, :n] = s.Ahat @ s.Ahat.T + eps * np.eye(n)
    K[:n, n:] = s.Ahat @ s.B.T
    K[n:, :n] = s.B @ s.Ahat.T
    K[n:, n:] = s.B @ s.B.T
    rhs = np.concatenate([s.ftil, s.g])
    d = 1.0 / np.sqrt(np.abs(np.diag(K)))
    sol = d * np.linalg.solve(K * d[:, None] * d[None, :], d * rhs)
    return s.Ahat.T @ sol[:n] + s.B.T @ sol[n:]


def solve_bc_eliminated(s, return_status=False):
    """The algorithm the HIP kernels implement ("per-element solve"):

    split w = (w1, v) with w1 = (w_0, w_1); B = [B1 B2]; w1 = d - C v with
    C = B1^{-1} B2, d = B1^{-1} g.  Then v minimises
        1/2 |d - C v|^2 + 1/2 |v|^2 + gt/2 |fbar - Abar v|^2,
        Abar = Ahat2 - Ahat1 C,  fbar = ftil - Ahat1 d,
    i.e. the SPD (M-2) system
        (eps (I + C^T C) + Abar^T Abar) v = Abar^T fbar + eps C^T d,  eps = 1/gt,
    solved by Jacobi-scaled Cholesky.
    """
    M = s.M
    B1 = s.B[:, :2]
    B2 = s.B[:, 2:]
    det = B1[0, 0] * B1[1, 1] - B1[0, 1] * B1[1, 0]
    B1i = np.array([[B1[1, 1], -B1[0, 1]], [-B1[1, 0], B1[0, 0]]]) / det
    d = B1i @ s.g
    w = np.zeros(M)
    if M == 2:
        w[:] = d
        return (w, 0) if return_status else w
    C = B1i @ B2
    eps = 1.0 / s.gamma_t
    Abar = s.Ahat[:, 2:] - s.Ahat[:, :2] @ C
    fbar = s.ftil - s.Ahat[:, :2] @ d
    S = Abar.T @ Abar + eps * (np.eye(M - 2) + C.T @ C)
    r = Abar.T @ fbar + eps * (C.T @ d)
    dj = 1.0 / np.sqrt(np.diag(S))
    Ss = S * dj[:, None] * dj[None, :]
    status = 0
    try:
        Lc = np.linalg.cholesky(Ss)
        y = np.linalg.solve(Lc, dj * r)
        v = dj * np.linalg.solve(Lc.T, y)
        if not np.all(np.isfinite(v)):
            raise np.linalg.LinAlgError("non-finite")
        w[2:] = v
        w[:2] = d - C @ v
    except np.linalg.LinAlgError:
        status = 1
        w[:] = linear_fallback_coef(s.g[0], s.g[1], M)
    return (w, status) if return_status else w


def linear_fallback_coef(g_l, g_r, M):
    """Legendre coefficients of the linear interpolant the reference falls back
    to on an exception (Dual.py:164-169): (g_l+g_r)/2 L_0 + (g_r-g_l)/2 L_1."""
    w = np.zeros(M)
    w[0] = 0.5 * (g_l + g_r)
    if M > 1:
        w[1] = 0.5 * (g_r - g_l)
    return w


def boundary_values(i, ne, a, b, u_l, u_r, global_domain, bc_left=0.0, bc_right=0.0):
    """Dual.py:65-75 + 150-151: on a global-boundary element whose end point
    equals the global end point exactly, the Dirichlet value replaces the FEM
    nodal value."""
    g_l = bc_left if (i == 0 and a == global_domain[0]) else u_l
    g_r = bc_right if (i == ne - 1 and b == global_domain[1]) else u_r
    return g_l, g_r


def enhance_all(nodes, values, M, gamma, n=12, rhs=poisson_rhs, global_domain=None,
                solver="bc_elim", coef_a=None, coef_da=None, bc_left=0.0, bc_right=0.0):
    """Closed-form restatement of ``solve_lssvr_subproblems`` (Dual.py:139-169):
    returns W float64[ne, M] (row i = ``lssvr_functions[i].coef``) and status."""
    nodes = np.asarray(nodes, dtype=np.float64)
    values = np.asarray(values, dtype=np.float64)
    ne = len(nodes) - 1
    if global_domain is None:
        global_domain = (nodes[0], nodes[-1])
    W = np.zeros((ne, M))
    status = np.zeros(ne, dtype=np.int32)
    fn = {"bc_elim": solve_bc_eliminated, "primal": solve_primal_kkt, "dual": solve_dual_gram}[solver]
    for i in range(ne):
        a, b = nodes[i], nodes[i + 1]
        g_l, g_r = boundary_values(i, ne, a, b, values[i], values[i + 1], global_domain,
                                   bc_left, bc_right)
        s = element_system(a, b, g_l, g_r, M, gamma, n, rhs, coef_a, coef_da)
        if solver == "bc_elim":
            W[i], status[i] = fn(s, return_status=True)
        else:
            W[i] = fn(s)
    return W, status


def enhance_all_vec(nodes, values, M, gamma, n=12, rhs=poisson_rhs, global_domain=None,
                    coef_a=None, coef_da=None, bc_left=0.0, bc_right=0.0, chunk=200000):
    """Batched float64 primal-KKT solve of every element (same rows as
    :func:`element_system`, same equilibration as :func:`solve_primal_kkt`); used by
    the full-size parity tests where the per-element loop is too slow."""
    nodes = np.asarray(nodes, dtype=np.float64)
    values = np.asarray(values, dtype=np.float64)
    ne = len(nodes) - 1
    if global_domain is None:
        global_domain = (nodes[0], nodes[-1])
    W = np.zeros((ne, M))
    for s0 in range(0, ne, chunk):
        s1 = min(ne, s0 + chunk)
        a = nodes[s0:s1]
        b = nodes[s0 + 1:s1 + 1]
        gl = values[s0:s1].copy()
        gr = values[s0 + 1:s1 + 1].copy()
        if s0 == 0 and a[0] == global_domain[0]:
            gl[0] = bc_left
        if s1 == ne and b[-1] == global_domain[1]:
            gr[-1] = bc_right
        oldlen = b - a
        off = (b * -1.0 - a * 1.0) / oldlen
        scl = 2.0 / oldlen
        step = oldlen / (n - 1)
        k = np.arange(n, dtype=np.float64)
        x = k[None, :] * step[:, None]
        x = x + a[:, None]
        x[:, -1] = b
        t = off[:, None] + scl[:, None] * x
        m = s1 - s0
        L = np.zeros((m, n, M)); D1 = np.zeros((m, n, M)); D2 = np.zeros((m, n, M))
        L[..., 0] = 1.0
        if M > 1:
            L[..., 1] = t
            D1[..., 1] = 1.0
        for p in range(1, M - 1):
            L[..., p + 1] = ((2 * p + 1) * t * L[..., p] - p * L[..., p - 1]) / (p + 1)
            D1[..., p + 1] = D1[..., p - 1] + (2 * p + 1) * L[..., p]
            D2[..., p + 1] = D2[..., p - 1] + (2 * p + 1) * D1[..., p]
        if coef_a is None:
            Ahat = -D2
        else:
            Ahat = -(coef_a(x)[..., None] * D2) - (coef_da(x) / scl[:, None])[..., None] * D1
        ta = off + scl * a
        tb = off + scl * b
        Bm = np.zeros((m, 2, M))
        for row, tt in ((0, ta), (1, tb)):
            Bm[:, row, 0] = 1.0
            if M > 1:
                Bm[:, row, 1] = tt
            for p in range(1, M - 1):
                Bm[:, row, p + 1] = ((2 * p + 1) * tt * Bm[:, row, p] - p * Bm[:, row, p - 1]) / (p + 1)
        f = np.asarray(rhs(x), dtype=np.float64)
        ftil = f / (scl * scl)[:, None]
        eps = 1.0 / (float(gamma) * scl ** 4)
        K = np.zeros((m, M + 2, M + 2))
        K[:, :M, :M] = np.einsum("ekp,ekq->epq", Ahat, Ahat) + eps[:, None, None] * np.eye(M)[None]
        K[:, :M, M:] = np.transpose(Bm, (0, 2, 1))
        K[:, M:, :M] = Bm
        r = np.zeros((m, M + 2))
        r[:, :M] = np.einsum("ekp,ek->ep", Ahat, ftil)
        r[:, M] = gl
        r[:, M + 1] = gr
        W[s0:s1] = np.linalg.solve(K, r[..., None])[..., 0][:, :M]
    return W


# --------------------------------------------------------------------------
# evaluate_solution, Dual.py:176-203
# --------------------------------------------------------------------------
def locate_elements(nodes, xq):
    """First j with nodes[j] <= x <= nodes[j+1] (Dual.py:182-183); below / above the
    mesh -> element 0 / ne-1 (Dual.py:192-201); NaN -> -1 (no branch taken)."""
    nodes = np.asarray(nodes, dtype=np.float64)
    xq = np.asarray(xq, dtype=np.float64)
    ne = len(nodes) - 1
    j = np.searchsorted(nodes, xq, side="left") - 1
    j = np.clip(j, 0, ne - 1).astype(np.int64)
    j[np.isnan(xq)] = -1
    return j


def locate_elements_scan(nodes, xq):
    """Literal O(P*ne) scan of Dual.py:180-201, for small cases."""
    out = np.full(len(xq), -1, dtype=np.int64)
    ne = len(nodes) - 1
    for i, xi in enumerate(xq):
        for j in range(ne):
            if nodes[j] <= xi <= nodes[j + 1]:
                out[i] = j
                break
        else:
            if xi < nodes[0]:
                out[i] = 0
            elif xi > nodes[-1]:
                out[i] = ne - 1
    return out


def evaluate_solution(nodes, W, xq):
    """u(x) = Legendre(W[j], [x_j, x_{j+1}])(x): mapdomain then Clenshaw
    (_polybase.py:513-515); returns (u float64[P], elem int64[P])."""
    nodes = np.asarray(nodes, dtype=np.float64)
    xq = np.asarray(xq, dtype=np.float64)
    j = locate_elements(nodes, xq)
    u = np.zeros(len(xq))
    for i, (xi, ji) in enumerate(zip(xq, j)):
        if ji < 0:
            continue
        off, scl = mapparms(nodes[ji], nodes[ji + 1])
        u[i] = clenshaw(off + scl * xi, W[ji])
    return u, j


def evaluate_solution_vec(nodes, W, xq):
    """Vectorised form of :func:`evaluate_solution` (same arithmetic order)."""
    nodes = np.asarray(nodes, dtype=np.float64)
    xq = np.asarray(xq, dtype=np.float64)
    j = locate_elements(nodes, xq)
    jj = np.maximum(j, 0)
    a = nodes[jj]
    b = nodes[jj + 1]
    oldlen = b - a
    off = (b * -1.0 - a * 1.0) / oldlen
    scl = 2.0 / oldlen
    t = off + scl * xq
    M = W.shape[1]
    c = W[jj]
    if M == 1:
        u = c[:, 0] + 0 * t
    elif M == 2:
        u = c[:, 0] + c[:, 1] * t
    else:
        nd = M
        c0 = c[:, -2].copy()
        c1 = c[:, -1].copy()
        for i in range(3, M + 1):
            tmp = c0
            nd = nd - 1
            c0 = c[:, -i] - (c1 * (nd - 1)) / nd
            c1 = tmp + (c1 * t * (2 * nd - 1)) / nd
        u = c0 + c1 * t
    u = np.where(j < 0, 0.0, u)
    return u, j


# --------------------------------------------------------------------------
# P1 FEM step, Dual.py:110-137 (scikit-fem semantics per SURVEY.md Appendix C)
# --------------------------------------------------------------------------
def gauss_rule01(nquad=2):
    """Gauss-Legendre rule mapped to [0,1] (scikit-fem's default for P1 is 2 points:
    xi = 1/2 -+ 1/(2 sqrt 3), w = 1/2 -- SURVEY.md Appendix C)."""
    xg, wg = np.polynomial.legendre.leggauss(int(nquad))
    return 0.5 * (1.0 + xg), 0.5 * wg


def quad_points(nodes, nquad=2):
    nodes = np.asarray(nodes, dtype=np.float64)
    xi, _ = gauss_rule01(nquad)
    a = nodes[:-1]
    h = nodes[1:] - a
    return a[:, None] + h[:, None] * xi[None, :]


def p1_assemble_local(nodes, rhs=poisson_rhs, coef_a=None, nquad=2):
    """Element-local P1 stiffness and load (Dual.py:117-128).

    k_e = abar_e/h [[1,-1],[-1,1]] (abar_e = quadrature mean of a, 1 for Poisson; the
    reference's two minus signs cancel), f_e[j] = h sum_q w_q f(x_q) phi_j(xi_q) with
    the ``nquad``-point Gauss rule on [0,1].  Returns (kdiag[ne], fl[ne], fr[ne]):
    kdiag = abar_e/h, fl/fr = load on the element's left/right node.
    """
    nodes = np.asarray(nodes, dtype=np.float64)
    xi, wt = gauss_rule01(nquad)
    h = nodes[1:] - nodes[:-1]
    xq = quad_points(nodes, nquad)
    fq = np.asarray(rhs(xq), dtype=np.float64)
    sl = np.zeros_like(h)
    sr = np.zeros_like(h)
    am = np.zeros_like(h)
    for k in range(len(xi)):
        sl = sl + (wt[k] * (1.0 - xi[k])) * fq[:, k]
        sr = sr + (wt[k] * xi[k]) * fq[:, k]
        if coef_a is not None:
            am = am + wt[k] * coef_a(xq[:, k])
    abar = np.ones_like(h) if coef_a is None else am
    return abar / h, h * sl, h * sr


def p1_scatter(kdiag, fl, fr):
    """Scatter element-local pieces to the global tridiagonal system:
    diag[ne+1], off[ne] (off[i] couples nodes i,i+1), load[ne+1]."""
    ne = len(kdiag)
    diag = np.zeros(ne + 1)
    load = np.zeros(ne + 1)
    diag[:-1] += kdiag
    diag[1:] += kdiag
    load[:-1] += fl
    load[1:] += fr
    return diag, -kdiag, load


def thomas_dirichlet(diag, off, load, u0=0.0, u1=0.0):
    """Dirichlet on both end dofs (``enforce`` with D=all boundary dofs, Dual.py:129)
    then a tridiagonal solve of the interior (stands in for ``solve``, Dual.py:130)."""
    n = len(diag)
    u = np.zeros(n)
    u[0], u[-1] = u0, u1
    m = n - 2
    if m <= 0:
        return u
    d = diag[1:-1].copy()
    r = load[1:-1].copy()
    lo = off[1:-1].copy()       # couples interior i, i+1 (global nodes i+1, i+2)
    r[0] -= off[0] * u0
    r[-1] -= off[-1] * u1
    for i in range(1, m):
        wgt = lo[i - 1] / d[i - 1]
        d[i] -= wgt * lo[i - 1]
        r[i] -= wgt * r[i - 1]
    x = np.zeros(m)
    x[-1] = r[-1] / d[-1]
    for i in range(m - 2, -1, -1):
        x[i] = (r[i] - lo[i] * x[i + 1]) / d[i]
    u[1:-1] = x
    return u


def banded_dirichlet(diag, off, load, u0=0.0, u1=0.0):
    """Same system as :func:`thomas_dirichlet`, LAPACK banded solve (fast for 1e7 dofs)."""
    from scipy.linalg import solve_banded
    n = len(diag)
    u = np.zeros(n)
    u[0], u[-1] = u0, u1
    m = n - 2
    if m <= 0:
        return u
    r = load[1:-1].copy()
    r[0] -= off[0] * u0
    r[-1] -= off[-1] * u1
    ab = np.zeros((3, m))
    ab[1] = diag[1:-1]
    ab[0, 1:] = off[1:-1]
    ab[2, :-1] = off[1:-1]
    u[1:-1] = solve_banded((1, 1), ab, r)
    return u


def fem_p1_solve(nodes, rhs=poisson_rhs, coef_a=None, nquad=2):
    """``solve_fem`` (Dual.py:110-137) -> nodal values float64[ne+1]."""
    kdiag, fl, fr = p1_assemble_local(nodes, rhs, coef_a, nquad)
    diag, off, load = p1_scatter(kdiag, fl, fr)
    if len(diag) > 4096:
        return banded_dirichlet(diag, off, load)
    return thomas_dirichlet(diag, off, load)


_GOLDEN_V1_XI = (0.5 - 0.5 / np.sqrt(3.0), 0.5 + 0.5 / np.sqrt(3.0))


def fem_p1_solve_golden_v1(nodes, rhs=poisson_rhs):
    """FROZEN: the P1 stand-in exactly as it was when ``tests/golden/G*.npz`` were written
    (2-point Gauss load in this operation order, Thomas elimination at every size).  The
    fixtures' ``values_sel`` are its output bit for bit; ``oracle/gen_golden.py`` calls this and
    nothing else for nodal values, so a regeneration reproduces the committed inputs.  Do not
    edit -- :func:`fem_p1_solve` is the living restatement (any quadrature order, LAPACK banded
    solve on large meshes: nodal values 3e-11 away at 1e5 elements)."""
    nodes = np.asarray(nodes, dtype=np.float64)
    a = nodes[:-1]
    b = nodes[1:]
    h = b - a
    x1 = a + h * _GOLDEN_V1_XI[0]
    x2 = a + h * _GOLDEN_V1_XI[1]
    f1 = rhs(x1)
    f2 = rhs(x2)
    kdiag = np.ones_like(h) / h
    hw = 0.5 * h
    fl = hw * (f1 * (1.0 - _GOLDEN_V1_XI[0]) + f2 * (1.0 - _GOLDEN_V1_XI[1]))
    fr = hw * (f1 * _GOLDEN_V1_XI[0] + f2 * _GOLDEN_V1_XI[1])
    diag, off, load = p1_scatter(kdiag, fl, fr)
    return thomas_dirichlet(diag, off, load)


# --------------------------------------------------------------------------
# the reference's own SLSQP loop = the CPU baseline (Dual.py:20-98, 139-169)
# --------------------------------------------------------------------------
def slsqp_element(rhs, a, b, u_l, u_r, M, gamma, n=12, left=False, right=False,
                  global_domain=(-1.0, 1.0), rng=None, bc_left=0.0, bc_right=0.0,
                  coef_a=None, coef_da=None):
    """One ``lssvr_primal`` call restated step for step: same unknown vector
    [w(M), e(n)], same objective (Dual.py:46-49), same constraint vector built
    point by point through ``Legendre.deriv(2)`` (Dual.py:43-44,51-78), same start
    (Dual.py:81) and the same SLSQP options with finite-difference derivatives
    (Dual.py:87-88).  Returns (coef[M], success).

    ``coef_a`` / ``coef_da`` (callables): the residual of BASELINE config 5,
    ``-(a u')' - f = -a u'' - a' u' - f`` -- an EXTENSION of Dual.py:43-44 (the reference
    hard-codes ``-u''``) written the way the reference writes its own residual: one more
    ``series.deriv(1)(xk)`` per point."""
    from numpy.polynomial.legendre import Legendre
    from scipy.optimize import minimize

    dom = [a, b]
    pts = np.linspace(a, b, n)
    g_l = bc_left if (left and a == global_domain[0]) else u_l
    g_r = bc_right if (right and b == global_domain[1]) else u_r

    def cost(z):
        return 0.5 * np.linalg.norm(z[:M]) ** 2 + gamma / 2 * np.sum(z[M:M + n] ** 2)

    def eq(z):
        series = Legendre(z[:M], dom)
        slack = z[M:M + n]
        rows = []
        for k, xk in enumerate(pts):
            if coef_a is None:
                rows.append(-series.deriv(2)(xk) - rhs(xk) + slack[k])
            else:
                rows.append(-coef_a(xk) * series.deriv(2)(xk) - coef_da(xk) * series.deriv(1)(xk)
                            - rhs(xk) + slack[k])
        rows.append(series(a) - g_l)
        rows.append(series(b) - g_r)
        return np.array(rows)

    rand = np.random.rand(M) if rng is None else rng.random(M)
    z0 = np.concatenate([rand * 0.01, np.zeros(n)])
    res = minimize(cost, x0=z0, constraints={"type": "eq", "fun": eq}, method="SLSQP",
                   options={"maxiter": 1000, "ftol": 1e-12})
    return res.x[:M].copy(), bool(res.success)


def slsqp_loop(nodes, values, M, gamma, n=12, rhs=poisson_rhs, global_domain=None,
               elements=None, seed=0):
    """``solve_lssvr_subproblems`` (Dual.py:139-169) over ``elements`` (default all)."""
    nodes = np.asarray(nodes, dtype=np.float64)
    ne = len(nodes) - 1
    if global_domain is None:
        global_domain = (float(nodes[0]), float(nodes[-1]))
    if elements is None:
        elements = range(ne)
    rng = np.random.default_rng(seed)
    out = []
    ok = []
    for i in elements:
        c, s = slsqp_element(rhs, nodes[i], nodes[i + 1], values[i], values[i + 1], M, gamma, n,
                             left=(i == 0), right=(i == ne - 1),
                             global_domain=global_domain, rng=rng)
        out.append(c)
        ok.append(s)
    return np.array(out), np.array(ok)


# --------------------------------------------------------------------------
# norms
# --------------------------------------------------------------------------
def rel_l2_coef(c, c_ref):
    """Relative L2([a,b]) distance of two Legendre series on the same element:
    |sum d_p L_p|^2 = (h/2) sum d_p^2 * 2/(2p+1)  (h cancels in the ratio)."""
    c = np.asarray(c, dtype=np.float64)
    c_ref = np.asarray(c_ref, dtype=np.float64)
    wgt = 1.0 / (2.0 * np.arange(c.shape[-1]) + 1.0)
    num = np.sqrt(np.sum((c - c_ref) ** 2 * wgt, axis=-1))
    den = np.sqrt(np.sum(c_ref ** 2 * wgt, axis=-1))
    return num / den


def rel_l2_bubble(c, c_ref):
    """The same weighted norm restricted to the ENHANCEMENT, p >= 2, relative to the bubble's
    own norm: |sum_{p>=2} d_p L_p| / |sum_{p>=2} c_ref_p L_p|.

    Why it exists: on fine meshes the linear part (w_0, w_1) carries all but ~1.2 h^2 of the
    element polynomial's norm (1.5e-10 on 1e5 elements of [-1,1], 1.5e-14 on 1e7), so
    :func:`rel_l2_coef` cannot tell a correct enhancement from none at all there: with
    ``c[..., 2:] = 0`` it reads 1.5e-14, this function reads exactly 1.  Rows whose reference
    has no bubble (M = 2, or an element in the linear fallback) give 0 when ``c`` has none either,
    inf otherwise."""
    c = np.asarray(c, dtype=np.float64)
    c_ref = np.asarray(c_ref, dtype=np.float64)
    wgt = 1.0 / (2.0 * np.arange(c.shape[-1]) + 1.0)
    wgt[:2] = 0.0
    num = np.sqrt(np.sum((c - c_ref) ** 2 * wgt, axis=-1))
    den = np.sqrt(np.sum(c_ref ** 2 * wgt, axis=-1))
    with np.errstate(divide="ignore", invalid="ignore"):
        out = np.where(den > 0, num / np.where(den > 0, den, 1.0), np.where(num > 0, np.inf, 0.0))
    return out


def rel_l2_global(W, W_ref, nodes):
    """Relative L2 over the whole mesh of the piecewise polynomial W vs W_ref."""
    h = np.diff(np.asarray(nodes, dtype=np.float64))
    wgt = 1.0 / (2.0 * np.arange(W.shape[1]) + 1.0)
    num = np.sum(h[:, None] * (W - W_ref) ** 2 * wgt[None, :])
    den = np.sum(h[:, None] * W_ref ** 2 * wgt[None, :])
    return float(np.sqrt(num / den))


# --------------------------------------------------------------------------
# BASELINE config 5: -(a u')' = f, manufactured u = sin(pi x)
# --------------------------------------------------------------------------
def varcoef_params(seed=20260130, K=8):
    rng = np.random.default_rng(seed)
    c = rng.uniform(-1.0, 1.0, K)
    phi = rng.uniform(0.0, 2.0 * np.pi, K)
    return c, phi


def varcoef_functions(c, phi):
    """a(x) = 1 + 0.5 sum_k c_k sin(k pi x + phi_k)/k, a'(x), and
    f = -a' u' - a u'' for u = sin(pi x)."""
    k = np.arange(1, len(c) + 1, dtype=np.float64)

    def a(x):
        x = np.asarray(x, dtype=np.float64)
        return 1.0 + 0.5 * np.sum(c * np.sin(k * np.pi * x[..., None] + phi) / k, axis=-1)

    def da(x):
        x = np.asarray(x, dtype=np.float64)
        return 0.5 * np.pi * np.sum(c * np.cos(k * np.pi * x[..., None] + phi), axis=-1)

    def f(x):
        x = np.asarray(x, dtype=np.float64)
        return -da(x) * np.pi * np.cos(np.pi * x) + a(x) * np.pi ** 2 * np.sin(np.pi * x)

    return a, da, f
