"""Prototype (numpy float64) of the Chebyshev-moment form of the per-element solve.

Rows of the Poisson residual are pure polynomials of t, so in the Chebyshev basis
    G_ij = sum_k T_i(t_k) T_j(t_k) = 1/2 (m_{i+j} + m_{|i-j|}),  m_d = sum_k T_d(t_k):
the Gram needs 2 MR - 1 moments per point instead of MR (MR+1)/2 products.
Boundary rows to first order in (1 + t_a, 1 - t_b).  Checked against the 60-digit truth.
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from numpy.polynomial import chebyshev as ch, legendre as lg
from oracle import lssvr_oracle as orc
from oracle import closed_form_mp as cf

_tab = {}
def tables(M):
    """X: L''_{j+2} = sum_i X[i,j] T_i  (MR x MR, upper triangular);  Xi = X^-1; a_p = p(p+1)/2."""
    if M in _tab: return _tab[M]
    MR = M - 2
    X = np.zeros((MR, MR))
    for j in range(MR):
        c = np.zeros(j + 3); c[j + 2] = 1.0
        d2 = lg.legder(c, 2)                # Legendre series of L''_{j+2}
        cheb = ch.poly2cheb(lg.leg2poly(d2))
        X[:len(cheb), j] = cheb
    Xi = np.linalg.inv(X)
    Xi[np.abs(Xi) < 1e-300] = 0
    for i in range(MR):
        for j in range(MR):
            if (i + j) % 2: X[i, j] = 0; Xi[i, j] = 0
    a = np.array([(j + 2) * (j + 3) / 2.0 for j in range(MR)])
    _tab[M] = (X, Xi, a)
    return _tab[M]

def solve_cheb(s, first_order_ridge=True):
    M, n = s.M, s.n
    MR = M - 2
    X, Xi, a = tables(M)
    t = s.t
    eps = 1.0 / s.gamma_t
    ta = s.off + s.scl * s.a
    tb = s.off + s.scl * s.b
    ea = 1.0 + ta; eb = 1.0 - tb          # exact (Sterbenz)
    sig = 0.5 * (ea + eb); dl = 0.5 * (ea - eb)
    idet = 0.5 * (1.0 + sig + sig * sig)  # 1/(tb - ta)
    gl, gr = s.g
    d0 = (tb * gl - ta * gr) * idet
    d1 = (gr - gl) * idet
    if MR == 0:
        return np.array([d0, d1])
    # C (2 x MR) to first order, v-basis
    ev = (np.arange(MR) % 2 == 0)
    C0 = np.where(ev, 1.0 - a * sig, (a - 1.0) * dl)
    C1 = np.where(ev, a * dl, 1.0 - (a - 1.0) * sig)
    # z-basis: v = Xi z
    C0z = C0 @ Xi; C1z = C1 @ Xi
    N = Xi.T @ Xi
    # moments
    D = 2 * MR - 1
    Tk = np.zeros((n, max(D, 2)))
    Tk[:, 0] = 1.0; Tk[:, 1] = t
    for d in range(2, D):
        Tk[:, d] = 2 * t * Tk[:, d - 1] - Tk[:, d - 2]
    m = 0.5 * Tk.sum(0)
    G = np.zeros((MR, MR))
    for i in range(MR):
        for j in range(MR):
            G[i, j] = m[i + j] + m[abs(i - j)]
    phi = -s.ftil
    r = -(Tk[:, :MR].T @ phi)             # rows are -T: (Ahat w)_k = -T_k^T z ; r = Ahat^T ftil = -T^T ftil
    r = -(Tk[:, :MR].T @ s.ftil)
    # Abar = Ahat2 - Ahat1 C = Ahat2 (Poisson), fbar = ftil
    S = G + eps * (N + np.outer(C0z, C0z) + np.outer(C1z, C1z))
    rhs = r + eps * (C0z * d0 + C1z * d1)
    # LDL^T (no pivoting), as the kernel
    A = S.copy(); y = rhs.copy()
    L = np.eye(MR); dd = np.zeros(MR)
    for j in range(MR):
        dd[j] = A[j, j]
        L[j + 1:, j] = A[j + 1:, j] / dd[j]
        A[j + 1:, j + 1:] -= np.outer(L[j + 1:, j], A[j, j + 1:])
    y = np.linalg.solve(L, rhs); z = np.linalg.solve(L.T, y / dd)
    v = Xi @ z
    w = np.zeros(M)
    w[2:] = v
    w[0] = d0 - C0 @ v
    w[1] = d1 - C1 @ v
    return w

def check(nodes, values, M, gamma, n, sel, gd=None):
    ne = len(nodes) - 1
    gd = gd or (nodes[0], nodes[-1])
    errs = []; errs_o = []
    for i in sel:
        gl, gr = orc.boundary_values(i, ne, nodes[i], nodes[i + 1], values[i], values[i + 1], gd)
        s = orc.element_system(nodes[i], nodes[i + 1], gl, gr, M, gamma, n)
        tr = cf.solve_truth(s)
        w = solve_cheb(s)
        wo = orc.solve_bc_eliminated(s)
        errs.append(orc.rel_l2_coef(w[None], tr[None])[0]); errs_o.append(orc.rel_l2_coef(wo[None], tr[None])[0])
    return max(errs), max(errs_o)

if __name__ == "__main__" and len(sys.argv) == 1:
    rng = np.random.default_rng(0)
    cases = [("C1 ne8 M5 n5", np.linspace(-1, 1, 9), 5, 1e4, 5),
             ("default ne24 M8 n12", np.linspace(-1, 1, 25), 8, 1e4, 12),
             ("ne24 M9 n16", np.linspace(-1, 1, 25), 9, 1e4, 16),
             ("ne4096 M9 n16", np.linspace(-1, 1, 4097), 9, 1e4, 16),
             ("ne1e5 narrow M9 n16", np.linspace(-1, 1, 100001), 9, 1e4, 16),
             ("classdef ne4 M12 n12 g1e6", np.linspace(-1, 1, 5), 12, 1e6, 12),
             ("M22 n40", np.linspace(-1, 1, 25), 22, 1e4, 40),
             ("M16 n20", np.linspace(-1, 1, 25), 16, 1e4, 20),
             ("M33 n64", np.linspace(-1, 1, 25), 33, 1e4, 64),
             ("M33 n64 ne1e5", np.linspace(-1, 1, 100001), 33, 1e4, 64),
             ("M9 n7 (n = MR)", np.linspace(-1, 1, 25), 9, 1e4, 7),
             ("M9 n200", np.linspace(-1, 1, 25), 9, 1e4, 200),
             ("gamma 1e-2 h=0.5", np.linspace(-1, 1, 5), 9, 1e-2, 16),
             ("gamma 1 h=2", np.linspace(-1, 1, 2), 9, 1.0, 16),
             ]
    for name, nodes, M, gamma, n in cases:
        values = np.sin(np.pi * nodes); values[0] = values[-1] = 0
        ne = len(nodes) - 1
        sel = sorted(set([0, 1, ne // 3, ne // 2, ne - 1]) & set(range(ne)))
        e, eo = check(nodes, values, M, gamma, n, sel)
        print(f"{name:28s} cheb-moment {e:.2e}   bc-eliminated oracle {eo:.2e}")
    # wide domains
    for ne, half in ((100008, 4167.0), (10000008, 416667.0)):
        nodes_all = None
        sel = [0, 1, ne // 3, ne // 2 + 7, ne - 2, ne - 1]
        step = 2 * half / ne
        errs = []; errs_o = []
        for i in sel:
            a = i * step - half; b = (i + 1) * step - half if i + 1 < ne else half
            if i == 0: a = -half
            gl = 0.0 if i == 0 else np.sin(np.pi * a); gr = 0.0 if i == ne - 1 else np.sin(np.pi * b)
            s = orc.element_system(a, b, gl, gr, 9, 1e4, 16)
            tr = cf.solve_truth(s)
            errs.append(orc.rel_l2_coef(solve_cheb(s)[None], tr[None])[0])
            errs_o.append(orc.rel_l2_coef(orc.solve_bc_eliminated(s)[None], tr[None])[0])
        print(f"wide ne={ne:9d} M9 n16       cheb-moment {max(errs):.2e}   bc-eliminated oracle {max(errs_o):.2e}")
    # random stress
    worst = 0; worst_o = 0
    for it in range(300):
        M = int(rng.integers(3, 23)); n = int(rng.integers(max(M - 2, 2), 3 * M + 4))
        gamma = 10.0 ** rng.uniform(0, 8); h = 10.0 ** rng.uniform(-6, 0.3); x0 = rng.uniform(-1e3, 1e3) * h * 10 ** rng.uniform(0, 3)
        a, b = x0, x0 + h
        s = orc.element_system(a, b, rng.normal(), rng.normal(), M, gamma, n)
        tr = cf.solve_truth(s)
        e = orc.rel_l2_coef(solve_cheb(s)[None], tr[None])[0]
        eo = orc.rel_l2_coef(orc.solve_bc_eliminated(s)[None], tr[None])[0]
        if e > worst: worst = e; wc = (M, n, gamma, h, x0, e, eo)
        worst_o = max(worst_o, eo)
    print("stress worst cheb", wc, " oracle worst", worst_o)


# ---------------------------------------------------------------------------------------------
# the same algorithm arranged exactly as the HIP kernel does it (x2 scaling, product moments,
# structured first-order ridge, generated constant tables)
# ---------------------------------------------------------------------------------------------
_gen = None
def gen_tables():
    global _gen
    if _gen is None:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gen"))
        import gen_cheb_tables as g
        Y, a, al, b, N = g.compute_tables()
        X = g.compute_inverse(Y)
        f = lambda m: np.array([[float(x) for x in r] for r in m])
        _gen = (f(Y), np.array([float(x) for x in a]), np.array([float(x) for x in al]),
                np.array([float(x) for x in b]), f(N))
        global _genX
        _genX = f(X)
    return _gen

_genX = None

def solve_cheb_kernel(s):
    Y, a, al, bb, N = gen_tables()
    M, n = s.M, s.n
    MR = M - 2
    t = s.t
    hh = 0.5 * (s.b - s.a)
    inv_scl2 = hh * hh
    eps2 = 2.0 * inv_scl2 * inv_scl2 * (1.0 / s.gamma)
    ta = s.off + s.scl * s.a
    tb = s.off + s.scl * s.b
    ea = 1.0 + ta; eb = 1.0 - tb
    sig = 0.5 * (ea + eb); dl = 0.5 * (ea - eb)
    idet = 0.5 * (1.0 + sig * (1.0 + sig))
    gl, gr = s.g
    d0 = (tb * gl - ta * gr) * idet
    d1 = (gr - gl) * idet
    if MR == 0:
        return np.array([d0, d1])
    phi2 = -2.0 * (s.f * inv_scl2)
    T = np.zeros((n, MR)); T[:, 0] = 1.0
    if MR > 1: T[:, 1] = t
    for d in range(2, MR):
        T[:, d] = (t + t) * T[:, d - 1] - T[:, d - 2]
    m = T.sum(0); m[0] = n
    P = (T[:, MR - 1:MR] * T).sum(0)        # P[j] = sum T_{MR-1} T_j
    r = (T * phi2[:, None]).sum(0)
    Mom = np.zeros(2 * MR - 1)
    Mom[:MR] = m
    for j in range(1, MR):
        Mom[MR - 1 + j] = 2.0 * P[j] - m[MR - 1 - j]
    S = np.zeros((MR, MR)); rhs = np.zeros(MR)
    es = eps2 * sig; ed = eps2 * dl
    e_d0 = eps2 * d0; e_d1 = eps2 * d1
    q_ev = eps2 * (dl * d1 - sig * d0); q_od = eps2 * (dl * d0 - sig * d1)
    for i in range(MR):
        for k in range(i + 1):
            v = Mom[i + k] + Mom[i - k]
            Q = al[i] * bb[k] + bb[i] * al[k]
            if (i + k) % 2 == 0:
                v += eps2 * (N[i, k] + al[i] * al[k]) - es * Q
            else:
                v += ed * Q
            S[i, k] = S[k, i] = v
        if i % 2 == 0:
            rhs[i] = r[i] + al[i] * e_d0 + bb[i] * q_ev
        else:
            rhs[i] = r[i] + al[i] * e_d1 + bb[i] * q_od
    A = S.copy()
    L = np.eye(MR); dd = np.zeros(MR)
    for j in range(MR):
        dd[j] = A[j, j]
        L[j + 1:, j] = A[j + 1:, j] / dd[j]
        A[j + 1:, j + 1:] -= np.outer(L[j + 1:, j], A[j, j + 1:])
    y = np.linalg.solve(L, rhs); z = np.linalg.solve(L.T, y / dd)
    v = Y[:MR, :MR] @ z
    ev = (np.arange(MR) % 2 == 0)
    C0 = np.where(ev, 1.0 - a[:MR] * sig, (a[:MR] - 1.0) * dl)
    C1 = np.where(ev, a[:MR] * dl, 1.0 - (a[:MR] - 1.0) * sig)
    w = np.zeros(M)
    w[2:] = v
    w[0] = d0 - C0 @ v
    w[1] = d1 - C1 @ v
    return w


# ---------------------------------------------------------------------------------------------
# RIDGE-DOMINATED elements (gamma scl^4 below ridge_gamma_scl4(M), round 4): the kernels' cold path
# (enhance_small_cheb.hpp::cheb_ridge_solve, enhance_large_cheb.hip::ridge_wave_solve).  Same moments,
# mapped back to the Legendre-bubble basis with the exact table X = Y^-1 (rho_j = sum_i X[i][j] T_i):
#     S2_v = X^T (m_{i+k} + m_{|i-k|}) X + eps2 (I + C^T C),   rhs2_v = X^T r2 + eps2 C^T d,
# exact boundary rows (Legendre recurrence).  In the Chebyshev basis the ridge is eps (N + C_z^T C_z),
# N = Y^T Y, and once eps outweighs the Gram the solve inherits cond(Y)^2.
# ---------------------------------------------------------------------------------------------
def ridge_gamma_scl4(M):
    """lssvr_device.hpp::ridge_gamma_scl4"""
    return 3.0e-4 if M <= 12 else (1.0e-4 if M <= 17 else 3.0e-5)


def solve_ridge_kernel(s):
    gen_tables()
    M, n = s.M, s.n
    MR = M - 2
    X = _genX[:MR, :MR]
    t = s.t
    hh = 0.5 * (s.b - s.a)
    inv_scl2 = hh * hh
    eps2 = 2.0 * inv_scl2 * inv_scl2 * (1.0 / s.gamma)
    ta = s.off + s.scl * s.a
    tb = s.off + s.scl * s.b
    idet = 1.0 / (tb - ta)
    gl, gr = s.g
    d0 = (tb * gl - ta * gr) * idet
    d1 = (gr - gl) * idet
    if MR == 0:
        return np.array([d0, d1])
    La = np.zeros(M); Lb = np.zeros(M); La[0] = Lb[0] = 1.0; La[1] = ta; Lb[1] = tb
    for p in range(1, M - 1):
        La[p + 1] = ((2 * p + 1) * ta * La[p] - p * La[p - 1]) / (p + 1)
        Lb[p + 1] = ((2 * p + 1) * tb * Lb[p] - p * Lb[p - 1]) / (p + 1)
    C0 = (tb * La[2:] - ta * Lb[2:]) * idet
    C1 = (Lb[2:] - La[2:]) * idet
    phi2 = -2.0 * (s.f * inv_scl2)
    T = np.zeros((n, MR)); T[:, 0] = 1.0
    if MR > 1: T[:, 1] = t
    for d in range(2, MR):
        T[:, d] = (t + t) * T[:, d - 1] - T[:, d - 2]
    m = T.sum(0); m[0] = n
    P = (T[:, MR - 1:MR] * T).sum(0)
    r = (T * phi2[:, None]).sum(0)
    Mom = np.zeros(2 * MR - 1)
    Mom[:MR] = m
    for j in range(1, MR):
        Mom[MR - 1 + j] = 2.0 * P[j] - m[MR - 1 - j]
    G2 = np.array([[Mom[i + k] + Mom[abs(i - k)] for k in range(MR)] for i in range(MR)])
    S = X.T @ (G2 @ X) + eps2 * (np.eye(MR) + np.outer(C0, C0) + np.outer(C1, C1))
    rhs = X.T @ r + eps2 * (C0 * d0 + C1 * d1)
    A = S.copy()
    L = np.eye(MR); dd = np.zeros(MR)
    for j in range(MR):
        dd[j] = A[j, j]
        L[j + 1:, j] = A[j + 1:, j] / dd[j]
        A[j + 1:, j + 1:] -= np.outer(L[j + 1:, j], A[j, j + 1:])
    y = np.linalg.solve(L, rhs); v = np.linalg.solve(L.T, y / dd)
    w = np.zeros(M)
    w[2:] = v
    w[0] = d0 - C0 @ v
    w[1] = d1 - C1 @ v
    return w


def solve_kernel_dispatch(s):
    """What the Poisson kernels do per element: the ridge form below the threshold, else the moment form."""
    if s.M >= 5 and s.gamma_t < ridge_gamma_scl4(s.M):          # (kRidgeMinM: Y is diagonal up to M = 4)
        return solve_ridge_kernel(s)
    return solve_cheb_kernel(s)


def ridge_sweep(seed=5, reps=3):
    """Both forms against the 60-digit minimiser across gamma scl^4 (smooth: h = 0.5; rough: h = 100, fifty
    periods of the right-hand side per element): where the threshold of ridge_gamma_scl4 comes from."""
    rng = np.random.default_rng(seed)
    print("M  n   h     g*scl^4 | direct Gram | moment form | ridge form | dispatch      (coef / bubble)")
    for (M, n) in [(5, 5), (9, 16), (14, 28), (22, 44), (33, 64)]:
        for h in [0.5, 100.0]:
            for g4 in [1e-12, 1e-9, 1e-6, 1e-5, 3e-5, 1e-4, 3e-4, 1e-3, 1e-2, 1.0, 1e4]:
                gamma = g4 / (2.0 / h) ** 4
                worst = np.zeros((4, 2))
                for _ in range(reps):
                    x0 = rng.uniform(-30, 30) * h
                    s = orc.element_system(x0, x0 + h, rng.normal(), rng.normal(), M, gamma, n)
                    tr = cf.solve_truth(s)
                    for i, f in enumerate([orc.solve_bc_eliminated, solve_cheb_kernel, solve_ridge_kernel,
                                           solve_kernel_dispatch]):
                        w = f(s)
                        worst[i] = np.maximum(worst[i], [orc.rel_l2_coef(w[None], tr[None])[0],
                                                         orc.rel_l2_bubble(w[None], tr[None])[0]])
                print(f"{M:2d} {n:2d} {h:6.1f} {g4:8.0e} | " + " | ".join(f"{w[0]:.1e}/{w[1]:.1e}" for w in worst))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "ridge":
    ridge_sweep()
