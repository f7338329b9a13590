"""Host-side mirror of the reference's solve-then-enhance call surface.

Same names, argument meaning and error behaviour as
``1D-Possion/Hybrid-FEM-LSSVR-Dual.py`` ("Dual.py"):

* :func:`lssvr_primal`                 -- Dual.py:20-98
* :class:`FEMLSSVRPrimalSolver`        -- Dual.py:100-203
  (``solve_fem``, ``solve_lssvr_subproblems``, ``solve``, ``evaluate_solution``;
  attributes ``fem_nodes``, ``fem_values``, ``lssvr_functions``)
* :func:`true_solution`, :func:`poisson_rhs`, ``main_boundary_condition_*`` -- Dual.py:8-18

What differs is where the arithmetic runs: every step is a hand-written gfx950
kernel behind ``liblssvr_hip.so``; this module only moves arrays to the device,
chooses the right-hand-side mode and wraps results.  There is no CPU fallback:
without the HIP library or without a GPU every solve raises.

Additions over the reference (all keyword-only, defaults reproduce the reference):
``n_colloc`` (hard-coded 12 at Dual.py:40), ``rhs`` (``poisson_rhs`` at Dual.py:157),
``nquad`` (scikit-fem's default 2-point rule for P1), ``mesh`` (a scikit-fem style
mesh/basis or node array instead of ``np.linspace(a, b, num_fem_nodes)``), ``fem_solver``
(``"bands"``: the assembled float64 tridiagonal system, as ``enforce`` + ``solve`` see it;
``"flux"``: exact-structure prefix-scan solve), ``solver``
(``ops.SOLVER_PRIMAL`` default; ``ops.SOLVER_SHARED``: uniform meshes only, one shared operator
applied per element).
"""
from __future__ import annotations

import numpy as np
from numpy.polynomial.legendre import Legendre

from . import ops
from .mesh import LineMesh, P1Basis, as_line_mesh


# --------------------------------------------------------------------------
# problem definition (Dual.py:8-18)
# --------------------------------------------------------------------------
def true_solution(x):
    """Dual.py:8-9."""
    return np.sin(np.pi * x)


class SinRHS:
    """f(x) = amp * sin(omega * x): callable like the reference's ``poisson_rhs``
    (numpy arithmetic, Dual.py:11-12) and recognised by the facade, which then
    evaluates it inside the kernels instead of tabulating it on the host."""

    def __init__(self, amp, omega):
        self.amp = float(amp)
        self.omega = float(omega)

    def __call__(self, x):
        return self.amp * np.sin(self.omega * x)

    def __repr__(self):
        return f"SinRHS(amp={self.amp!r}, omega={self.omega!r})"


poisson_rhs = SinRHS(np.pi ** 2, np.pi)          # Dual.py:11-12


def main_boundary_condition_left(x):
    """Dual.py:14-15."""
    return 0.0


def main_boundary_condition_right(x):
    """Dual.py:17-18."""
    return 0.0


# --------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------
def _torch():
    import torch
    return torch


def _device(device):
    torch = _torch()
    if not torch.cuda.is_available():
        raise RuntimeError("hybrid_fem_lssvr_amd needs an MI355X (no GPU visible); "
                           "the HIP path has no CPU fallback")
    return torch.device(device)


def _to_dev(a, device):
    torch = _torch()
    return torch.as_tensor(np.ascontiguousarray(np.asarray(a, dtype=np.float64)), device=device)


def _rhs_mode(rhs, x_dev, n_colloc, M=None):
    """Keyword arguments of ops.enhance for the right-hand side: ``rhs=(amp, omega)`` for the named
    f, else the callable (Dual.py:20 ``rhs_func``) tabulated on the host at np.linspace's points
    (Dual.py:40).  The callable ALWAYS sees the element-major array ``x[e, k]`` -- points along the last
    axis, as in the reference, where it receives the 1-D array of one element's n points (Dual.py:40-44): a
    callable that returns a per-point vector of shape (n,), or otherwise relies on the last axis being the
    points, broadcasts the same at every degree (up to M = 22 it used to see the transposed
    array).  The TABLE is then handed over point-major (``t[k, e]``) for the lane-per-element kernels
    (M <= 22), which read that layout at full HBM rate, element-major otherwise; same values either way."""
    if isinstance(rhs, SinRHS):
        return dict(rhs=(rhs.amp, rhs.omega))
    pm = M is not None and int(M) <= 22
    xc = ops.colloc_points(x_dev, n_colloc)                    # np.linspace per element: [ne, n]
    f = np.asarray(rhs(xc.cpu().numpy()), dtype=np.float64)
    if f.shape != tuple(xc.shape):
        f = np.broadcast_to(f, tuple(xc.shape))
    if pm:
        f = np.ascontiguousarray(f.T)
    return dict(rhs_values=_to_dev(f, x_dev.device), point_major=pm)


def _enhance(x, u, M, gamma, n_colloc, *, global_domain, bc, solver, uniform_rtol=1e-9, **kw):
    """ops.enhance, or -- for ``solver=ops.SOLVER_SHARED`` -- the uniform-mesh shortcut: checks that
    every element length is within ``uniform_rtol`` of the mean, builds the shared operator with
    the general kernel and applies it (``lssvr_enhance_shared``)."""
    if solver != ops.SOLVER_SHARED:
        return ops.enhance(x, u, M, gamma, n_colloc, global_domain=global_domain, bc=bc, solver=solver, **kw)
    ne = x.numel() - 1
    if ne < 1:
        raise ValueError("need at least one element")
    hs = x[1:] - x[:-1]
    h0 = float(((x[-1] - x[0]) / ne).item())
    dev_rel = float(((hs / h0 - 1.0).abs().max()).item())
    if not dev_rel <= uniform_rtol:
        raise ValueError(f"solver='shared' needs a uniform mesh: element lengths deviate by {dev_rel:.2e} "
                         f"(> uniform_rtol = {uniform_rtol:.1e}); use the general solver")
    op = ops.build_shared_operator(h0, M, gamma, n_colloc, device=x.device)
    return ops.enhance_shared(x, u, op, M, n_colloc, global_domain=global_domain, bc=bc, **kw)


class EnhancedSolution:
    """Per-element Legendre coefficients on the device plus what is needed to use
    them: ``W`` float64[ne, M] (row i = ``lssvr_functions[i].coef``), ``nodes``
    float64[ne+1], ``status`` int32[ne] (1 = linear-interpolant fallback)."""

    def __init__(self, nodes, W, status):
        self.nodes = nodes
        self.W = W
        self.status = status

    @property
    def n_fallback(self):
        return int((self.status != 0).sum().item())

    def evaluate(self, x_points, return_elements=False):
        """``evaluate_solution`` (Dual.py:176-203) on the device."""
        xq = _to_dev(np.asarray(x_points, dtype=np.float64).reshape(-1), self.W.device)
        u, elem = ops.evaluate(self.nodes, self.W, xq, want_elem=True)
        u = u.cpu().numpy()
        if return_elements:
            return u, elem.cpu().numpy()
        return u


class _ElementFunctions:
    """``solver.lssvr_functions``: a read-only sequence whose item i is
    ``Legendre(W[i], [x_i, x_{i+1}])`` (Dual.py:95, 163) -- built lazily from the device
    coefficients, so a 1e7-element solve does not create 1e7 Python objects."""

    def __init__(self, nodes_host, W_dev):
        self._nodes = nodes_host
        self._W_dev = W_dev
        self._W = None

    def _host(self):
        if self._W is None:
            self._W = self._W_dev.cpu().numpy()
        return self._W

    def __len__(self):
        return self._W_dev.shape[0]

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        n = len(self)
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError(i)
        return Legendre(self._host()[i], [self._nodes[i], self._nodes[i + 1]])

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]


def enhance_elements(mesh, nodal_values, M, gamma, *, n_colloc=12, rhs=poisson_rhs,
                     global_domain=None, bc=(0.0, 0.0), device="cuda:0", solver=ops.SOLVER_PRIMAL):
    """Batched entry: every element of ``mesh`` (scikit-fem style mesh / basis, or
    node array) gets its LSSVR polynomial in one launch.  Returns EnhancedSolution."""
    dev = _device(device)
    m = as_line_mesh(mesh)
    x = _to_dev(m.nodes, dev)
    u = _to_dev(nodal_values, dev)
    if u.numel() != x.numel():
        raise ValueError("nodal_values must have one value per mesh node")
    if global_domain is None:
        global_domain = (float(m.nodes[0]), float(m.nodes[-1]))
    kw = _rhs_mode(rhs, x, n_colloc, M)
    W, st = _enhance(x, u, int(M), float(gamma), int(n_colloc), global_domain=global_domain,
                     bc=bc, solver=solver, **kw)
    return EnhancedSolution(x, W, st)


def enhance_elements_hetero(mesh, nodal_values, M, gamma, *, n_colloc=12, rhs=poisson_rhs,
                            global_domain=None, bc=(0.0, 0.0), device="cuda:0"):
    """Per-element ``M`` / ``gamma`` / ``n_colloc`` (the reference has one
    ``lssvr_M`` / ``lssvr_gamma`` per mesh, Dual.py:101).  Each of the three is a scalar or an
    array with one entry per element.  Elements are grouped by (M, n_colloc); every group is
    one ``lssvr_enhance_subset`` launch writing straight into the rows of a zero-padded
    ``W[ne, max M]`` (a Legendre series with trailing zeros evaluates identically), so
    ``EnhancedSolution.evaluate`` works unchanged.  ``lssvr_functions``-style access:
    ``Legendre(W[i, :M_i], [x_i, x_{i+1}])``."""
    torch = _torch()
    dev = _device(device)
    m = as_line_mesh(mesh)
    ne = m.nelements
    x = _to_dev(m.nodes, dev)
    u = _to_dev(nodal_values, dev)
    if u.numel() != x.numel():
        raise ValueError("nodal_values must have one value per mesh node")
    if global_domain is None:
        global_domain = (float(m.nodes[0]), float(m.nodes[-1]))
    Ms = np.broadcast_to(np.asarray(M, dtype=np.int64), (ne,))
    ns = np.broadcast_to(np.asarray(n_colloc, dtype=np.int64), (ne,))
    gs = np.array(np.broadcast_to(np.asarray(gamma, dtype=np.float64), (ne,)))
    if np.any(gs <= 0):
        raise ValueError("gamma must be > 0 for every element")
    W = torch.zeros((ne, int(Ms.max())), dtype=torch.float64, device=dev)
    st = torch.zeros((ne,), dtype=torch.int32, device=dev)
    gv = _to_dev(gs, dev)
    for Mg, ng in sorted({(int(a), int(b)) for a, b in zip(Ms, ns)}):
        ids_h = np.nonzero((Ms == Mg) & (ns == ng))[0].astype(np.int64)
        ids = torch.as_tensor(ids_h, device=dev)
        if isinstance(rhs, SinRHS):
            kw = dict(rhs=(rhs.amp, rhs.omega))
        else:
            # tabulate f at this group's collocation points (np.linspace per element)
            xc = ops.colloc_points(x, ng)[ids]
            f = np.asarray(rhs(xc.cpu().numpy()), dtype=np.float64)
            kw = dict(rhs_values=_to_dev(np.broadcast_to(f, tuple(xc.shape)), dev))
        ops.enhance_subset(x, u, Mg, 1.0, ng, W, elem_ids=ids, gamma_values=gv,
                           global_domain=global_domain, bc=bc, status=st, **kw)
    sol = EnhancedSolution(x, W, st)
    sol.degrees = Ms.copy()
    return sol


# --------------------------------------------------------------------------
# lssvr_primal (Dual.py:20-98)
# --------------------------------------------------------------------------
def lssvr_primal(rhs_func, domain_range, u_xmin, u_xmax, M, gamma,
                 is_left_boundary=False, is_right_boundary=False,
                 global_domain_range=(-1, 1), *, n_colloc=12, device="cuda:0"):
    """One element's LSSVR polynomial; same signature as Dual.py:20-22.

    Returns ``numpy.polynomial.legendre.Legendre(coef, domain_range)`` like the
    reference (Dual.py:95-98).  The equality-constrained QP of Dual.py:46-78 is
    solved in closed form on the GPU instead of by SLSQP; a breakdown of the
    factorisation prints the reference's warning (Dual.py:90-92) and returns the
    linear interpolant of the boundary values (Dual.py:164-169)."""
    dev = _device(device)
    xmin, xmax = float(domain_range[0]), float(domain_range[1])
    x = _to_dev([xmin, xmax], dev)
    u = _to_dev([u_xmin, u_xmax], dev)
    # the kernel derives the flags from the element's global index (Dual.py:150-151)
    off = 0 if is_left_boundary else 1
    ne_global = off + 1 if is_right_boundary else off + 2
    kw = _rhs_mode(rhs_func, x, n_colloc, M)
    W, st = ops.enhance(x, u, int(M), float(gamma), int(n_colloc), elem_offset=off,
                        ne_global=ne_global,
                        global_domain=(float(global_domain_range[0]), float(global_domain_range[1])),
                        bc=(main_boundary_condition_left(global_domain_range[0]),
                            main_boundary_condition_right(global_domain_range[1])), **kw)
    if int(st[0].item()) != 0:
        print("Warning: Optimization may not have converged: factorisation breakdown "
              "(linear interpolant returned)")
    return Legendre(W[0].cpu().numpy(), [domain_range[0], domain_range[1]])


# --------------------------------------------------------------------------
# FEMLSSVRPrimalSolver (Dual.py:100-203)
# --------------------------------------------------------------------------
class FEMLSSVRPrimalSolver:
    def __init__(self, num_fem_nodes=5, lssvr_M=12, lssvr_gamma=1e6, global_domain=(-1, 1), *,
                 n_colloc=12, rhs=poisson_rhs, nquad=2, mesh=None, device="cuda:0",
                 solver=ops.SOLVER_PRIMAL, fem_solver="bands"):
        # Dual.py:101-108
        self.num_fem_nodes = num_fem_nodes
        self.lssvr_M = lssvr_M
        self.lssvr_gamma = lssvr_gamma
        self.global_domain = global_domain
        self.fem_nodes = None
        self.fem_values = None
        self.lssvr_functions = []
        # additions
        self.n_colloc = n_colloc
        self.rhs = rhs
        self.nquad = nquad
        self.mesh = None if mesh is None else as_line_mesh(mesh)
        self.device = device
        self.solver_id = solver
        if fem_solver not in ("bands", "flux"):
            raise ValueError("fem_solver must be 'bands' or 'flux'")
        self.fem_solver = fem_solver
        self.enhanced = None            # EnhancedSolution after solve_lssvr_subproblems
        self._x_dev = None
        self._u_dev = None

    # ---- Dual.py:110-137 --------------------------------------------------------------
    def solve_fem(self):
        """P1 finite-element solve: mesh, element-local assembly, Dirichlet on all
        boundary dofs, tridiagonal solve, nodal values.  Returns ``(u_fem, basis)``
        like the reference (``basis`` is scikit-fem shaped, see ``mesh.py``)."""
        dev = _device(self.device)
        if self.mesh is None:
            m = LineMesh.from_nodes(
                np.linspace(self.global_domain[0], self.global_domain[1], self.num_fem_nodes))
        else:
            m = self.mesh
        basis = P1Basis(m)
        x = _to_dev(m.nodes, dev)
        if isinstance(self.rhs, SinRHS):
            bands = ops.p1_assemble(x, self.nquad, rhs=(self.rhs.amp, self.rhs.omega), want_local=True)
        else:
            xq = ops.quad_points(x, self.nquad)
            fq = _to_dev(self.rhs(xq.cpu().numpy()), dev)
            bands = ops.p1_assemble(x, self.nquad, rhs_quad=fq, want_local=True)
        u0 = main_boundary_condition_left(self.global_domain[0])
        u1 = main_boundary_condition_right(self.global_domain[1])
        if self.fem_solver == "flux":
            # exact-structure solution of A = D^T K D by the element-flux prefix scan
            u = ops.p1_flux_solve(bands["kloc"], bands["load"], u0, u1)
        else:
            # the ASSEMBLED float64 bands (rounded diagonal included) through `enforce` + `solve`
            # semantics (Dual.py:129-130).  Which matrix scikit-fem itself assembles is an
            # assumption of SURVEY.md Appendix C -- parity unpinned, the package is not importable
            # here (the reference negates both forms, Dual.py:117-124: -K u = -b; this code
            # assembles +K u = +b, the same u)
            u = ops.tridiag_dirichlet_solve(bands["diag"], bands["off"], bands["load"], u0, u1)
        self.bands = bands
        self._x_dev, self._u_dev = x, u
        u_fem = u.cpu().numpy()
        self.fem_nodes = m.p[0]
        self.fem_values = u_fem.copy()
        return u_fem, basis

    # ---- Dual.py:139-169 --------------------------------------------------------------
    def solve_lssvr_subproblems(self):
        """Solve LSSVR with the primal method in each element (one kernel launch)."""
        if self.fem_nodes is None or self.fem_values is None:
            raise RuntimeError("fem_nodes / fem_values are not set: call solve_fem() first "
                               "or assign them (Dual.py:139 reads only these attributes)")
        dev = _device(self.device)
        nodes = np.asarray(self.fem_nodes, dtype=np.float64)
        x = self._x_dev
        if x is None or x.numel() != nodes.size or not np.array_equal(x.cpu().numpy(), nodes):
            x = _to_dev(nodes, dev)
        u = _to_dev(self.fem_values, dev)       # the attribute is authoritative (may be user-set)
        kw = _rhs_mode(self.rhs, x, self.n_colloc, self.lssvr_M)
        W, st = _enhance(x, u, int(self.lssvr_M), float(self.lssvr_gamma), int(self.n_colloc),
                         global_domain=(float(self.global_domain[0]), float(self.global_domain[1])),
                         bc=(main_boundary_condition_left(self.global_domain[0]),
                             main_boundary_condition_right(self.global_domain[1])),
                         solver=self.solver_id, **kw)
        self.enhanced = EnhancedSolution(x, W, st)
        nbad = self.enhanced.n_fallback
        if nbad:
            bad = np.nonzero(st.cpu().numpy())[0]
            for i in bad[:10]:
                print(f"Error in element {i+1}: factorisation breakdown, linear interpolant used")
            if nbad > 10:
                print(f"... and {nbad - 10} more elements")
        self.lssvr_functions = _ElementFunctions(nodes, W)

    # ---- Dual.py:171-174 --------------------------------------------------------------
    def solve(self):
        """Complete solution: FEM + LSSVR."""
        self.solve_fem()
        self.solve_lssvr_subproblems()

    # ---- Dual.py:176-203 --------------------------------------------------------------
    def evaluate_solution(self, x_points):
        """Evaluate the hybrid solution at given points (float64 result; the
        reference's ``zeros_like`` dtype trap for integer input is not reproduced)."""
        if self.enhanced is None:
            raise RuntimeError("call solve() or solve_lssvr_subproblems() first")
        x_points = np.asarray(x_points)
        out = self.enhanced.evaluate(x_points.reshape(-1))
        return out.reshape(x_points.shape)

    def element_indices(self, x_points):
        """Element each query point is evaluated in (Dual.py:182-201 rule)."""
        _, elem = self.enhanced.evaluate(np.asarray(x_points).reshape(-1), return_elements=True)
        return elem
