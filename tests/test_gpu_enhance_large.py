"""GPU parity tests of the wave-per-element / f64-MFMA mapping (M up to 33)."""
import numpy as np
import pytest

from oracle import lssvr_oracle as orc
from oracle import closed_form_mp as cf

pytestmark = pytest.mark.gpu

TOL_TRUTH = 1e-13
TOL_REF = 1e-10


def _t(a, dev):
    import torch
    return torch.as_tensor(np.ascontiguousarray(a), device=dev)


def _enhance(dev, nodes, values, M, gamma, n, **kw):
    import torch
    from hybrid_fem_lssvr_amd import ops
    W, st = ops.enhance(_t(nodes, dev), _t(values, dev), M, gamma, n, **kw)
    torch.cuda.synchronize()
    return W.cpu().numpy(), st.cpu().numpy()


def test_golden_degree32(dev, golden):
    """BASELINE config 4's element problem (M=33, 64 points) vs the reference's SLSQP
    output and the 60-digit minimiser (fixture G5)."""
    g = golden("G5_ne24_M33_n64")
    ne, M, n, gamma = int(g["ne"]), int(g["M"]), int(g["n"]), float(g["gamma"])
    nodes = np.linspace(-1.0, 1.0, ne + 1)
    values = orc.fem_p1_solve(nodes)
    values[g["elements"]] = g["values_sel"][:, 0]
    values[g["elements"] + 1] = g["values_sel"][:, 1]
    W, st = _enhance(dev, nodes, values, M, gamma, n)
    assert np.all(st == 0)
    Wsel = W[g["elements"]]
    assert orc.rel_l2_coef(Wsel, g["coef_truth"]).max() <= TOL_TRUTH
    assert orc.rel_l2_coef(Wsel, g["coef_ref"]).max() <= TOL_REF


@pytest.mark.parametrize("M", [15, 16, 17, 18, 19, 23, 24, 31, 32, 33])
def test_large_degrees_vs_oracle(dev, M):
    rng = np.random.default_rng(500 + M)
    ne = 203
    nodes = np.cumsum(np.concatenate([[-0.7], rng.uniform(0.01, 0.08, ne)]))
    values = np.sin(np.pi * nodes) + 0.01 * rng.standard_normal(ne + 1)
    n = {15: 16, 16: 31, 17: 32, 18: 33, 19: 40, 23: 48, 24: 64, 31: 65, 32: 96, 33: 64}[M]
    gd = (nodes[0], nodes[-1])
    from hybrid_fem_lssvr_amd import ops
    # M <= 22 would take the lane kernel by default: force the wave / MFMA mapping here
    W, st = _enhance(dev, nodes, values, M, 1e4, n, global_domain=gd, solver=ops.SOLVER_PRIMAL_WAVE)
    assert np.all(st == 0)
    Wo = orc.enhance_all_vec(nodes, values, M, 1e4, n, global_domain=gd)
    err = orc.rel_l2_coef(W, Wo)
    assert err.max() <= 1e-11, (M, err.max())
    if cf.HAVE_MP:
        sel = [0, ne // 2, ne - 1]
        tr = cf.truth_all(nodes, values, M, 1e4, n, orc.poisson_rhs, gd, sel)
        assert orc.rel_l2_coef(W[sel], tr).max() <= TOL_TRUTH


@pytest.mark.parametrize("M", [2, 3, 5, 9, 14, 18, 22])
def test_wave_mapping_matches_lane_mapping(dev, M):
    """The two mappings of the same algorithm (solver 0 vs 2) agree to rounding."""
    from hybrid_fem_lssvr_amd import ops
    ne, n = 1500, max(16, 2 * M)
    nodes = np.linspace(-1, 1, ne + 1)
    values = orc.fem_p1_solve(nodes)
    W0, s0 = _enhance(dev, nodes, values, M, 1e4, n)
    W2, s2 = _enhance(dev, nodes, values, M, 1e4, n, solver=ops.SOLVER_PRIMAL_WAVE)
    assert np.all(s0 == 0) and np.all(s2 == 0)
    assert orc.rel_l2_coef(W2, W0).max() <= 1e-12


def test_boundary_rows_general_recurrence_branch(dev):
    """|x|/h ~ 1e10 makes t(xmin), t(xmax) miss -1, +1 by ~1e-6, which sends the
    boundary rows of the wave kernel through the general recurrence instead of the
    near-one series; both must reproduce the oracle, which mirrors the same float64
    abscissae (Dual.py:66-75 evaluates u(xmin) through the same mapdomain)."""
    ne, M, n = 64, 20, 24
    nodes = 1.0e6 + 1.0e-4 * np.arange(ne + 1)
    values = np.cos(np.arange(ne + 1) * 0.1)
    gd = (nodes[0], nodes[-1])
    from hybrid_fem_lssvr_amd import ops
    W, st = _enhance(dev, nodes, values, M, 1e4, n, global_domain=gd, solver=ops.SOLVER_PRIMAL_WAVE)
    Wo = orc.enhance_all_vec(nodes, values, M, 1e4, n, global_domain=gd)
    ok = st == 0
    assert ok.sum() >= ne - 2
    assert orc.rel_l2_coef(W[ok], Wo[ok]).max() <= 1e-9


def test_full_size_config4_sample(dev, note):
    """BASELINE config 4 (1e5 elements, degree 32, 64 points): all elements solved,
    boundary rows exact, three 2000-element windows against the batched float64 oracle and
    sampled elements against the 60-digit minimiser -- the whole polynomial and, separately, the
    enhancement relative to its own norm (oracle.rel_l2_bubble: the bubble is 1.3e-10 of the
    polynomial's norm at this h, invisible to rel_l2_coef), plus every element's leading bubble
    coefficient against its asymptotic value."""
    ne, M, n = 100000, 33, 64
    nodes = np.linspace(-1, 1, ne + 1)
    values = orc.fem_p1_solve(nodes)
    W, st = _enhance(dev, nodes, values, M, 1e4, n)
    assert np.all(st == 0)
    sgn = (-1.0) ** np.arange(M)
    assert np.max(np.abs(W @ sgn - np.concatenate([[0.0], values[1:-1]]))) < 1e-12
    assert np.max(np.abs(W.sum(1) - np.concatenate([values[1:-1], [0.0]]))) < 1e-12
    sel = np.array([0, 1, 31337, 50000, 99999])
    worst = 0.0
    for s0 in (0, 49000, ne - 2000):
        Wo = orc.enhance_all_vec(nodes[s0:s0 + 2001], values[s0:s0 + 2001], M, 1e4, n, global_domain=(-1.0, 1.0))
        if s0 + 2000 < ne:
            Wo, Wg = Wo[:1999], W[s0:s0 + 1999]      # (the window's last element is not the mesh's last)
        else:
            Wg = W[s0:]
        if s0 > 0:
            Wo, Wg = Wo[1:], Wg[1:]                  # (nor its first the mesh's first)
        assert orc.rel_l2_coef(Wg, Wo).max() <= 1e-11
        worst = max(worst, orc.rel_l2_bubble(Wg, Wo).max())
    note("config 4 bubble vs batched oracle, 6e3 elements", worst, 4e-14)
    assert worst <= 4e-14, worst                # measured 3.9e-15
    xm = 0.5 * (nodes[:-1] + nodes[1:])
    lead = -(2.0 / 3.0) * (np.pi ** 2 / 8.0) * (2.0 / ne) ** 2 * np.sin(np.pi * xm)
    big = np.abs(np.sin(np.pi * xm)) > 1e-3
    dev_lead = np.max(np.abs(W[big, 2] / lead[big] - 1.0))
    note("config 4 max |w_2 / asymptote - 1| over all elements", dev_lead)
    assert dev_lead < 1e-5
    if cf.HAVE_MP:
        tr = cf.truth_all(nodes, values, M, 1e4, n, orc.poisson_rhs, (-1.0, 1.0), sel)
        assert orc.rel_l2_coef(W[sel], tr).max() <= TOL_TRUTH
        bt = orc.rel_l2_bubble(W[sel], tr).max()
        note("config 4 bubble vs 60-digit minimiser", bt, 2e-14)
        assert bt <= 2e-14, bt                  # measured 2.4e-15


def test_fallback_status_large(dev):
    import torch
    from hybrid_fem_lssvr_amd import ops
    ne, M, n = 40, 24, 32
    nodes = np.linspace(0, 1, ne + 1)
    nodes[11] = nodes[10]
    values = np.sin(nodes)
    values[30] = np.nan
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    W, st = ops.enhance(_t(nodes, dev), _t(values, dev), M, 1e4, n, fail_count=cnt,
                        global_domain=(0.0, 1.0))
    torch.cuda.synchronize()
    st = st.cpu().numpy()
    W = W.cpu().numpy()
    assert set(np.nonzero(st)[0]) == {10, 29, 30}
    assert int(cnt.item()) == 3
    assert np.allclose(W[10, :2], [0.5 * (values[10] + values[11]), 0.5 * (values[11] - values[10])])
    assert np.all(W[10, 2:] == 0)


@pytest.mark.parametrize("M,n,ne", [(33, 64, 24), (33, 64, 1003), (25, 48, 77), (9, 16, 130), (16, 31, 35), (4, 7, 19),
                                    (3, 5, 9), (2, 4, 5), (22, 30, 41)])
def test_moment_wave_mapping_vs_default(dev, M, n, ne):
    """LSSVR_SOLVER_PRIMAL_MOMENT (the kernel sequence of csrc/enhance_large_cheb.hip / enhance_large_parity.hip
    forced for any M: Chebyshev-moment Gram, four systems per wave in the DPP LDL^T) against the oracle and the
    default kernels: every tail of the 4-element rounds, degrees from 1 to 32."""
    from hybrid_fem_lssvr_amd import ops
    rng = np.random.default_rng(900 + M + ne)
    nodes = np.cumsum(np.concatenate([[-0.7], rng.uniform(0.01, 0.08, ne)]))
    values = np.sin(np.pi * nodes) + 0.01 * rng.standard_normal(ne + 1)
    gd = (nodes[0], nodes[-1])
    W, st = _enhance(dev, nodes, values, M, 1e4, n, global_domain=gd, solver=ops.SOLVER_PRIMAL_MOMENT)
    assert np.all(st == 0)
    Wo = orc.enhance_all_vec(nodes, values, M, 1e4, n, global_domain=gd)
    assert orc.rel_l2_coef(W, Wo).max() <= 1e-11
    Wd, _ = _enhance(dev, nodes, values, M, 1e4, n, global_domain=gd)
    assert orc.rel_l2_coef(W, Wd).max() <= 1e-11
    if cf.HAVE_MP:
        sel = [0, ne // 2, ne - 1]
        tr = cf.truth_all(nodes, values, M, 1e4, n, orc.poisson_rhs, gd, sel)
        assert orc.rel_l2_coef(W[sel], tr).max() <= TOL_TRUTH
    # tabulated rhs == in-kernel sin
    import torch
    x = _t(nodes, dev)
    f = _t(orc.poisson_rhs(ops.colloc_points(x, n).cpu().numpy()), dev)
    W2, _ = ops.enhance(x, _t(values, dev), M, 1e4, n, global_domain=gd, rhs_values=f, solver=ops.SOLVER_PRIMAL_MOMENT)
    torch.cuda.synchronize()
    assert orc.rel_l2_coef(W2.cpu().numpy(), W).max() <= 1e-12


def test_workspace_free_entry_matches_two_kernel_path(dev, golden):
    """Above M = 22 `ops.enhance` takes lssvr_enhance_ws with a workspace (moments kernel + solve
    kernel); `work=False` -- what the workspace-free C entry lssvr_enhance launches -- runs the
    single f64-MFMA kernel.  Same minimiser: both against the golden truth and each other; the
    fused StepPlan (assembly + enhancement) of both degrees classes as well."""
    import torch
    from hybrid_fem_lssvr_amd import ops
    g = golden("G5_ne24_M33_n64")
    ne, M, n, gamma = int(g["ne"]), int(g["M"]), int(g["n"]), float(g["gamma"])
    nodes = np.linspace(-1.0, 1.0, ne + 1)
    values = orc.fem_p1_solve(nodes)
    values[g["elements"]] = g["values_sel"][:, 0]
    values[g["elements"] + 1] = g["values_sel"][:, 1]
    x, u = _t(nodes, dev), _t(values, dev)
    W2, s2 = ops.enhance(x, u, M, gamma, n, global_domain=(-1.0, 1.0))
    W1, s1 = ops.enhance(x, u, M, gamma, n, global_domain=(-1.0, 1.0), work=False)
    plan = ops.StepPlan(x, u, M, gamma, n, global_domain=(-1.0, 1.0))
    W3, s3 = plan.launch()
    torch.cuda.synchronize()
    assert int(s1.sum()) == 0 and int(s2.sum()) == 0 and int(s3.sum()) == 0
    for W in (W1, W2, W3):
        assert orc.rel_l2_coef(W.cpu().numpy()[g["elements"]], g["coef_truth"]).max() <= TOL_TRUTH
    assert orc.rel_l2_coef(W1.cpu().numpy(), W2.cpu().numpy()).max() <= 1e-12
    assert torch.equal(W2, W3)
    bands = orc.p1_scatter(*orc.p1_assemble_local(nodes))
    assert np.allclose(plan.bands["diag"].cpu().numpy(), bands[0], rtol=1e-15)
    # a workspace that is given but too small is an error (ABI 4; ABI 3 silently ran the workspace-free
    # kernel, whose accuracy differs in the near-square regime): nothing is launched, W stays as it is
    from hybrid_fem_lssvr_amd import _capi
    small = torch.empty(16, dtype=torch.float64, device=dev)
    W4 = torch.full((len(nodes) - 1, M), 7.0, dtype=torch.float64, device=dev)
    with pytest.raises(_capi.LssvrHipError, match="lssvr_enhance_work_bytes"):
        ops.enhance(x, u, M, gamma, n, global_domain=(-1.0, 1.0), work=small, out=W4)
    torch.cuda.synchronize()
    assert bool((W4 == 7.0).all())


# (M, n, h, bar of the refined two-kernel path, floor of the unrefined single kernel): numbers of
# scripts/proto/qr_nearsquare.py and of the first GPU run; the bar is ~5x what was measured.
NEAR_SQUARE = [
    (33, 31, 1.0 / 12, 3e-13, 1e-8),     # measured on the MI355X: refined 4.8e-14 (3 steps), single kernel 1.1e-6
    (33, 32, 1.0 / 12, 1e-14, 1e-11),    # 7.5e-16 / 2.1e-8 (2 steps)
    (33, 33, 1.0 / 12, 1e-14, 1e-12),
    (33, 35, 1.0 / 12, 1e-14, 1e-13),
    (33, 38, 1.0 / 12, 1e-14, 0.0),      # 1 step
    (33, 44, 1.0 / 12, 1e-14, 0.0),
    (28, 26, 1.0 / 12, 1e-14, 1e-13),
    (24, 22, 1.0 / 12, 1e-14, 0.0),
    (33, 33, 0.5, 2e-13, 1e-10),         # 3.6e-14 / 1.2e-7: coarse elements, the problem's own conditioning shows
    (33, 36, 0.5, 1e-14, 1e-12),
]


@pytest.mark.parametrize("M,n,h,bar,plain_floor", NEAR_SQUARE)
def test_near_square_refinement(dev, M, n, h, bar, plain_floor):
    """About as many equispaced collocation points as bubble coefficients (n = M-2 ... M+12): the
    normal equations lose up to ten digits (3e-6 at M = 33, n = 31); the
    two-kernel path of lssvr_enhance_ws follows the solve with 1-3 steps of the corrected
    semi-normal equations (point residual through the rows: residual_kernel + solve4_kernel<2>) and
    meets the 60-digit minimiser again.  `plain_floor` > 0: the unrefined single kernel (work=False)
    is demonstrably worse there -- the test would notice the refinement silently not running."""
    if not cf.HAVE_MP:
        pytest.skip("mpmath missing")
    from hybrid_fem_lssvr_amd import _capi
    ne = int(round(2.0 / h))
    nodes = np.linspace(-1.0, 1.0, ne + 1)
    values = np.sin(np.pi * nodes)
    lib = _capi.load()
    steps_expected = 3 if n - (M - 2) <= 1 else 2 if n - (M - 2) <= 4 else 1
    need = lib.lssvr_enhance_work_bytes(ne, M, n, 0)
    assert need == ne * (96 + 32) * 8, (need, steps_expected)
    W, st = _enhance(dev, nodes, values, M, 1e4, n, global_domain=(-1.0, 1.0))
    assert np.all(st == 0)
    sel = sorted({0, ne // 3, ne // 2, ne - 1})
    tr = cf.truth_all(nodes, values, M, 1e4, n, orc.poisson_rhs, (-1.0, 1.0), sel)
    err = orc.rel_l2_coef(W[sel], tr).max()
    assert err <= bar, err
    W0, st0 = _enhance(dev, nodes, values, M, 1e4, n, global_domain=(-1.0, 1.0), work=False)
    assert np.all(st0 == 0)
    err0 = orc.rel_l2_coef(W0[sel], tr).max()
    print("near-square M=%d n=%d h=%g: refined %.1e, single kernel %.1e" % (M, n, h, err, err0))
    assert err0 >= plain_floor, err0
    # tabulated right-hand side takes the same path
    import torch
    from hybrid_fem_lssvr_amd import ops
    x = _t(nodes, dev)
    f = _t(orc.poisson_rhs(ops.colloc_points(x, n).cpu().numpy()), dev)
    W2, _ = ops.enhance(x, _t(values, dev), M, 1e4, n, global_domain=(-1.0, 1.0), rhs_values=f)
    torch.cuda.synchronize()
    assert orc.rel_l2_coef(W2.cpu().numpy()[sel], tr).max() <= 5 * bar


def test_refinement_leaves_well_posed_sizes_alone(dev):
    """n - (M-2) > 14 (every BASELINE configuration): no refinement kernels, 96-double workspace."""
    from hybrid_fem_lssvr_amd import _capi
    lib = _capi.load()
    assert lib.lssvr_enhance_work_bytes(1000, 33, 64, 0) == 1000 * 96 * 8
    assert lib.lssvr_enhance_work_bytes(1000, 33, 46, 0) == 1000 * 96 * 8
    assert lib.lssvr_enhance_work_bytes(1000, 33, 45, 0) == 1000 * 128 * 8
    assert lib.lssvr_enhance_work_bytes(1000, 22, 20, 0) == 0


# (M, n, x0, h, ne): the parity-split solve kernel (csrc/enhance_large_parity.hip) over its regimes:
# both parities of the number of bubble coefficients (padding column of the odd block), every tail of
# the four-element rounds, point sets from exactly symmetric to strongly asymmetric in float64
# (contraction rate of the coupling iteration 1e-13 ... 4e-4: one to several corrections), and
# elements beyond the first-order range of the boundary rows (cold exact path inside the kernel).
PARITY_CASES = [
    (33, 64, -1.0, 1.0 / 12, 24),
    (33, 64, -4166.0, 1.0 / 12, 37),        # rho = 3e-6: block solve alone is 5e-12 off
    (33, 64, 416666.0, 1.0 / 12, 5),
    (33, 64, 0.9999, 2.0e-7, 9),            # |x|/h = 5e6: rho = 4e-4, several corrections
    (33, 62, -0.7, 0.03, 3),                # n = 2 (M-2) exactly: the gate of the kernel
    (32, 60, -4166.0, 1.0 / 12, 2),         # 30 bubble coefficients: 15 + 15
    (31, 64, 12.5, 0.125, 1),               # 29: 15 + 14
    (24, 44, -4166.0, 1.0 / 12, 6),         # 22: 11 + 11
    (23, 42, 3.0, 0.01, 7),                 # 21: 11 + 10
    (33, 96, 1.0e8, 1.0, 6),                # |x|/h = 1e8: boundary rows by the exact recurrence
    (27, 80, -3.0e7, 0.5, 4),
]


@pytest.mark.parametrize("M,n,x0,h,ne", PARITY_CASES)
def test_parity_split_regimes(dev, M, n, x0, h, ne):
    import torch
    from hybrid_fem_lssvr_amd import ops
    assert n >= 2 * (M - 2) and M > 22                      # (the regime the kernel is launched for)
    rng = np.random.default_rng(int(M * 1000 + n + ne))
    nodes = x0 + h * np.arange(ne + 1)
    values = np.sin(np.pi * nodes) + 0.01 * rng.standard_normal(ne + 1)
    gd = (nodes[0], nodes[-1])
    W, st = _enhance(dev, nodes, values, M, 1e4, n, global_domain=gd)
    assert np.all(st == 0)
    W1, st1 = _enhance(dev, nodes, values, M, 1e4, n, global_domain=gd, work=False)   # single f64-MFMA kernel
    assert np.all(st1 == 0)
    assert orc.rel_l2_coef(W, W1).max() <= 1e-12
    if cf.HAVE_MP:
        sel = sorted({0, ne // 2, ne - 1})
        tr = cf.truth_all(nodes, values, M, 1e4, n, orc.poisson_rhs, gd, sel)
        err = orc.rel_l2_coef(W[sel], tr).max()
        assert err <= TOL_TRUTH, err
    # tabulated right-hand side: same kernels, same answer
    x = _t(nodes, dev)
    f = _t(orc.poisson_rhs(ops.colloc_points(x, n).cpu().numpy()), dev)
    W2, _ = ops.enhance(x, _t(values, dev), M, 1e4, n, global_domain=gd, rhs_values=f)
    torch.cuda.synchronize()
    assert orc.rel_l2_coef(W2.cpu().numpy(), W).max() <= 1e-12


def test_parity_split_per_element_gamma_and_failures(dev):
    """gamma_values / fail_count / status through the persistent parity kernel: a degenerate element
    (zero length) and a NaN nodal value fall back, everything else matches the oracle."""
    import torch
    from hybrid_fem_lssvr_amd import ops
    ne, M, n = 43, 33, 64
    nodes = np.linspace(0.0, 1.0, ne + 1)
    nodes[11] = nodes[10]
    values = np.sin(nodes)
    values[30] = np.nan
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    W, st = ops.enhance(_t(nodes, dev), _t(values, dev), M, 1e4, n, fail_count=cnt, global_domain=(0.0, 1.0))
    torch.cuda.synchronize()
    st = st.cpu().numpy()
    assert set(np.nonzero(st)[0]) == {10, 29, 30} and int(cnt.item()) == 3
    good = [0, 5, 20, 42]
    Wo = orc.enhance_all_vec(nodes, np.nan_to_num(values), M, 1e4, n, global_domain=(0.0, 1.0))
    assert orc.rel_l2_coef(W.cpu().numpy()[good], Wo[good]).max() <= 1e-11


def test_moment_solver_needs_workspace(dev):
    """LSSVR_SOLVER_PRIMAL_MOMENT is the kernel sequence with a workspace in between: without one the
    call is refused (no silent switch to another kernel), with the default per-device buffer it runs."""
    from hybrid_fem_lssvr_amd import ops, _capi
    nodes = np.linspace(-1.0, 1.0, 25)
    values = np.sin(np.pi * nodes)
    with pytest.raises(_capi.LssvrHipError, match="workspace"):
        _enhance(dev, nodes, values, 9, 1e4, 16, global_domain=(-1.0, 1.0), solver=ops.SOLVER_PRIMAL_MOMENT, work=False)
    lib = _capi.load()
    assert lib.lssvr_enhance_work_bytes(24, 9, 16, ops.SOLVER_PRIMAL_MOMENT) == 24 * 96 * 8
    assert lib.lssvr_enhance_work_bytes(24, 9, 16, ops.SOLVER_PRIMAL) == 0
    W, st = _enhance(dev, nodes, values, 9, 1e4, 16, global_domain=(-1.0, 1.0), solver=ops.SOLVER_PRIMAL_MOMENT)
    Wd, _ = _enhance(dev, nodes, values, 9, 1e4, 16, global_domain=(-1.0, 1.0))
    assert np.all(st == 0) and orc.rel_l2_coef(W, Wd).max() <= 1e-12


def test_default_workspace_is_handed_between_streams_in_event_order(dev):
    """the default workspace above M = 22 is ONE buffer per device (not one per stream handle for the
    life of the process).  Launches on different streams that go through it are ordered by the event recorded
    after the previous user's launch: interleaved launches on two streams with DIFFERENT inputs must each return
    their own result (a race on the moments between the kernels of a launch would mix them), the cache holds one
    entry, and release_workspaces() empties it."""
    import torch
    from hybrid_fem_lssvr_amd import ops
    M, n, ne = 33, 64, 3000
    nodes = np.linspace(-1.0, 1.0, ne + 1)
    xa = _t(nodes, dev)
    ua = _t(np.sin(np.pi * nodes), dev)
    ub = _t(np.cos(1.7 * nodes), dev)
    ref_a, _ = ops.enhance(xa, ua, M, 1e4, n, global_domain=(-1.0, 1.0))
    ref_b, _ = ops.enhance(xa, ub, M, 1e4, n, global_domain=(-1.0, 1.0))
    torch.cuda.synchronize()
    ref_a, ref_b = ref_a.clone(), ref_b.clone()
    sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    outs = []
    for rep in range(6):                       # alternate streams; nothing synchronises the host in between
        for s, u in ((sa, ua), (sb, ub)):
            with torch.cuda.stream(s):
                W, st = ops.enhance(xa, u, M, 1e4, n, global_domain=(-1.0, 1.0))
                outs.append((W, u is ua))
    torch.cuda.synchronize()
    for W, is_a in outs:
        assert torch.equal(W, ref_a if is_a else ref_b)
    assert len(ops._WORK) == 1
    ops.release_workspaces()
    assert len(ops._WORK) == 0
