"""GPU parity tests of the hot path: HIP kernels (through the C ABI) vs the oracle,
vs the committed golden vectors (reference SLSQP output + 60-digit closed form),
and size-independent properties at BASELINE's full sizes.

Tolerances (BASELINE.md section 4 / SURVEY.md section 8(c)):
  * vs the extended-precision minimiser of the reference's QP ("truth"): 1e-13 relative L2
    per element polynomial;
  * vs the reference's own SLSQP output: 1e-10 (3e-10 on config 1 and on the class
    defaults, where the reference itself is 1e-10 / 2.6e-10 from the exact minimiser);
  * element / node indices: exact.
"""
import numpy as np
import pytest

from oracle import lssvr_oracle as orc
from oracle import closed_form_mp as cf

pytestmark = pytest.mark.gpu

TOL_TRUTH = 1e-13
TOL_REF = 1e-10
# The enhancement itself (Legendre coefficients p >= 2) relative to ITS OWN norm
# (oracle.rel_l2_bubble): on fine meshes the linear part hides it from rel_l2_coef by 1.2 h^2
# (a kernel returning no enhancement at all reads 1.5e-14 there on 1e7 elements).  A float64
# solve gets the bubble to ~3e-16 of itself at every h.  Measured on the MI355X (round 3,
# gpurun_out/measured_bars.log): config 2 / 3 against the batched float64 oracle 1.6e-15 / 1.9e-15
# over every element, against the 60-digit minimiser 3.1e-16 / 4.3e-16; bars at ~10x that.
TOL_BUBBLE_TRUTH = 5e-15
TOL_BUBBLE_ORACLE = 2e-14


def _t(a, dev):
    import torch
    return torch.as_tensor(np.ascontiguousarray(a), device=dev)


def _enhance(dev, nodes, values, M, gamma, n, **kw):
    import torch
    from hybrid_fem_lssvr_amd import ops
    W, st = ops.enhance(_t(nodes, dev), _t(values, dev), M, gamma, n, **kw)
    torch.cuda.synchronize()
    return W.cpu().numpy(), st.cpu().numpy()


SMALL_GOLDEN = [
    ("G1_c1_ne8_M5_n5", 3e-10),
    ("G2_default_ne24_M8_n12", TOL_REF),
    ("G3_ne24_M9_n16", TOL_REF),
    ("G4_ne4096_M9_n16", TOL_REF),
    ("G6a_wide_ne100008_M9_n16", TOL_REF),
    ("G6b_wide_ne10000008_M9_n16", 2e-10),
    ("G8_classdefaults_ne4_M12_n12", 3e-10),
]


@pytest.mark.parametrize("name,tol_ref", SMALL_GOLDEN)
def test_golden_small_degree(dev, golden, name, tol_ref):
    g = golden(name)
    lo, hi, ne = float(g["lo"]), float(g["hi"]), int(g["ne"])
    M, n, gamma = int(g["M"]), int(g["n"]), float(g["gamma"])
    nodes = np.linspace(lo, hi, ne + 1)                       # Dual.py:112
    assert np.array_equal(nodes[g["elements"]], g["nodes_sel"][:, 0])
    values = np.sin(np.pi * nodes)
    values[g["elements"]] = g["values_sel"][:, 0]
    values[g["elements"] + 1] = g["values_sel"][:, 1]
    W, st = _enhance(dev, nodes, values, M, gamma, n, global_domain=(lo, hi))
    assert np.all(st == 0)
    Wsel = W[g["elements"]]
    err_truth = orc.rel_l2_coef(Wsel, g["coef_truth"])
    err_ref = orc.rel_l2_coef(Wsel, g["coef_ref"])
    assert err_truth.max() <= TOL_TRUTH, err_truth
    assert err_ref.max() <= tol_ref, err_ref


@pytest.mark.parametrize("M", list(range(2, 23)))
def test_every_small_degree_vs_oracle(dev, M):
    """Each template instantiation of the lane-per-element kernel, non-uniform mesh."""
    rng = np.random.default_rng(100 + M)
    ne = 777
    nodes = np.cumsum(np.concatenate([[-1.3], rng.uniform(0.002, 0.05, ne)]))
    values = np.sin(np.pi * nodes) + 0.01 * rng.standard_normal(ne + 1)
    n = max(M + 3, 6) if M <= 14 else 2 * M
    gd = (nodes[0], nodes[-1])
    W, st = _enhance(dev, nodes, values, M, 1e4, n, global_domain=gd)
    assert np.all(st == 0)
    Wo = orc.enhance_all_vec(nodes, values, M, 1e4, n, global_domain=gd)
    err = orc.rel_l2_coef(W, Wo)
    assert err.max() <= 1e-12, (M, err.max())
    # Dirichlet value replaces the nodal value on the two global-boundary elements only
    assert abs(orc.clenshaw(-1.0, W[0]) - 0.0) < 1e-12
    assert abs(orc.clenshaw(1.0, W[-1]) - 0.0) < 1e-12
    if cf.HAVE_MP:
        sel = [0, 1, ne // 2, ne - 1]
        tr = cf.truth_all(nodes, values, M, 1e4, n, orc.poisson_rhs, gd, sel)
        assert orc.rel_l2_coef(W[sel], tr).max() <= TOL_TRUTH


@pytest.mark.parametrize("n", [2, 3, 5, 12, 16, 33, 64, 200])
def test_collocation_counts(dev, n):
    ne, M = 130, 6
    nodes = np.linspace(-1, 1, ne + 1)
    values = orc.fem_p1_solve(nodes)
    W, st = _enhance(dev, nodes, values, M, 1e4, n)
    assert np.all(st == 0)
    # boundary rows hold whatever n is
    sgn = (-1.0) ** np.arange(M)
    assert np.max(np.abs(W @ sgn - np.concatenate([[0.0], values[1:-1]]))) < 1e-12
    assert np.max(np.abs(W.sum(1) - np.concatenate([values[1:-1], [0.0]]))) < 1e-12
    if n < M - 2:
        # fewer collocation points than bubble coefficients: the primal Gram is rank deficient
        # (float64 KKT / normal equations are O(1) wrong there, oracle included); the library
        # routes these calls to the dual Gram solver, checked against the 60-digit minimiser.
        # No BASELINE configuration is in this regime (5>=3, 12>=6, 16>=7, 64>=31, 12>=10).
        assert np.all(np.isfinite(W))
        if cf.HAVE_MP:
            sel = [0, 1, 64, 129]
            tr = cf.truth_all(nodes, values, M, 1e4, n, orc.poisson_rhs, (-1.0, 1.0), sel)
            assert orc.rel_l2_coef(W[sel], tr).max() <= 1e-13
        return
    Wo = orc.enhance_all_vec(nodes, values, M, 1e4, n)
    assert orc.rel_l2_coef(W, Wo).max() <= 1e-12


def test_rhs_array_matches_in_kernel_rhs(dev):
    import torch
    from hybrid_fem_lssvr_amd import ops
    ne, M, n = 1000, 9, 16
    nodes = np.linspace(-3, 5, ne + 1)
    values = np.sin(np.pi * nodes)
    x = _t(nodes, dev)
    xc = ops.colloc_points(x, n)
    xc_h = xc.cpu().numpy()
    # np.linspace per element, bit for bit (Dual.py:40)
    for e in (0, 1, 17, ne - 1):
        assert np.array_equal(xc_h[e], np.linspace(nodes[e], nodes[e + 1], n))
    f = _t(orc.poisson_rhs(xc_h), dev)
    W1, _ = ops.enhance(x, _t(values, dev), M, 1e4, n, rhs_values=f)
    W2, _ = ops.enhance(x, _t(values, dev), M, 1e4, n)
    torch.cuda.synchronize()
    err = orc.rel_l2_coef(W1.cpu().numpy(), W2.cpu().numpy())
    assert err.max() <= 1e-13


def test_in_kernel_sin_accuracy(dev):
    """f = pi^2 sin(pi x) in-kernel vs numpy, through a problem whose answer is f:
    M=3, huge gamma: w_2 -> least-squares fit of -u'' = f  (indirect), plus a direct
    check via the array path on a wide domain where |pi x| ~ 1.3e6."""
    import torch
    from hybrid_fem_lssvr_amd import ops
    ne, M, n = 4096, 9, 16
    nodes = np.linspace(-416667.0, -416667.0 + ne / 12.0, ne + 1)
    values = np.sin(np.pi * nodes)
    x = _t(nodes, dev)
    f = _t(orc.poisson_rhs(ops.colloc_points(x, n).cpu().numpy()), dev)
    W1, _ = ops.enhance(x, _t(values, dev), M, 1e4, n, rhs_values=f, global_domain=(-416667.0, 416667.0),
                        ne_global=10000008)
    W2, _ = ops.enhance(x, _t(values, dev), M, 1e4, n, global_domain=(-416667.0, 416667.0),
                        ne_global=10000008)
    torch.cuda.synchronize()
    assert orc.rel_l2_coef(W1.cpu().numpy(), W2.cpu().numpy()).max() <= 1e-13


def test_empty_and_single_element(dev):
    import torch
    from hybrid_fem_lssvr_amd import ops
    x = _t(np.array([0.25]), dev)
    W, st = ops.enhance(x, x.clone(), 9, 1e4, 16, global_domain=(0.0, 1.0))
    assert W.shape == (0, 9) and st.shape == (0,)
    nodes = np.array([-1.0, 1.0])
    W, st = _enhance(dev, nodes, np.array([0.3, -0.2]), 9, 1e4, 16)
    # single element touches both global boundaries: both nodal values are replaced by 0
    Wo = orc.enhance_all_vec(nodes, np.array([0.3, -0.2]), 9, 1e4, 16)
    assert orc.rel_l2_coef(W, Wo).max() <= 1e-12
    assert abs(orc.clenshaw(-1.0, W[0])) < 1e-13 and abs(orc.clenshaw(1.0, W[0])) < 1e-13


def test_shard_offsets_boundary_flags(dev):
    """Elements sharded across ranks: only global element 0 / ne-1 see the Dirichlet
    values (Dual.py:150-151), and shards stitch to the single-shard answer bit for bit."""
    ne, M, n = 1001, 9, 16
    nodes = np.linspace(-1, 1, ne + 1)
    values = np.cos(nodes)          # non-zero at the boundary nodes on purpose
    Wfull, _ = _enhance(dev, nodes, values, M, 1e4, n, global_domain=(-1.0, 1.0))
    parts = []
    cuts = [0, 250, 251, 777, ne]
    for s0, s1 in zip(cuts[:-1], cuts[1:]):
        Wp, _ = _enhance(dev, nodes[s0:s1 + 1], values[s0:s1 + 1], M, 1e4, n, elem_offset=s0,
                         ne_global=ne, global_domain=(-1.0, 1.0))
        parts.append(Wp)
    assert np.array_equal(np.concatenate(parts), Wfull)
    assert abs(orc.clenshaw(-1.0, Wfull[0])) < 1e-13          # Dirichlet 0, not cos(-1)
    assert abs(orc.clenshaw(1.0, Wfull[1]) - values[2]) < 1e-12


def test_fallback_status_on_degenerate_elements(dev):
    """Zero-length / NaN elements cannot be factorised: status 1 and the linear
    interpolant of (g_l, g_r) (Dual.py:164-169), neighbours untouched."""
    import torch
    from hybrid_fem_lssvr_amd import ops
    ne, M, n = 300, 9, 16
    nodes = np.linspace(0, 3, ne + 1)
    nodes[100] = nodes[99]                 # element 99 has h = 0
    values = np.sin(nodes)
    values[200] = np.nan                   # elements 199, 200 see a NaN nodal value
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    W, st = ops.enhance(_t(nodes, dev), _t(values, dev), M, 1e4, n, fail_count=cnt,
                        global_domain=(0.0, 3.0))
    torch.cuda.synchronize()
    W, st = W.cpu().numpy(), st.cpu().numpy()
    bad = {99, 199, 200}
    assert set(np.nonzero(st)[0]) == bad
    assert int(cnt.item()) == 3
    assert np.allclose(W[99, :2], [0.5 * (values[99] + values[100]), 0.5 * (values[100] - values[99])])
    assert np.all(W[99, 2:] == 0)
    good = np.array([e for e in range(ne) if e not in bad and e not in (98, 100)])
    Wo = orc.enhance_all_vec(np.linspace(0, 3, ne + 1), np.sin(np.linspace(0, 3, ne + 1)), M, 1e4, n)
    keep = good[(good < 98) | (good > 201)]
    assert orc.rel_l2_coef(W[keep], Wo[keep]).max() <= 1e-12


def test_full_size_config2_properties(dev, note):
    """BASELINE config 2 (1e5 elements on [-1,1], degree 8, 16 points): every element
    against the batched float64 oracle -- the whole polynomial AND the enhancement on its own
    (the bubble is 1.3e-10 of the norm here: rel_l2_coef alone would pass with no bubble at
    all) --, boundary rows, and the L2 error vs sin(pi x)."""
    ne, M, n = 100000, 9, 16
    nodes = np.linspace(-1, 1, ne + 1)
    values = orc.fem_p1_solve(nodes)
    W, st = _enhance(dev, nodes, values, M, 1e4, n)
    assert np.all(st == 0)
    Wo = orc.enhance_all_vec(nodes, values, M, 1e4, n)
    assert orc.rel_l2_coef(W, Wo).max() <= 1e-12
    assert orc.rel_l2_global(W, Wo, nodes) <= 1e-13
    bub = orc.rel_l2_bubble(W, Wo).max()
    note("config 2 bubble vs batched oracle, every element", bub, TOL_BUBBLE_ORACLE)
    assert bub <= TOL_BUBBLE_ORACLE, bub
    Wz = W.copy()
    Wz[:, 2:] = 0.0                     # what the old metric could not see
    assert orc.rel_l2_coef(Wz, Wo).max() <= 1e-9 and orc.rel_l2_bubble(Wz, Wo).min() == 1.0
    # boundary rows: u(x_e) = g_l, u(x_{e+1}) = g_r
    sgn = (-1.0) ** np.arange(M)
    assert np.max(np.abs(W @ sgn - np.concatenate([[0.0], values[1:-1]]))) < 1e-13
    assert np.max(np.abs(W.sum(1) - np.concatenate([values[1:-1], [0.0]]))) < 1e-13
    # L2 error against the exact solution equals the CPU restatement's to 1e-10
    xq = np.linspace(-1, 1, 20001)
    u_gpu, _ = orc.evaluate_solution_vec(nodes, W, xq)
    u_cpu, _ = orc.evaluate_solution_vec(nodes, Wo, xq)
    ex = orc.true_solution(xq)
    e_gpu = np.linalg.norm(u_gpu - ex) / np.linalg.norm(ex)
    e_cpu = np.linalg.norm(u_cpu - ex) / np.linalg.norm(ex)
    assert abs(e_gpu - e_cpu) <= 1e-10 * max(e_cpu, 1e-300) + 1e-16
    if cf.HAVE_MP:
        sel = [0, 1, 49999, 50000, 99999]
        tr = cf.truth_all(nodes, values, M, 1e4, n, orc.poisson_rhs, (-1.0, 1.0), sel)
        assert orc.rel_l2_coef(W[sel], tr).max() <= TOL_TRUTH
        bt = orc.rel_l2_bubble(W[sel], tr).max()
        note("config 2 bubble vs 60-digit minimiser", bt, TOL_BUBBLE_TRUTH)
        assert bt <= TOL_BUBBLE_TRUTH, bt


def test_full_size_config3_single_gpu(dev, note):
    """BASELINE config 3's mesh (1e7 elements, degree 8, 16 points) on one GPU: every element
    solved, boundary rows exact everywhere, a 6e4-element sample against the batched oracle and
    a few elements against the 60-digit minimiser -- whole polynomial and, separately, the
    enhancement relative to its own norm (1.3e-14 of the polynomial's at this h: invisible to
    rel_l2_coef).  A device-side check covers EVERY element's bubble: w_2 = -(h^2/8) u''(mid) to
    O(h^2), i.e. w_2 / (pi^2 h^2 sin(pi x_mid) / 8) -> 1.  Shard stitching is covered by
    test_shard_offsets_boundary_flags and tests/test_distributed_gloo.py."""
    import torch
    from hybrid_fem_lssvr_amd import ops
    ne, M, n = 10000000, 9, 16
    nodes = np.linspace(-1, 1, ne + 1)
    values = np.sin(np.pi * nodes)
    values[0] = values[-1] = 0.0
    W, st = ops.enhance(_t(nodes, dev), _t(values, dev), M, 1e4, n, global_domain=(-1.0, 1.0))
    assert int(st.sum().item()) == 0
    sgn = torch.as_tensor((-1.0) ** np.arange(M), device=dev)
    left = (W @ sgn).cpu().numpy()
    right = W.sum(1).cpu().numpy()
    assert np.max(np.abs(left - values[:-1])) < 1e-13
    assert np.max(np.abs(right - values[1:])) < 1e-13
    worst = 0.0
    for s0 in (0, 4999000, ne - 20000):
        sl = slice(s0, s0 + 20000)
        Wo = orc.enhance_all_vec(nodes[s0:s0 + 20001], values[s0:s0 + 20001], M, 1e4, n,
                                 global_domain=(-1.0, 1.0))
        Wg = W[sl].cpu().numpy()
        assert orc.rel_l2_coef(Wg, Wo).max() <= 1e-12
        worst = max(worst, orc.rel_l2_bubble(Wg, Wo).max())
    note("config 3 bubble vs batched oracle, 6e4 elements", worst, TOL_BUBBLE_ORACLE)
    assert worst <= TOL_BUBBLE_ORACLE, worst
    # every element: the leading bubble coefficient against its asymptotic value (the element
    # mid point away from the zeros of sin, where the ratio is 0/0)
    xm = 0.5 * (nodes[:-1] + nodes[1:])
    # bubble = (f/2)(x-a)(b-x) = (f h^2 / 8)(1 - t^2) + O(h^3),  1 - t^2 = (2/3)(L_0 - L_2)
    lead = -(2.0 / 3.0) * (np.pi ** 2 / 8.0) * (2.0 / ne) ** 2 * np.sin(np.pi * xm)
    w2 = W[:, 2].cpu().numpy()
    big = np.abs(np.sin(np.pi * xm)) > 1e-3
    ratio = w2[big] / lead[big]
    note("config 3 max |w_2 / asymptote - 1| over all elements", np.max(np.abs(ratio - 1.0)))
    assert np.max(np.abs(ratio - 1.0)) < 1e-5
    if cf.HAVE_MP:
        sel = [0, 1, 5000000, ne - 1]
        tr = cf.truth_all(nodes, values, M, 1e4, n, orc.poisson_rhs, (-1.0, 1.0), sel)
        Ws = W[sel].cpu().numpy()
        assert orc.rel_l2_coef(Ws, tr).max() <= TOL_TRUTH
        bt = orc.rel_l2_bubble(Ws, tr).max()
        note("config 3 bubble vs 60-digit minimiser", bt, TOL_BUBBLE_TRUTH)
        assert bt <= TOL_BUBBLE_TRUTH, bt


def test_enhance_sharded_single_rank_chunks(dev):
    """distributed.enhance_sharded on one rank (no process group needed): the chunked,
    stream-overlapped compute + stitch path reproduces the single launch bit for bit."""
    import torch
    from hybrid_fem_lssvr_amd import ops
    from hybrid_fem_lssvr_amd.distributed import ShardPlan, enhance_sharded
    ne, M, n = 10007, 9, 16
    nodes = np.linspace(-1, 1, ne + 1)
    x, u = _t(nodes, dev), _t(np.cos(3 * nodes), dev)
    Wref, stref = ops.enhance(x, u, M, 1e4, n, global_domain=(-1.0, 1.0))
    for chunks in (1, 3, 8):
        Wl, st, Wg = enhance_sharded(x, u, ShardPlan(ne, 1), 0, M, 1e4, n,
                                     global_domain=(-1.0, 1.0), chunks=chunks)
        torch.cuda.synchronize()
        assert torch.equal(Wg, Wref) and torch.equal(Wl, Wref) and torch.equal(st, stref)
    Wl, st, Wg = enhance_sharded(x, u, ShardPlan(ne, 1), 0, M, 1e4, n, global_domain=(-1.0, 1.0),
                                 gather=False)
    torch.cuda.synchronize()
    assert Wg is None and torch.equal(Wl, Wref)


def test_step_refuses_rank_deficient_primal(dev):
    """n_colloc < M-2: the BC-eliminated primal Gram is rank deficient and float64 returns O(1)
    errors with status OK (oracle: 2.06 relative at M=17, n=12).  lssvr_enhance and
    lssvr_enhance_varcoef reroute to the dual solver (tests/test_gpu_enhance_dual.py); the fused
    lssvr_step is primal-only and must refuse."""
    import torch
    from hybrid_fem_lssvr_amd import ops, _capi
    nodes = np.linspace(-1, 1, 25)
    x, u = _t(nodes, dev), _t(np.sin(np.pi * nodes), dev)
    # (refused when the plan is bound -- lssvr_step_plan_create validates what lssvr_step validates -- and by the
    # unbound call as well)
    with pytest.raises(_capi.LssvrHipError, match="rank deficient"):
        ops.StepPlan(x, u, 17, 1e4, 12, global_domain=(-1.0, 1.0))
    ok = ops.StepPlan(x, u, 14, 1e4, 12, global_domain=(-1.0, 1.0))
    bad = list(ok._cargs)
    bad[9] = type(bad[9])(17)                                  # M of the bare lssvr_step argument tuple
    assert ok._step(*bad, torch.cuda.current_stream().cuda_stream) == -5
    assert b"rank deficient" in ok.lib.lssvr_last_error()
    # the boundary of the regime (n == M-2) is accepted
    ops.StepPlan(x, u, 14, 1e4, 12, global_domain=(-1.0, 1.0)).launch()
    torch.cuda.synchronize()
    # the rerouted call is exact where the primal solver would be O(1) wrong
    W, st = ops.enhance(x, u, 17, 1e4, 12, global_domain=(-1.0, 1.0))
    torch.cuda.synchronize()
    assert int(st.sum()) == 0
    if cf.HAVE_MP:
        sel = [0, 11, 23]
        tr = cf.truth_all(nodes, np.sin(np.pi * nodes), 17, 1e4, 12, orc.poisson_rhs, (-1.0, 1.0), sel)
        assert orc.rel_l2_coef(W.cpu().numpy()[sel], tr).max() <= 1e-13


@pytest.mark.parametrize("n", [257, 1024, 4096])
def test_in_kernel_sin_many_points(dev, n):
    """The in-kernel right-hand side carries (sin, cos) by a rotation that is re-seeded every 64
    points: RHS_SIN must agree with the tabulated numpy values at the 1e-13 bar up to the ABI's
    largest collocation count (an unseeded rotation drifts to 1.5e-13 in sin at n = 4096)."""
    import torch
    from hybrid_fem_lssvr_amd import ops
    ne, M = 300, 9
    nodes = np.linspace(-3, 5, ne + 1) * 1.0
    nodes[1:-1] += 1e-3 * np.sin(np.arange(1, ne))           # non-uniform h up to ~2 x the mean
    values = np.sin(np.pi * nodes)
    x = _t(nodes, dev)
    f = _t(orc.poisson_rhs(ops.colloc_points(x, n).cpu().numpy()), dev)
    W1, s1 = ops.enhance(x, _t(values, dev), M, 1e4, n, rhs_values=f)
    W2, s2 = ops.enhance(x, _t(values, dev), M, 1e4, n)
    torch.cuda.synchronize()
    assert int(s1.sum()) == 0 and int(s2.sum()) == 0
    assert orc.rel_l2_coef(W1.cpu().numpy(), W2.cpu().numpy()).max() <= 1e-13


def test_wrapper_buffer_validation(dev):
    """Caller buffers travel as raw pointers: every wrapper rejects wrong sizes in Python."""
    import torch
    from hybrid_fem_lssvr_amd import ops
    nodes = np.linspace(-1, 1, 25)
    x, u = _t(nodes, dev), _t(np.sin(np.pi * nodes), dev)
    gd = (-1.0, 1.0)
    small = torch.empty((23, 9), dtype=torch.float64, device=dev)
    st_small = torch.empty(23, dtype=torch.int32, device=dev)
    op = ops.build_shared_operator(2.0 / 24, 9, 1e4, 16, device=dev)
    for call in (
        lambda: ops.enhance(x, u, 9, 1e4, 16, global_domain=gd, out=small),
        lambda: ops.enhance(x, u, 9, 1e4, 16, global_domain=gd, status=st_small),
        lambda: ops.enhance(x, u, 9, 1e4, 16, global_domain=gd, rhs_values=torch.zeros(24 * 15, dtype=torch.float64, device=dev)),
        lambda: ops.enhance_shared(x, u, op, 9, 16, global_domain=gd, out=small),
        lambda: ops.enhance_shared(x, u, op, 9, 16, global_domain=gd, status=st_small),
        lambda: ops.enhance_shared(x, u, op, 9, 16, global_domain=gd, rhs_values=torch.zeros(7, dtype=torch.float64, device=dev)),
        lambda: ops.enhance_shared(x, u, op[:, :8].contiguous(), 9, 16, global_domain=gd),
        lambda: ops.enhance_profiled(x, u, 9, 1e4, 16, global_domain=gd, out=small),
        lambda: ops.enhance_profiled(x, u, 9, 1e4, 16, global_domain=gd, status=st_small),
        lambda: ops.StepPlan(x, u, 9, 1e4, 16, global_domain=gd, out=small),
        lambda: ops.StepPlan(x, u, 9, 1e4, 16, global_domain=gd, status=st_small),
        lambda: ops.enhance_varcoef(x, u, 9, 1e4, 16, *(torch.ones((24, 16), dtype=torch.float64, device=dev),) * 3,
                                    global_domain=gd, out=small),
        lambda: ops.enhance_varcoef(x, u, 9, 1e4, 16, *(torch.ones((24, 16), dtype=torch.float64, device=dev),) * 3,
                                    global_domain=gd, status=st_small),
    ):
        with pytest.raises(ValueError):
            call()


# --------------------------------------------------------------------------------------------
# cold paths of the lane kernel (enhance_small_cheb.hpp): exact boundary rows on per-lane scratch
# (cheb_slow_build, taken by a wave with a_max max(|1+t_a|, |1-t_b|) >= 1e-6) and the per-point
# sin (taken by a wave with any |delta| > 1e-7, i.e. |omega x| >~ 1e9)
# --------------------------------------------------------------------------------------------
def _boundary_rows_slow(nodes, M):
    """The kernel's own predicate, restated with numpy's mapparms arithmetic (Dual.py:56 ->
    polyutils.mapparms / mapdomain): per element, then per 64-element wave."""
    a, b = nodes[:-1], nodes[1:]
    oldlen = b - a
    off = (b * -1.0 - a * 1.0) / oldlen
    scl = 2.0 / oldlen
    ea = 1.0 + (off + scl * a)
    eb = 1.0 - (off + scl * b)
    slow = 0.5 * M * (M - 1) * np.maximum(np.abs(ea), np.abs(eb)) >= 1.0e-6
    pad = np.zeros(-(-len(slow) // 64) * 64, dtype=bool)
    pad[:len(slow)] = slow
    return slow, pad.reshape(-1, 64).any(1)


COLD_CASES = [(M, ratio) for M in (3, 9, 14, 22) for ratio in (1.0e8, 1.0e10)]


@pytest.mark.parametrize("M,ratio", COLD_CASES)
def test_lane_kernel_exact_boundary_rows_cold_path(dev, note, M, ratio):
    """|x|/h = 1e8 and 1e10: t(xmin), t(xmax) miss -1, +1 by ~|x|/h eps, the first-order boundary
    rows no longer hold to 1e-13 and the wave runs cheb_slow_build (the exact Legendre recurrence
    at the float64 abscissae numpy's mapdomain gives, Dual.py:66-75) -- the default solver, M <= 22.
    The test restates the kernel's predicate, so it says which branch each case took: every case
    is on the cold path except (M = 3, 1e8), which sits just inside the first-order range and
    pins that side of the switch.  Against the 60-digit minimiser of the reference's QP at the
    same float64 abscissae, the float64 oracle, and the tabulated-rhs path."""
    import torch
    from hybrid_fem_lssvr_amd import ops
    ne, h, n = 150, 0.1, max(16, 2 * M)          # three waves, the last one partial
    rng = np.random.default_rng(int(M * 7 + np.log10(ratio)))
    nodes = ratio * h + h * np.arange(ne + 1)
    values = np.sin(np.pi * nodes) + 0.01 * rng.standard_normal(ne + 1)
    gd = (nodes[0], nodes[-1])
    slow_el, slow_wave = _boundary_rows_slow(nodes, M)
    if (M, ratio) == (3, 1.0e8):
        assert not slow_wave.any()
    else:
        assert slow_wave.all(), (M, ratio, slow_el.mean())
    W, st = _enhance(dev, nodes, values, M, 1e4, n, global_domain=gd)
    assert np.all(st == 0)
    Wo = orc.enhance_all_vec(nodes, values, M, 1e4, n, global_domain=gd)
    eo = orc.rel_l2_coef(W, Wo).max()
    note("cold boundary rows M=%d |x|/h=%.0e vs float64 oracle" % (M, ratio), eo, 1e-14)
    assert eo <= 1e-14, eo                      # measured 1.7e-16 .. 9.2e-16 over the eight cases
    # boundary rows hold at the float64 end-point abscissae (what Dual.py:66-75 evaluates)
    x = _t(nodes, dev)
    ul, _ = ops.evaluate(x, _t(W, dev), x[:-1].contiguous())
    assert np.max(np.abs(ul.cpu().numpy()[1:] - values[1:-1])) < 1e-11    # u_e(x_e) = g_l (interior)
    xc = ops.colloc_points(x, n)
    f = _t(orc.poisson_rhs(xc.cpu().numpy()), dev)
    W2, st2 = ops.enhance(x, _t(values, dev), M, 1e4, n, global_domain=gd, rhs_values=f)
    torch.cuda.synchronize()
    assert int(st2.sum()) == 0
    et = orc.rel_l2_coef(W2.cpu().numpy(), W).max()
    note("cold boundary rows M=%d |x|/h=%.0e in-kernel rhs vs tabulated" % (M, ratio), et, 2e-15)
    assert et <= 2e-15, et                      # measured <= 1.6e-16
    if cf.HAVE_MP:
        sel = [0, 1, 63, 64, ne // 2, ne - 1]
        tr = cf.truth_all(nodes, values, M, 1e4, n, orc.poisson_rhs, gd, sel)
        err = orc.rel_l2_coef(W[sel], tr).max()
        note("cold boundary rows M=%d |x|/h=%.0e vs 60-digit minimiser" % (M, ratio), err, 2e-15)
        assert err <= 2e-15, err                # measured 1.1e-16 .. 2.3e-16


@pytest.mark.parametrize("h_right", [8.0, 4000.0])
def test_lane_kernel_cold_path_is_per_wave(dev, note, h_right):
    """One mesh, far from the origin on the left and near it on the right: some waves take the
    exact boundary rows, others the first-order ones; both agree with the oracle and the switch
    leaves no seam.  h_right = 8: gamma scl^4 = 39, the moment form; h_right = 4000 (round 3's first
    contact, gpurun_out/pytest_r3a.log: 7.6e-13 against the 60-digit minimiser): gamma scl^4 = 6e-10,
    RIDGE-DOMINATED -- those waves solve in the Legendre-bubble basis (cheb_ridge_solve, round 4)."""
    M, n = 9, 16
    left = 3.0e7 + 0.05 * np.arange(200)                       # |x|/h = 6e8: cold boundary rows
    right = left[-1] + np.cumsum(np.full(250, h_right))        # |x|/h = 4e6 (7.5e3): first-order rows
    nodes = np.concatenate([left, right])
    ne = len(nodes) - 1
    values = np.cos(0.37 * np.arange(ne + 1))
    gd = (nodes[0], nodes[-1])
    _, slow_wave = _boundary_rows_slow(nodes, M)
    assert slow_wave[:3].all() and not slow_wave[4:].any()
    ridge = _ridge_dominated(nodes, M, 1e4)
    assert not ridge[:199].any() and ridge[199:].all() == (h_right > 100.0) and ridge[199:].any() == (h_right > 100.0)
    W, st = _enhance(dev, nodes, values, M, 1e4, n, global_domain=gd)
    assert np.all(st == 0)
    Wo = orc.enhance_all_vec(nodes, values, M, 1e4, n, global_domain=gd)
    assert orc.rel_l2_coef(W, Wo).max() <= 1e-11
    if cf.HAVE_MP:
        sel = [0, 191, 192, 198, 199, 200, 255, 256, ne - 1]   # 198 / 199: the last fine, the first coarse element
        tr = cf.truth_all(nodes, values, M, 1e4, n, orc.poisson_rhs, gd, sel)
        err = orc.rel_l2_coef(W[sel], tr).max()
        note("cold path per wave, h_right=%g vs 60-digit minimiser" % h_right, err, TOL_TRUTH)
        assert err <= TOL_TRUTH, err


def _ridge_dominated(nodes, M, gamma):
    """The kernels' predicate (lssvr_device.hpp::ridge_dominated), restated: gamma scl^4 below
    ridge_gamma_scl4(M) -- the element is solved in the Legendre-bubble basis."""
    h = np.diff(nodes)
    thr = 3.0e-4 if M <= 12 else (1.0e-4 if M <= 17 else 3.0e-5)
    g4 = np.asarray(gamma, dtype=np.float64) * (2.0 / h) ** 4
    return (M >= 5) & (g4 < thr)          # (M <= 4: Y is diagonal, the moment form needs no help; kRidgeMinM)


# gamma scl^4 of the sweep x (M, n): the verdict's table of round 3 (Chebyshev-moment form 2.6e-13 at M = 9 ..
# 3e-11 at M = 22 / 33 where the float64 KKT solve holds 1e-15) and its two decades towards the crossover.
RIDGE_SWEEP = [(M, n, g4) for (M, n) in [(3, 4), (4, 6), (5, 5), (9, 16), (14, 28), (22, 44), (33, 64)]
               for g4 in (1e-3, 1e-6, 1e-9, 1e-12)]


@pytest.mark.skipif(not cf.HAVE_MP, reason="mpmath missing")
@pytest.mark.parametrize("M,n,g4", RIDGE_SWEEP)
def test_ridge_dominated_sweep(dev, note, M, n, g4):
    """gamma scl^4 << 1 (coarse elements with a small penalty): in the Chebyshev basis the ridge is
    eps (N + C_z^T C_z), N = Y^T Y, and the moment form inherits cond(Y)^2 (round 3: 2.6e-13 at M = 9,
    3e-11 at M = 22 and 33, against 1e-15 of a float64 KKT solve).  Round 4: below
    ridge_gamma_scl4(M) the kernels solve in the Legendre-bubble basis from the same moments
    (cheb_ridge_solve per lane, ridge_wave_solve per wave above M = 22).  Every entry -- in-kernel and
    tabulated right-hand side (both layouts), the fused step, a subset launch -- against the 60-digit
    minimiser of the reference's QP (Dual.py:46-78) at 1e-13; at M = 33 the problem's own conditioning
    shows at gamma scl^4 = 1e-6 (float64 KKT / direct Gram: 4e-13 .. 6e-13 on this mesh): the bar there
    is 4 x the float64 oracle's own error, and never below 1e-13."""
    import torch
    from hybrid_fem_lssvr_amd import ops
    ne, h = 150, 0.5                            # three waves, the last one partial
    gamma = g4 / (2.0 / h) ** 4
    rng = np.random.default_rng(M * 1000 + n)
    nodes = -1.0 + h * np.arange(ne + 1)
    values = np.sin(np.pi * nodes) + 0.01 * rng.standard_normal(ne + 1)
    gd = (nodes[0], nodes[-1])
    ridge = _ridge_dominated(nodes, M, gamma)
    expect = (g4 < 1e-4) and M >= 5           # 1e-3 (and M <= 4 always): the moment form, below: the ridge form
    assert ridge.all() == expect and ridge.any() == expect
    x, u = _t(nodes, dev), _t(values, dev)
    sel = [0, 1, 63, 64, 127, 128, ne - 1]
    tr = cf.truth_all(nodes, values, M, gamma, n, orc.poisson_rhs, gd, sel)
    W, st = ops.enhance(x, u, M, gamma, n, global_domain=gd)
    xc = ops.colloc_points(x, n)
    f = _t(orc.poisson_rhs(xc.cpu().numpy()), dev)
    W2, st2 = ops.enhance(x, u, M, gamma, n, global_domain=gd, rhs_values=f)
    W3, _ = ops.enhance(x, u, M, gamma, n, global_domain=gd, rhs_values=f.t().contiguous(), point_major=True)
    plan = ops.StepPlan(x, u, M, gamma, n, global_domain=gd)
    W4, st4 = plan.launch()
    W5 = torch.zeros((ne, M + 3), dtype=torch.float64, device=dev)
    st5 = torch.full((ne,), -7, dtype=torch.int32, device=dev)
    ops.enhance_subset(x, u, M, gamma, n, W5, elem_ids=_t(np.array(sel, dtype=np.int64), dev), global_domain=gd,
                       status=st5)
    torch.cuda.synchronize()
    assert int(st.sum()) == 0 and int(st2.sum()) == 0 and int(st4.sum()) == 0
    st5 = st5.cpu().numpy()
    assert np.all(st5[sel] == 0) and np.all(np.delete(st5, sel) == -7)
    W5 = W5.cpu().numpy()
    assert np.all(W5[:, M:] == 0.0) and np.all(np.delete(W5, sel, axis=0) == 0.0)
    bar = TOL_TRUTH
    oracle_err = max(orc.rel_l2_coef(orc.enhance_all_vec(nodes, values, M, gamma, n, global_domain=gd)[sel], tr).max(),
                     max(orc.rel_l2_coef(orc.solve_primal_kkt(orc.element_system(
                         nodes[i], nodes[i + 1], *orc.boundary_values(i, ne, nodes[i], nodes[i + 1], values[i],
                                                                      values[i + 1], gd), M, gamma, n))[None],
                                         tr[k][None])[0] for k, i in enumerate(sel)))
    if M == 33:
        bar = max(bar, 4.0 * oracle_err)
    worst = 0.0
    for Wx in (W.cpu().numpy(), W2.cpu().numpy(), W3.cpu().numpy(), W4.cpu().numpy(), W5[:, :M]):
        worst = max(worst, orc.rel_l2_coef(Wx[sel], tr).max(), orc.rel_l2_bubble(Wx[sel], tr).max())
    note("ridge sweep M=%d n=%d gamma scl^4=%.0e vs 60-digit minimiser (float64 oracle: %.1e)" % (M, n, g4, oracle_err),
         worst, bar)
    assert worst <= bar, (worst, bar, oracle_err)
    assert torch.equal(W, W4)


@pytest.mark.skipif(not cf.HAVE_MP, reason="mpmath missing")
@pytest.mark.parametrize("M,n", [(9, 16), (20, 18), (33, 64), (33, 40)])
def test_ridge_dominated_lanes_inside_a_wave(dev, note, M, n):
    """Ridge-dominated and ordinary elements side by side in one wave (alternating element lengths and a
    per-element gamma through the subset entry): a ridge lane takes the Legendre-bubble solve, its
    neighbours keep the moment form -- including the near-square refinement at (20, 18) and (33, 40) --
    and the two-kernel path's solve kernels leave the rows moments_kernel wrote alone."""
    import torch
    from hybrid_fem_lssvr_amd import ops
    ne = 200
    rng = np.random.default_rng(M + n)
    coarse = np.arange(ne) % 3 == 0
    hs = np.where(coarse, 40.0, 0.25)                          # scl^4 = 6e-6 / 4096
    nodes = np.concatenate([[-3.0], -3.0 + np.cumsum(hs)])
    values = 0.3 * rng.standard_normal(ne + 1)
    gd = (nodes[0], nodes[-1])
    gam = np.where(coarse, 1e-5, 1e3) * 10.0 ** rng.uniform(-1, 1, ne)   # gamma scl^4 ~ 6e-11 / 4e6
    gam[5::7] = 1e-14                                           # a few fine elements ridge-dominated through gamma
    ridge = _ridge_dominated(nodes, M, gam)
    assert ridge[:64].any() and not ridge[:64].all() and ridge[0] and not ridge[1] and ridge[5]
    x, u = _t(nodes, dev), _t(values, dev)
    W = torch.zeros((ne, M), dtype=torch.float64, device=dev)
    st = torch.full((ne,), -7, dtype=torch.int32, device=dev)
    fc = torch.zeros(1, dtype=torch.int32, device=dev)
    ops.enhance_subset(x, u, M, 1.0, n, W, gamma_values=_t(gam, dev), global_domain=gd, status=st, fail_count=fc)
    torch.cuda.synchronize()
    assert int(fc.item()) == 0 and int(st.abs().sum()) == 0
    W = W.cpu().numpy()
    sel = [0, 1, 2, 3, 5, 6, 63, 64, 65, 66, ne - 2, ne - 1]
    worst_r, worst_o = 0.0, 0.0
    for i in sel:
        gl, gr = orc.boundary_values(i, ne, nodes[i], nodes[i + 1], values[i], values[i + 1], gd)
        s = orc.element_system(nodes[i], nodes[i + 1], gl, gr, M, gam[i], n)
        tr = cf.solve_truth(s)
        e = float(orc.rel_l2_coef(W[i][None], tr[None])[0])
        if ridge[i]:
            worst_r = max(worst_r, e)
        else:
            worst_o = max(worst_o, e)
    note("mixed wave M=%d n=%d: ridge lanes vs 60-digit minimiser" % (M, n), worst_r, TOL_TRUTH)
    note("mixed wave M=%d n=%d: ordinary lanes vs 60-digit minimiser" % (M, n), worst_o, TOL_TRUTH)
    assert worst_r <= TOL_TRUTH and worst_o <= TOL_TRUTH, (worst_r, worst_o)


@pytest.mark.parametrize("x0,h", [(1.0e9, 512.0), (-3.0e8, 1.0 / 12), (2.5e9, 0.7)])
def test_in_kernel_sin_per_point_branch(dev, note, x0, h):
    """|omega x| ~ 1e9 .. 8e9: numpy's argument rounding fl(pi x_k) differs from the carried
    rotation angle by more than 1e-7, so the wave evaluates sin at every point (sin_tab on the
    exactly rounded argument, Dual.py:11-12 / 43-44) instead of the rotation + first-order
    correction.  (1e9, 512): boundary rows on the first-order path, per-point sin only;
    (-3e8, 1/12) and (2.5e9, 0.7): both cold paths together.  Against the tabulated path (numpy's
    own sin at the same abscissae) and the 60-digit minimiser."""
    import torch
    from hybrid_fem_lssvr_amd import ops
    ne, M, n = 140, 9, 16
    nodes = x0 + h * np.arange(ne + 1)
    values = np.sin(np.pi * nodes)
    gd = (nodes[0], nodes[-1])
    # the kernel's predicate for the per-point branch, restated
    a = nodes[:-1]
    step = (nodes[1:] - a) / (n - 1)
    k = np.arange(n - 1, dtype=np.float64)
    xk = k[None, :] * step[:, None] + a[:, None]
    delta = (np.pi * xk - (np.pi * a)[:, None]) - k[None, :] * (np.pi * step)[:, None]
    assert (np.abs(delta).max(1) > 1.0e-7).any()
    _, slow_wave = _boundary_rows_slow(nodes, M)
    assert slow_wave.all() == (h < 100.0) and slow_wave.any() == (h < 100.0)
    x, u = _t(nodes, dev), _t(values, dev)
    W1, s1 = ops.enhance(x, u, M, 1e4, n, global_domain=gd)
    xc = ops.colloc_points(x, n).cpu().numpy()
    W2, s2 = ops.enhance(x, u, M, 1e4, n, global_domain=gd, rhs_values=_t(orc.poisson_rhs(xc), dev))
    torch.cuda.synchronize()
    assert int(s1.sum()) == 0 and int(s2.sum()) == 0
    W1, W2 = W1.cpu().numpy(), W2.cpu().numpy()
    d = orc.rel_l2_coef(W1, W2).max()
    db = orc.rel_l2_bubble(W1, W2).max()
    note("per-point sin x0=%.1e h=%g: in-kernel vs tabulated (whole / bubble)" % (x0, h), d)
    note("per-point sin x0=%.1e h=%g: bubble" % (x0, h), db)
    assert d <= 1e-14, d                        # measured 1.8e-16 .. 1.7e-15
    assert db <= 2e-14, db                      # measured 4.2e-16 .. 2.2e-15
    if cf.HAVE_MP:
        sel = [0, 1, 64, ne - 1]
        tr = cf.truth_all(nodes, values, M, 1e4, n, orc.poisson_rhs, gd, sel)
        err = orc.rel_l2_coef(W1[sel], tr).max()
        note("per-point sin x0=%.1e h=%g vs 60-digit minimiser" % (x0, h), err, 1e-14)
        assert err <= 1e-14, err                # measured 1.0e-16 .. 1.3e-15


# (M, n, h): about as many equispaced points as bubble coefficients, M <= 22 -- the lane kernel's own
# refinement (round 3; measured without it: 2.0e-12 at (22, 20, 0.5), 7e-14 at (22, 21, 0.5) and
# (20, 18, 0.5), 1.3e-14 at (22, 20, 1/12); with it <= 5.5e-16 everywhere)
LANE_NEAR_SQUARE = [(22, 20, 0.5), (22, 21, 0.5), (22, 24, 0.5), (20, 18, 0.5), (18, 16, 0.5), (16, 14, 0.5),
                    (14, 12, 0.5), (22, 20, 1.0 / 12), (20, 18, 1.0 / 12), (15, 13, 0.25)]


@pytest.mark.skipif(not cf.HAVE_MP, reason="mpmath missing")
@pytest.mark.parametrize("M,n,h", LANE_NEAR_SQUARE)
def test_lane_kernel_near_square_refinement(dev, note, M, n, h):
    """n ~ M - 2 below M = 23: 1-2 steps of the corrected semi-normal equations inside the lane kernel
    (enhance_small_refine_kernel: residual through the rows, the factors re-used) bring the result
    to the 60-digit minimiser at the same bar as everywhere else -- in-kernel and tabulated (both
    layouts) right-hand sides, the fused step, a subset launch."""
    import torch
    from hybrid_fem_lssvr_amd import ops
    ne = 150
    rng = np.random.default_rng(M * 100 + n)
    nodes = -1.0 + h * np.arange(ne + 1)
    values = np.sin(np.pi * nodes) + 0.01 * rng.standard_normal(ne + 1)
    gd = (nodes[0], nodes[-1])
    x, u = _t(nodes, dev), _t(values, dev)
    sel = [0, 1, 63, 64, ne - 1]
    tr = cf.truth_all(nodes, values, M, 1e4, n, orc.poisson_rhs, gd, sel)
    W, st = ops.enhance(x, u, M, 1e4, n, global_domain=gd)
    xc = ops.colloc_points(x, n)
    f = _t(orc.poisson_rhs(xc.cpu().numpy()), dev)
    W2, _ = ops.enhance(x, u, M, 1e4, n, global_domain=gd, rhs_values=f)
    W3, _ = ops.enhance(x, u, M, 1e4, n, global_domain=gd, rhs_values=f.t().contiguous(), point_major=True)
    plan = ops.StepPlan(x, u, M, 1e4, n, global_domain=gd)
    W4, _ = plan.launch()
    W5 = torch.zeros((ne, M), dtype=torch.float64, device=dev)
    ops.enhance_subset(x, u, M, 1e4, n, W5, elem_ids=_t(np.array(sel, dtype=np.int64), dev), global_domain=gd)
    torch.cuda.synchronize()
    assert int(st.sum()) == 0
    worst = 0.0
    for Wx in (W, W2, W3, W4, W5):
        worst = max(worst, orc.rel_l2_coef(Wx.cpu().numpy()[sel], tr).max())
    note("lane near-square M=%d n=%d h=%.3g vs 60-digit minimiser" % (M, n, h), worst, 5e-15)
    assert worst <= 5e-15, worst
    assert torch.equal(W2, W3) and torch.equal(W, W4)


@pytest.mark.skipif(not cf.HAVE_MP, reason="mpmath missing")
@pytest.mark.parametrize("seed", [31, 32, 33])
def test_random_parity_sweep(dev, note, seed):
    """scripts/stress_parity.py as a driver-run test: 40 random (M in [2, 33], n in [2, 80], gamma in
    [1e-2, 1e8], h in [1e-7, 10], x0 up to 1e6) per seed on non-uniform 12-element meshes, every route
    the default solver can take (lane kernel incl. its cold paths and near-square refinement, moment /
    parity-split / refined two-kernel path, dual solver below the rank boundary), each against the
    60-digit minimiser of the reference's QP (Dual.py:46-78) at the same float64 abscissae.
    Bar: 1e-13 where gamma scl^4 >= 1 and h <= 1 (every regime a mesh refinement produces) AND where the
    ridge dominates outright (gamma scl^4 <= 1e-8: round 3's docstring called that regime "the problem's
    own cond(A) u" -- it was the moment form's cond(Y)^2, a float64 KKT solve holds 1e-15 there; since
    round 4 those elements are solved in the Legendre-bubble basis).  In between (1e-8 < gamma scl^4 < 1,
    or elements many periods of the right-hand side long) the high-degree directions are Gram-dominated
    and the low ones ridge-dominated: there the float64 KKT solve itself reads 1e-13 .. 1e-11 at
    M >= 22 (scripts/proto/cheb_moment.py ridge): 1e-12 (measured over the three seeds: 1.4e-14)."""
    import torch
    from hybrid_fem_lssvr_amd import ops
    rng = np.random.default_rng(seed)
    worst_tight, worst_loose = 0.0, 0.0
    for _ in range(40):
        M = int(rng.integers(2, 34))
        n = int(rng.integers(2, 81))
        if n < M - 2 and n > 64:
            continue
        gamma = 10.0 ** rng.uniform(-2, 8)
        h = 10.0 ** rng.uniform(-7, 1)
        x0 = rng.choice([-1, 1]) * 10.0 ** rng.uniform(-1, 6) * rng.choice([0, 1, 1])
        ne = 12
        nodes = x0 + h * np.cumsum(np.concatenate([[0.0], rng.uniform(0.5, 1.5, ne)]))
        if not np.all(np.diff(nodes) > 0):
            continue
        values = np.sin(np.pi * nodes) + 0.1 * rng.standard_normal(ne + 1)
        gd = (nodes[0], nodes[-1])
        W, st = ops.enhance(_t(nodes, dev), _t(values, dev), M, gamma, n, global_domain=gd)
        torch.cuda.synchronize()
        W, st = W.cpu().numpy(), st.cpu().numpy()
        sel = [0, 5, 11]
        assert np.all(st[sel] == 0), (M, n, gamma, h, x0)
        tr = cf.truth_all(nodes, values, M, gamma, n, orc.poisson_rhs, gd, sel)
        err = float(orc.rel_l2_coef(W[sel], tr).max())
        hd = np.diff(nodes)
        well_posed = gamma * (2.0 / hd.max()) ** 4 >= 1.0 and hd.max() <= 1.0
        ridge_only = gamma * (2.0 / hd.min()) ** 4 <= 1.0e-8
        if well_posed or ridge_only:
            worst_tight = max(worst_tight, err)
            assert err <= 1e-13, (err, M, n, gamma, h, x0)
        else:
            worst_loose = max(worst_loose, err)
            assert err <= 1e-12, (err, M, n, gamma, h, x0)
    note("random sweep seed %d: worst, well-posed or ridge-dominated regime" % seed, worst_tight, 1e-13)
    note("random sweep seed %d: worst, mixed regime (1e-8 < gamma scl^4 < 1 or h > 1)" % seed, worst_loose, 1e-12)
