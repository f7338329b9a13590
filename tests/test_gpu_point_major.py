"""GPU: point-major tables (LSSVR_RHS_ARRAY_PM / LSSVR_TABLE_POINT_MAJOR, ABI 4) against the
element-major ones.  The two layouts hold the same values, every kernel does the same arithmetic
on them in the same order, so the results are required to be BIT-equal; the element-major path is
itself pinned against the oracle / the goldens elsewhere (test_gpu_enhance.py, test_gpu_fem_eval.py).
The reference has no tabulated inputs at all (its f is a Python callable, Dual.py:20,157): the
tables are this boundary's way of passing ``rhs_func`` -- and config 5's a, a' -- across a C ABI."""
import numpy as np
import pytest

from oracle import lssvr_oracle as orc

pytestmark = pytest.mark.gpu


def _t(a, dev):
    import torch
    return torch.as_tensor(np.ascontiguousarray(a), device=dev)


@pytest.mark.parametrize("M,n,ne", [(2, 5, 70), (3, 2, 64), (9, 16, 1000), (9, 17, 129), (9, 3, 5),
                                    (14, 30, 333), (22, 44, 200), (24, 50, 77), (33, 64, 130), (33, 33, 9)])
def test_poisson_table_layouts_agree(dev, M, n, ne):
    """Tabulated f: element-major rows, point-major columns and the in-kernel f give the same W
    (the first two bit for bit) through the lane kernel (M <= 22) and the moment / MFMA kernels."""
    import torch
    from hybrid_fem_lssvr_amd import ops
    rng = np.random.default_rng(M * 100 + n)
    nodes = np.cumsum(np.concatenate([[-2.0], rng.uniform(0.01, 0.06, ne)]))
    values = np.sin(np.pi * nodes) + 0.01 * rng.standard_normal(ne + 1)
    gd = (nodes[0], nodes[-1])
    x, u = _t(nodes, dev), _t(values, dev)
    xc = ops.colloc_points(x, n)
    xp = ops.colloc_points(x, n, point_major=True)
    assert xp.shape == (n, ne) and torch.equal(xp, xc.t().contiguous())
    for e in (0, ne - 1):
        assert np.array_equal(xp[:, e].cpu().numpy(), np.linspace(nodes[e], nodes[e + 1], n))   # Dual.py:40
    f_em = _t(orc.poisson_rhs(xc.cpu().numpy()), dev)
    f_pm = f_em.t().contiguous()
    We, se = ops.enhance(x, u, M, 1e4, n, global_domain=gd, rhs_values=f_em)
    Wp, sp = ops.enhance(x, u, M, 1e4, n, global_domain=gd, rhs_values=f_pm, point_major=True)
    Wk, _ = ops.enhance(x, u, M, 1e4, n, global_domain=gd)
    torch.cuda.synchronize()
    assert int(se.sum()) == 0 and int(sp.sum()) == 0
    assert torch.equal(We, Wp)
    assert orc.rel_l2_coef(Wp.cpu().numpy(), Wk.cpu().numpy()).max() <= 1e-12
    if M > 22:           # the single f64-MFMA kernel (no workspace) reads the same tables
        W1, _ = ops.enhance(x, u, M, 1e4, n, global_domain=gd, rhs_values=f_em, work=False)
        W2, _ = ops.enhance(x, u, M, 1e4, n, global_domain=gd, rhs_values=f_pm, point_major=True, work=False)
        assert torch.equal(W1, W2)
    if n <= 64:          # the dual Gram solver
        W1, _ = ops.enhance(x, u, M, 1e4, n, global_domain=gd, rhs_values=f_em, solver=ops.SOLVER_DUAL)
        W2, _ = ops.enhance(x, u, M, 1e4, n, global_domain=gd, rhs_values=f_pm, point_major=True,
                            solver=ops.SOLVER_DUAL)
        assert torch.equal(W1, W2)


@pytest.mark.parametrize("M,n,ne", [(3, 5, 100), (9, 16, 1000), (9, 18, 131), (9, 7, 64), (12, 12, 300),
                                    (20, 32, 90), (26, 40, 50)])
def test_varcoef_table_layouts_agree(dev, M, n, ne):
    """BASELINE config 5's three tables in both layouts: bit-equal W; and against the float64 oracle."""
    import torch
    from hybrid_fem_lssvr_amd import ops
    a, da, f = orc.varcoef_functions(*orc.varcoef_params())
    nodes = np.linspace(-1, 1, ne + 1)
    values = orc.fem_p1_solve(nodes, rhs=f, coef_a=a)
    x, u = _t(nodes, dev), _t(values, dev)
    xc = ops.colloc_points(x, n).cpu().numpy()
    tabs = [_t(t(xc), dev) for t in (a, da, f)]
    We, se = ops.enhance_varcoef(x, u, M, 1e4, n, *tabs, global_domain=(-1.0, 1.0))
    Wp, sp = ops.enhance_varcoef(x, u, M, 1e4, n, *[t.t().contiguous() for t in tabs], global_domain=(-1.0, 1.0),
                                 point_major=True)
    torch.cuda.synchronize()
    assert int(se.sum()) == 0 and int(sp.sum()) == 0
    assert torch.equal(We, Wp)
    Wo = orc.enhance_all_vec(nodes, values, M, 1e4, n, rhs=f, coef_a=a, coef_da=da, global_domain=(-1.0, 1.0))
    assert orc.rel_l2_coef(Wp.cpu().numpy(), Wo).max() <= (1e-12 if M <= 22 else 1e-10)


def test_point_major_subset_and_shared(dev):
    """Subset launches index a point-major table by the position in elem_ids (t[k*nsub + pos]);
    the shared-operator kernel reads point-major columns directly."""
    import torch
    from hybrid_fem_lssvr_amd import ops
    ne, M, n = 500, 9, 16
    nodes = np.linspace(-1, 1, ne + 1)
    x, u = _t(nodes, dev), _t(np.sin(np.pi * nodes), dev)
    ids = torch.as_tensor(np.random.default_rng(3).permutation(ne)[:137].astype(np.int64), device=dev)
    xc = ops.colloc_points(x, n)
    f_sub = _t(orc.poisson_rhs(xc[ids].cpu().numpy()), dev)                 # [nsub, n]
    W1 = torch.zeros((ne, M), dtype=torch.float64, device=dev)
    W2 = torch.zeros_like(W1)
    ops.enhance_subset(x, u, M, 1e4, n, W1, elem_ids=ids, rhs_values=f_sub, global_domain=(-1.0, 1.0))
    ops.enhance_subset(x, u, M, 1e4, n, W2, elem_ids=ids, rhs_values=f_sub.t().contiguous(), point_major=True,
                       global_domain=(-1.0, 1.0))
    torch.cuda.synchronize()
    assert torch.equal(W1, W2) and float(W1.abs().sum()) > 0
    op = ops.build_shared_operator(2.0 / ne, M, 1e4, n, device=dev)
    f_all = _t(orc.poisson_rhs(xc.cpu().numpy()), dev)
    S1, _ = ops.enhance_shared(x, u, op, M, n, rhs_values=f_all, global_domain=(-1.0, 1.0))
    S2, _ = ops.enhance_shared(x, u, op, M, n, rhs_values=f_all.t().contiguous(), point_major=True,
                               global_domain=(-1.0, 1.0))
    torch.cuda.synchronize()
    assert torch.equal(S1, S2)


@pytest.mark.parametrize("M,n,pm", [(9, 16, True), (9, 16, False), (5, 9, True), (12, 20, False), (14, 24, True)])
def test_fused_varcoef_step_equals_separate_launches(dev, M, n, pm):
    """lssvr_step_varcoef (a-weighted P1 assembly + variable-coefficient enhancement: one grid for
    M <= 12, two launches above) == lssvr_p1_assemble + lssvr_enhance_varcoef, bit for bit, both layouts."""
    import torch
    from hybrid_fem_lssvr_amd import ops
    a, da, f = orc.varcoef_functions(*orc.varcoef_params())
    ne = 1237
    nodes = np.linspace(-1, 1, ne + 1)
    values = orc.fem_p1_solve(nodes, rhs=f, coef_a=a)
    x, u = _t(nodes, dev), _t(values, dev)
    xc = ops.colloc_points(x, n, point_major=pm).cpu().numpy()
    xq = ops.quad_points(x, 2).cpu().numpy()
    tabs = [_t(t(xc), dev) for t in (a, da, f)]
    fq, aq = _t(f(xq), dev), _t(a(xq), dev)
    plan = ops.StepPlanVarcoef(x, u, M, 1e4, n, *tabs, fq, aq, point_major=pm, global_domain=(-1.0, 1.0))
    W, st = plan.launch()
    torch.cuda.synchronize()
    W2, st2 = ops.enhance_varcoef(x, u, M, 1e4, n, *tabs, global_domain=(-1.0, 1.0), point_major=pm)
    b2 = ops.p1_assemble(x, 2, rhs_quad=fq, a_quad=aq)
    torch.cuda.synchronize()
    assert int(st.sum()) == 0 and torch.equal(W, W2) and torch.equal(st, st2)
    for k in ("diag", "off", "load"):
        assert torch.equal(plan.bands[k], b2[k]), k
    # the assembled system solves to the nodal values the oracle's P1 step gives
    uu = ops.tridiag_dirichlet_solve(plan.bands["diag"], plan.bands["off"], plan.bands["load"]).cpu().numpy()
    assert np.max(np.abs(uu - values)) < 1e-9
