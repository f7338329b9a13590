"""GPU tests of LSSVR_SOLVER_DUAL: north_star's Gram form K = Z Z^T + I/gamma, (n+2) solve.

Accuracy gate: the dual system is well conditioned when there are fewer rows than Legendre
coefficients (n + 2 <= M) -- exactly where the primal normal equations are rank deficient --
and ill conditioned otherwise (cond ~ |A A^T| gamma scl^4), SURVEY.md Appendix B.3."""
import numpy as np
import pytest

from oracle import lssvr_oracle as orc
from oracle import closed_form_mp as cf

pytestmark = pytest.mark.gpu


def _run(dev, nodes, values, M, gamma, n, **kw):
    import torch
    from hybrid_fem_lssvr_amd import ops
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
    W, st = ops.enhance(t(nodes), t(values), M, gamma, n, solver=ops.SOLVER_DUAL, **kw)
    torch.cuda.synchronize()
    return W.cpu().numpy(), st.cpu().numpy()


@pytest.mark.skipif(not cf.HAVE_MP, reason="mpmath missing")
@pytest.mark.parametrize("M,n,tol", [(12, 6, 1e-13), (12, 10, 1e-12), (20, 8, 1e-13), (24, 16, 1e-13),
                                     (32, 20, 1e-13), (32, 12, 1e-13), (9, 3, 1e-13), (6, 2, 1e-13),
                                     (32, 29, 1e-5)])
def test_dual_exact_where_primal_is_rank_deficient(dev, M, n, tol):
    """n + 2 <= M: the dual solve reaches the 60-digit minimiser (the float64 primal forms are
    O(1) wrong there).  (32, 29): 29 equispaced points against degree 31 is an ill-conditioned
    Vandermonde whatever the formulation -- numpy's equilibrated LU reaches 1e-9, LDL^T 2e-6."""
    ne = 37
    nodes = np.linspace(-1, 1, ne + 1)
    values = orc.fem_p1_solve(nodes)
    W, st = _run(dev, nodes, values, M, 1e4, n)
    assert np.all(st == 0)
    sel = [0, 1, 18, 36]
    tr = cf.truth_all(nodes, values, M, 1e4, n, orc.poisson_rhs, (-1.0, 1.0), sel)
    assert orc.rel_l2_coef(W[sel], tr).max() <= tol
    # boundary rows hold
    sgn = (-1.0) ** np.arange(M)
    btol = 1e-12 if tol < 1e-10 else 1e-6
    assert np.max(np.abs(W @ sgn - np.concatenate([[0.0], values[1:-1]]))) < btol
    assert np.max(np.abs(W.sum(1) - np.concatenate([values[1:-1], [0.0]]))) < btol


def test_dual_on_reference_configurations(dev, golden):
    """More rows than coefficients (every BASELINE configuration): the Gram form loses
    digits to its conditioning -- same behaviour as the float64 numpy restatement
    (oracle.solve_dual_gram) -- which is why PRIMAL is the default."""
    for name, tol in (("G3_ne24_M9_n16", 1e-10), ("G2_default_ne24_M8_n12", 1e-10), ("G1_c1_ne8_M5_n5", 1e-7)):
        g = golden(name)
        ne, M, n = int(g["ne"]), int(g["M"]), int(g["n"])
        nodes = np.linspace(-1, 1, ne + 1)
        values = np.concatenate([g["values_sel"][:, 0], g["values_sel"][-1:, 1]])
        W, st = _run(dev, nodes, values, M, float(g["gamma"]), n)
        ok = st == 0
        assert ok.sum() >= ne - 1
        assert orc.rel_l2_coef(W[ok], g["coef_truth"][ok]).max() <= tol, name


def test_dual_matches_primal_at_moderate_conditioning(dev):
    from hybrid_fem_lssvr_amd import ops
    import torch
    ne, M, n = 2001, 9, 16
    nodes = np.linspace(-4, 4, ne + 1)
    values = np.sin(np.pi * nodes)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
    Wp, _ = ops.enhance(t(nodes), t(values), M, 1.0, n, global_domain=(-4.0, 4.0))
    Wd, sd = ops.enhance(t(nodes), t(values), M, 1.0, n, global_domain=(-4.0, 4.0), solver=ops.SOLVER_DUAL)
    torch.cuda.synchronize()
    assert int(sd.sum()) == 0
    assert orc.rel_l2_coef(Wd.cpu().numpy(), Wp.cpu().numpy()).max() <= 1e-7


def test_dual_limits_are_argument_errors(dev):
    import torch
    from hybrid_fem_lssvr_amd import _capi, ops
    x = torch.linspace(0, 1, 11, dtype=torch.float64, device=dev)
    with pytest.raises(_capi.LssvrHipError, match="n_colloc = 30"):
        ops.enhance(x, x, 9, 1e4, 30, global_domain=(0.0, 1.0), solver=ops.SOLVER_DUAL)
    with pytest.raises(_capi.LssvrHipError, match="M = 33"):
        ops.enhance(x, x, 33, 1e4, 12, global_domain=(0.0, 1.0), solver=ops.SOLVER_DUAL)
