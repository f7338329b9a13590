"""GPU parity tests of the heterogeneous-mesh entry (lssvr_enhance_subset):
per-element gamma, degree and collocation count on non-uniform meshes.  The reference has one
(lssvr_M, lssvr_gamma) per mesh (Dual.py:101) and a uniform mesh (Dual.py:112), so there is no
reference output for a mixed mesh: every element is checked against the float64 oracle solve of
ITS OWN problem (the same per-element QP the golden tests pin), i.e. parity per element."""
import numpy as np
import pytest
from numpy.polynomial.legendre import Legendre

from oracle import lssvr_oracle as orc

pytestmark = pytest.mark.gpu


def _oracle_element(nodes, values, i, M, gamma, n, gd):
    ne = len(nodes) - 1
    gl, gr = orc.boundary_values(i, ne, nodes[i], nodes[i + 1], values[i], values[i + 1], gd)
    return orc.solve_primal_kkt(orc.element_system(nodes[i], nodes[i + 1], gl, gr, M, gamma, n))


def _mesh(ne, seed):
    rng = np.random.default_rng(seed)
    nodes = np.cumsum(np.concatenate([[-1.3], rng.uniform(0.02, 0.09, ne)]))
    values = np.sin(np.pi * nodes) + 0.02 * rng.standard_normal(ne + 1)
    return rng, nodes, values


def test_per_element_gamma(dev):
    import hybrid_fem_lssvr_amd as pkg
    rng, nodes, values = _mesh(300, 11)
    ne = 300
    gam = 10.0 ** rng.uniform(2, 7, ne)
    sol = pkg.enhance_elements_hetero(nodes, values, 9, gam, n_colloc=16)
    W = sol.W.cpu().numpy()
    assert sol.n_fallback == 0
    gd = (nodes[0], nodes[-1])
    for i in range(ne):
        wo = _oracle_element(nodes, values, i, 9, gam[i], 16, gd)
        assert orc.rel_l2_coef(W[i][None], wo[None]).max() <= 1e-12, i


def test_mixed_degree_and_colloc(dev):
    import hybrid_fem_lssvr_amd as pkg
    rng, nodes, values = _mesh(240, 12)
    ne = 240
    choices = [(5, 8), (9, 16), (12, 12), (14, 20), (22, 40), (25, 48), (33, 64)]
    pick = rng.integers(0, len(choices), ne)
    Ms = np.array([choices[k][0] for k in pick])
    ns = np.array([choices[k][1] for k in pick])
    gam = 10.0 ** rng.uniform(3, 5, ne)
    sol = pkg.enhance_elements_hetero(nodes, values, Ms, gam, n_colloc=ns)
    W = sol.W.cpu().numpy()
    assert W.shape == (ne, 33) and sol.n_fallback == 0
    gd = (nodes[0], nodes[-1])
    for i in range(ne):
        wo = _oracle_element(nodes, values, i, int(Ms[i]), gam[i], int(ns[i]), gd)
        assert np.all(W[i, Ms[i]:] == 0.0)
        tol = 1e-12 if Ms[i] <= 22 else 1e-11
        assert orc.rel_l2_coef(W[i, :Ms[i]][None], wo[None]).max() <= tol, (i, Ms[i])
    # the padded rows evaluate like the per-element numpy series (Dual.py:176-203 rule)
    xq = np.sort(rng.uniform(nodes[0], nodes[-1], 500))
    uq, elem = sol.evaluate(xq, return_elements=True)
    ref = np.array([Legendre(W[e, :Ms[e]], [nodes[e], nodes[e + 1]])(x) for x, e in zip(xq, elem)])
    assert np.max(np.abs(uq - ref)) <= 1e-13 * max(1.0, np.max(np.abs(ref)))


def test_subset_leaves_other_rows_alone(dev):
    import torch
    from hybrid_fem_lssvr_amd import ops
    _, nodes, values = _mesh(130, 13)
    x = torch.as_tensor(nodes, device=dev)
    u = torch.as_tensor(values, device=dev)
    gd = (nodes[0], nodes[-1])
    for M, n in [(9, 16), (33, 64)]:
        W = torch.full((130, M + 3), 7.0, dtype=torch.float64, device=dev)
        st = torch.full((130,), -5, dtype=torch.int32, device=dev)
        ids = torch.as_tensor(np.array([129, 0, 64, 3, 77], dtype=np.int64), device=dev)
        touched = np.zeros(130, dtype=bool)
        touched[[129, 0, 64, 3, 77]] = True
        Wfull, _ = ops.enhance(x, u, M, 1e4, n, global_domain=gd, work=False)
        Wdef, _ = ops.enhance(x, u, M, 1e4, n, global_domain=gd)
        # work=False: the workspace-free kernels, bit-identical to the full launch with work=False;
        # default: above M = 22 the moment / solve pair (ABI 4), within rounding of both
        for work in (False, None):
            W.fill_(7.0)
            st.fill_(-5)
            ops.enhance_subset(x, u, M, 1e4, n, W, elem_ids=ids, global_domain=gd, status=st, work=work)
            torch.cuda.synchronize()
            Wh, sth = W.cpu().numpy(), st.cpu().numpy()
            assert np.all(Wh[~touched] == 7.0) and np.all(sth[~touched] == -5)
            assert np.all(Wh[touched][:, M:] == 7.0) and np.all(sth[touched] == 0)
            if work is False:
                assert np.array_equal(Wh[touched][:, :M], Wfull.cpu().numpy()[touched])
            assert orc.rel_l2_coef(Wh[touched][:, :M], Wdef.cpu().numpy()[touched]).max() <= 1e-12
            assert orc.rel_l2_coef(Wh[touched][:, :M], Wfull.cpu().numpy()[touched]).max() <= 1e-12


def test_subset_argument_errors(dev):
    import torch
    from hybrid_fem_lssvr_amd import ops, _capi
    _, nodes, values = _mesh(20, 14)
    x = torch.as_tensor(nodes, device=dev)
    u = torch.as_tensor(values, device=dev)
    W = torch.zeros((20, 9), dtype=torch.float64, device=dev)
    with pytest.raises(ValueError):
        ops.enhance_subset(x, u, 12, 1e4, 16, W, global_domain=(nodes[0], nodes[-1]))   # ldw < M
    W = torch.zeros((20, 12), dtype=torch.float64, device=dev)
    with pytest.raises(_capi.LssvrHipError):
        ops.enhance_subset(x, u, 12, 1e4, 5, W, global_domain=(nodes[0], nodes[-1]))    # n < M-2


def test_subset_out_of_range_ids_touch_nothing(dev):
    """An id outside [0, ne_mesh) must not become an out-of-bounds access: the element is
    skipped (no load, no store) and counted in fail_count; valid ids of the same launch are
    enhanced as usual (lane kernel M = 9; moment + parity-split solve kernels M = 33; the f64-MFMA kernel's
    range check is covered by test_large_degree_subset_runs_the_two_kernel_path)."""
    import torch
    from hybrid_fem_lssvr_amd import ops
    _, nodes, values = _mesh(90, 15)
    x = torch.as_tensor(nodes, device=dev)
    u = torch.as_tensor(values, device=dev)
    gd = (nodes[0], nodes[-1])
    for M, n in [(9, 16), (33, 64)]:
        # guard rows around W and status catch a stray write next to the arrays
        Wbig = torch.full((92, M), 7.0, dtype=torch.float64, device=dev)
        stbig = torch.full((92,), -5, dtype=torch.int32, device=dev)
        W, st = Wbig[1:91], stbig[1:91]
        fc = torch.zeros(1, dtype=torch.int32, device=dev)
        ids = torch.as_tensor(np.array([5, -1, 90, 17, 1 << 40, -(1 << 50), 89], dtype=np.int64), device=dev)
        ops.enhance_subset(x, u, M, 1e4, n, W, elem_ids=ids, global_domain=gd, status=st, fail_count=fc)
        torch.cuda.synchronize()
        assert int(fc.item()) == 4
        Wh, sth = Wbig.cpu().numpy(), stbig.cpu().numpy()
        good = np.zeros(92, dtype=bool)
        good[[6, 18, 90]] = True
        assert np.all(Wh[~good] == 7.0) and np.all(sth[~good] == -5)
        assert np.all(sth[good] == 0)
        # (the default subset launch above M = 22 is the moment / solve pair, like the default full launch)
        Wfull, _ = ops.enhance(x, u, M, 1e4, n, global_domain=gd)
        assert np.array_equal(Wh[good], Wfull.cpu().numpy()[[5, 17, 89]])


@pytest.mark.parametrize("M,n", [(33, 64), (33, 40), (33, 31), (24, 44), (27, 27)])
def test_large_degree_subset_runs_the_two_kernel_path(dev, M, n):
    """lssvr_enhance_subset_ws (ABI 4): above M = 22 a subset group runs as moments + solve kernels --
    the parity-split solve (n >= 2 (M-2)), the full four-systems-per-wave solve, and the refined
    near-square regime -- with rows / status / gamma by mesh index and the workspace by position.
    Against the single f64-MFMA kernel (work=False), the float64 oracle of each element's own problem
    and, in the refined regime, the 60-digit minimiser (where the MFMA kernel is 1e-6 .. 1e-10 off)."""
    import torch
    from hybrid_fem_lssvr_amd import ops
    from oracle import closed_form_mp as cf
    rng, nodes, values = _mesh(150, 40 + M + n)
    ne = 150
    gd = (nodes[0], nodes[-1])
    x, u = torch.as_tensor(nodes, device=dev), torch.as_tensor(values, device=dev)
    ids_h = np.concatenate([rng.permutation(ne)[:61], [ne + 5, -1]]).astype(np.int64)     # two ids out of range
    ids = torch.as_tensor(ids_h, device=dev)
    gam_h = 10.0 ** rng.uniform(3, 5, ne)
    gam = torch.as_tensor(gam_h, device=dev)
    out = {}
    for tag, work in (("split", None), ("mfma", False)):
        W = torch.full((ne, M + 2), 7.0, dtype=torch.float64, device=dev)
        st = torch.full((ne,), -5, dtype=torch.int32, device=dev)
        cnt = torch.zeros(1, dtype=torch.int32, device=dev)
        ops.enhance_subset(x, u, M, 1.0, n, W, elem_ids=ids, gamma_values=gam, global_domain=gd, status=st,
                           fail_count=cnt, work=work)
        torch.cuda.synchronize()
        out[tag] = (W.cpu().numpy(), st.cpu().numpy(), int(cnt.item()))
    good = ids_h[:-2]
    for tag in out:
        W, st, cnt = out[tag]
        assert cnt == 2                                             # the two out-of-range ids, nothing else
        assert np.all(st[good] == 0)
        rest = np.setdiff1d(np.arange(ne), good)
        assert np.all(W[rest] == 7.0) and np.all(st[rest] == -5)    # untouched rows stay untouched
        assert np.all(W[good][:, M:] == 7.0)                        # ... and so do the pad columns (ldw = M + 2)
    near_square = n - (M - 2) <= 14
    agree = orc.rel_l2_coef(out["split"][0][good][:, :M], out["mfma"][0][good][:, :M]).max()
    assert agree <= (1e-4 if near_square else 1e-12), agree
    sel = good[:4]
    if near_square and cf.HAVE_MP:
        tr = np.array([cf.solve_truth(orc.element_system(
            nodes[i], nodes[i + 1], *orc.boundary_values(int(i), ne, nodes[i], nodes[i + 1], values[i], values[i + 1], gd),
            M, gam_h[i], n)) for i in sel])
        assert orc.rel_l2_coef(out["split"][0][sel][:, :M], tr).max() <= 2e-12
    else:
        for i in sel:
            wo = _oracle_element(nodes, values, int(i), M, gam_h[i], n, gd)
            assert orc.rel_l2_coef(out["split"][0][i, :M][None], wo[None]).max() <= 1e-11
