"""CPU: host-side logic of the facade -- mesh adapters, shard plan, RHS objects."""
import numpy as np
import pytest


def test_line_mesh_shapes_and_adapters():
    from hybrid_fem_lssvr_amd import LineMesh, P1Basis, as_line_mesh
    m = LineMesh.from_nodes(np.linspace(-1, 1, 25))
    assert m.p.shape == (1, 25) and m.t.shape == (2, 24) and m.t.dtype == np.int32
    assert np.array_equal(m.t[:, 5], [5, 6])                 # element i <-> nodes (i, i+1)
    b = P1Basis(m)
    assert b.N == 25 and b.nelems == 24 and np.array_equal(b.get_dofs(), [0, 24])
    assert np.allclose(b.interpolator(np.arange(25.0))(m.nodes), np.arange(25.0))

    class SkfemLikeMesh:                                      # duck-typed scikit-fem MeshLine
        p = np.linspace(0, 1, 5).reshape(1, -1)
        t = np.vstack([np.arange(4), np.arange(1, 5)])

    class SkfemLikeBasis:
        mesh = SkfemLikeMesh()

    assert np.array_equal(as_line_mesh(SkfemLikeMesh()).nodes, np.linspace(0, 1, 5))
    assert np.array_equal(as_line_mesh(SkfemLikeBasis()).nodes, np.linspace(0, 1, 5))
    assert as_line_mesh([0.0, 0.5, 2.0]).nelements == 2

    class Shuffled(SkfemLikeMesh):
        t = np.vstack([np.arange(4)[::-1], np.arange(1, 5)[::-1]])

    with pytest.raises(ValueError, match="chain connectivity"):
        as_line_mesh(Shuffled())
    with pytest.raises(ValueError, match="ascending"):
        LineMesh.from_nodes([0.0, 1.0, 0.5])
    with pytest.raises(ValueError):
        LineMesh.from_nodes([0.0])


def test_shard_plan_partitions_exactly():
    from hybrid_fem_lssvr_amd import ShardPlan
    for ne, world in ((10, 3), (10000000, 8), (7, 8), (100008, 4), (1, 1), (0, 2)):
        plan = ShardPlan(ne, world)
        b = [plan.bounds(r) for r in range(world)]
        assert b[0][0] == 0 and b[-1][1] == ne
        assert all(b[r][1] == b[r + 1][0] for r in range(world - 1))
        sizes = [plan.size(r) for r in range(world)]
        assert max(sizes) - min(sizes) <= 1 and max(sizes) == plan.max_size
        for e in {0, ne // 2, max(ne - 1, 0)} if ne else ():
            r = plan.owner_of_element(e)
            assert b[r][0] <= e < b[r][1]
        for r in range(world):
            sl = plan.node_slice(r)
            assert sl.stop - sl.start == sizes[r] + 1


def test_rhs_objects_mirror_reference_arithmetic():
    import hybrid_fem_lssvr_amd as pkg
    x = np.linspace(-3, 3, 101)
    assert np.array_equal(pkg.poisson_rhs(x), np.pi ** 2 * np.sin(np.pi * x))   # Dual.py:12
    assert np.array_equal(pkg.true_solution(x), np.sin(np.pi * x))               # Dual.py:9
    assert pkg.main_boundary_condition_left(-1) == 0.0 and pkg.main_boundary_condition_right(1) == 0.0
    assert pkg.poisson_rhs.amp == np.pi ** 2 and pkg.poisson_rhs.omega == np.pi


def test_solver_surface_and_loud_failure_without_gpu():
    import torch
    import hybrid_fem_lssvr_amd as pkg
    s = pkg.FEMLSSVRPrimalSolver()
    assert (s.num_fem_nodes, s.lssvr_M, s.lssvr_gamma, s.global_domain) == (5, 12, 1e6, (-1, 1))
    assert s.fem_nodes is None and s.fem_values is None and s.lssvr_functions == []
    for name in ("solve_fem", "solve_lssvr_subproblems", "solve", "evaluate_solution"):
        assert callable(getattr(s, name))
    with pytest.raises(RuntimeError, match="solve_fem"):
        s.solve_lssvr_subproblems()
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            s.solve()
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            pkg.lssvr_primal(pkg.poisson_rhs, [-1, 0], 0.0, 0.0, 5, 1e4)


def test_bench_cpu_baseline_sample_and_budget():
    """bench.py's CPU baseline (the reference's per-element SLSQP loop, oracle restatement): one job per element
    carrying only that element's data, finished elements / wall time, a central sample for degree 32, and jobs still
    running at the budget are abandoned and NOT counted."""
    import importlib
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    bench = importlib.import_module("bench")        # (importable by name: its pool pickles the worker function)
    nodes, values, gd = bench.wide_mesh(2400)
    assert abs((nodes[1] - nodes[0]) - 1.0 / 12.0) < 1e-12 and values[0] == 0.0 and values[-1] == 0.0
    r = bench.cpu_baseline(nodes, values, gd, per_core=1)
    assert r["kind"] == "port" and r["cores"] >= 1 and r["value"] > 0 and r["wall_s"] < 60
    assert "evenly spaced" in r["sample"] and "converged" in r["sample"]
    # an impossible budget: nothing finishes, nothing is counted, the line says so
    r0 = bench.cpu_baseline(nodes, values, gd, per_core=1, M=33, n=64, central=True, budget_s=0.05)
    assert r0["value"] == 0.0 and "abandoned" in r0["sample"] and "nearest the origin" in r0["sample"]
