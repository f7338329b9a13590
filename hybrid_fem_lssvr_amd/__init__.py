"""MI355X-native per-element LSSVR enhancement for hybrid FEM-LSSVR.

Drop-in for the solve-then-enhance path of maryambabaei/hybrid-FEM-LSSVR
(``1D-Possion/Hybrid-FEM-LSSVR-Dual.py``): same class / function names and
argument meaning, with the per-element SLSQP loop replaced by hand-written
gfx950 kernels behind a C ABI (``include/lssvr_hip.h``).
"""
from . import _capi, ops  # noqa: F401
from .mesh import LineMesh, P1Basis, as_line_mesh  # noqa: F401
from .solver import (  # noqa: F401
    EnhancedSolution,
    FEMLSSVRPrimalSolver,
    SinRHS,
    enhance_elements,
    enhance_elements_hetero,
    lssvr_primal,
    main_boundary_condition_left,
    main_boundary_condition_right,
    poisson_rhs,
    true_solution,
)
from .distributed import (  # noqa: F401
    ShardPlan,
    allgather_rows,
    enhance_sharded,
    solve_fem_sharded,
    solve_sharded,
)

__version__ = "0.1.0"
