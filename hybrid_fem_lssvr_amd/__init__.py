"""MI355X-native per-element LSSVR enhancement for hybrid FEM-LSSVR.

Drop-in for the solve-then-enhance path of maryambabaei/hybrid-FEM-LSSVR
(``1D-Possion/Hybrid-FEM-LSSVR-Dual.py``): same class / function names and
argument meaning, with the per-element SLSQP loop replaced by hand-written
gfx950 kernels behind a C ABI (``include/lssvr_hip.h``).
"""
from . import _capi, ops  # noqa: F401

__all__ = ["ops"]
__version__ = "0.1.0"
