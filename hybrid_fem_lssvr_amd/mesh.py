"""scikit-fem-shaped line mesh / P1 basis containers and adapters.

The reference builds ``MeshLine(np.linspace(a, b, n))`` and
``Basis(m, ElementLineP1())`` (Dual.py:112-114).  scikit-fem may or may not be
installed next to this package, so the facade accepts anything that *looks*
like those objects (duck typing) and carries its own minimal stand-ins with the
same attribute shapes:

* ``mesh.p``  float64, shape (1, n_nodes)  -- node coordinates (``m.p[0]``, Dual.py:134)
* ``mesh.t``  int32,   shape (2, n_elems)  -- connectivity ``[[0..n-2],[1..n-1]]``
* ``basis.mesh``, ``basis.N`` (dofs), ``basis.get_dofs()`` -> boundary dofs (Dual.py:129)

Element / node numbering is part of the parity contract: element ``i`` has nodes
``(i, i+1)`` (Dual.py:143-147).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass
class LineMesh:
    """Stand-in for ``skfem.MeshLine``: ascending nodes, chain connectivity."""

    p: np.ndarray
    t: np.ndarray

    @classmethod
    def from_nodes(cls, nodes):
        nodes = np.ascontiguousarray(np.asarray(nodes, dtype=np.float64).reshape(-1))
        if nodes.size < 2:
            raise ValueError("a line mesh needs at least two nodes")
        if not np.all(np.diff(nodes) > 0):
            raise ValueError("nodes must be strictly ascending")
        n = nodes.size
        t = np.vstack([np.arange(0, n - 1), np.arange(1, n)]).astype(np.int32)
        return cls(p=nodes.reshape(1, -1), t=t)

    @property
    def nodes(self):
        return self.p[0]

    @property
    def nelements(self):
        return self.t.shape[1]

    @property
    def nvertices(self):
        return self.p.shape[1]


@dataclass
class P1Basis:
    """Stand-in for ``skfem.Basis(mesh, ElementLineP1())`` (Dual.py:113-114)."""

    mesh: LineMesh

    @property
    def N(self):
        return self.mesh.nvertices

    @property
    def nelems(self):
        return self.mesh.nelements

    def get_dofs(self):
        """All boundary dofs, as ``basis.get_dofs()`` without arguments (Dual.py:129)."""
        return np.array([0, self.N - 1], dtype=np.int64)

    def interpolator(self, u):
        """Piecewise-linear interpolant; at the nodes it returns ``u`` (Dual.py:133-135)."""
        nodes = self.mesh.nodes
        u = np.asarray(u, dtype=np.float64)

        def interp(x):
            x = np.asarray(x, dtype=np.float64).reshape(-1)
            return np.interp(x, nodes, u)

        return interp


def as_line_mesh(obj):
    """Accept a LineMesh, a skfem-like mesh (``.p``, ``.t``), a skfem-like basis
    (``.mesh``), or a plain 1-D array of node coordinates.  Returns a LineMesh whose
    element ``i`` is ``(i, i+1)``; anything else is rejected rather than silently
    renumbered (indices are part of the parity contract)."""
    if isinstance(obj, LineMesh):
        return obj
    if hasattr(obj, "mesh") and hasattr(obj.mesh, "p"):
        obj = obj.mesh
    if hasattr(obj, "p"):
        p = np.asarray(obj.p, dtype=np.float64)
        if p.ndim != 2 or p.shape[0] != 1:
            raise ValueError(f"expected a 1-D mesh with p of shape (1, n), got {p.shape}")
        nodes = np.ascontiguousarray(p[0])
        if hasattr(obj, "t"):
            t = np.asarray(obj.t)
            n = nodes.size
            chain = np.vstack([np.arange(0, n - 1), np.arange(1, n)])
            if t.shape != chain.shape or not np.array_equal(t, chain):
                raise ValueError("mesh.t is not the chain connectivity [[0..n-2],[1..n-1]]; "
                                 "renumber the mesh first (element i must be nodes (i, i+1))")
        return LineMesh.from_nodes(nodes)
    return LineMesh.from_nodes(obj)
