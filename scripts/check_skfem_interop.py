#!/usr/bin/env python3
"""Checks SURVEY.md Appendix C -- the scikit-fem 11 behaviour the P1 half of the path was written
from memory against -- whenever scikit-fem IS importable.  It is not in the build image (nothing is
ever installed from here); there the script says so and exits 0.

With the package present it compares, on the reference's own demo mesh (Dual.py:206-213):
  * MeshLine(p).p / .t shapes, dtypes and connectivity with hybrid_fem_lssvr_amd.mesh.LineMesh;
  * the assembled stiffness / load of Dual.py:117-128 with the oracle's element-local P1 assembly
    (2-point Gauss) -- including the sign convention;
  * basis.get_dofs() == the two end nodes, and the nodal values of enforce + solve with the
    oracle's P1 solve.
Needs no GPU (CPU oracle only)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    try:
        import skfem
        from skfem import Basis, BilinearForm, ElementLineP1, LinearForm, MeshLine, enforce, solve
        from skfem.helpers import dot, grad
    except Exception as exc:
        print(f"scikit-fem is not importable here ({type(exc).__name__}: {exc}); Appendix C stays unverified. "
              "Nothing to do.")
        return 0
    import numpy as np
    from oracle import lssvr_oracle as orc
    from hybrid_fem_lssvr_amd.mesh import LineMesh

    ok = True

    def check(name, cond, detail=""):
        nonlocal ok
        print(("PASS " if cond else "FAIL ") + name + (f"  [{detail}]" if detail else ""))
        ok = ok and bool(cond)

    nodes = np.linspace(-1.0, 1.0, 25)
    m = MeshLine(nodes)
    mine = LineMesh(nodes)
    check("MeshLine.p shape (1, n)", m.p.shape == mine.p.shape == (1, 25))
    check("MeshLine.t == [[i], [i+1]]", np.array_equal(np.sort(m.t, axis=0), mine.t), f"dtype {m.t.dtype}")
    check("MeshLine.p[0] are the nodes", np.array_equal(m.p[0], nodes))
    basis = Basis(m, ElementLineP1())

    @BilinearForm
    def laplace(u, v, _):
        return -dot(grad(u), grad(v))

    @LinearForm
    def load(v, w):
        return -(np.pi ** 2) * np.sin(np.pi * w.x[0]) * v

    A = laplace.assemble(basis)
    b = load.assemble(basis)
    kd, fl, fr = orc.p1_assemble_local(nodes)
    diag, off, ld = orc.p1_scatter(kd, fl, fr)
    Ad = A.toarray()
    check("stiffness = -(oracle bands)  (both reference forms are negated)",
          np.allclose(np.diag(Ad), -diag, rtol=1e-13) and np.allclose(np.diag(Ad, 1), -off, rtol=1e-13),
          f"max |diag + oracle| = {np.max(np.abs(np.diag(Ad) + diag)):.2e}")
    check("load = -(oracle load): default quadrature of Basis(P1) is the 2-point Gauss rule",
          np.allclose(b, -ld, rtol=1e-12, atol=1e-15), f"max diff {np.max(np.abs(b + ld)):.2e}")
    D = basis.get_dofs()
    dofs = np.sort(np.asarray(D.flatten() if hasattr(D, "flatten") else D))
    check("get_dofs() = the two end nodes", np.array_equal(dofs, [0, 24]), str(dofs))
    u = solve(*enforce(A, b, D=D))
    u_or = orc.fem_p1_solve(nodes)
    check("enforce + solve == oracle P1 solve to 1e-13", np.max(np.abs(u - u_or)) < 1e-13,
          f"max diff {np.max(np.abs(u - u_or)):.2e}")
    check("max nodal error 3.274e-6 (SURVEY.md Appendix B)",
          abs(np.max(np.abs(u - np.sin(np.pi * nodes))) - 3.274e-6) < 2e-9)
    print("scikit-fem", getattr(skfem, "__version__", "?"), "->", "all assumptions hold" if ok else "MISMATCH")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
