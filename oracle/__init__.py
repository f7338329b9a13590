"""CPU oracle for the per-element LSSVR enhancement path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and there only as the checker / the reported CPU baseline.
The product path (``hybrid_fem_lssvr_amd``) never imports this package and
fails loudly when its HIP library is missing.

Parity pinning: the reference (maryambabaei/hybrid-FEM-LSSVR) ships no tests,
golden vectors or fixtures.  The oracle is therefore pinned
against outputs of the reference itself, generated in the build container by
``oracle/gen_golden.py`` (which imports ``lssvr_primal`` /
``FEMLSSVRPrimalSolver`` from the reference's own file) and committed as
``tests/golden/*.npz``.
"""
