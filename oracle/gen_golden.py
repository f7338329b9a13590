"""Generate ``tests/golden/*.npz`` from the reference itself.

TEST INFRASTRUCTURE ONLY.  Runs in the BUILD CONTAINER only (it reads
``/root/reference``, which does not exist on the GPU box); the fixtures it
writes are data -- inputs and expected outputs -- and are committed.

How the reference is imported: the module does
``from skfem import *`` at import time (Dual.py:5-6) and scikit-fem is not
installed in this image, but ``lssvr_primal``, ``solve_lssvr_subproblems`` and
``evaluate_solution`` only use numpy/scipy.  Two empty placeholder modules named
``skfem`` / ``skfem.helpers`` are registered so the import statement succeeds;
nothing from them is ever called (``solve_fem``, the only skfem user, is never
run -- nodal values are assigned to ``solver.fem_nodes/fem_values`` directly).
The reference hard-codes 12 collocation points (Dual.py:40); other counts are
obtained, without editing the reference, by giving the loaded module a forwarding
``np`` whose ``linspace(a, b, 12)`` returns ``linspace(a, b, n)``.

Every fixture stores, per element: the inputs, the reference's SLSQP output
(seeded, Dual.py:81 draws from numpy's global RNG), and the extended-precision
closed-form minimiser of the same QP (``oracle/closed_form_mp.py``).

Nodal values come from ``oracle.lssvr_oracle.fem_p1_solve_golden_v1`` -- a FROZEN copy of the P1
stand-in as it was when the committed fixtures were written -- so that a regeneration
reproduces their inputs bit for bit (``--verify`` checks that, and reruns the reference on
the stored inputs; tests/test_oracle_golden.py does the same on every CPU run in the build
container).

Usage:  python oracle/gen_golden.py [--only G1,G2,...] | --verify
"""
from __future__ import annotations

import argparse
import importlib.util
import io
import contextlib
import os
import sys
import time
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import lssvr_oracle as orc          # noqa: E402
from oracle import closed_form_mp as cf         # noqa: E402

REF_FILE = "/root/reference/1D-Possion/Hybrid-FEM-LSSVR-Dual.py"
OUT_DIR = os.path.join(ROOT, "tests", "golden")


def load_reference():
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.dont_write_bytecode = True
    if "skfem" not in sys.modules:
        pk = types.ModuleType("skfem")
        pk.__all__ = []
        hp = types.ModuleType("skfem.helpers")
        hp.dot = None
        hp.grad = None
        pk.helpers = hp
        sys.modules["skfem"] = pk
        sys.modules["skfem.helpers"] = hp
    spec = importlib.util.spec_from_file_location("ref_dual", REF_FILE)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class _NpForward:
    """Forwards everything to numpy; only ``linspace(a, b, 12)`` is re-counted."""

    def __init__(self, n):
        self._n = n

    def __getattr__(self, name):
        return getattr(np, name)

    def linspace(self, a, b, num=50, *args, **kw):
        if num == 12:
            num = self._n
        return np.linspace(a, b, num, *args, **kw)


@contextlib.contextmanager
def colloc_count(mod, n):
    if n == 12:
        yield
        return
    old = mod.np
    mod.np = _NpForward(n)
    try:
        yield
    finally:
        mod.np = old


def run_reference(mod, nodes, values, M, gamma, n, elements, global_domain, seed0=1234):
    """Reference coefficients for ``elements`` through the reference's own loop body
    (Dual.py:143-162), seeding numpy's global RNG per element."""
    ne = len(nodes) - 1
    coefs, msgs, secs = [], [], []
    with colloc_count(mod, n):
        for i in elements:
            np.random.seed(seed0 + int(i) % 100000)
            buf = io.StringIO()
            t0 = time.perf_counter()
            with contextlib.redirect_stdout(buf):
                fn = mod.lssvr_primal(
                    mod.poisson_rhs, [nodes[i], nodes[i + 1]], values[i], values[i + 1], M, gamma,
                    is_left_boundary=(i == 0), is_right_boundary=(i == ne - 1),
                    global_domain_range=global_domain)
            secs.append(time.perf_counter() - t0)
            coefs.append(np.array(fn.coef, dtype=np.float64))
            assert tuple(fn.domain) == (nodes[i], nodes[i + 1])
            msgs.append(buf.getvalue().strip())
    return np.array(coefs), msgs, np.array(secs)


def make_case(mod, name, lo, hi, ne, M, gamma, n, elements=None, truth=True):
    nodes = np.linspace(lo, hi, ne + 1)                      # Dual.py:112
    values = orc.fem_p1_solve_golden_v1(nodes)               # stands in for Dual.py:127-135 (FROZEN)
    if elements is None:
        elements = np.arange(ne)
    elements = np.asarray(elements, dtype=np.int64)
    gd = (lo, hi)
    t0 = time.perf_counter()
    ref, msgs, secs = run_reference(mod, nodes, values, M, gamma, n, elements, gd)
    tru = cf.truth_all(nodes, values, M, gamma, n, orc.poisson_rhs, gd, elements) if truth else None
    err = orc.rel_l2_coef(ref, tru) if truth else None
    print(f"{name}: ne={ne} M={M} n={n} gamma={gamma:g} elems={len(elements)} "
          f"ref {secs.mean()*1e3:.0f} ms/el  ref-vs-truth relL2 max {np.max(err):.2e} "
          f"warn={sum(bool(m) for m in msgs)}  ({time.perf_counter()-t0:.1f}s)")
    np.savez(os.path.join(OUT_DIR, name + ".npz"),
             lo=np.float64(lo), hi=np.float64(hi), ne=np.int64(ne), M=np.int64(M),
             gamma=np.float64(gamma), n=np.int64(n), elements=elements,
             nodes_sel=np.stack([nodes[elements], nodes[elements + 1]], 1),
             values_sel=np.stack([values[elements], values[elements + 1]], 1),
             coef_ref=ref, coef_truth=tru, ref_seconds=secs,
             ref_warned=np.array([bool(m) for m in msgs]))
    return nodes, values, ref


def make_eval_case(mod, name):
    """G7: ``evaluate_solution`` (Dual.py:176-203) on the demo grid plus node-coincident,
    out-of-range and NaN points; expected element indices from the literal scan."""
    lo, hi, ne, M, gamma, n = -1.0, 1.0, 24, 8, 1e4, 12
    nodes = np.linspace(lo, hi, ne + 1)
    values = orc.fem_p1_solve_golden_v1(nodes)
    solver = mod.FEMLSSVRPrimalSolver(ne + 1, lssvr_M=M, lssvr_gamma=gamma, global_domain=(lo, hi))
    solver.fem_nodes = nodes
    solver.fem_values = values
    np.random.seed(777)
    with contextlib.redirect_stdout(io.StringIO()):
        solver.solve_lssvr_subproblems()                      # Dual.py:139
    W = np.array([f.coef for f in solver.lssvr_functions])
    xq = np.concatenate([
        np.linspace(-1, 1, 201),                              # Dual.py:209
        nodes,                                                # node-coincident
        np.array([-1.5, -1.0000001, 1.0000001, 1.75, 0.0, np.nextafter(nodes[3], 2.0),
                  np.nextafter(nodes[3], -2.0)]),
    ])
    u = solver.evaluate_solution(xq)                          # Dual.py:176
    elem = orc.locate_elements_scan(nodes, xq)
    np.savez(os.path.join(OUT_DIR, name + ".npz"), nodes=nodes, values=values, W=W, xq=xq,
             u_ref=u, elem=elem, M=np.int64(M), gamma=np.float64(gamma), n=np.int64(n))
    uo, eo = orc.evaluate_solution(nodes, W, xq)
    print(f"{name}: P={len(xq)} oracle-vs-ref max|du|={np.max(np.abs(uo-u)):.2e} "
          f"elem equal={np.array_equal(eo, elem)}")


def rerun_on_stored_inputs(mod, g, limit=None):
    """The reference on a fixture's STORED inputs (``nodes_sel``, ``values_sel``; same per-element
    seeds as :func:`run_reference`): returns coefficients to compare with ``coef_ref`` -- bit for
    bit on the numpy / scipy versions that wrote the fixture.  ``limit``: first k elements only."""
    lo, hi, ne = float(g["lo"]), float(g["hi"]), int(g["ne"])
    M, n, gamma = int(g["M"]), int(g["n"]), float(g["gamma"])
    out = []
    with colloc_count(mod, n):
        for k, i in enumerate(g["elements"][:limit]):
            np.random.seed(1234 + int(i) % 100000)
            with contextlib.redirect_stdout(io.StringIO()):
                fn = mod.lssvr_primal(
                    mod.poisson_rhs, [g["nodes_sel"][k, 0], g["nodes_sel"][k, 1]],
                    g["values_sel"][k, 0], g["values_sel"][k, 1], M, gamma,
                    is_left_boundary=(int(i) == 0), is_right_boundary=(int(i) == ne - 1),
                    global_domain_range=(lo, hi))
            out.append(np.array(fn.coef, dtype=np.float64))
    return np.array(out)


def rerun_eval_case(mod, g):
    """G7 on its stored inputs: the reference's loop (seed 777) and ``evaluate_solution``."""
    nodes, values = g["nodes"], g["values"]
    solver = mod.FEMLSSVRPrimalSolver(len(nodes), lssvr_M=int(g["M"]), lssvr_gamma=float(g["gamma"]),
                                      global_domain=(float(nodes[0]), float(nodes[-1])))
    solver.fem_nodes = nodes
    solver.fem_values = values
    np.random.seed(777)
    with colloc_count(mod, int(g["n"])), contextlib.redirect_stdout(io.StringIO()):
        solver.solve_lssvr_subproblems()
    W = np.array([f.coef for f in solver.lssvr_functions])
    return W, solver.evaluate_solution(g["xq"])


def verify():
    """Every committed fixture against (1) the frozen P1 stand-in (inputs) and (2) the reference
    rerun on the stored inputs (outputs), bit for bit.  The 1e7-element mesh of G6b takes ~30 s of
    Thomas elimination in Python; ``tests/test_oracle_golden.py`` runs the same checks without it."""
    mod = load_reference()
    bad = 0
    for fn in sorted(os.listdir(OUT_DIR)):
        if not fn.endswith(".npz"):
            continue
        g = dict(np.load(os.path.join(OUT_DIR, fn), allow_pickle=False))
        if "elements" not in g:
            W, u = rerun_eval_case(mod, g)
            ok_in = np.array_equal(orc.fem_p1_solve_golden_v1(g["nodes"]), g["values"])
            ok_out = np.array_equal(W, g["W"]) and np.array_equal(u, g["u_ref"], equal_nan=True)
        else:
            nodes = np.linspace(float(g["lo"]), float(g["hi"]), int(g["ne"]) + 1)
            v = orc.fem_p1_solve_golden_v1(nodes)
            e = g["elements"]
            ok_in = (np.array_equal(nodes[e], g["nodes_sel"][:, 0]) and np.array_equal(v[e], g["values_sel"][:, 0])
                     and np.array_equal(v[e + 1], g["values_sel"][:, 1]))
            ok_out = np.array_equal(rerun_on_stored_inputs(mod, g), g["coef_ref"])
        print(f"{fn}: inputs regenerate bit-equal: {ok_in}; reference rerun bit-equal: {ok_out}")
        bad += (not ok_in) + (not ok_out)
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--verify", action="store_true",
                    help="check the committed fixtures instead of writing them")
    args = ap.parse_args()
    if args.verify:
        sys.exit(1 if verify() else 0)
    only = set(filter(None, args.only.split(",")))
    os.makedirs(OUT_DIR, exist_ok=True)
    mod = load_reference()

    def want(k):
        return not only or k in only

    if want("G1"):   # BASELINE config 1
        make_case(mod, "G1_c1_ne8_M5_n5", -1.0, 1.0, 8, 5, 1e4, 5)
    if want("G2"):   # reference __main__ (Dual.py:208-212)
        make_case(mod, "G2_default_ne24_M8_n12", -1.0, 1.0, 24, 8, 1e4, 12)
    if want("G3"):   # deg 8 / 16 pts
        make_case(mod, "G3_ne24_M9_n16", -1.0, 1.0, 24, 9, 1e4, 16)
    if want("G4"):   # [-1,1], 4096 el, sampled
        make_case(mod, "G4_ne4096_M9_n16", -1.0, 1.0, 4096, 9, 1e4, 16,
                  elements=[0, 1, 511, 1024, 2047, 2048, 3333, 4095])
    if want("G5"):   # deg 32 / 64 pts
        make_case(mod, "G5_ne24_M33_n64", -1.0, 1.0, 24, 33, 1e4, 64, elements=[0, 7, 23])
    if want("G6"):   # wide domain, h = 1/12 (SURVEY.md finding 5)
        make_case(mod, "G6a_wide_ne100008_M9_n16", -4167.0, 4167.0, 100008, 9, 1e4, 16,
                  elements=[0, 1, 12345, 50003, 50004, 77777, 100006, 100007])
        make_case(mod, "G6b_wide_ne10000008_M9_n16", -416667.0, 416667.0, 10000008, 9, 1e4, 16,
                  elements=[0, 1, 1234567, 5000003, 5000004, 7777777, 10000006, 10000007])
    if want("G7"):
        make_eval_case(mod, "G7_eval_default")
    if want("G8"):   # class defaults (Dual.py:101)
        make_case(mod, "G8_classdefaults_ne4_M12_n12", -1.0, 1.0, 4, 12, 1e6, 12)


if __name__ == "__main__":
    main()
