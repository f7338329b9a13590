"""Prints the measured forward differences that tests/test_gpu_fem_eval.py pins (tridiagonal
solver vs LAPACK, config-5 enhancement vs plain P1) -- run on the GPU box, copy into the tests."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import lssvr_oracle as orc
from hybrid_fem_lssvr_amd import ops

dev = torch.device("cuda:0")
t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
for ne in [1, 2, 3, 24, 511, 512, 513, 514, 1025, 16385, 100000, 1234567]:
    nodes = np.linspace(-1, 1, ne + 1)
    kd, fl, fr = orc.p1_assemble_local(nodes)
    diag, off, load = orc.p1_scatter(kd, fl, fr)
    u = ops.tridiag_dirichlet_solve(t(diag), t(off), t(load), 0.25, -0.5).cpu().numpy()
    ref = orc.banded_dirichlet(diag, off, load, 0.25, -0.5)
    print("tridiag ne=%8d max|u-ref| = %.3e  scale %.3e" % (ne, np.max(np.abs(u - ref)), np.max(np.abs(ref))), flush=True)
c, phi = orc.varcoef_params()
a, da, f = orc.varcoef_functions(c, phi)
for ne, M, n in ((2000, 9, 16), (300, 20, 32), (100, 26, 40)):
    nodes = np.linspace(-1, 1, ne + 1)
    values = orc.fem_p1_solve(nodes, rhs=f, coef_a=a)
    x = t(nodes)
    xc = ops.colloc_points(x, n).cpu().numpy()
    W, st = ops.enhance_varcoef(x, t(values), M, 1e4, n, t(a(xc)), t(da(xc)), t(f(xc)), global_domain=(-1.0, 1.0))
    W = W.cpu().numpy()
    xq = np.linspace(-1, 1, 4001)
    uq, _ = orc.evaluate_solution_vec(nodes, W, xq)
    p1 = np.interp(xq, nodes, values)
    ex = np.sin(np.pi * xq)
    print("c5 ne=%d M=%d: |hybrid-ex| %.3e  |p1-ex| %.3e  ratio %.2f" % (ne, M, np.linalg.norm(uq - ex), np.linalg.norm(p1 - ex),
          np.linalg.norm(p1 - ex) / np.linalg.norm(uq - ex)), flush=True)
