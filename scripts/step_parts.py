"""The fused step, the enhancement alone and the P1 assembly alone, K = 200 launches each captured in a hipGraph
and replayed (BASELINE config 2): what the assembly adds to a step, free of host launch cost.
MI355X, round 3: fused step 7.7-7.8 us, enhancement only 7.45-7.49 us, assembly only 3.13 us (= the launch floor)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hybrid_fem_lssvr_amd import ops
dev = "cuda:0"
ne, M, n, K = 100008, 9, 16, 200
half = ne / 24.0
nodes = np.arange(ne + 1, dtype=np.float64) * (2 * half / ne) - half
x = torch.as_tensor(nodes, device=dev); u = torch.sin(np.pi * x)
plan = ops.StepPlan(x, u, M, 1e4, n, global_domain=(-half, half))
W = torch.empty((ne, M), dtype=torch.float64, device=dev); st = torch.empty(ne, dtype=torch.int32, device=dev)
def cap(fn):
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side): fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(K): fn()
    g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(9):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3 / K)
    return sorted(ts)
for r in range(2):
    a = cap(plan.launch)
    b = cap(lambda: ops.enhance(x, u, M, 1e4, n, global_domain=(-half, half), out=W, status=st))
    c = cap(lambda: ops.p1_assemble(x, nquad=2)) if hasattr(ops, "p1_assemble") else None
    print("graph, fused step       : median %.2f us min %.2f" % (a[4], a[0]))
    print("graph, enhancement only : median %.2f us min %.2f" % (b[4], b[0]))
    if c: print("graph, assembly only    : median %.2f us min %.2f" % (c[4], c[0]))
