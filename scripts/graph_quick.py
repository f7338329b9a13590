"""K steps of the hot path (BASELINE config 2: 100 008 elements, degree 8, 16 points) launched eagerly from Python
against the same K launches captured once in a hipGraph and replayed: per-step time between HIP events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hybrid_fem_lssvr_amd import ops
dev = "cuda:0"
ne, M, n, K = (int(sys.argv[1]) if len(sys.argv) > 1 else 100008), 9, 16, (int(sys.argv[2]) if len(sys.argv) > 2 else 200)
half = ne / 24.0
nodes = np.arange(ne + 1, dtype=np.float64) * (2 * half / ne) - half
x = torch.as_tensor(nodes, device=dev); u = torch.sin(np.pi * x)
plan = ops.StepPlan(x, u, M, 1e4, n, global_domain=(-half, half))
def timed(fn, reps=9):
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / K)
    return sorted(ts)
def eager():
    for _ in range(K): plan.launch()
eager(); torch.cuda.synchronize()
te = timed(eager)
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side): plan.launch()
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(K): plan.launch()
g.replay(); torch.cuda.synchronize()
tg = timed(g.replay)
print(f"eager  : median {te[len(te)//2]:.2f} us per step, min {te[0]:.2f}")
print(f"graph  : median {tg[len(tg)//2]:.2f} us per step, min {tg[0]:.2f}")
import time
torch.cuda.synchronize()
t0 = time.perf_counter(); eager(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host   : issuing {K} eager launches took {(t1-t0)/K*1e6:.2f} us per launch; drained {(t2-t1)*1e6:.0f} us later")
