"""Host cost of one launch of the headline step from Python (no device synchronisation inside the timed loop):
StepPlan.launch(), the same with the stream handle passed in, the bare ctypes call.  usage: host_launch_cost.py [K]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hybrid_fem_lssvr_amd import ops

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ne = 100008
x = torch.linspace(-ne / 24.0, ne / 24.0, ne + 1, dtype=torch.float64, device="cuda:0")
u = torch.sin(np.pi * x)
plan = ops.StepPlan(x, u, 9, 1e4, 16, global_domain=(float(x[0]), float(x[-1])))
for _ in range(200):
    plan.launch()
torch.cuda.synchronize()
st = torch.cuda.current_stream().cuda_stream


def timed(fn, reps=30):
    out = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        out.append(((t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6))
    out.sort()
    return out[len(out) // 2]


print("K = %d launches per bracket; (host enqueue us per launch, wall incl. final sync us per launch)" % K)
print("plan.launch()                 ", "%.2f %.2f" % timed(lambda: plan.launch()))
print("plan.launch(stream=handle)    ", "%.2f %.2f" % timed(lambda: plan.launch(stream=st)))
step, cargs = plan._step, plan._cargs
print("bare ctypes lssvr_step(*args) ", "%.2f %.2f" % timed(lambda: step(*cargs, st)))
if hasattr(plan, "_handle") and plan._handle:
    launch, h = plan._launch, plan._handle
    print("bare ctypes plan_launch(h, st)", "%.2f %.2f" % timed(lambda: launch(h, st)))
