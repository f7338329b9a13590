"""The fused STEP (ops.StepPlan: P1 assembly + enhancement, what bench.py's timed region launches) a few times --
target for rocprofv3 --pmc / --kernel-trace -- plus the stream probe that calibrates FETCH_SIZE / WRITE_SIZE.

usage: prof_step.py ne,M,n [reps] [wide|narrow] [probe_doubles]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hybrid_fem_lssvr_amd import ops, _capi

ne, M, n = (int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else (100008, 9, 16)))
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
domain = sys.argv[3] if len(sys.argv) > 3 else "wide"
probe = int(sys.argv[4]) if len(sys.argv) > 4 else 0
dev = torch.device("cuda:0")
lo, hi = (-ne / 24.0, ne / 24.0) if domain == "wide" else (-1.0, 1.0)
nodes = np.arange(ne + 1, dtype=np.float64) * ((hi - lo) / ne) + lo
nodes[-1] = hi
x = torch.as_tensor(nodes, device=dev)
u = torch.sin(np.pi * x)
plan = ops.StepPlan(x, u, M, 1e4, n, global_domain=(lo, hi))
for _ in range(reps):
    plan.launch()
torch.cuda.synchronize()
if probe:
    src = torch.zeros(probe, dtype=torch.float64, device=dev)
    dst = torch.empty_like(src)
    lib = _capi.load()
    for _ in range(reps):
        lib.lssvr_stream_probe(src.data_ptr(), dst.data_ptr(), probe, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
print("done", ne, M, n, "fallback", int(plan.status.sum()))
