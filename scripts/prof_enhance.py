"""Run only the kernels of interest a few times (target for rocprofv3 --pmc / --kernel-trace).

usage: prof_enhance.py ne,M,n [reps] [solver] [wide|narrow] [probe_doubles] [nowork]
(nowork: no workspace is handed over, so degree > 21 runs its single-kernel variants)
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hybrid_fem_lssvr_amd import ops, _capi

ne, M, n = (int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else (100008, 9, 16)))
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
solver = int(sys.argv[3]) if len(sys.argv) > 3 else 0
domain = sys.argv[4] if len(sys.argv) > 4 else "wide"
probe = int(sys.argv[5]) if len(sys.argv) > 5 else 0
work = False if (len(sys.argv) > 6 and sys.argv[6] == "nowork") else None
dev = torch.device("cuda:0")
lo, hi = (-ne / 24.0, ne / 24.0) if domain == "wide" else (-1.0, 1.0)
nodes = np.arange(ne + 1, dtype=np.float64) * ((hi - lo) / ne) + lo
nodes[-1] = hi
x = torch.as_tensor(nodes, device=dev)
u = torch.sin(np.pi * x)
W = torch.empty((ne, M), dtype=torch.float64, device=dev)
st = torch.empty(ne, dtype=torch.int32, device=dev)
for _ in range(reps):
    ops.enhance(x, u, M, 1e4, n, out=W, status=st, global_domain=(lo, hi), solver=solver, work=work)
torch.cuda.synchronize()
if probe:
    src = torch.zeros(probe, dtype=torch.float64, device=dev)
    dst = torch.empty_like(src)
    lib = _capi.load()
    for _ in range(reps):
        lib.lssvr_stream_probe(src.data_ptr(), dst.data_ptr(), probe, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
print("done", ne, M, n, "fallback", int(st.sum()))
