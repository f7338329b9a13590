"""BASELINE config 5 under rocprofv3: the variable-coefficient lane kernel a few times, then the
two byte-count probes (full-line stream, half-line row chunks) that calibrate FETCH_SIZE / WRITE_SIZE.

usage: prof_c5.py [ne] [reps] [wide|narrow] [point|element]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hybrid_fem_lssvr_amd import ops, _capi
import bench

ne = int(sys.argv[1]) if len(sys.argv) > 1 else 1000008
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
domain = sys.argv[3] if len(sys.argv) > 3 else "wide"
pm = (sys.argv[4] if len(sys.argv) > 4 else "point") == "point"
M, n = 9, 16
dev = torch.device("cuda:0")
lo, hi = (-ne / 24.0, ne / 24.0) if domain == "wide" else (-1.0, 1.0)
nodes = np.arange(ne + 1, dtype=np.float64) * ((hi - lo) / ne) + lo
nodes[-1] = hi
x = torch.as_tensor(nodes, device=dev)
u = torch.sin(np.pi * x)
a, da, f = bench._varcoef_device_tables(ops.colloc_points(x, n, point_major=pm))
W = torch.empty((ne, M), dtype=torch.float64, device=dev)
st = torch.empty(ne, dtype=torch.int32, device=dev)
for _ in range(reps):
    ops.enhance_varcoef(x, u, M, 1e4, n, a, da, f, global_domain=(lo, hi), out=W, status=st, point_major=pm)
torch.cuda.synchronize()
lib = _capi.load()
s = torch.cuda.current_stream().cuda_stream
probe = 12500000
src = torch.zeros(probe, dtype=torch.float64, device=dev)
dst = torch.empty_like(src)
rows = torch.empty(ne, dtype=torch.float64, device=dev)
for _ in range(reps):
    lib.lssvr_stream_probe(src.data_ptr(), dst.data_ptr(), probe, s)
    lib.lssvr_row_chunk_probe(f.data_ptr(), rows.data_ptr(), ne, n, 16, s)
    lib.lssvr_row_chunk_probe(a.data_ptr(), rows.data_ptr(), ne, n, 8, s)
torch.cuda.synchronize()
print("done", ne, "point-major" if pm else "element-major", "fallback", int(st.sum()))
