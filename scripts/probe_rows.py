"""Bandwidth of the row-chunk access pattern (8 or 16 columns of 64 rows per batch) against the
full-line stream copy: which request shape the tabulated inputs of config 5 should be read with."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hybrid_fem_lssvr_amd import _capi
lib = _capi.load()
dev = torch.device("cuda:0")
ne, n = 1000000, 16
src = torch.rand(ne * n, dtype=torch.float64, device=dev)
rows = torch.empty(ne, dtype=torch.float64, device=dev)
big = torch.rand(12500000, dtype=torch.float64, device=dev)
dst = torch.empty_like(big)
s = torch.cuda.current_stream().cuda_stream
def t(fn, reps=30):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) * 1e-3)
    return sorted(ts)[len(ts) // 2]
for chunk in (8, 16):
    dt = t(lambda: lib.lssvr_row_chunk_probe(src.data_ptr(), rows.data_ptr(), ne, n, chunk, s))
    print(f"row chunks of {chunk}: {dt*1e6:.1f} us = {ne*n*8/dt/1e12:.2f} TB/s read")
dt = t(lambda: lib.lssvr_stream_probe(big.data_ptr(), dst.data_ptr(), big.numel(), s))
print(f"stream copy: {dt*1e6:.1f} us = {2*big.numel()*8/dt/1e12:.2f} TB/s read+write")
