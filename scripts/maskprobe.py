import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hybrid_fem_lssvr_amd import ops, _capi
lib = _capi.load()
out = torch.zeros(8192 * 256, dtype=torch.float64, device="cuda:0")
st = torch.cuda.current_stream().cuda_stream
def t(mode, blocks=8192, iters=2048):
    lib.lssvr_fp64_probe(out.data_ptr(), blocks, iters, mode, st); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); lib.lssvr_fp64_probe(out.data_ptr(), blocks, iters, mode, st); e1.record(); e1.synchronize()
    return e0.elapsed_time(e1)
for act in (64, 48, 32, 16, 8):
    print("active lanes", act, "ms", round(t(100 + act), 3), " 1-wave/SIMD (1024 blocks of 256):", round(t(100 + act, 256, 16384), 3))
# f64 MFMA vs FP64 vector FMA on the same SIMD: overlapping pipes -> mixed ~ max(mfma, valu), shared -> ~ sum
B, I = 8192, 512
tm = t(1, B // 2, I)
tv = t(0, B // 2, 8 * I)
tx = t(2, B, I)
print(f"mfma-only {tm:.3f} ms  valu-only {tv:.3f} ms  mixed {tx:.3f} ms  (sum {tm+tv:.3f}, max {max(tm,tv):.3f})")
