"""stem conv forward / weight gradient at the cfg2 shape.  usage: python scripts/bench_stem.py"""
import sys, os
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
import torch
import mt3d_amd
from mt3d_amd.engine import ops
dt = torch.bfloat16
x = torch.rand((2, 1, 128, 128, 128), device="cuda")
w = torch.randn((32, 1, 3, 3, 3), device="cuda") * 0.2
b = torch.randn(32, device="cuda")
y = ops.Act.zeros(2, 128, 128, 128, 32, dt)
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
print(f"stem fwd {t(lambda: ops.stem_conv_fwd(x, w, b, y, (3, 3, 3))):.1f} us")
dw = torch.empty_like(w)
print(f"stem wgrad {t(lambda: ops.stem_conv_bwd_weight(x, y, dw, (3, 3, 3))):.1f} us")
