"""Microbenchmark of the stage-transition layers (stride-2 conv, transposed conv) at the cfg2 shapes, device time from
HIP events over 20 back-to-back launches.  usage: python scripts/bench_transitions.py"""
import sys, os
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
import torch
import mt3d_amd
from mt3d_amd.engine import ops, lib

dt = torch.bfloat16
def timeit(name, fn, flops, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f"{name:28s} {us:9.1f} us {flops / us / 1e6:8.1f} TF/s [{lib.load().rx_last_conv_kernel().decode()}]", flush=True)

n = 2
for ci, co, d in [(32, 64, 128), (64, 128, 64), (128, 256, 32)]:
    x = ops.Act(torch.randn((n, d, d, d, ci), device="cuda").to(dt))
    y = ops.Act(torch.randn((n, d // 2, d // 2, d // 2, co), device="cuda").to(dt))
    dx = ops.Act.zeros(n, d, d, d, ci, dt)
    w = torch.randn((co, ci, 3, 3, 3), device="cuda") * 0.05
    wf, wb = ops.pack_conv_weight(w, dt)
    dw = torch.empty_like(w)
    k, s = (3, 3, 3), (2, 2, 2)
    fl = 2.0 * n * (d // 2) ** 3 * ci * co * 27
    timeit(f"conv s2 {ci}->{co}@{d} fwd", lambda: ops.conv3d_fwd(x, wf, None, y, k, s), fl)
    timeit(f"conv s2 {ci}->{co}@{d} dgrad", lambda: ops.conv3d_bwd_data(y, wb, dx, k, s), fl)
    timeit(f"conv s2 {ci}->{co}@{d} wgrad", lambda: ops.conv3d_bwd_weight(x, y, dw, k, s), fl)
for ci, co, d in [(64, 32, 64), (128, 64, 32), (256, 128, 16)]:
    x = ops.Act(torch.randn((n, d, d, d, ci), device="cuda").to(dt))
    y = ops.Act(torch.randn((n, 2 * d, 2 * d, 2 * d, co), device="cuda").to(dt))
    dx = ops.Act.zeros(n, d, d, d, ci, dt)
    w = torch.randn((ci, co, 2, 2, 2), device="cuda") * 0.05
    wf, wb = ops.pack_convT_weight(w, dt)
    dw = torch.empty_like(w)
    s = (2, 2, 2)
    fl = 2.0 * n * d ** 3 * ci * co * 8
    timeit(f"convT {ci}->{co}@{d}->{2*d} fwd", lambda: ops.convT3d_fwd(x, wf, None, y, s), fl)
    timeit(f"convT {ci}->{co}@{d}->{2*d} dgrad", lambda: ops.convT3d_bwd_data(y, wb, dx, s), fl)
    timeit(f"convT {ci}->{co}@{d}->{2*d} wgrad", lambda: ops.convT3d_bwd_weight(x, y, dw, s), fl)
