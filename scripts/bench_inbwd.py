"""Isolated InstanceNorm backward (rx_instnorm_act_bwd: colreduce + finalize + apply) at the cfg2 shapes; run under
rocprofv3 --kernel-trace --stats to split the three kernels.  usage: python scripts/bench_inbwd.py [--iters N]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
import mt3d_amd  # noqa: F401
from mt3d_amd.engine import ops
iters = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else 20
dt = torch.bfloat16
for c, d in [(32, 128), (64, 64), (128, 32), (256, 16)]:
    n = 2
    mk = lambda: ops.Act(torch.randn((n, d, d, d, c), device="cuda").to(dt))
    g, y, out, dy, dres = mk(), mk(), mk(), mk(), mk()
    stats = torch.empty((n, c, 2), device="cuda")
    ops.instnorm_stats(y, stats)
    for name, fn in [("no-res", lambda: ops.instnorm_act_bwd(g, y, stats, None, dy, 0.01)),
                     ("res   ", lambda: ops.instnorm_act_bwd(g, y, stats, out, dy, 0.01, dres, False))]:
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        nt = 3 if name.startswith("no") else 5
        mb = nt * n * d ** 3 * c * 2 / 1e6
        print(f"{c:4d}ch @{d:3d}^3 {name}: {us:7.1f} us for the three launches; {mb:7.1f} MB algorithmic -> {mb / us / 1e3 * 1e3:6.2f} GB/ms")
