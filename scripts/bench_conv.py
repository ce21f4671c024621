"""Per-layer microbenchmark of the conv entry points (fwd / bwd-data / bwd-weight) at the cfg2 shapes.
usage: python scripts/bench_conv.py [filter ...] [--iters N]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
import mt3d_amd
from mt3d_amd.engine import ops, lib

SHAPES = [  # name, ci, co, dims, batch
    ("32->32@128", 32, 32, (128, 128, 128), 2),
    ("64->32@128", 64, 32, (128, 128, 128), 2),
    ("64->64@64", 64, 64, (64, 64, 64), 2),
    ("128->64@64", 128, 64, (64, 64, 64), 2),
    ("128->128@32", 128, 128, (32, 32, 32), 2),
    ("256->256@16", 256, 256, (16, 16, 16), 2),
    ("320->320@8", 320, 320, (8, 8, 8), 2),
    ("320->320@4", 320, 320, (4, 4, 4), 2),
    ("512->512@8", 512, 512, (8, 8, 8), 2),
    ("512->512@4", 512, 512, (4, 4, 4), 2),
]
flt = [a for a in sys.argv[1:] if not a.startswith("--") and not a.isdigit()]
iters = 10
if "--iters" in sys.argv:
    iters = int(sys.argv[sys.argv.index("--iters") + 1])
KINDS = ("fwd", "fwdst", "dgrad", "dgradacc", "wgrad")   # dgradacc: dx += (accumulate)      # fwdst: conv + InstanceNorm statistics (rx_conv3d_fwd_stats)
kinds = [f for f in flt if f in KINDS] or ["fwd", "dgrad", "wgrad"]
names = [f for f in flt if f not in KINDS]
dt = torch.bfloat16
ZEROS = "--zeros" in sys.argv      # all-zero operands: the chip holds a higher clock (MI355X_MICROARCH.md, DVFS give-back) -- the gap to
                                   # the random-data figure is power management, not the kernel
for name, ci, co, dims, n in SHAPES:
    if names and not any(f in name for f in names):
        continue
    mk = torch.zeros if ZEROS else torch.randn
    x = ops.Act(mk((n, *dims, ci), device="cuda").to(dt))
    y = ops.Act(mk((n, *dims, co), device="cuda").to(dt))
    dx = ops.Act.zeros(n, *dims, ci, dt)
    w = mk((co, ci, 3, 3, 3), device="cuda") * 0.05
    wf, wb = ops.pack_conv_weight(w, dt)
    dw = torch.empty_like(w)
    k, s = (3, 3, 3), (1, 1, 1)
    flops = 2.0 * n * dims[0] * dims[1] * dims[2] * ci * co * 27
    stats = torch.empty((n, co, 2), device="cuda")
    fns = {"fwd": lambda: ops.conv3d_fwd(x, wf, None, y, k, s), "fwdst": lambda: ops.conv3d_fwd_stats(x, wf, None, y, k, s, stats), "dgrad": lambda: ops.conv3d_bwd_data(y, wb, dx, k, s), "dgradacc": lambda: ops.conv3d_bwd_data(y, wb, dx, k, s, True),
           "wgrad": lambda: ops.conv3d_bwd_weight(x, y, dw, k, s)}
    for kind in kinds:
        fn = fns[kind]
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        print(f"{name:14s} {kind:6s} {us:9.1f} us  {flops / us / 1e6:8.1f} TF/s  [{lib.load().rx_last_conv_kernel().decode()}]", flush=True)
