"""Per-layer microbenchmark of the cfg2 conv layers that do NOT run on the stride-1 3x3x3 halo kernels: strided 3x3x3, 1x1x1
projections, decoder convs on the concat (Ci = 2 Co), transposed convs.  usage: python scripts/bench_conv_misc.py [--iters N]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
import mt3d_amd  # noqa
from mt3d_amd.engine import ops, lib
iters = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else 20
dt = torch.bfloat16
n = 2
L = []
for ci, co, d in ((32, 64, 128), (64, 128, 64), (128, 256, 32), (256, 512, 16), (512, 512, 8)):
    L.append((f"s2 3x3x3 {ci}->{co} in {d}^3", "conv", ci, co, (d,) * 3, (3, 3, 3), (2, 2, 2)))
    L.append((f"1x1x1 skip {ci}->{co} @{d // 2}^3", "conv", ci, co, (d // 2,) * 3, (1, 1, 1), (1, 1, 1)))
for c, d in ((64, 64), (128, 32), (256, 16), (512, 8)):
    L.append((f"decoder 3x3x3 {2 * c}->{c} @{d}^3", "conv", 2 * c, c, (d,) * 3, (3, 3, 3), (1, 1, 1)))
for ci, co, d in ((512, 512, 4), (512, 256, 8), (256, 128, 16), (128, 64, 32), (64, 32, 64)):
    L.append((f"convT {ci}->{co} from {d}^3", "convT", ci, co, (d,) * 3, None, (2, 2, 2)))
def timed(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
for name, kind, ci, co, dims, k, s in L:
    if kind == "conv":
        od = tuple((d + 2 * ((kk - 1) // 2) - kk) // ss + 1 for d, kk, ss in zip(dims, k, s))
        x = ops.Act(torch.randn((n, *dims, ci), device="cuda").to(dt))
        y = ops.Act(torch.randn((n, *od, co), device="cuda").to(dt))
        dx = ops.Act.zeros(n, *dims, ci, dt)
        w = torch.randn((co, ci, *k), device="cuda") * 0.05
        wf, wb = ops.pack_conv_weight(w, dt)
        dw = torch.empty_like(w)
        fl = 2.0 * n * od[0] * od[1] * od[2] * ci * co * k[0] * k[1] * k[2]
        fns = [("fwd", lambda: ops.conv3d_fwd(x, wf, None, y, k, s)), ("dgrad", lambda: ops.conv3d_bwd_data(y, wb, dx, k, s)),
               ("wgrad", lambda: ops.conv3d_bwd_weight(x, y, dw, k, s))]
    else:
        od = tuple(d * 2 for d in dims)
        x = ops.Act(torch.randn((n, *dims, ci), device="cuda").to(dt))
        y = ops.Act(torch.randn((n, *od, co), device="cuda").to(dt))
        dx = ops.Act.zeros(n, *dims, ci, dt)
        w = torch.randn((ci, co, 2, 2, 2), device="cuda") * 0.05
        wf, wb = ops.pack_convT_weight(w, dt)
        dw = torch.empty_like(w)
        fl = 2.0 * n * od[0] * od[1] * od[2] * ci * co
        fns = [("fwd", lambda: ops.convT3d_fwd(x, wf, None, y, s)), ("dgrad", lambda: ops.convT3d_bwd_data(y, wb, dx, s)),
               ("wgrad", lambda: ops.convT3d_bwd_weight(x, y, dw, s))]
    for kn, fn in fns:
        us = timed(fn)
        print(f"{name:34s} {kn:6s} {us:8.1f} us {fl / us / 1e6:8.1f} TF/s  [{lib.load().rx_last_conv_kernel().decode()}]", flush=True)
