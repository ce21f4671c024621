"""Isolated durations of the InstanceNorm passes at the cfg2 shapes (one kernel family per line, bytes = the tensors each pass
has to touch once).   usage: python scripts/bench_elem.py [filter ...] [--iters N]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
import mt3d_amd  # noqa: F401
from mt3d_amd.engine import ops

SHAPES = [("32@128", 32, (128, 128, 128), 2), ("64@64", 64, (64, 64, 64), 2), ("128@32", 128, (32, 32, 32), 2),
          ("256@16", 256, (16, 16, 16), 2), ("320@8", 320, (8, 8, 8), 2)]
flt = [a for a in sys.argv[1:] if not a.startswith("--") and not a.isdigit()]
iters = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else 20
dt = torch.bfloat16


def timed(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for name, c, dims, n in SHAPES:
    if flt and not any(f in name for f in flt):
        continue
    mk = lambda: ops.Act(torch.randn((n, *dims, c), device="cuda").to(dt))
    y, g, out, dy, res, dres = mk(), mk(), mk(), mk(), mk(), mk()
    stats = torch.empty((n, c, 2), device="cuda")
    m12 = torch.zeros((n, c, 2), device="cuda")
    ops.instnorm_stats(y, stats)
    ops.instnorm_act_fwd(y, stats, out, residual=res)
    B = y.tensor().numel() * 2 / 1e6      # MB per tensor
    rows = [
        ("stats (colreduce+finalize)", 1, lambda: ops.instnorm_stats(y, stats)),
        ("act_fwd", 2, lambda: ops.instnorm_act_fwd(y, stats, dy)),
        ("act_fwd +residual", 3, lambda: ops.instnorm_act_fwd(y, stats, dy, residual=res)),
        ("bwd reduce+apply (mask from xhat)", 5, lambda: ops.instnorm_act_bwd(g, y, stats, None, dy)),
        ("bwd apply only (mask from xhat)", 3, lambda: ops.instnorm_act_bwd_apply(g, y, stats, None, dy, m12)),
        ("bwd reduce+apply (mask from out)", 7, lambda: ops.instnorm_act_bwd(g, y, stats, out, dy)),
        ("bwd apply only (mask from out)", 4, lambda: ops.instnorm_act_bwd_apply(g, y, stats, out, dy, m12)),
        ("bwd_res reduce+apply (g' written once)", 7, lambda: ops.instnorm_act_bwd_res(g, y, stats, out, dy, dres)),
    ]
    for label, passes, fn in rows:
        us = timed(fn)
        print(f"{name:8s} {label:42s} {us:8.1f} us  {passes * B / us:6.2f} TB/s  ({passes} x {B:.0f} MB)", flush=True)
