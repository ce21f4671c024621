"""Do an MFMA-bound weight-gradient kernel (side stream) and an HBM-bound InstanceNorm pass (main stream) overlap on this part?
Times A alone, B alone, and A || B (two streams, same number of launches each)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
import mt3d_amd  # noqa
from mt3d_amd.engine import ops
dt = torch.bfloat16
cases = {"64@64": (64, 64, (64, 64, 64)), "32@128": (32, 32, (128, 128, 128)), "128@32": (128, 128, (32, 32, 32))}
name = sys.argv[1] if len(sys.argv) > 1 else "64@64"
ci, co, dims = cases[name]
n = 2
x = ops.Act(torch.randn((n, *dims, ci), device="cuda").to(dt))
dy = ops.Act(torch.randn((n, *dims, co), device="cuda").to(dt))
w = torch.randn((co, ci, 3, 3, 3), device="cuda") * 0.05
dw = torch.empty_like(w)
k, s = (3, 3, 3), (1, 1, 1)
ws2 = torch.empty(ops.workspace().numel(), dtype=torch.uint8, device="cuda")
# HBM-bound pass: InstanceNorm apply + LeakyReLU on a separate tensor of the same size (and a big one: 32ch @128^3)
ey = ops.Act(torch.randn((n, 128, 128, 128, 32), device="cuda").to(dt))
eo = ops.Act.zeros(n, 128, 128, 128, 32, dt)
stats = torch.zeros((n, 32, 2), device="cuda"); stats[..., 1] = 1.0
wf, wb = ops.pack_conv_weight(w, dt)
dx = ops.Act.zeros(n, *dims, ci, dt)
A = lambda: ops.conv3d_bwd_weight(x, dy, dw, k, s, ws=ws2)
variants = {"in_act_fwd(268MB)": lambda: ops.instnorm_act_fwd(ey, stats, eo), "dgrad(same layer)": lambda: ops.conv3d_bwd_data(dy, wb, dx, k, s)}
side = torch.cuda.Stream()
def timed(fa, fb, reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if fa is not None:
        with torch.cuda.stream(side):
            for _ in range(reps): fa()
    if fb is not None:
        for _ in range(reps): fb()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6
reps = 40
for bname, B in variants.items():
    for f in (A, B): f()
    timed(A, B, 200)            # clocks up
    ta, tb, tab = (min(timed(*p, reps) for _ in range(3)) for p in ((A, None), (None, B), (A, B)))
    print(f"{name}: wgrad alone {ta:.1f} us, {bname} alone {tb:.1f} us, both streams {tab:.1f} us per pair  (sum {ta + tb:.1f}, max {max(ta, tb):.1f})")
