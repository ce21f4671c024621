import sys, os
sys.path[:0] = ["/root/repo"]
import torch, mt3d_amd
from mt3d_amd.engine import ops, lib
dt = torch.bfloat16
def timeit(name, fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:40s} {e0.elapsed_time(e1) * 1e3 / iters:8.1f} us [{lib.load().rx_last_conv_kernel().decode()}]", flush=True)
n, d, ci, co = 2, 128, 32, 64
buf = torch.randn((n, d, d, d, 64), device="cuda").to(dt)
dense = ops.Act(torch.randn((n, d, d, d, 32), device="cuda").to(dt))
strided = ops.Act(buf, 32, 32)
y = ops.Act(torch.randn((n, 64, 64, 64, co), device="cuda").to(dt))
w = torch.randn((co, ci, 3, 3, 3), device="cuda") * 0.05
wf, wb = ops.pack_conv_weight(w, dt)
k, s = (3, 3, 3), (2, 2, 2)
for nm, x in (("dense", dense), ("slice of 64", strided)):
    timeit(f"s2 conv fwd, x {nm}", lambda: ops.conv3d_fwd(x, wf, None, y, k, s))
    timeit(f"s2 conv dgrad, dx {nm}", lambda: ops.conv3d_bwd_data(y, wb, x, k, s))
    timeit(f"s2 conv dgrad acc, dx {nm}", lambda: ops.conv3d_bwd_data(y, wb, x, k, s, True))
    dw = torch.empty_like(w)
    timeit(f"s2 conv wgrad, x {nm}", lambda: ops.conv3d_bwd_weight(x, y, dw, k, s))
    stats = torch.empty((n, 32, 2), device="cuda")
    o = ops.Act(torch.empty_like(dense.t))
    timeit(f"instnorm stats, y {nm}", lambda: ops.instnorm_stats(x, stats))
    timeit(f"in_act_fwd, out {nm}", lambda: ops.instnorm_act_fwd(dense, stats, x, 0.01))
    p = ops.Act.empty(n, 64, 64, 64, 32, dt)
    timeit(f"avgpool_fwd, x {nm}", lambda: ops.avgpool_fwd(x, p, (2, 2, 2)))
    timeit(f"avgpool_bwd acc, dx {nm}", lambda: ops.avgpool_bwd(p, x, (2, 2, 2), True))
