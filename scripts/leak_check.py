"""does a long run grow?  300 train steps of cfg1 (64^3, batch 2, bf16, launch programs): device memory (allocated / reserved) and
host RSS at steps 20, 100, 200, 300, plus an eval forward every 50 steps."""
import os, sys, resource
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
import bench
import mt3d_amd  # noqa
from mt3d_amd.builders.build_network_from_config import NetworkFromConfig
from mt3d_amd.training.losses.losses import LOSS_FN_MAP
from mt3d_amd.training.optim import EngineAdamW, clip_and_step
w = dict(bench.WORKLOADS["cfg1"])
torch.manual_seed(0)
net = NetworkFromConfig(bench.make_mgr(w)).cuda(); net.compute_dtype = torch.bfloat16
loss_fn = LOSS_FN_MAP["BCEDiceLoss"](0.5, 0.5)
params = list(net.parameters())
opt = EngineAdamW(params, model=None, lr=1e-3, weight_decay=0.0)
x, t = bench.synthetic_batch(w, 2, 1234, "cuda")
marks = {}
for step in range(1, 301):
    net.train()
    out = net(x); loss = loss_fn(out["sheet"], t["sheet"]); loss.backward()
    clip_and_step(opt, params, 3); opt.zero_grad(set_to_none=True)
    if step % 50 == 0:
        net.eval()
        with torch.no_grad():
            net(x)
    if step in (20, 100, 200, 300):
        torch.cuda.synchronize()
        marks[step] = (torch.cuda.memory_allocated(), torch.cuda.memory_reserved(), resource.getrusage(resource.RUSAGE_SELF).ru_maxrss)
        print(step, [round(v / 2**20, 1) for v in marks[step][:2]], "MiB dev, host maxrss", round(marks[step][2] / 1024, 1), "MiB, loss", float(loss))
a, b = marks[100], marks[300]
assert b[0] <= a[0] + (1 << 20), "device allocations grow"
assert b[2] <= a[2] * 1.02 + 8 * 1024, "host memory grows"
print("OK: no growth between step 100 and step 300")
