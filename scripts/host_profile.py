"""cProfile of the host side of the cfg2 train step: where the ~2000 C-ABI enqueues per step spend their Python time.
Usage (GPU box): python scripts/host_profile.py > gpurun_out/host_profile.txt"""
import cProfile, pstats, sys, io, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
import bench
import mt3d_amd  # noqa: F401
from mt3d_amd.builders.build_network_from_config import NetworkFromConfig
from mt3d_amd.training.losses.losses import LOSS_FN_MAP
from mt3d_amd.training.optim import EngineAdamW, clip_and_step

w = dict(bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"])
torch.manual_seed(0)
net = NetworkFromConfig(bench.make_mgr(w)).cuda(); net.compute_dtype = torch.bfloat16; net.train()
loss_fns = {k: LOSS_FN_MAP[v.get("loss_fn", "BCEDiceLoss")](**v.get("loss_kwargs", {})) for k, v in w["tasks"].items()}
params = list(net.parameters())
opt = EngineAdamW(params, model=None, lr=1e-3, weight_decay=0.0)
x, t = bench.synthetic_batch(w, w["batch"], 1234, "cuda")

def step():
    out = net(x)
    loss = 0.0
    for k, gt in t.items():
        loss = loss + loss_fns[k](out[k], gt)
    loss.backward()
    clip_and_step(opt, params, 3); opt.zero_grad(set_to_none=True)

for _ in range(5):
    step()
torch.cuda.synchronize()
# host-only time per step: enqueue 6 steps without syncing and time the host side
t0 = time.perf_counter()
for _ in range(6):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3*(t1-t0)/6:.2f} ms/step; with drain {1e3*(t2-t0)/6:.2f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(6):
    step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(40)
print(s.getvalue())
