"""how long does the host need to ENQUEUE one train step vs how long the GPU needs to finish it?"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
import bench
import mt3d_amd
from mt3d_amd.builders.build_network_from_config import NetworkFromConfig
from mt3d_amd.training.losses.losses import LOSS_FN_MAP
WL = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
w = dict(bench.WORKLOADS[WL])
torch.manual_seed(0)
net = NetworkFromConfig(bench.make_mgr(w)).cuda(); net.compute_dtype = torch.bfloat16; net.train()
loss_fn = LOSS_FN_MAP["BCEDiceLoss"](0.5, 0.5)
params = list(net.parameters())
from mt3d_amd.training.optim import EngineAdamW, clip_and_step
opt = EngineAdamW(params, model=None, lr=1e-3, weight_decay=0.0)
x, t = bench.synthetic_batch(w, 2, 1234, "cuda")
def step():
    out = net(x); loss = loss_fn(out["sheet"], t["sheet"]); loss.backward()
    clip_and_step(opt, params, 3); opt.zero_grad(set_to_none=True)
for _ in range(6): step()
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"host enqueue {1e3*(t1-t0):.2f} ms, until GPU idle {1e3*(t2-t0):.2f} ms")
# split: forward only / backward only host cost
torch.cuda.synchronize(); t0 = time.perf_counter(); out = net(x); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"forward: host {1e3*(t1-t0):.2f} ms, gpu-complete {1e3*(t2-t0):.2f} ms")
loss = loss_fn(out["sheet"], t["sheet"]); torch.cuda.synchronize()
t0 = time.perf_counter(); loss.backward(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"backward: host {1e3*(t1-t0):.2f} ms, gpu-complete {1e3*(t2-t0):.2f} ms")
out = net(x); loss = loss_fn(out["sheet"], t["sheet"]); loss.backward(); torch.cuda.synchronize()
t0 = time.perf_counter(); clip_and_step(opt, params, 3); opt.zero_grad(set_to_none=True); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"clip+adamw+zero_grad: host {1e3*(t1-t0):.2f} ms, gpu-complete {1e3*(t2-t0):.2f} ms")
out = net(x); torch.cuda.synchronize()
t0 = time.perf_counter(); loss = loss_fn(out["sheet"], t["sheet"]); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"loss fwd: host {1e3*(t1-t0):.2f} ms, gpu-complete {1e3*(t2-t0):.2f} ms")
print("programs:", sum(1 for p in net._plans.values() for s in p._pstate.values() if s.get("prog") is not None))
