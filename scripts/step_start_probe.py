"""Is the main stream idle at the start of a step?  Events around the optimizer end, the first forward kernel (stem) and the
end of the forward, against host timestamps (how far the host runs ahead)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
import bench
import mt3d_amd
from mt3d_amd.builders.build_network_from_config import NetworkFromConfig
from mt3d_amd.training.losses.losses import LOSS_FN_MAP
w = dict(bench.WORKLOADS["cfg2"])
torch.manual_seed(0)
net = NetworkFromConfig(bench.make_mgr(w)).cuda(); net.compute_dtype = torch.bfloat16; net.train()
loss_fn = LOSS_FN_MAP["BCEDiceLoss"](0.5, 0.5)
params = list(net.parameters())
from mt3d_amd.training.optim import EngineAdamW, clip_and_step
opt = EngineAdamW(params, model=None, lr=1e-3, weight_decay=0.0)
x, t = bench.synthetic_batch(w, 2, 1234, "cuda")
E = lambda: torch.cuda.Event(enable_timing=True)
rec = []
def step(i, probe):
    r = dict(h0=time.perf_counter(), e0=E(), e_stem0=E(), e_stem1=E(), e_fwd=E(), e_bwd=E(), e_end=E())
    r["e0"].record()
    if probe:
        plan = next(iter(net._plans.values()))
        orig = plan.fwd[0]
        def first():
            r["e_stem0"].record(); orig(); r["e_stem1"].record()
        plan.fwd[0] = first
    out = net(x)
    if probe:
        plan.fwd[0] = orig
    r["e_fwd"].record()
    loss = loss_fn(out["sheet"], t["sheet"]); loss.backward()
    r["e_bwd"].record()
    clip_and_step(opt, params, 3); opt.zero_grad(set_to_none=True)
    r["e_end"].record()
    r["h1"] = time.perf_counter()
    rec.append(r)
for i in range(3): step(i, False)
torch.cuda.synchronize(); rec.clear()
for i in range(8): step(i, True)
torch.cuda.synchronize()
base_e, base_h = rec[0]["e0"], rec[0]["h0"]
for i, r in enumerate(rec):
    g = lambda k: base_e.elapsed_time(r[k])
    print(f"step {i}: host start {1e3*(r['h0']-base_h):7.2f} end {1e3*(r['h1']-base_h):7.2f} | gpu start {g('e0'):7.2f} stem [{g('e_stem0'):7.2f},{g('e_stem1'):7.2f}] fwd_end {g('e_fwd'):7.2f} bwd_end {g('e_bwd'):7.2f} step_end {g('e_end'):7.2f}")
