"""Sliding-window inference throughput of the cfg2 network (bf16 eval forwards, 128^3 patches, 50 % overlap) on a
synthetic 256^3 volume resident in HBM.  usage: python scripts/bench_inference.py [--size 256] [--batch 2]"""
import argparse, os, sys, time
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
import torch
import bench as B
import mt3d_amd  # noqa: F401
from mt3d_amd.builders.build_network_from_config import NetworkFromConfig
from mt3d_amd.inference import SlidingWindowInferer, all_positions

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=256)
ap.add_argument("--batch", type=int, default=2)
a = ap.parse_args()
w = dict(B.WORKLOADS["cfg2"])
torch.manual_seed(0)
net = NetworkFromConfig(B.make_mgr(w)).cuda()
vol = torch.rand((1, a.size, a.size, a.size), device="cuda")
run = SlidingWindowInferer(net, None, (128, 128, 128), batch_size=a.batch, overlap=0.5, compute_dtype=torch.bfloat16)
run.accumulate(vol); torch.cuda.synchronize()
t0 = time.perf_counter()
sums, cnt = run.accumulate(vol)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
n = len(all_positions(vol.shape[1:], (128, 128, 128), 0.5))
print(f"{n} patches of 128^3 in {dt * 1e3:.1f} ms -> {n / dt:.1f} patches/s, {a.size ** 3 / dt / 1e6:.1f} Mvoxel/s of volume")

# where a batch's time goes (each phase synchronised: sums to more than the pipelined loop above)
pos = all_positions(vol.shape[1:], (128, 128, 128), 0.5)[:a.batch]
def timed(fn, n=10):
    fn(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3, r
net.eval(); net.compute_dtype = torch.bfloat16
t_cut, patches = timed(lambda: torch.stack([vol[:, z:z + 128, y:y + 128, x:x + 128] for z, y, x in pos]).contiguous())
t_fwd, raw = timed(lambda: net.forward_logits(patches))
plan = net.plan_for(patches.shape, torch.bfloat16, patches.device, False)
t_plan, _ = timed(lambda: plan.run_forward(patches, apply_act=False))
def acc():
    for name in raw:
        pred = torch.sigmoid(raw[name].float())
        for b, (z, y, x) in enumerate(pos):
            sums[name][:, z:z + 128, y:y + 128, x:x + 128] += pred[b]
    for z, y, x in pos:
        cnt[z:z + 128, y:y + 128, x:x + 128] += 1.0
t_acc, _ = timed(acc)
print(f"per batch of {a.batch}: cut + stack {t_cut:.2f} ms, forward_logits {t_fwd:.2f} ms (plan.run_forward alone {t_plan:.2f}), activate + accumulate {t_acc:.2f} ms")
