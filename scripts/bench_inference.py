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
