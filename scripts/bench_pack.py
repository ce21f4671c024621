"""Isolated weight re-pack (rx_pack_conv_weight) at the cfg2 weight shapes.  usage: python scripts/bench_pack.py"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
import mt3d_amd  # noqa: F401
from mt3d_amd.engine import ops
for co, ci in [(32, 32), (64, 64), (128, 128), (256, 256), (512, 512), (512, 1024)]:
    w = torch.randn((co, ci, 3, 3, 3), device="cuda")
    wf, wb = ops.pack_conv_weight(w, torch.bfloat16)
    for _ in range(3):
        ops.pack_conv_weight(w, torch.bfloat16, wf, wb)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.pack_conv_weight(w, torch.bfloat16, wf, wb)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    mb = w.numel() * 8 / 1e6
    print(f"pack {co:4d}x{ci:4d}x27: {us:7.1f} us  {mb:7.1f} MB -> {mb / us / 1e3 * 1e3:5.2f} GB/ms")
