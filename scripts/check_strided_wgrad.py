"""Full-size check of the stride-2 weight gradient (32->64 @128^3 -> 64^3, batch 2, x read through a channel slice of a
wider buffer): LDS-halo kernel vs generic kernel vs torch (fp32 conv backward on the device)."""
import os, sys
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
import torch, torch.nn.functional as F
import mt3d_amd
from mt3d_amd.engine import ops
torch.manual_seed(0)
dt = torch.bfloat16
for (ci, co, d, ld, c0) in [(32, 64, 128, 64, 32), (32, 64, 128, 32, 0), (64, 128, 64, 128, 64)]:
    n = 2
    buf = torch.randn((n, d, d, d, ld), device="cuda").to(dt)
    x = ops.Act(buf, c0, ci)
    gy = ops.Act((torch.randn((n, d // 2, d // 2, d // 2, co), device="cuda") * 0.1).to(dt))
    k, s = (3, 3, 3), (2, 2, 2)
    dw_new = torch.empty((co, ci, 3, 3, 3), device="cuda"); dw_old = torch.empty_like(dw_new)
    os.environ.pop("RX_NO_STRIDED_WGH", None)
    ops.conv3d_bwd_weight(x, gy, dw_new, k, s)
    os.environ["RX_NO_STRIDED_WGH"] = "1"
    ops.conv3d_bwd_weight(x, gy, dw_old, k, s)
    os.environ.pop("RX_NO_STRIDED_WGH", None)
    xr = x.tensor().permute(0, 4, 1, 2, 3).float().contiguous()
    gr = gy.tensor().permute(0, 4, 1, 2, 3).float().contiguous()
    ref = torch.nn.grad.conv3d_weight(xr, (co, ci, 3, 3, 3), gr, stride=2, padding=1)
    torch.cuda.synchronize()
    rel = lambda a, b: ((a - b).norm() / b.norm()).item()
    print(f"ci={ci} co={co} d={d} ld={ld} c0={c0}:  halo vs torch {rel(dw_new, ref):.2e}   generic vs torch {rel(dw_old, ref):.2e}   halo vs generic {rel(dw_new, dw_old):.2e}")
