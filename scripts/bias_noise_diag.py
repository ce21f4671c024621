"""conv biases under InstanceNorm have an analytically zero gradient: how large is what the engine delivers (bf16 vs fp32 compute)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
import mt3d_amd  # noqa
def _mgr(patch, tasks, cin, batch, autoconfigure, model_config):      # what NetworkFromConfig reads from a ConfigManager
    from types import SimpleNamespace
    return SimpleNamespace(tasks=tasks, train_patch_size=tuple(patch), train_batch_size=batch, in_channels=cin, vram_max=16.0,
                           autoconfigure=autoconfigure, model_config=dict(model_config), verbose=False)
from mt3d_amd.builders.build_network_from_config import NetworkFromConfig
patch = (14, 256, 256) if (len(sys.argv) < 2 or sys.argv[1] == "ink") else (64, 64, 64)
B = int(sys.argv[2]) if len(sys.argv) > 2 else 3
tasks = {"ink": {"channels": 1, "activation": "none"}}
mgr = _mgr(patch, tasks, 1, B, True, {"conv_bias": True, "squeeze_excitation": len(sys.argv) < 4})
torch.manual_seed(0)
net = NetworkFromConfig(mgr).cuda()
gen = torch.Generator(device="cuda").manual_seed(7)
x = torch.rand((B, 1, *patch), device="cuda", generator=gen)
g = torch.randn((B, 1, *patch), device="cuda", generator=gen) * 1e-3
res = {}
for dt in (torch.float32, torch.bfloat16):
    net.compute_dtype = dt
    for p in net.parameters():
        p.grad = None
    out = net(x)
    torch.autograd.backward([out["ink"]], [g])
    res[dt] = {n: p.grad.detach().double().clone() for n, p in net.named_parameters() if p.grad is not None}
for n in res[torch.float32]:
    if not n.endswith(".conv.bias"):
        continue
    w = n[:-4] + "weight"
    for dt in res:
        bg, wg = res[dt][n], res[dt][w]
        wc = wg.flatten(1).norm(dim=1)
        r = (bg.abs() / wc.clamp_min(1e-30))
        print(f"{str(dt):15s} {n:60s} |b|={bg.norm():.3e} |w|={wg.norm():.3e} max per-channel |b_c|/|w_c|={r.max():.3e} at c={int(r.argmax())} (b={bg[int(r.argmax())]:.3e}, wc={wc[int(r.argmax())]:.3e})")
    if "stem" in n:
        print("  stem bias grad fp32:", res[torch.float32][n][:16].tolist())
        print("  stem bias grad bf16:", res[torch.bfloat16][n][:16].tolist())
