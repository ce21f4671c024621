"""InstanceNorm nets are per-sample: the gradient of a batch is the sum of the gradients of its samples run alone."""
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
patch = {"ink": (14, 256, 256), "c64": (64, 64, 64), "c128": (128, 128, 128)}[sys.argv[1]]
B = int(sys.argv[2])
dts = {"fp32": torch.float32, "bf16": torch.bfloat16}
order = sys.argv[3].split(",")
se = "nose" not in sys.argv
bias = "nobias" not in sys.argv
tasks = {"ink": {"channels": 1, "activation": "none"}}
mgr = _mgr(patch, tasks, 1, B, True, {"conv_bias": bias, "squeeze_excitation": se})
torch.manual_seed(0)
net = NetworkFromConfig(mgr).cuda()
gen = torch.Generator(device="cuda").manual_seed(7)
x = torch.rand((B, 1, *patch), device="cuda", generator=gen)
g = torch.randn((B, 1, *patch), device="cuda", generator=gen) * 1e-3
def grads(xx, gg):
    for p in net.parameters():
        p.grad = None
    out = net(xx)
    torch.autograd.backward([out["ink"]], [gg])
    return out["ink"].detach().double().clone(), {n: p.grad.detach().double().clone() for n, p in net.named_parameters() if p.grad is not None}
for name in order:
    net.compute_dtype = dts[name]
    o, full = grads(x, g)
    acc = None
    outs = []
    for i in range(B):
        oi, gi = grads(x[i:i + 1].contiguous(), g[i:i + 1].contiguous())
        outs.append(oi)
        acc = gi if acc is None else {n: acc[n] + gi[n] for n in acc}
    oe = ((o - torch.cat(outs)).norm() / o.norm()).item()
    worst = sorted(((((full[n] - acc[n]).norm() / acc[n].norm().clamp_min(1e-30)).item(), n) for n in full if not n.endswith(".conv.bias")), reverse=True)[:4]
    print(f"{name} B={B}: logits batch vs singles rel {oe:.3e}; stem |w| batch {full['shared_encoder.stem.convs.0.conv.weight'].norm():.4f} singles {acc['shared_encoder.stem.convs.0.conv.weight'].norm():.4f}; worst non-prenorm-bias grads:")
    for e, n in worst:
        print(f"    {e:.3e} {n}")
