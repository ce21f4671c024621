"""a patch far beyond the BASELINE sizes (256^3, batch 2: 1.07e9 elements / 2.1 GB per full-resolution tensor): does anything
index with 32 bits?  determinism over two steps, finite gradients, sample independence (batch swap), one SGD step reduces the loss."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
import mt3d_amd  # noqa
def _mgr(patch, tasks, cin, batch, autoconfigure, model_config):      # what NetworkFromConfig reads from a ConfigManager
    from types import SimpleNamespace
    return SimpleNamespace(tasks=tasks, train_patch_size=tuple(patch), train_batch_size=batch, in_channels=cin, vram_max=16.0,
                           autoconfigure=autoconfigure, model_config=dict(model_config), verbose=False)
from mt3d_amd.builders.build_network_from_config import NetworkFromConfig
from mt3d_amd.training.losses.losses import BCEDiceLoss
P = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
patch = (P, P, P)
tasks = {"sheet": {"channels": 1, "activation": "none", "loss_fn": "BCEDiceLoss", "loss_kwargs": {"alpha": 0.5, "beta": 0.5}}}
mgr = _mgr(patch, tasks, 1, B, True, {})
torch.manual_seed(0)
net = NetworkFromConfig(mgr).cuda()
net.compute_dtype = torch.bfloat16
print("stages", net.num_stages, "features", net.features_per_stage, flush=True)
gen = torch.Generator(device="cuda").manual_seed(3)
x = torch.rand((B, 1, *patch), device="cuda", generator=gen)
t = (torch.rand((B, 1, *patch), device="cuda", generator=gen) > 0.8).float()
loss_fn = BCEDiceLoss(0.5, 0.5)
def run(xx, tt):
    net.zero_grad(set_to_none=True)
    out = net(xx)["sheet"]
    l = loss_fn(out, tt)
    l.backward()
    return out.detach().clone(), l.item(), {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
t0 = time.time()
o1, l1, g1 = run(x, t)
torch.cuda.synchronize(); print(f"first step {time.time() - t0:.1f} s, loss {l1:.5f}, mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
o2, l2, g2 = run(x, t)
assert torch.equal(o1, o2) and l1 == l2, "forward not deterministic"
for n in g1:
    assert torch.isfinite(g1[n]).all(), n
    assert torch.equal(g1[n], g2[n]), f"gradient not deterministic: {n}"
if B > 1:
    o3, l3, g3 = run(x.flip(0).contiguous(), t.flip(0).contiguous())
    assert torch.equal(o3.flip(0), o1), "sample independence (batch swap) violated"
if B > 2:      # tensors beyond 2^31 bytes: every sample alone (a batch-1 plan, far below the limit) must give the batch's logits
    net.eval()
    with torch.no_grad():
        whole = net(x)["sheet"]
        for i in range(B):
            one = net(x[i:i + 1].contiguous())["sheet"]
            d = (one - whole[i:i + 1]).abs().max().item()
            print(f"sample {i} alone vs in the batch: max |diff| {d:.3e}", flush=True)
            assert d <= 1e-2 * whole.abs().max().item(), (i, d)
    net.train()
opt = torch.optim.SGD(net.parameters(), lr=0.05)
losses = []
for _ in range(3):
    opt.zero_grad(set_to_none=True)
    l = loss_fn(net(x)["sheet"], t); l.backward(); opt.step(); losses.append(l.item())
assert losses[-1] < losses[0], losses
torch.cuda.synchronize(); t0 = time.time()
for _ in range(3):
    opt.zero_grad(set_to_none=True)
    l = loss_fn(net(x)["sheet"], t); l.backward(); opt.step()
torch.cuda.synchronize()
print(f"OK {P}^3 batch {B}: losses {losses}, {(time.time() - t0) / 3 * 1e3:.1f} ms per step (SGD), peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
