"""Same network, same input, two builds of the plan: every fusion / specialised kernel of round 1 ON (default) vs OFF (environment
knobs).  usage: consistency_check.py dump OUT.pt  (run once per mode, the knobs are read once per process)  |  compare A.pt B.pt"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import torch
if sys.argv[1] == "errors":      # errors REF.pt A.pt B.pt ...: rel-L2 of selected tensors against a reference (e.g. the fp32 parity mode)
    ref = torch.load(sys.argv[2])
    for f in sys.argv[3:]:
        a = torch.load(f)
        rels = {k: ((a[k].double() - ref[k].double()).norm() / ref[k].double().norm().clamp(min=1e-30)).item() for k in ref}
        worst = sorted(rels.items(), key=lambda kv: -kv[1])[:4]
        print(os.path.basename(f), "worst vs reference:", [(k.split("/")[0] + "/" + k.split(".")[-3] + "." + k.split(".")[-1], round(v, 4)) for k, v in worst])
    sys.exit(0)
if sys.argv[1] == "compare":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    worst = 0.0
    for k in a:
        x, y = a[k].double(), b[k].double()
        if y.norm() < 1e-12:
            continue
        cos = (x.flatten() @ y.flatten() / (x.norm() * y.norm())).item()
        rel = ((x - y).norm() / y.norm()).item()
        worst = max(worst, rel)
        if rel > 5e-2 or cos < 0.998:
            print(f"MISMATCH {k}: rel {rel:.3e} cos {cos:.6f}")
            sys.exit(1)
    print(f"consistent: {len(a)} tensors, worst rel-L2 {worst:.2e}")
    sys.exit(0)
import mt3d_amd
import resenc_oracle as oracle
from mt3d_amd.builders.build_network_from_config import NetworkFromConfig
tasks = {"sheet": {"channels": 1, "activation": "none", "loss_fn": "BCEDiceLoss", "loss_kwargs": {"alpha": 0.5, "beta": 0.5}},
         "normals": {"channels": 3, "activation": "none", "loss_fn": "MaskedCosineLoss"}}
out = {}
for patch, batch in [((48, 80, 96), 3), ((64, 64, 64), 1)]:
    mgr = oracle.make_mgr(patch, tasks, 1, batch, True, {})
    torch.manual_seed(1)
    net = NetworkFromConfig(mgr).cuda()
    net.compute_dtype = torch.float32 if os.environ.get("CC_FP32") == "1" else torch.bfloat16
    x, t = oracle.synthetic_batch(batch, 1, patch, tasks, 3)
    o = net(x.cuda())
    loss = oracle.train_loss(o, {k: v.cuda() for k, v in t.items()}, tasks)
    loss.backward()
    tag = "x".join(map(str, patch))
    for k, v in o.items():
        out[f"{tag}/logits/{k}"] = v.detach().float().cpu()
    for n, p in net.named_parameters():
        if p.grad is not None:
            out[f"{tag}/grad/{n}"] = p.grad.detach().float().cpu()
torch.save(out, sys.argv[2])
print("dumped", len(out), "tensors")
