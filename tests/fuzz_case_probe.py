"""TEST-SIDE DEBUGGING TOOL: one draw of tests/test_fuzz_gpu.py in fp32, per-parameter gradient cosine / norm ratio against the oracle.
usage: RX_FUZZ_SEED=<s> python tests/fuzz_case_probe.py <index>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import torch
import mt3d_amd  # noqa
import resenc_oracle as oracle
import test_fuzz_gpu as fz
from mt3d_amd.builders.build_network_from_config import NetworkFromConfig
i = int(sys.argv[1])
c = fz.configs()[i]
print(c)
mgr = oracle.make_mgr(c["patch"], c["tasks"], c["cin"], c["batch"], False, c["mc"])
torch.manual_seed(100 + i); ref = oracle.NetworkFromConfig(mgr)
torch.manual_seed(100 + i); net = NetworkFromConfig(mgr).cuda(); net.compute_dtype = torch.float32
x, t = oracle.synthetic_batch(c["batch"], c["cin"], c["patch"], c["tasks"], 7 + i)
o_r = ref(x); o_n = net(x.cuda())
t = fz.targets_for(c, o_r, 7 + i)
l_r = oracle.train_loss(o_r, t, c["tasks"]); l_n = oracle.train_loss(o_n, {k: v.cuda() for k, v in t.items()}, c["tasks"])
l_r.backward(); l_n.backward()
for k in o_r:
    print("logits", k, ((o_n[k].cpu() - o_r[k].detach()).norm() / o_r[k].detach().norm()).item())
pr, pn = dict(ref.named_parameters()), dict(net.named_parameters())
for n in pr:
    if pr[n].grad is None: continue
    a, b = pn[n].grad.double().flatten().cpu(), pr[n].grad.double().flatten()
    if b.norm() < 1e-9: continue
    print(f"{n:70s} cos {(a @ b / (a.norm() * b.norm())).item():.6f} ratio {(a.norm() / b.norm()).item():.4f} |ref| {b.norm().item():.3e} shape {tuple(pr[n].shape)}")
plan = next(iter(net._plans.values()))
print([ (r.kind, tuple(r.a['y'].act.dims), r.a['y'].act.c) if 'y' in r.a else r.kind for r in plan.enc_tape[:10]])
