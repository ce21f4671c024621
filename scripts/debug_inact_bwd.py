"""debug helper: re-run the fused IN+LReLU(+residual) backward kernel on the engine's real buffers
and compare with torch autograd (fp64, CPU)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import torch, torch.nn.functional as F
import resenc_oracle as oracle
from golden_cases import CASES
from helpers import rel_l2
import mt3d_amd
from mt3d_amd.builders.build_network_from_config import NetworkFromConfig
from mt3d_amd.engine import ops

case = sys.argv[1] if len(sys.argv) > 1 else "auto_aniso_bias"
c = CASES[case]
mgr = oracle.make_mgr(c["patch"], c["tasks"], c["in_channels"], c["batch"], c["autoconfigure"], c["model_config"])
torch.manual_seed(c["seed"]); net = NetworkFromConfig(mgr).cuda()
x, t = oracle.synthetic_batch(c["batch"], c["in_channels"], c["patch"], c["tasks"], c["data_seed"])
o_n = net(x.cuda()); oracle.train_loss(o_n, {k: v.cuda() for k, v in t.items()}, c["tasks"]).backward()
plan = list(net._plans.values())[0]
recs = [r for tape in [plan.enc_tape] + plan.dec_tapes for r in tape if r.kind == "inact"]
for i, r in enumerate(recs):
    a = r.a
    y = a["y"].act.to_ncdhw().double().cpu().requires_grad_(True)
    res = a["res"].act.to_ncdhw().double().cpu() if a["res"] is not None else None
    g = a["out"].gact.to_ncdhw().double().cpu()
    pre = F.instance_norm(y, eps=a["eps"]) + (res if res is not None else 0)
    out = F.leaky_relu(pre, a["slope"]) if a["slope"] != 1.0 else pre
    (dy_ref,) = torch.autograd.grad(out, y, g)
    st = a["stats"].double().cpu()
    mean = y.detach().mean(dim=(2, 3, 4)); var = y.detach().var(dim=(2, 3, 4), unbiased=False)
    est = ((st[..., 1] - (var + a["eps"]).rsqrt()) / (var + a["eps"]).rsqrt()).abs().max().item()
    fo = rel_l2(a["out"].act.to_ncdhw().double().cpu(), out.detach())
    dy = ops.Act.zeros(*a["y"].act.dims, a["y"].act.c, a["y"].act.dtype)
    ops.instnorm_act_bwd(a["out"].gact, a["y"].act, a["stats"], a["out"].act if a["slope"] != 1.0 else None, dy, a["slope"], None, False)
    near0 = (out.detach().abs() < 1e-6).sum().item()
    print(f"{i:3d} dims={tuple(y.shape[1:])} res={'y' if res is not None else 'n'} stats_err={est:.1e} out_err={fo:.1e} dy_err={rel_l2(dy.to_ncdhw().double().cpu(), dy_ref):.2e} near0={near0}")
