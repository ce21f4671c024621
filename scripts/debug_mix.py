import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import torch, torch.nn.functional as F
import resenc_oracle as oracle
from golden_cases import CASES
from helpers import rel_l2
import mt3d_amd
from mt3d_amd.builders.build_network_from_config import NetworkFromConfig

c = CASES["auto_aniso_bias"]
mgr = oracle.make_mgr(c["patch"], c["tasks"], c["in_channels"], c["batch"], c["autoconfigure"], c["model_config"])
torch.manual_seed(c["seed"]); ref = oracle.NetworkFromConfig(mgr).double()
torch.manual_seed(c["seed"]); net = NetworkFromConfig(mgr).cuda()
x, t = oracle.synthetic_batch(c["batch"], c["in_channels"], c["patch"], c["tasks"], c["data_seed"])
blk = ref.shared_encoder.stages[2].blocks[1]
S = {}
def fh(name):
    def h(m, i, o):
        S[name + ".in"] = i[0].detach().clone(); S[name + ".out"] = o.detach().clone()
        o.register_hook(lambda g: S.__setitem__(name + ".g", g.clone()))
    return h
blk.register_forward_hook(fh("blk")); blk.conv1.register_forward_hook(fh("c1")); blk.conv2.conv.register_forward_hook(fh("y2"))
o = ref(x.double()); oracle.train_loss(o, {k: v.double() for k, v in t.items()}, c["tasks"]).backward()
o_n = net(x.cuda()); oracle.train_loss(o_n, {k: v.cuda() for k, v in t.items()}, c["tasks"]).backward()
plan = list(net._plans.values())[0]
tape = plan.enc_tape
ia, ca = tape[31].a, tape[30].a
mine = dict(g=ia["out"].gact.to_ncdhw().double().cpu(), y2=ia["y"].act.to_ncdhw().double().cpu(),
            res=ia["res"].act.to_ncdhw().double().cpu(), out=ia["out"].act.to_ncdhw().double().cpu(),
            a1=ca["x"].act.to_ncdhw().double().cpu(), ga1=ca["x"].gact.to_ncdhw().double().cpu(),
            gres=ia["res"].gact.to_ncdhw().double().cpu())
orc = dict(g=S["blk.g"], y2=S["y2.out"], res=S["blk.in"], out=S["blk.out"], a1=S["c1.out"], ga1=S["c1.g"])
for k in ("g", "y2", "res", "out", "a1", "ga1"):
    print(f"{k:4s} mine vs oracle rel {rel_l2(mine[k], orc[k]):.3e}   max abs diff {(mine[k]-orc[k]).abs().max().item():.3e}  max|oracle| {orc[k].abs().max().item():.3e}")
w = blk.conv2.conv.weight.detach()
def F_(g, y2, res):
    y = y2.clone().requires_grad_(True)
    out = F.leaky_relu(F.instance_norm(y, eps=1e-5) + res, 0.01)
    (dy,) = torch.autograd.grad(out, y, g)
    return F.conv_transpose3d(dy, w, padding=1), dy
for gn in ("mine", "orc"):
    for tn in ("mine", "orc"):
        G = (mine if gn == "mine" else orc)["g"]; T = mine if tn == "mine" else orc
        ga1, dy = F_(G, T["y2"], T["res"])
        print(f"g={gn:4s} tensors={tn:4s}: vs oracle ga1 {rel_l2(ga1, orc['ga1']):.3e}  vs my ga1 {rel_l2(ga1, mine['ga1']):.3e}")
d = mine["g"] - orc["g"]
print("g diff: per-channel mean/std of diff", (d.mean(dim=(2,3,4)).abs().max()).item(), d.std().item(), " g std", orc["g"].std().item())
