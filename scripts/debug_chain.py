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

c = CASES["auto_aniso_bias"]
mgr = oracle.make_mgr(c["patch"], c["tasks"], c["in_channels"], c["batch"], c["autoconfigure"], c["model_config"])
torch.manual_seed(c["seed"]); net = NetworkFromConfig(mgr).cuda()
x, t = oracle.synthetic_batch(c["batch"], c["in_channels"], c["patch"], c["tasks"], c["data_seed"])
o_n = net(x.cuda()); oracle.train_loss(o_n, {k: v.cuda() for k, v in t.items()}, c["tasks"]).backward()
plan = list(net._plans.values())[0]
tape = plan.enc_tape
for i, r in enumerate(tape):
    a = r.a
    desc = r.kind
    if r.kind == "conv":
        desc += f" x={a['x'].name} y={a['y'].name} k={a['kernel']} s={a['stride']} w={tuple(a['pk']['param'].shape)}"
    if r.kind == "inact":
        desc += f" y={a['y'].name} out={a['out'].name} res={a['res'].name if a['res'] is not None else None} slope={a['slope']}"
    if r.kind == "pool":
        desc += f" x={a['x'].name} y={a['y'].name}"
    print(i, desc)
# replay stage-2 block-1: find the inact with res whose out dims (4,8,8), second such
cands = [i for i, r in enumerate(tape) if r.kind == "inact" and r.a["res"] is not None and r.a["out"].act.dims[1:] == (4, 8, 8)]
print("residual inacts at (4,8,8):", cands)
i = cands[1]
ia, ca = tape[i].a, tape[i - 1].a
assert tape[i - 1].kind == "conv"
dy = ops.Act.zeros(*ia["y"].act.dims, ia["y"].act.c, torch.float32)
ops.instnorm_act_bwd(ia["out"].gact, ia["y"].act, ia["stats"], ia["out"].act, dy, ia["slope"], None, False)
gx = ops.Act.zeros(*ca["x"].act.dims, ca["x"].act.c, torch.float32)
ops.conv3d_bwd_data(dy, ca["pk"]["w_bwd"], gx, ca["kernel"], ca["stride"], False)
print("replayed g(a1) vs stored:", rel_l2(gx.to_ncdhw().cpu(), ca["x"].gact.to_ncdhw().cpu()))
w = ca["pk"]["param"].detach().double().cpu()
ref = F.conv_transpose3d(dy.to_ncdhw().double().cpu(), w, padding=1)
print("replayed g(a1) vs torch:", rel_l2(gx.to_ncdhw().cpu(), ref), " stored vs torch:", rel_l2(ca["x"].gact.to_ncdhw().cpu(), ref))
wb = ca["pk"]["w_bwd"].double().cpu()
print("w_bwd pack ok:", torch.equal(wb, w.reshape(w.shape[0], w.shape[1], 27).permute(2, 1, 0)))
