"""debug helper: gradient w.r.t. every residual-block output, HIP engine vs fp64 oracle."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import torch
import resenc_oracle as oracle
from golden_cases import CASES
from helpers import rel_l2
import mt3d_amd
from mt3d_amd.builders.build_network_from_config import NetworkFromConfig

case = sys.argv[1] if len(sys.argv) > 1 else "auto_aniso_bias"
c = CASES[case]
mgr = oracle.make_mgr(c["patch"], c["tasks"], c["in_channels"], c["batch"], c["autoconfigure"], c["model_config"])
torch.manual_seed(c["seed"]); ref = oracle.NetworkFromConfig(mgr).double()
torch.manual_seed(c["seed"]); net = NetworkFromConfig(mgr).cuda()
x, t = oracle.synthetic_batch(c["batch"], c["in_channels"], c["patch"], c["tasks"], c["data_seed"])
grads, acts, names = [], [], []
def mk(name):
    def hook(m, inp, out):
        i = len(grads); grads.append(None); acts.append(out.detach()); names.append(name)
        out.register_hook(lambda g, i=i: grads.__setitem__(i, g))
    return hook
for n, m in ref.named_modules():
    if isinstance(m, (oracle.BasicBlockD, oracle.ConvDropoutNormReLU)) and ".encoder." not in n:
        m.register_forward_hook(mk(n))
o_r = ref(x.double()); oracle.train_loss(o_r, {k: v.double() for k, v in t.items()}, c["tasks"]).backward()
o_n = net(x.cuda()); oracle.train_loss(o_n, {k: v.cuda() for k, v in t.items()}, c["tasks"]).backward()
plan = list(net._plans.values())[0]
recs = [r for tape in [plan.enc_tape] + plan.dec_tapes for r in tape if r.kind == "inact"]
# oracle call order: a block's hook fires after its inner CDNRs; engine order: inact records (skip, conv1, conv2+res fused)
# match by shape & value instead: for each engine inact out, find the oracle activation with the smallest forward error
for i, r in enumerate(recs):
    out = r.a["out"]
    a = out.act.to_ncdhw().double().cpu()
    best = min(range(len(acts)), key=lambda j: rel_l2(a, acts[j]) if acts[j].shape == a.shape else 1e9)
    fe = rel_l2(a, acts[best])
    ge = rel_l2(out.gact.to_ncdhw().double().cpu(), grads[best]) if out.gact is not None and grads[best] is not None else float("nan")
    print(f"{i:3d} {names[best]:55s} fwd {fe:.2e} grad {ge:.2e} res={'y' if r.a['res'] is not None else 'n'}")
