"""debug helper: per-parameter gradient error of the HIP engine vs the CPU oracle (fp64)."""
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
dtype = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}[sys.argv[2] if len(sys.argv) > 2 else "fp32"]
c = CASES[case]
mgr = oracle.make_mgr(c["patch"], c["tasks"], c["in_channels"], c["batch"], c["autoconfigure"], c["model_config"])
torch.manual_seed(c["seed"]); ref = oracle.NetworkFromConfig(mgr).double()
torch.manual_seed(c["seed"]); net = NetworkFromConfig(mgr).cuda()
net.compute_dtype = dtype
x, t = oracle.synthetic_batch(c["batch"], c["in_channels"], c["patch"], c["tasks"], c["data_seed"])
o_r = ref(x.double()); l_r = oracle.train_loss(o_r, {k: v.double() for k, v in t.items()}, c["tasks"]); l_r.backward()
o_n = net(x.cuda()); l_n = oracle.train_loss(o_n, {k: v.cuda() for k, v in t.items()}, c["tasks"]); l_n.backward()
for k in o_r:
    print("logits", k, "%.3e" % rel_l2(o_n[k].cpu(), o_r[k].detach()))
print("loss", l_r.item(), l_n.item())
pr, pn = dict(ref.named_parameters()), dict(net.named_parameters())
for n in pr:
    if pr[n].grad is None:
        continue
    print("%.3e |g|=%.3e %s" % (rel_l2(pn[n].grad.cpu(), pr[n].grad), pr[n].grad.norm().item(), n))
