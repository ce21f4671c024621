"""debug helper: per-InstanceNorm (mean, rstd) of the HIP engine vs the fp64 oracle, in call order."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import torch
import resenc_oracle as oracle
from golden_cases import CASES
import mt3d_amd
from mt3d_amd.builders.build_network_from_config import NetworkFromConfig

case = sys.argv[1] if len(sys.argv) > 1 else "auto_aniso_bias"
c = CASES[case]
mgr = oracle.make_mgr(c["patch"], c["tasks"], c["in_channels"], c["batch"], c["autoconfigure"], c["model_config"])
torch.manual_seed(c["seed"]); ref = oracle.NetworkFromConfig(mgr).double()
torch.manual_seed(c["seed"]); net = NetworkFromConfig(mgr).cuda()
x, t = oracle.synthetic_batch(c["batch"], c["in_channels"], c["patch"], c["tasks"], c["data_seed"])
rec = []
def hook(m, inp, out):
    v = inp[0]
    mean = v.mean(dim=tuple(range(2, v.dim())))
    var = v.var(dim=tuple(range(2, v.dim())), unbiased=False)
    rec.append((mean, (var + m.eps).rsqrt(), var, out.detach()))
for m in ref.modules():
    if isinstance(m, torch.nn.InstanceNorm3d):
        m.register_forward_hook(hook)
ref(x.double())
out = net(x.cuda())
plan = list(net._plans.values())[0]
recs = [r for tape in [plan.enc_tape] + plan.dec_tapes for r in tape if r.kind == "inact"]
print(len(rec), len(recs))
for i, (r, (mean, rstd, var, o)) in enumerate(zip(recs, rec)):
    st = r.a["stats"].double().cpu()
    em = (st[..., 0] - mean).abs().max().item()
    er = ((st[..., 1] - rstd) / rstd).abs().max().item()
    y = r.a["y"].act.to_ncdhw().double().cpu()
    print(f"{i:3d} C={mean.shape[1]:4d} dims={tuple(y.shape[2:])} |mean err|={em:.2e} rel rstd err={er:.2e} "
          f"min var={var.min().item():.3e} max mean^2/var={(mean**2/var).max().item():.2e}")
