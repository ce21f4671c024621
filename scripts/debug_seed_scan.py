import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import torch
import resenc_oracle as oracle
from golden_cases import CASES
from helpers import rel_l2
import mt3d_amd
from mt3d_amd.builders.build_network_from_config import NetworkFromConfig
for case in sys.argv[1:]:
    c = CASES[case]
    mgr = oracle.make_mgr(c["patch"], c["tasks"], c["in_channels"], c["batch"], c["autoconfigure"], c["model_config"])
    for ds in [c["data_seed"], 21, 22, 23, 24, 25]:
        torch.manual_seed(c["seed"]); ref = oracle.NetworkFromConfig(mgr).double()
        torch.manual_seed(c["seed"]); net = NetworkFromConfig(mgr).cuda()
        x, t = oracle.synthetic_batch(c["batch"], c["in_channels"], c["patch"], c["tasks"], ds)
        o_r = ref(x.double()); oracle.train_loss(o_r, {k: v.double() for k, v in t.items()}, c["tasks"]).backward()
        o_n = net(x.cuda()); oracle.train_loss(o_n, {k: v.cuda() for k, v in t.items()}, c["tasks"]).backward()
        pr, pn = dict(ref.named_parameters()), dict(net.named_parameters())
        worst = max(rel_l2(pn[n].grad.cpu(), pr[n].grad) for n in pr if pr[n].grad is not None and pr[n].grad.norm() > 1e-6)
        print(case, "data_seed", ds, "worst grad rel %.2e" % worst, "logits %.2e" % max(rel_l2(o_n[k].cpu(), o_r[k].detach()) for k in o_r), flush=True)
