"""One tiny forward+backward of the hot path on the HIP engine, checked against the golden vector
made by the real reference and against the CPU oracle (used by __graft_entry__.smoke())."""
import torch

import resenc_oracle as oracle
from golden_cases import CASES
from helpers import load_golden, rel_l2


def run_smoke(device):
    import mt3d_amd  # noqa: F401
    from mt3d_amd.builders.build_network_from_config import NetworkFromConfig
    case = "auto16_2head"
    c = CASES[case]
    g = load_golden(case)
    mgr = oracle.make_mgr(c["patch"], c["tasks"], c["in_channels"], c["batch"], True, {})
    torch.manual_seed(c["seed"])
    net = NetworkFromConfig(mgr).to(device)
    torch.manual_seed(c["seed"])
    ref = oracle.NetworkFromConfig(mgr)
    x = torch.from_numpy(g["x"])
    targets = {k[len("target."):]: torch.from_numpy(v) for k, v in g.items() if k.startswith("target.")}
    out = net(x.to(device))
    o_ref = ref(x)
    for k in out:
        r_gold = rel_l2(out[k].cpu(), g[f"logits.{k}"])
        r_orc = rel_l2(out[k].cpu(), o_ref[k].detach())
        assert r_gold < 2e-4 and r_orc < 2e-4, (k, r_gold, r_orc)
    loss = oracle.train_loss(out, {k: v.to(device) for k, v in targets.items()}, c["tasks"])
    loss.backward()
    l_ref = oracle.train_loss(o_ref, targets, c["tasks"])
    l_ref.backward()
    assert abs(loss.item() - l_ref.item()) < 1e-4
    pn, pr = dict(net.named_parameters()), dict(ref.named_parameters())
    worst = 0.0
    for n in pr:
        if pr[n].grad is None:
            assert pn[n].grad is None
        elif pr[n].grad.norm() > 1e-6:
            worst = max(worst, rel_l2(pn[n].grad.cpu(), pr[n].grad))
    assert worst < 2e-3, worst
    print(f"smoke ok: loss {loss.item():.6f} (oracle {l_ref.item():.6f}), worst grad rel-l2 {worst:.2e}")
