"""TEST INFRASTRUCTURE ONLY.  GPU-side half of the seed curation for tests/test_network_gpu.py::test_fp32_live_oracle_32cube_two_steps: for data seeds
that already have LeakyReLU mask margin on the CPU (oracle fp32 vs fp64 < 2e-5, /oracle/scan_seeds.py procedure), print
the engine's worst fp32 parameter-gradient distance from the fp64 oracle.  Usage: python oracle/seed_margin_gpu.py cube32|widened 99 103 ..."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import torch  # noqa: E402
import resenc_oracle as oracle  # noqa: E402
import mt3d_amd  # noqa: E402,F401
from mt3d_amd.builders.build_network_from_config import NetworkFromConfig  # noqa: E402


def rel_l2(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()
    return ((a - b).norm() / b.norm().clamp(min=1e-30)).item()


which = sys.argv[1]
if which == "cube32":       # tests/test_network_gpu.py::test_fp32_live_oracle_32cube_two_steps
    tasks = {"sheet": {"channels": 1, "activation": "sigmoid", "loss_fn": "BCEDiceLoss", "loss_kwargs": {"alpha": 0.5, "beta": 0.5}}}
    patch, cin, batch, wseed = (32, 32, 32), 1, 1, 5
    mgr = oracle.make_mgr(patch, tasks, cin, batch, True, {})
else:                       # "widened": tests/test_variants_gpu.py::test_widened_configs_six_inputs_twelve_class_head
    from golden_cases import _manual
    tasks = {"seg": {"channels": 12, "activation": "softmax", "loss_fn": "BCEDiceLoss", "loss_kwargs": {"alpha": 0.5, "beta": 0.5}}}
    patch, cin, batch, wseed = (16, 16, 16), 6, 2, 11
    mgr = oracle.make_mgr(patch, tasks, cin, batch, False, _manual())
for ds in [int(a) for a in sys.argv[2:]]:
    torch.manual_seed(wseed)
    ref = oracle.NetworkFromConfig(mgr).double()
    torch.manual_seed(wseed)
    net = NetworkFromConfig(mgr).cuda()
    x, t = oracle.synthetic_batch(batch, cin, patch, tasks, ds)
    o_r = ref(x.double())
    oracle.train_loss(o_r, {k: v.double() for k, v in t.items()}, tasks).backward()
    o_n = net(x.cuda())
    oracle.train_loss(o_n, {k: v.cuda() for k, v in t.items()}, tasks).backward()
    pr, pn = dict(ref.named_parameters()), dict(net.named_parameters())
    worst = max((rel_l2(pn[n].grad, pr[n].grad), n) for n in pr if pr[n].grad is not None and pr[n].grad.norm() > 1e-6)
    print(f"data_seed {ds}: worst gradient rel-L2 vs fp64 oracle {worst[0]:.2e} ({worst[1]}), logits {max(rel_l2(o_n[k], o_r[k].detach()) for k in o_r):.2e}",
          flush=True)
