"""GPU (-m gpu): topology variants of the reference's config surface against the CPU oracle (fp32 mode, live):
BottleneckD encoder, ResidualBlock decoder, plain-conv encoder (the reference's `basic_encoder_block:
"ResidualBlock"` quirk), ReLU, 2-D networks, multi-conv decoder stages, and the cfg5-style 320-cap / 2-input /
non-power-of-two (20^3) manual topology in fp16."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import resenc_oracle as oracle
from helpers import rel_l2

from golden_cases import CASES, UNPINNED_CASES, _manual as manual

ONE = {"sheet": {"channels": 1, "activation": "sigmoid", "loss_fn": "BCEDiceLoss", "loss_kwargs": {"alpha": 0.5, "beta": 0.5}}}

# every variant is a curated case (oracle/golden_cases.py): seeds chosen with LeakyReLU/ReLU mask margin by oracle/scan_seeds.py,
# so the live-oracle gradient comparison carries the north-star 1e-3 bar.  The first eight are ALSO golden cases: the oracle is
# pinned bit-for-bit to the real reference on them (tests/test_oracle_vs_reference.py, tests/golden/<name>.npz) and
# tests/test_network_gpu.py::test_fp32_matches_reference_golden replays the fixtures on the engine.  The squeeze_excite* ones
# are PARITY UNPINNED (third-party SqueezeExcite: checked against the oracle's restatement of its published source).
VARIANTS = {k: CASES[k] for k in ("bottleneck_enc", "resdec_softmax", "plain_relu_2conv", "two_d", "aniso_kernels", "no_stem", "no_stem_plain", "odd_channels")}
VARIANTS.update(UNPINNED_CASES)


@pytest.fixture(scope="module")
def NetworkFromConfig():
    import mt3d_amd  # noqa: F401
    from mt3d_amd.builders.build_network_from_config import NetworkFromConfig as N
    return N


def _run(NetworkFromConfig, patch, cin, tasks, mc, autoconf=False, batch=2, seed=11, dtype=torch.float32, data_seed=5):
    mgr = oracle.make_mgr(patch, tasks, cin, batch, autoconf, mc)
    torch.manual_seed(seed)
    ref = oracle.NetworkFromConfig(mgr)
    torch.manual_seed(seed)
    net = NetworkFromConfig(mgr).cuda()
    assert list(ref.state_dict().keys()) == list(net.state_dict().keys())
    x, t = oracle.synthetic_batch(batch, cin, patch, tasks, data_seed)
    net.compute_dtype = dtype
    o_r, o_n = ref(x), net(x.cuda())
    l_r = oracle.train_loss(o_r, t, tasks)
    l_n = oracle.train_loss(o_n, {k: v.cuda() for k, v in t.items()}, tasks)
    l_r.backward()
    l_n.backward()
    return ref, net, o_r, o_n, l_r, l_n


@pytest.mark.parametrize("name", list(VARIANTS))
def test_variant_fp32_matches_oracle(NetworkFromConfig, name):
    v = VARIANTS[name]
    ref, net, o_r, o_n, l_r, l_n = _run(NetworkFromConfig, v["patch"], v["in_channels"], v["tasks"], v["model_config"],
                                        batch=v["batch"], seed=v["seed"], data_seed=v["data_seed"])
    for k in o_r:
        assert o_n[k].shape == o_r[k].shape
        assert rel_l2(o_n[k].cpu(), o_r[k].detach()) < 2e-4, (name, k)
    assert abs(l_r.item() - l_n.item()) < 1e-4
    pr, pn = dict(ref.named_parameters()), dict(net.named_parameters())
    for n in pr:
        assert (pr[n].grad is None) == (pn[n].grad is None), n
        if pr[n].grad is not None and pr[n].grad.norm() > 1e-6:
            assert rel_l2(pn[n].grad.cpu(), pr[n].grad) < 1e-3, (name, n, rel_l2(pn[n].grad.cpu(), pr[n].grad))
    ref.eval(); net.eval()
    x, _ = oracle.synthetic_batch(v["batch"], v["in_channels"], v["patch"], v["tasks"], v["data_seed"])
    with torch.no_grad():
        e_r, e_n = ref(x), net(x.cuda())
    for k in e_r:
        assert rel_l2(e_n[k].cpu(), e_r[k]) < 2e-4, (name, k)


def test_cfg5_style_320cap_two_inputs_fp16(NetworkFromConfig):
    """BASELINE configs[4] shrunk: manual 6-stage topology capped at 320 features, 2 input channels, patch with
    non-power-of-two sizes down to a 1^3... here 40^3 -> bottleneck 5^3 is emulated by 20^3 -> stages 20,10,5."""
    mc = manual(features_per_stage=[32, 64, 320], n_blocks_per_stage=[1, 2, 2])
    ref, net, o_r, o_n, l_r, l_n = _run(NetworkFromConfig, (20, 20, 20), 2, ONE, mc, batch=1, dtype=torch.float16)
    assert rel_l2(o_n["sheet"].cpu(), o_r["sheet"].detach()) < 8e-3
    pr, pn = dict(ref.named_parameters()), dict(net.named_parameters())
    for n in pr:
        if pr[n].grad is not None and pr[n].grad.norm() > 1e-5:
            a, b = pn[n].grad.double().flatten().cpu(), pr[n].grad.double().flatten()
            assert (a @ b / (a.norm() * b.norm())).item() > 0.98, n


def test_unsupported_configs_fail_loudly(NetworkFromConfig):
    from mt3d_amd.engine.plan import UnsupportedConfig
    mgr = oracle.make_mgr((16, 16, 16), ONE, 1, 1, False, manual(nonlin="nn.Tanh"))      # (odd channel counts run: odd_channels)
    with pytest.raises((UnsupportedConfig, ValueError, KeyError, AttributeError, TypeError)):
        net = NetworkFromConfig(mgr).cuda()
        net(torch.zeros(1, 1, 16, 16, 16, device="cuda"))
    # (round 3: any in_channels, 5- / 7-wide kernels and strides 3 / 4 run -- goldens wide_in_stem, big_kernels, stride4_mixed)
    mgr = oracle.make_mgr((16, 16, 16), ONE, 1, 1, False, manual(kernel_sizes=[[9, 3, 3], [3, 3, 3], [3, 3, 3]]))
    net = NetworkFromConfig(mgr).cuda()
    with pytest.raises(UnsupportedConfig):                                                  # kernels wider than 7 have no tap table
        net(torch.zeros(1, 1, 16, 16, 16, device="cuda"))


def test_droppath_training_and_eval(NetworkFromConfig):
    """PARITY UNPINNED (third-party DropPath).  Training: the engine's per-sample factors are forced to known values
    (one sample dropped in the first block, kept and rescaled elsewhere) and handed to the oracle's DropPath; eval: identity."""
    patch, tasks = (16, 16, 16), ONE
    mc = manual(squeeze_excitation=True, stochastic_depth_p=0.2)
    mgr = oracle.make_mgr(patch, tasks, 1, 2, False, mc)
    torch.manual_seed(5)
    ref = oracle.NetworkFromConfig(mgr)
    torch.manual_seed(5)
    net = NetworkFromConfig(mgr).cuda()
    net.compute_dtype = torch.float32
    from mt3d_amd.engine import plan as plan_mod
    forced = []
    blocks_r = [m for m in ref.modules() if isinstance(m, oracle.DropPath)]

    def draw(self, g):
        if g["scale"] is None or not self.net.training:
            return None
        i = len(forced)
        v = torch.tensor([0.0, 1.25] if i == 0 else [1.25, 1.25], dtype=torch.float32)
        forced.append(v)
        g["scale"].copy_(v)
        return g["scale"]
    orig = plan_mod.Plan._draw_path_scale
    plan_mod.Plan._draw_path_scale = draw
    try:
        x, t = oracle.synthetic_batch(2, 1, patch, tasks, 5)
        o_n = net(x.cuda())
        assert len(forced) == len(blocks_r) > 0
        for m, v in zip(blocks_r, forced):
            m.forced_scale = v
        o_r = ref(x)
        l_r = oracle.train_loss(o_r, t, tasks)
        l_n = oracle.train_loss(o_n, {k: v.cuda() for k, v in t.items()}, tasks)
        l_r.backward()
        l_n.backward()
    finally:
        plan_mod.Plan._draw_path_scale = orig
    for k in o_r:
        assert rel_l2(o_n[k].cpu(), o_r[k].detach()) < 2e-4
    pr, pn = dict(ref.named_parameters()), dict(net.named_parameters())
    for n in pr:
        assert (pr[n].grad is None) == (pn[n].grad is None), n
        if pr[n].grad is not None and pr[n].grad.norm() > 1e-6:
            # (data seed 5 has mask margin for this net and these forced factors: oracle fp32 vs fp64 2.3e-6)
            assert rel_l2(pn[n].grad.cpu(), pr[n].grad) < 1e-3, (n, rel_l2(pn[n].grad.cpu(), pr[n].grad))
    # the random draw itself: per-sample values in {0, 1/keep}
    net(x.cuda())
    plan = next(iter(net._plans.values()))
    scales = [r.a["gate"]["scale"] for r in plan.enc_tape if r.kind == "inact" and r.a["gate"] is not None]
    assert len(scales) == len(blocks_r)
    for sc in scales:
        assert all(abs(v) < 1e-6 or abs(v - 1.25) < 1e-6 for v in sc.cpu().tolist())
    ref.eval(); net.eval()
    with torch.no_grad():
        e_r, e_n = ref(x), net(x.cuda())
    for k in e_r:
        assert rel_l2(e_n[k].cpu(), e_r[k]) < 2e-4


@pytest.mark.parametrize("classes,data_seed", [(12, 1), (40, 1), (117, 1)])      # 117: more than 64 classes (round 3: chunks of 64 + a softmax pass)
def test_widened_configs_six_inputs_twelve_class_head(NetworkFromConfig, classes, data_seed):
    """round 2 widening: in_channels up to 8 (the stem's VALU kernels above 4) and task heads up to 64
    channels (softmax over 12 / 40 classes here; above 16 the head's weight gradient runs in chunks of 16 output channels),
    against the oracle in fp32 mode.  The multi-channel BCE-Dice loss takes torch's path (the HIP loss kernels cover C <= 8)."""
    tasks = {"seg": {"channels": classes, "activation": "softmax", "loss_fn": "BCEDiceLoss", "loss_kwargs": {"alpha": 0.5, "beta": 0.5}}}
    ref, net, o_r, o_n, l_r, l_n = _run(NetworkFromConfig, (16, 16, 16), 6, tasks, manual(), batch=2, seed=11, data_seed=data_seed)
    assert rel_l2(o_n["seg"].cpu(), o_r["seg"].detach()) < 2e-4
    assert torch.equal(o_n["seg"].cpu().argmax(1), o_r["seg"].argmax(1))
    assert abs(l_r.item() - l_n.item()) < 1e-4
    pr, pn = dict(ref.named_parameters()), dict(net.named_parameters())
    for n in pr:
        assert (pr[n].grad is None) == (pn[n].grad is None), n
        if pr[n].grad is not None and pr[n].grad.norm() > 1e-6:
            # (data seed 1 has mask margin for this net on the CPU -- oracle fp32 vs fp64 2.5e-6 -- and on the engine, 1.8e-6
            # against the fp64 oracle: oracle/seed_margin_gpu.py widened; seed 5 flips one mask in the engine's summation order)
            # (117 classes: the data seed is not curated for mask margin -- 3e-3 there)
            assert rel_l2(pn[n].grad.cpu(), pr[n].grad) < (1e-3 if classes <= 64 else 3e-3), (n, rel_l2(pn[n].grad.cpu(), pr[n].grad))
    ref.eval(); net.eval()
    x, _ = oracle.synthetic_batch(2, 6, (16, 16, 16), tasks, 1)
    with torch.no_grad():
        e_r, e_n = ref(x), net(x.cuda())
    assert rel_l2(e_n["seg"].cpu(), e_r["seg"]) < 2e-4
