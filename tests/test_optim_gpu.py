"""GPU (-m gpu): `EngineAdamW` -- AdamW update + gradient clipping + weight re-pack in one pass (csrc rx_adamw_pack /
rx_adamw_flat) -- against `torch.optim.AdamW(fused=True)` + `clip_grad_norm_` + the engine's own re-pack."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import resenc_oracle as oracle
from golden_cases import CASES
from helpers import rel_l2


def _setup():
    import mt3d_amd  # noqa: F401
    from mt3d_amd.builders.build_network_from_config import NetworkFromConfig
    from mt3d_amd.engine import lib
    lib.require_device()
    c = CASES["auto16_2head"]
    mgr = oracle.make_mgr(c["patch"], c["tasks"], c["in_channels"], c["batch"], c["autoconfigure"], c["model_config"])
    torch.manual_seed(c["seed"])
    net = NetworkFromConfig(mgr).cuda()
    x, targets = oracle.synthetic_batch(c["batch"], c["in_channels"], c["patch"], c["tasks"], 7)
    return net, c, mgr, x.cuda(), {k: v.cuda() for k, v in targets.items()}, NetworkFromConfig


def _train(net, c, x, targets, opt, steps, clip):
    losses = []
    for _ in range(steps):
        net.train()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = net(x)
        loss = oracle.train_loss(out, targets, c["tasks"])
        loss.backward()
        clip(opt)
        opt.step()
        opt.zero_grad(set_to_none=True)
        losses.append(loss.item())
    return losses


def test_engine_adamw_tracks_torch_adamw():
    from mt3d_amd.training.optim import EngineAdamW
    net_a, c, mgr, x, targets, N = _setup()
    torch.manual_seed(c["seed"])
    net_b = N(mgr).cuda()
    pa, pb = [p for p in net_a.parameters()], [p for p in net_b.parameters()]
    oa = torch.optim.AdamW(pa, lr=1e-3, weight_decay=0.01, fused=True)
    ob = EngineAdamW(pb, model=net_b, lr=1e-3, weight_decay=0.01)
    # ONE step from identical weights and (bitwise) identical gradients: the update itself to fp32 round-off, clipping active
    _train(net_a, c, x, targets, oa, 1, lambda o: torch.nn.utils.clip_grad_norm_(pa, 0.05))
    _train(net_b, c, x, targets, ob, 1, lambda o: o.clip_grad_norm(0.05))
    for (n, a), b in zip(net_a.named_parameters(), pb):
        if a.grad is None and not oa.state.get(a):
            assert torch.equal(a, b), n
            continue
        assert (a - b).abs().max().item() <= 2e-6 * max(1.0, a.abs().max().item()), n
    sa, sb = oa.state[pa[0]], ob.state[pb[0]]
    assert rel_l2(sb["exp_avg"].cpu(), sa["exp_avg"].cpu()) < 1e-5 and rel_l2(sb["exp_avg_sq"].cpu(), sa["exp_avg_sq"].cpu()) < 1e-5
    # the packed copies written by the optimizer equal a fresh pack of the updated parameters -> a fresh module agrees
    fresh = N(mgr).cuda()
    fresh.load_state_dict(net_b.state_dict())

    def fwd(m):
        m.train()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            return m(x)
    o1, o2 = fwd(net_b), fwd(fresh)
    for k in o1:
        assert torch.equal(o1[k], o2[k]), k
    plan = [p for p in net_b._plans.values() if p.needs_grad][0]
    assert all(e.get("event") is None for e in plan.packs)


def test_engine_adamw_trains_like_torch_over_several_steps():
    from mt3d_amd.training.optim import EngineAdamW
    net_a, c, mgr, x, targets, N = _setup()
    torch.manual_seed(c["seed"])
    net_b = N(mgr).cuda()
    pa, pb = [p for p in net_a.parameters()], [p for p in net_b.parameters()]
    la = _train(net_a, c, x, targets, torch.optim.AdamW(pa, lr=1e-3, weight_decay=0.0, fused=True), 6,
                lambda o: torch.nn.utils.clip_grad_norm_(pa, 3))
    ob = EngineAdamW(pb, model=net_b, lr=1e-3, weight_decay=0.0)
    lb = _train(net_b, c, x, targets, ob, 6, lambda o: o.clip_grad_norm(3))
    assert la[-1] < la[0]
    for a, b in zip(la, lb):          # Adam amplifies round-off on noise-level gradients: trajectories agree loosely
        assert abs(a - b) < 3e-2 * max(1.0, abs(a)), (la, lb)
    sd = ob.state_dict()
    assert set(sd["state"][0]) >= {"step", "exp_avg", "exp_avg_sq"}


def test_clip_and_step_matches_clip_grad_norm_then_step():
    """training/optim.clip_and_step: clip(3) + fused AdamW with the clip scale inside the update kernel == the two calls"""
    import mt3d_amd  # noqa: F401
    from mt3d_amd.training.optim import clip_and_step
    torch.manual_seed(0)
    shapes = [(64, 32, 3, 3, 3), (64,), (7, 5), (128, 64, 1, 1, 1)]
    pa = [torch.nn.Parameter(torch.randn(s, device="cuda")) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    oa = torch.optim.AdamW(pa, lr=1e-2, weight_decay=0.01, fused=True)
    ob = torch.optim.AdamW(pb, lr=1e-2, weight_decay=0.01, fused=True)
    for it in range(4):
        scale = 10.0 if it % 2 == 0 else 0.01           # alternately clipped and not clipped
        for a, b in zip(pa, pb):
            g = torch.randn_like(a) * scale
            a.grad, b.grad = g.clone(), g.clone()
        na = clip_and_step(oa, pa, 3)
        nb = torch.nn.utils.clip_grad_norm_(pb, 3)
        ob.step()
        assert abs(na.item() - nb.item()) <= 1e-5 * nb.item()
        for a, b in zip(pa, pb):
            assert torch.allclose(a, b, rtol=2e-6, atol=2e-7), it
            assert torch.allclose(a.grad, b.grad, rtol=2e-6, atol=1e-8)      # .grad holds the clipped gradient afterwards
    assert not hasattr(oa, "grad_scale") and not hasattr(oa, "found_inf")


def test_engine_adamw_without_a_plan_clips_and_matches_torch():
    """EngineAdamW(model=None): every parameter goes through rx_adamw_flat with the clip coefficient -- what bench.py runs"""
    import mt3d_amd  # noqa: F401
    from mt3d_amd.training.optim import EngineAdamW, clip_and_step
    torch.manual_seed(1)
    shapes = [(64, 32, 3, 3, 3), (64,), (7, 5), (128, 64, 1, 1, 1), (3, 32, 1, 1, 1)]
    pa = [torch.nn.Parameter(torch.randn(s, device="cuda")) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    oa = EngineAdamW(pa, model=None, lr=1e-2, weight_decay=0.01)
    ob = torch.optim.AdamW(pb, lr=1e-2, weight_decay=0.01, fused=True)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(oa, T_max=4)
    sched_b = torch.optim.lr_scheduler.CosineAnnealingLR(ob, T_max=4)
    for it in range(4):
        scale = 10.0 if it % 2 == 0 else 0.01
        for a, b in zip(pa, pb):
            g = torch.randn_like(a) * scale
            a.grad, b.grad = g.clone(), g.clone()
        na = clip_and_step(oa, pa, 3)
        nb = torch.nn.utils.clip_grad_norm_(pb, 3)
        ob.step()
        sched.step(); sched_b.step()
        assert abs(float(na) - float(nb)) <= 1e-5 * float(nb)
        for a, b in zip(pa, pb):
            assert torch.allclose(a, b, rtol=3e-6, atol=3e-7), it
    sd = oa.state_dict()
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}


def test_engine_adamw_table_kernel_many_ragged_tensors():
    """rx_adamw_flat_multi's table kernel: more tensors than one launch's table holds (48), sizes around the 4096-element
    chunk (full chunks on 16-byte loads, ragged tails, tiny tensors) and a parameter whose storage is not 16-byte aligned."""
    import mt3d_amd  # noqa: F401
    from mt3d_amd.training.optim import EngineAdamW
    torch.manual_seed(2)
    sizes = [1, 3, 255, 4095, 4096, 4097, 8192, 12289, 70001] + [17 + 131 * i for i in range(50)]
    base = torch.randn(40000, device="cuda")
    odd = torch.nn.Parameter(base[1:1 + 20001])                   # data_ptr() % 16 == 4
    assert odd.data_ptr() % 16 != 0
    pa = [torch.nn.Parameter(torch.randn(n, device="cuda")) for n in sizes] + [odd]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    oa = EngineAdamW(pa, model=None, lr=1e-2, weight_decay=0.01)
    ob = torch.optim.AdamW(pb, lr=1e-2, weight_decay=0.01, fused=True)
    for it in range(3):
        for a, b in zip(pa, pb):
            g = torch.randn_like(a)
            a.grad, b.grad = g.clone(), g.clone()
        oa.step(); ob.step()
        for a, b in zip(pa, pb):
            assert torch.allclose(a, b, rtol=3e-6, atol=3e-7), (it, a.numel())
    assert torch.equal(base[0], base[0]) and torch.isfinite(base).all()
    for a, b in zip(pa[:3], pb[:3]):
        assert torch.allclose(oa.state[a]["exp_avg_sq"], ob.state[b]["exp_avg_sq"], rtol=1e-5, atol=1e-9)


def test_engine_grad_norm_clip_matches_torch():
    """rx_grad_norm_clip (EngineAdamW.clip_grad_norm): global L2 norm and min(1, max_norm / (norm + 1e-6)) over more tensors than one
    table launch holds, ragged sizes and a 16-byte-misaligned gradient; deterministic"""
    import mt3d_amd  # noqa: F401
    from mt3d_amd.training.optim import EngineAdamW
    torch.manual_seed(3)
    sizes = [1, 5, 4095, 4096, 4097, 100003] + [31 + 97 * i for i in range(55)]
    ps = [torch.nn.Parameter(torch.zeros(n, device="cuda")) for n in sizes]
    base = torch.randn(9000, device="cuda")
    for p in ps:
        p.grad = torch.randn_like(p) * 0.3
    ps.append(torch.nn.Parameter(torch.zeros(8191, device="cuda")))
    ps[-1].grad = base[1:8192]
    assert ps[-1].grad.data_ptr() % 16 != 0
    opt = EngineAdamW(ps, model=None, lr=1e-3)
    for max_norm in (3.0, 1e4):
        ref = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(p.grad.double()) for p in ps])).item()
        got = opt.clip_grad_norm(max_norm)
        coef = opt._clip.clone()
        got2 = opt.clip_grad_norm(max_norm)
        assert got.item() == got2.item() and torch.equal(coef, opt._clip)          # fixed summation order
        assert abs(got.item() - ref) <= 2e-6 * ref
        assert abs(coef.item() - min(1.0, max_norm / (ref + 1e-6))) <= 3e-6
        opt._clip = None
