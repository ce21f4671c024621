"""CPU: the oracle restatement reproduces every golden vector made from the REAL reference
(tests/golden/*.npz, generator: oracle/make_golden.py).  This is what pins the oracle on machines
where /root/reference does not exist (the GPU box)."""
import numpy as np
import pytest
import torch

import resenc_oracle as oracle
from golden_cases import CASES
from helpers import build_oracle, load_golden, rel_l2


@pytest.mark.parametrize("case", list(CASES))
def test_oracle_reproduces_reference_golden(case):
    g = load_golden(case)
    net, c, mgr = build_oracle(case)
    # same keys, same unique parameters, same seeded init as the reference
    assert sorted(net.state_dict().keys()) == g["state_dict_keys"]
    names = [n for n, _ in net.named_parameters()]
    assert names == g["param_names"]
    for (n, p), ck in zip(net.named_parameters(), g["init_checksums"]):
        assert p.numel() == int(ck[0]), n
        assert abs(p.detach().double().sum().item() - ck[1]) <= 1e-9 * max(1.0, abs(ck[1])), n
        assert abs(p.detach().double().norm().item() - ck[2]) <= 1e-9 * max(1.0, abs(ck[2])), n
    topo = g["topology"]
    assert net.num_stages == topo["num_stages"]
    assert list(net.features_per_stage) == topo["features_per_stage"]
    assert list(net.n_blocks_per_stage) == topo["n_blocks_per_stage"]

    x = torch.from_numpy(g["x"])
    targets = {k[len("target."):]: torch.from_numpy(v) for k, v in g.items() if k.startswith("target.")}
    # the synthetic generator is deterministic: the stored batch equals a regenerated one
    x2, t2 = oracle.synthetic_batch(c["batch"], c["in_channels"], c["patch"], c["tasks"], c["data_seed"])
    assert torch.equal(x, x2) and all(torch.equal(targets[k], t2[k]) for k in targets)

    net.train()
    out = net(x)
    for k, v in out.items():
        assert rel_l2(v, g[f"logits.{k}"]) < 1e-5, k
        # bit-exact decision map (north-star: exact argmax)
        ref = torch.from_numpy(g[f"logits.{k}"])
        if v.shape[1] == 1:
            assert torch.equal(v > 0, ref > 0)
        else:
            assert torch.equal(v.argmax(1), ref.argmax(1))
    loss = oracle.train_loss(out, targets, c["tasks"])
    assert abs(loss.item() - float(g["loss"])) < 1e-5
    loss.backward()
    for (n, p), ck in zip(net.named_parameters(), g["grad_checksums"]):
        if ck[0] == 0.0:
            assert p.grad is None, f"{n} must stay grad-less (unused deep-supervision head)"
            continue
        assert p.grad is not None, n
        l2 = p.grad.double().norm().item()
        assert abs(l2 - ck[2]) <= 2e-4 * max(ck[2], 1e-7), (n, l2, ck[2])
        key = f"grad.{n}"
        if key in g:
            assert rel_l2(p.grad, g[key]) < 2e-4 or ck[2] < 1e-6, n
    net.eval()
    with torch.no_grad():
        ev = net(x)
    for k, v in ev.items():
        assert rel_l2(v, g[f"eval.{k}"]) < 1e-5, k


def test_planner_known_answers():
    # SURVEY 3.3 [probe]: 64^3 -> 5 stages, 128^3 / 160^3 -> 6 stages, anisotropic (14,256,256)
    assert len(oracle.plan_pooling((64, 64, 64))[1]) == 5
    assert len(oracle.plan_pooling((128, 128, 128))[1]) == 6
    assert len(oracle.plan_pooling((160, 160, 160))[1]) == 6
    npool, strides, kernels = oracle.plan_pooling((14, 256, 256))
    assert strides[1] == (2, 2, 2) and strides[2] == (1, 2, 2)
    assert all(k == (3, 3, 3) for k in kernels)
    assert oracle.blocks_per_stage(6) == [1, 3, 4, 6, 6, 6]


def test_fp32_gradients_are_mask_discontinuous():
    """Why golden seeds are chosen with margin: the reference path's OWN fp32 gradients move by > 1e-3
    against its fp64 evaluation when a single LeakyReLU mask bit differs (data seed 29), and agree to
    ~5e-6 when none does (data seed 24).  Any implementation with a different fp32 summation order
    (threads, oneDNN version, MFMA) sees the same effect; the 1e-3 gradient bar therefore only holds
    on inputs whose masks have margin."""
    from golden_cases import CASES
    c = CASES["auto_aniso_bias"]
    mgr = oracle.make_mgr(c["patch"], c["tasks"], c["in_channels"], c["batch"], c["autoconfigure"], c["model_config"])

    def worst(data_seed):
        res = {}
        for dt in (torch.float32, torch.float64):
            torch.manual_seed(c["seed"])
            net = oracle.NetworkFromConfig(mgr).to(dt)
            x, t = oracle.synthetic_batch(c["batch"], c["in_channels"], c["patch"], c["tasks"], data_seed)
            out = net(x.to(dt))
            oracle.train_loss(out, {k: v.to(dt) for k, v in t.items()}, c["tasks"]).backward()
            res[dt] = {n: p.grad for n, p in net.named_parameters() if p.grad is not None}
        return max(rel_l2(res[torch.float32][n], res[torch.float64][n]) for n in res[torch.float32]
                   if res[torch.float64][n].norm() > 1e-6)

    assert worst(24) < 1e-4
    assert worst(29) > 1e-3
