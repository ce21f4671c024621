"""CPU: every draw of the randomized GPU tests (tests/test_fuzz_gpu.py) builds its execution plan on the meta device -- forward and
backward launch lists, padded buffers, shadow parameters -- in both compute types; structural invariants only (no kernel runs)."""
import pytest
import torch

import mt3d_amd  # noqa: F401
import resenc_oracle as oracle
import test_fuzz_gpu as fz
from mt3d_amd.builders.build_network_from_config import NetworkFromConfig
from mt3d_amd.engine.plan import Plan, UnsupportedConfig


def _all_draws():
    return ([("small", i, c) for i, c in enumerate(fz.configs())] + [("medium", i, c) for i, c in enumerate(fz.medium_configs())]
            + [("large", i, c) for i, c in enumerate(fz.large_configs())])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_every_fuzz_draw_plans_on_the_meta_device(dtype):
    built = 0
    for kind, i, c in _all_draws():
        mgr = oracle.make_mgr(c["patch"], c["tasks"], c["cin"], c["batch"], False, c["mc"])
        net = NetworkFromConfig(mgr)
        shape = (c["batch"], c["cin"], *c["patch"])
        try:
            plan = Plan(net.to("meta"), shape, dtype, "meta", needs_grad=True)
        except UnsupportedConfig:
            continue
        built += 1
        assert set(plan.outputs) == set(c["tasks"]), (kind, i)
        for name, info in c["tasks"].items():
            out = plan.outputs[name]
            assert out.shape[0] == c["batch"] and out.shape[1] == info["channels"], (kind, i, name)
        assert len(plan.fwd) > 0 and len(plan.bwd) > 0
        used = {id(p) for p in plan.params}
        # every parameter except the unused deep-supervision heads is an engine input
        for n, p in net.named_parameters():
            assert (id(p) in used) or ".seg_layers." in n, (kind, i, n)
        for e in plan._shadows:              # a shadow is at least as large as its parameter in every dimension
            assert all(a >= b for a, b in zip(e["sh"].shape, e["param"].shape)), (kind, i)
        ev = Plan(net, shape, dtype, "meta", needs_grad=False)
        assert len(ev.bwd) == 0
    assert built >= 55, built
