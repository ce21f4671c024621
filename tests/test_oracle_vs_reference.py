"""CPU, build container only: the oracle against the REAL reference imported from /root/reference
(skipped wherever that tree does not exist, e.g. the GPU box -- the golden fixtures cover it there)."""
import pytest
import torch

import ref_shim
import resenc_oracle as oracle
from golden_cases import CASES, TASKS_2HEAD

pytestmark = pytest.mark.skipif(not ref_shim.reference_available(), reason="reference tree not present")


def _pair(patch, tasks, in_channels=1, autoconfigure=True, model_config=None, seed=3):
    mgr = ref_shim.make_mgr(patch, tasks, in_channels, 1, autoconfigure, model_config)
    torch.manual_seed(seed)
    ref = ref_shim.build_reference_network(mgr)
    torch.manual_seed(seed)
    mine = oracle.NetworkFromConfig(mgr)
    return ref, mine


@pytest.mark.parametrize("case", list(CASES))
def test_state_dict_and_forward_backward_equal(case):
    c = CASES[case]
    ref, mine = _pair(c["patch"], c["tasks"], c["in_channels"], c["autoconfigure"], c["model_config"], c["seed"])
    sr, sm = ref.state_dict(), mine.state_dict()
    assert list(sr.keys()) == list(sm.keys())          # same keys in the same order
    for k in sr:
        assert torch.equal(sr[k], sm[k]), k            # same seeded init, bit for bit
    _, ref_losses = ref_shim.import_reference()
    x, targets = oracle.synthetic_batch(c["batch"], c["in_channels"], c["patch"], c["tasks"], c["data_seed"])
    ref.train(); mine.train()
    torch.manual_seed(77)                              # (channel dropout draws its masks from the global generator)
    o_r = ref(x)
    torch.manual_seed(77)
    o_m = mine(x)
    for k in o_r:
        assert torch.equal(o_r[k], o_m[k]), k          # identical op sequence -> identical bits
    # loss restatement vs the reference's loss classes
    l_m = oracle.train_loss(o_m, targets, c["tasks"])
    l_r = 0.0
    for name, gt in targets.items():
        info = c["tasks"][name]
        cls = getattr(ref_losses, info.get("loss_fn", "BCEDiceLoss"))
        l_r = l_r + cls(**info.get("loss_kwargs", {}))(o_r[name], gt) * info.get("weight", 1.0)
    assert abs(l_r.item() - l_m.item()) < 1e-6
    l_r.backward(); l_m.backward()
    for (n1, p1), (n2, p2) in zip(ref.named_parameters(), mine.named_parameters()):
        assert n1 == n2
        assert (p1.grad is None) == (p2.grad is None), n1
        if p1.grad is not None:
            assert torch.allclose(p1.grad, p2.grad, rtol=1e-4, atol=1e-7), n1


def test_autoconfig_topology_matches_reference_for_baseline_patches():
    for patch in [(64, 64, 64), (128, 128, 128), (160, 160, 160), (14, 256, 256), (8, 32, 32), (256, 256)]:
        mgr = ref_shim.make_mgr(patch, TASKS_2HEAD)
        ref = ref_shim.build_reference_network(mgr).to("meta") if False else None  # building 213M params is slow
        import builders.utils as ru
        _, s_ref, k_ref, _, _ = ru.get_pool_and_conv_props((1.0,) * len(patch), patch, 4, 999999)
        _, s_m, k_m = oracle.plan_pooling(patch, 4, (1.0,) * len(patch), 999999)
        assert s_ref == s_m and k_ref == k_m, patch
        assert ru.get_n_blocks_per_stage(len(s_ref)) == oracle.blocks_per_stage(len(s_m))
