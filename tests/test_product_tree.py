"""CPU: the product's parameter-container tree is key-for-key / bit-for-bit the reference's
(pinned through the golden fixtures), its engine plan has the right structure, and nothing in the
product computes without the HIP device."""
import os

import pytest
import torch

import mt3d_amd  # noqa: F401
from golden_cases import CASES
from helpers import load_golden
from mt3d_amd.builders.build_network_from_config import NetworkFromConfig
from mt3d_amd.builders.utils import get_n_blocks_per_stage, get_pool_and_conv_props
from mt3d_amd.engine.lib import RxError
from mt3d_amd.engine.plan import Plan, UnsupportedConfig
import resenc_oracle as oracle


def build_product(case):
    c = CASES[case]
    mgr = oracle.make_mgr(c["patch"], c["tasks"], c["in_channels"], c["batch"], c["autoconfigure"], c["model_config"])
    torch.manual_seed(c["seed"])
    return NetworkFromConfig(mgr), c


@pytest.mark.parametrize("case", list(CASES))
def test_state_dict_keys_and_seeded_init_match_reference(case):
    g = load_golden(case)
    net, c = build_product(case)
    assert sorted(net.state_dict().keys()) == g["state_dict_keys"]
    assert [n for n, _ in net.named_parameters()] == g["param_names"]
    for (n, p), ck in zip(net.named_parameters(), g["init_checksums"]):
        assert p.numel() == int(ck[0]), n
        assert abs(p.detach().double().sum().item() - ck[1]) <= 1e-9 * max(1.0, abs(ck[1])), n
        assert abs(p.detach().double().norm().item() - ck[2]) <= 1e-9 * max(1.0, abs(ck[2])), n
    topo = g["topology"]
    assert net.num_stages == topo["num_stages"]
    assert list(net.features_per_stage) == topo["features_per_stage"]


@pytest.mark.parametrize("case", list(CASES))
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_plan_structure_on_meta_device(case, dtype):
    g = load_golden(case)
    net, c = build_product(case)
    shape = (c["batch"], c["in_channels"], *c["patch"])
    plan = Plan(net.to("meta"), shape, dtype, "meta", needs_grad=True)
    used = {id(p) for p in plan.params}
    names_with_grad = {n for n, ck in zip(g["param_names"], g["grad_checksums"]) if ck[0] == 1.0}
    names_used = {n for n, p in net.named_parameters() if id(p) in used}
    assert names_used == names_with_grad          # unused deep-supervision heads are not engine inputs
    assert len(plan.fwd) > 0 and len(plan.bwd) > 0
    assert set(plan.outputs) == set(c["tasks"])
    for name, info in c["tasks"].items():
        spatial = c["patch"] if len(c["patch"]) == 3 else (1, *c["patch"])      # a 2-D net runs with a unit Z axis inside
        assert tuple(plan.outputs[name].shape) == (c["batch"], info["channels"], *spatial)
    ev = Plan(net, shape, dtype, "meta", needs_grad=False)
    assert len(ev.bwd) == 0 and all(e["w_bwd"] is None for e in ev.packs)


def test_planner_matches_oracle():
    for patch in [(64, 64, 64), (128, 128, 128), (160, 160, 160), (14, 256, 256), (8, 32, 32), (256, 256)]:
        a = get_pool_and_conv_props((1.0,) * len(patch), patch, 4, 999999)
        b = oracle.plan_pooling(patch, 4, (1.0,) * len(patch), 999999)
        assert (a[0], a[1], a[2]) == (b[0], b[1], b[2])
    assert get_n_blocks_per_stage(6) == [1, 3, 4, 6, 6, 6]


def test_no_cpu_fallback_and_loud_rejections():
    net, c = build_product("auto16_2head")
    with pytest.raises(RxError):
        net(torch.zeros(2, 1, 16, 16, 16))
    with pytest.raises(RuntimeError):
        net.shared_encoder(torch.zeros(2, 1, 16, 16, 16))       # containers never compute
    mgr = oracle.make_mgr((16, 16, 16), {"a": {"channels": 1}}, autoconfigure=False, model_config={})
    with pytest.raises(ValueError):
        NetworkFromConfig(mgr)
    mgr = oracle.make_mgr((16, 16, 16), {"a": {"channels": 1}}, model_config={"dropout_op_kwargs": {"p": 1.0}})
    with pytest.raises(UnsupportedConfig):          # (0 < p < 1 runs natively: tests/golden/dropout.npz)
        Plan(NetworkFromConfig(mgr).to("meta"), (1, 1, 16, 16, 16), torch.float32, "meta", True)
    mgr = oracle.make_mgr((16, 16, 16), {"a": {"channels": 1}}, model_config={"dropout_op_kwargs": {"p": 0.5}})
    plan = Plan(NetworkFromConfig(mgr).to("meta"), (1, 1, 16, 16, 16), torch.float32, "meta", True)
    assert len(plan._drops) >= 5 and all(d["keep"].shape[0] == 1 for d in plan._drops)


def test_yaml_string_blocks_do_not_crash_like_the_reference():
    # the reference's `is` comparison breaks for YAML-loaded (non-interned) block names; fixed here
    name = "".join(["Basic", "BlockD"])
    mc = {"basic_encoder_block": name, "basic_decoder_block": "".join(["Conv", "Block"]), "bottleneck_block": name,
          "features_per_stage": [32, 64], "num_stages": 2, "n_blocks_per_stage": [1, 1], "kernel_sizes": [3, 3],
          "n_conv_per_stage_decoder": [1], "strides": [1, 2]}
    net = NetworkFromConfig(oracle.make_mgr((16, 16, 16), {"a": {"channels": 1}}, autoconfigure=False, model_config=mc))
    assert net.shared_encoder.is_residual


def test_squeeze_excite_and_droppath_containers_match_the_oracle_tree():
    """PARITY UNPINNED options (third-party SqueezeExcite / DropPath): same state_dict keys, same seeded init as the
    oracle's restatement, and the plan wires them (fc parameters registered, gated InstanceNorm steps emitted)."""
    mc = {"squeeze_excitation": True, "stochastic_depth_p": 0.1}
    mgr = oracle.make_mgr((16, 16, 16), {"a": {"channels": 1}}, model_config=mc)
    torch.manual_seed(3)
    ref = oracle.NetworkFromConfig(mgr)
    torch.manual_seed(3)
    net = NetworkFromConfig(mgr)
    sr, sn = ref.state_dict(), net.state_dict()
    assert list(sr.keys()) == list(sn.keys())
    assert any("squeeze_excitation.fc1.weight" in k for k in sn)
    for k in sr:
        assert torch.equal(sr[k], sn[k]), k
    blk = net.shared_encoder.stages[1].blocks[0]
    assert blk.apply_se and blk.apply_stochastic_depth and blk.squeeze_excitation.rd_channels == 8
    plan = Plan(net.to("meta"), (2, 1, 16, 16, 16), torch.float32, "meta", True)
    gated = [r for r in plan.enc_tape if r.kind == "inact" and r.a["gate"] is not None]
    n_blocks = sum(len(st.blocks) for st in net.shared_encoder.stages)
    assert len(gated) == n_blocks and all(g.a["gate"]["se"] is not None and g.a["gate"]["keep_x"] == 1 for g in gated)
    fc = {id(p) for n, p in net.named_parameters() if "squeeze_excitation" in n}
    assert fc <= {id(p) for p in plan.params}


def test_reference_task_files_behave_as_upstream():
    """"tasks/*.yaml drop in unchanged" (north star), checked where the reference tree exists (this container): every task file of the
    reference goes through the reference's ConfigManager AND ours -- same outcome (the old-schema files raise KeyError on both sides,
    as upstream does today), same attribute values; the files that configure a network build it here and plan onto the HIP engine
    (meta device: structure only).  The YAML text itself is never copied into this repository."""
    import contextlib
    import glob
    import io
    import ref_shim
    if not ref_shim.reference_available():
        pytest.skip("reference tree not present")
    ref_shim.import_reference()
    from configuration.config_manager import ConfigManager as RefManager          # the reference's own (imports cleanly)
    from mt3d_amd.configuration.config_manager import ConfigManager
    files = sorted(glob.glob(os.path.join(ref_shim.REFERENCE_ROOT, "tasks", "*.yaml")))
    assert len(files) >= 5
    built = 0
    for f in files:
        outcome = []
        for cls in (RefManager, ConfigManager):
            try:
                with contextlib.redirect_stdout(io.StringIO()):
                    try:
                        m = cls(f, verbose=False)
                    except TypeError:
                        m = cls(f)
                outcome.append(("ok", m))
            except Exception as e:      # noqa: BLE001
                outcome.append((type(e).__name__, None))
        assert outcome[0][0] == outcome[1][0], (os.path.basename(f), outcome[0][0], outcome[1][0])
        if outcome[0][0] != "ok":
            continue
        r, m = outcome[0][1], outcome[1][1]
        for attr in ("model_name", "autoconfigure", "train_patch_size", "train_batch_size", "in_channels", "gradient_accumulation",
                     "optimizer", "initial_lr", "weight_decay", "max_epoch", "tr_val_split", "min_labeled_ratio", "min_bbox_percent"):
            assert getattr(r, attr) == getattr(m, attr), (os.path.basename(f), attr)
        assert list(r.tasks) == list(m.tasks) and r.model_config == m.model_config
        try:
            net = NetworkFromConfig(m)
        except ValueError:
            # autoconfigure false without the manual topology keys: the reference raises the same ValueError
            # (build_network_from_config.py:85-148)
            with pytest.raises(ValueError):
                with contextlib.redirect_stdout(io.StringIO()):
                    ref_shim.build_reference_network(r)
            continue
        shape = (m.train_batch_size, m.in_channels, *m.train_patch_size)
        plan = Plan(net.to("meta"), shape, torch.bfloat16, "meta", needs_grad=True)
        assert len(plan.fwd) > 50 and len(plan.bwd) > 50 and set(plan.outputs) == set(m.tasks)
        built += 1
    assert built >= 2          # dumb.yaml (128^3, two heads, SE) and ink.yaml (14 x 256 x 256, SE)


def test_oversized_activation_is_refused_before_any_launch():
    """kernels address a SAMPLE of a tensor with 32 bits (and step between samples with 64): the plan refuses a tensor with more than
    2^31 bytes per sample instead of risking a device fault; the batch does not count (scripts/big_patch_check.py ran 256^3 at batch 4)"""
    mgr = oracle.make_mgr((384, 384, 384), {"a": {"channels": 1}}, 1, 1, True, {})
    with pytest.raises(UnsupportedConfig):
        Plan(NetworkFromConfig(mgr).to("meta"), (1, 1, 384, 384, 384), torch.bfloat16, "meta", True)
    mgr = oracle.make_mgr((256, 256, 256), {"a": {"channels": 1}}, 1, 4, True, {})
    Plan(NetworkFromConfig(mgr).to("meta"), (4, 1, 256, 256, 256), torch.bfloat16, "meta", True)       # 4.3 GB per tensor: fine
    Plan(NetworkFromConfig(mgr).to("meta"), (1, 1, 256, 256, 256), torch.bfloat16, "meta", False)      # fine
