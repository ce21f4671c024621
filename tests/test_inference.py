"""Sliding-window inference (**parity unpinned**, see oracle/inference_oracle.py).

CPU: the position rule (helpers.py:200-216) against hand-derived answers and the oracle restatement; blending / cast
arithmetic of the product against the numpy oracle on synthetic sums.
GPU: the whole pipeline (HIP engine forwards, device accumulators) against the numpy oracle driven by the CPU oracle
network with identical weights -- fp32 compute mode, tolerance 2e-4 on the blended floats, +-1 on the integer casts."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import inference_oracle as ioracle   # noqa: E402
import resenc_oracle as oracle      # noqa: E402


def _product():
    import mt3d_amd  # noqa: F401
    from mt3d_amd import inference
    return inference


def test_positions_known_answers():
    inf = _product()
    assert inf.generate_positions(0, 64, 16, 8) == [0, 8, 16, 24, 32, 40, 48]
    assert inf.generate_positions(0, 50, 16, 8) == [0, 8, 16, 24, 32, 34]          # last patch forced to end at 50
    assert inf.generate_positions(0, 16, 16, 8) == [0]
    assert inf.generate_positions(3, 100, 32, 20) == [3, 23, 43, 63, 68]
    for args in [(0, 41, 14, 7), (0, 129, 128, 64), (0, 200, 64, 48), (5, 77, 9, 4)]:
        assert inf.generate_positions(*args) == ioracle.generate_positions(*args)
    with pytest.raises(ValueError):
        inf.generate_positions(0, 10, 16, 8)
    pos = inf.all_positions((20, 16, 24), (16, 16, 16), 0.5)
    assert pos == [(z, 0, x) for z in (0, 4) for x in (0, 8)]


def test_blend_and_cast_match_oracle():
    inf = _product().SlidingWindowInferer
    rng = np.random.default_rng(0)
    cnt = rng.integers(0, 4, size=(6, 7, 8)).astype(np.float32)
    seg = rng.random((1, 6, 7, 8)).astype(np.float32) * cnt            # sums of probabilities
    nrm = rng.normal(size=(3, 6, 7, 8)).astype(np.float32) * cnt
    targets = {"sheet": {"channels": 1}, "normals": {"channels": 3}}
    for name, s in (("sheet", seg), ("normals", nrm)):
        b = inf.blend(name, torch.from_numpy(s), torch.from_numpy(cnt)).numpy()
        ref = s.copy()
        mask = cnt > 0
        if name == "normals":
            mag = np.sqrt((ref ** 2).sum(0)) + 1e-8
            for k in range(3):
                ref[k][mask] /= mag[mask]
            fin_ref = np.clip((ref + 1.0) / 2.0 * 65535.0, 0, 65535).astype(np.uint16)
        else:
            ref[..., mask] /= cnt[mask]
            fin_ref = np.clip(ref * 255.0, 0, 255).astype(np.uint8)
        assert np.allclose(b, ref, rtol=1e-6, atol=1e-7), name
        fin = inf.cast_final(name, torch.from_numpy(ref)).numpy().astype(fin_ref.dtype)
        assert np.array_equal(fin, fin_ref), name
    assert targets


@pytest.mark.gpu
def test_pipeline_matches_oracle_on_the_device():
    import mt3d_amd  # noqa: F401
    from mt3d_amd.builders.build_network_from_config import NetworkFromConfig
    from mt3d_amd.engine import lib
    lib.require_device()
    inf = _product()
    tasks = {"sheet": {"channels": 1, "activation": "sigmoid"}, "normals": {"channels": 3, "activation": "none"}}
    mgr = oracle.make_mgr((16, 16, 16), tasks, 1, 2, True, {})
    torch.manual_seed(3)
    ref_net = oracle.NetworkFromConfig(mgr).eval()
    torch.manual_seed(3)
    net = NetworkFromConfig(mgr).cuda()
    g = torch.Generator().manual_seed(5)
    vol = torch.rand((1, 20, 28, 36), generator=g)
    runner = inf.SlidingWindowInferer(net, tasks, (16, 16, 16), batch_size=2, overlap=0.5, compute_dtype=torch.float32)
    got = runner(vol)
    pos = inf.all_positions(vol.shape[1:], (16, 16, 16), 0.5)

    def predict(patches):
        with torch.no_grad():
            ref_net.train()          # logits (the activation is the inference loop's job)
            out = ref_net(torch.from_numpy(patches))
        return {k: v.numpy() for k, v in out.items()}
    blended, final = ioracle.sliding_window(vol.numpy(), predict, tasks, (16, 16, 16), 2, pos)
    for name in tasks:
        assert got[name].shape == blended[name].shape
        err = np.abs(got[name] - blended[name]).max()
        assert err < 2e-4, (name, err)
        d = np.abs(got[name + "_final"].astype(np.int64) - final[name].astype(np.int64)).max()
        assert d <= 1, (name, d)
    assert got["sheet_final"].dtype == np.uint8 and got["normals_final"].dtype == np.uint16


@pytest.mark.gpu
def test_inference_with_stochastic_depth_is_deterministic_and_eval_mode():
    """the inferer used to put the module into train mode to get raw logits, which also switched DropPath on
    (random residual branches dropped and rescaled at inference).  With `stochastic_depth_p > 0` two runs must agree
    bit for bit, equal the oracle's EVAL-mode network, and leave the module's mode untouched.  (DropPath / SqueezeExcite:
    parity unpinned -- third-party classes, the checker is the oracle's restatement.)"""
    import mt3d_amd  # noqa: F401
    from mt3d_amd.builders.build_network_from_config import NetworkFromConfig
    from golden_cases import _manual
    inf = _product()
    tasks = {"sheet": {"channels": 1, "activation": "sigmoid"}}
    mc = _manual(squeeze_excitation=True, stochastic_depth_p=0.5)
    mgr = oracle.make_mgr((16, 16, 16), tasks, 1, 2, False, mc)
    torch.manual_seed(3)
    ref_net = oracle.NetworkFromConfig(mgr).eval()
    torch.manual_seed(3)
    net = NetworkFromConfig(mgr).cuda().train()
    g = torch.Generator().manual_seed(6)
    vol = torch.rand((1, 16, 24, 32), generator=g)
    runner = inf.SlidingWindowInferer(net, tasks, (16, 16, 16), batch_size=2, overlap=0.5, compute_dtype=torch.float32)
    a = runner(vol)
    assert net.training                       # mode restored
    b = runner(vol)
    assert np.array_equal(a["sheet"], b["sheet"])
    pos = inf.all_positions(vol.shape[1:], (16, 16, 16), 0.5)

    def predict(patches):
        with torch.no_grad():
            out = ref_net(torch.from_numpy(patches))          # eval mode: DropPath is the identity, sigmoid applied by the net
        return {k: torch.logit(v.double().clamp(1e-12, 1 - 1e-12)).float().numpy() for k, v in out.items()}
    blended, _ = ioracle.sliding_window(vol.numpy(), predict, tasks, (16, 16, 16), 2, pos)
    assert np.abs(a["sheet"] - blended["sheet"]).max() < 2e-4
    # forward_logits == train-mode logits of a net without stochastic depth, whatever the mode
    x = torch.rand((2, 1, 16, 16, 16), generator=g).cuda()
    net.eval()
    l1 = net.forward_logits(x)["sheet"]
    with torch.no_grad():
        act = net(x)["sheet"]
    assert torch.allclose(torch.sigmoid(l1), act, atol=1e-6)


def test_output_store_layout_matches_the_reference_rules(tmp_path, monkeypatch):
    """the output side (reference inference.py:66-113, 214-263) on the CPU: `write_store` needs device accumulators, so the patch
    loop is replaced by synthetic sums / counts; what is checked is the store itself -- array names, shapes (single channel squeezed),
    chunking by patch size, dtypes, values readable through an independent reader path, the refusal to overwrite."""
    import json
    inf = _product()
    import mt3d_amd  # noqa: F401
    from mt3d_amd.dataloading import zarr_lite
    rng = np.random.default_rng(1)
    Z, Y, X = 20, 24, 40
    cnt = rng.integers(0, 3, size=(Z, Y, X)).astype(np.float32)
    sums = {"sheet": torch.from_numpy(rng.random((1, Z, Y, X)).astype(np.float32) * cnt),
            "normals": torch.from_numpy(rng.normal(size=(3, Z, Y, X)).astype(np.float32) * cnt)}
    targets = {"sheet": {"channels": 1, "activation": "sigmoid"}, "normals": {"channels": 3, "activation": "none"}}
    runner = inf.SlidingWindowInferer(model=None, targets=targets, patch_size=(16, 16, 16), device="cpu")
    monkeypatch.setattr(runner, "accumulate", lambda volume: (sums, torch.from_numpy(cnt)))
    store = runner.write_store(None, str(tmp_path / "out"))
    assert store.endswith("predictions.zarr") and json.load(open(os.path.join(store, ".zgroup"))) == {"zarr_format": 2}
    assert sorted(os.listdir(store)) == sorted([".zgroup", "sheet_sum", "sheet_count", "sheet_final", "normals_sum", "normals_count",
                                                "normals_final"])
    sheet = zarr_lite.open(os.path.join(store, "sheet_final"))
    assert sheet.shape == (Z, Y, X) and sheet.dtype == np.uint8 and tuple(sheet.chunks) == (16, 16, 16)
    nrm = zarr_lite.open(os.path.join(store, "normals_final"))
    assert nrm.shape == (3, Z, Y, X) and nrm.dtype == np.uint16 and tuple(nrm.chunks) == (3, 16, 16, 16)
    blended = zarr_lite.open(os.path.join(store, "sheet_sum"))[:, :, :]
    want = sums["sheet"][0].numpy().copy()
    want[cnt > 0] /= cnt[cnt > 0]
    assert np.allclose(blended, want, rtol=1e-6, atol=1e-7)
    assert np.array_equal(sheet[:, :, :], np.clip(want * 255.0, 0, 255).astype(np.uint8))
    assert np.array_equal(zarr_lite.open(os.path.join(store, "normals_count"))[:, :, :], cnt)
    with pytest.raises(FileExistsError):
        runner.write_store(None, str(tmp_path / "out"))


@pytest.mark.gpu
@pytest.mark.parametrize("i", range(8))
def test_random_volumes_match_the_oracle(i):
    """randomized version of the pipeline test: volume extents (at, just above and far above the patch, per axis), overlap,
    batch size (full and ragged last batch), 1-2 input channels, sigmoid / softmax / raw heads -- blended float volumes to 2e-4,
    the integer casts to 1 level"""
    import random
    import mt3d_amd  # noqa: F401
    from mt3d_amd.builders.build_network_from_config import NetworkFromConfig
    from mt3d_amd.engine import lib
    lib.require_device()
    inf = _product()
    rng = random.Random(1000 + i)
    patch = tuple(rng.choice([8, 16]) for _ in range(3))
    cin = rng.choice([1, 2])
    tasks = {"a": {"channels": rng.choice([1, 2, 3]), "activation": rng.choice(["sigmoid", "softmax", "none"])}}
    if tasks["a"]["channels"] == 1 and tasks["a"]["activation"] == "softmax":
        tasks["a"]["activation"] = "sigmoid"
    if rng.random() < 0.5:
        tasks["b"] = {"channels": 3, "activation": "none"}
    mgr = oracle.make_mgr(patch, tasks, cin, 2, True, {})
    torch.manual_seed(3 + i)
    ref_net = oracle.NetworkFromConfig(mgr).eval()
    torch.manual_seed(3 + i)
    net = NetworkFromConfig(mgr).cuda()
    dims = tuple(p + rng.choice([0, 1, 3, p // 2, p, 2 * p + 5]) for p in patch)
    overlap = rng.choice([0.25, 0.5, 0.75])
    bs = rng.choice([1, 2, 3, 4])
    vol = torch.rand((cin, *dims), generator=torch.Generator().manual_seed(5 + i))
    got = inf.SlidingWindowInferer(net, tasks, patch, batch_size=bs, overlap=overlap, compute_dtype=torch.float32)(vol)
    pos = inf.all_positions(dims, patch, overlap)

    def predict(patches):
        with torch.no_grad():
            ref_net.train()          # logits (the activation is the inference loop's job)
            out = ref_net(torch.from_numpy(patches))
        return {k: v.numpy() for k, v in out.items()}
    blended, final = ioracle.sliding_window(vol.numpy(), predict, tasks, patch, bs, pos)
    for name in tasks:
        assert got[name].shape == blended[name].shape, (dims, patch)
        err = np.abs(got[name] - blended[name]).max()
        assert err < 2e-4, (name, err, dims, patch, overlap, bs)
        d = np.abs(got[name + "_final"].astype(np.int64) - final[name].astype(np.int64)).max()
        assert d <= 1, (name, d)
