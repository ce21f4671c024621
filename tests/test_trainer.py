"""Trainer plug-in surface (reference train.py:19-120): hooks on CPU, a short real run on the GPU."""
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "tasks", "synthetic_sheet.yaml")


def test_hooks_and_config_on_cpu():
    import mt3d_amd  # noqa: F401
    from mt3d_amd.builders.build_network_from_config import NetworkFromConfig
    from mt3d_amd.configuration.config_manager import ConfigManager
    from mt3d_amd.train import BaseTrainer
    tr = BaseTrainer(CFG, verbose=False)
    assert isinstance(tr.mgr, ConfigManager) and tr.mgr.train_patch_size == (32, 32, 32) and tr.mgr.in_channels == 1
    model = tr._build_model()
    assert isinstance(model, NetworkFromConfig) and model.num_stages == 4
    assert set(tr._build_loss()) == {"sheet"}
    opt = tr._get_optimizer(model)
    assert isinstance(opt, torch.optim.AdamW) and isinstance(tr._get_scheduler(opt), torch.optim.lr_scheduler.CosineAnnealingLR)
    ds = tr._configure_dataset()
    item = ds[0]
    assert item["image"].shape == (1, 32, 32, 32) and item["sheet"].shape == (1, 32, 32, 32)
    train, val = tr._configure_dataloaders(ds)
    assert len(train) > 0 and len(val) > 0
    with pytest.raises(KeyError):
        import tempfile, yaml
        with tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False) as f:
            yaml.safe_dump({"tr_params": {}}, f)          # the reference's OLD schema -> KeyError, as upstream
        ConfigManager(f.name, verbose=False)


@pytest.mark.gpu
def test_short_training_run_learns_and_checkpoints(tmp_path):
    import yaml
    import mt3d_amd  # noqa: F401
    from mt3d_amd.train import BaseTrainer
    cfg = yaml.safe_load(open(CFG))
    cfg["tr_setup"]["ckpt_out_base"] = str(tmp_path / "ckpt")
    cfg["tr_setup"]["tensorboard_log_dir"] = str(tmp_path / "tb")
    cfg["tr_config"]["max_epoch"] = 3
    p = tmp_path / "cfg.yaml"
    yaml.safe_dump(cfg, open(p, "w"))
    os.chdir(tmp_path)

    class Rec(BaseTrainer):
        losses = []

        def _log(self, *a):
            s = " ".join(str(x) for x in a)
            if s.startswith("[Train]"):
                self.losses.append(float(s.split("sheet: ")[1].split(" ")[0]))

    tr = Rec(str(p), verbose=False)
    model = tr.train()
    assert len(tr.losses) == 3 and tr.losses[-1] < tr.losses[0]          # it learns the synthetic task
    assert tr.last_patches_per_sec > 0
    files = sorted(os.listdir(tmp_path / "ckpt"))
    assert files == ["synthetic_sheet_1.pth", "synthetic_sheet_2.pth", "synthetic_sheet_3.pth"]
    ck = torch.load(tmp_path / "ckpt" / files[-1], weights_only=True)
    assert set(ck) == {"model", "optimizer", "scheduler", "epoch"} and ck["epoch"] == 2
    assert sorted(ck["model"].keys()) == sorted(model.state_dict().keys())
    # resume: weights come back bit-exactly, also from a torch.compile-style `_orig_mod.` checkpoint
    cfg["tr_setup"]["checkpoint_path"] = str(tmp_path / "ckpt" / files[-1])
    cfg["tr_config"]["max_epoch"] = 3
    yaml.safe_dump(cfg, open(p, "w"))
    tr2 = BaseTrainer(str(p), verbose=False)
    m2 = tr2.train()                                                    # start_epoch == max_epoch -> no step
    for (k, a), (_, b) in zip(model.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a.cpu(), b.cpu()), k
    assert set(BaseTrainer._strip_compile_prefix({"_orig_mod.a": 1})) == {"a"}


@pytest.mark.gpu
def test_training_from_zarr_volumes_with_squeeze_excite(tmp_path):
    """BaseTrainer fed by ZarrSegmentationDataset3D (zarr v2 stores written by zarr_lite: the `zarr` package is absent) on a
    config with `squeeze_excitation: true` (tasks/dumb.yaml:53 of the reference): a smooth synthetic sheet is learnt."""
    import numpy as np
    import yaml
    import mt3d_amd  # noqa: F401
    from mt3d_amd.dataloading import zarr_lite
    from mt3d_amd.train import BaseTrainer
    rng = np.random.default_rng(0)
    D = 64
    z, y, x = np.meshgrid(np.arange(D), np.arange(D), np.arange(D), indexing="ij")
    sheet = (np.abs(((y + 6 * np.sin(x / 9.0) + 4 * np.cos(z / 7.0)) % 16) - 8) < 2.5)
    img = (sheet * 140 + rng.integers(0, 80, size=sheet.shape)).astype(np.uint8)
    zarr_lite.write_array(str(tmp_path / "img.zarr"), img, (32, 32, 32), compressor="zlib")
    zarr_lite.write_array(str(tmp_path / "sheet.zarr"), (sheet * 255).astype(np.uint8), (32, 32, 32), compressor="zlib")
    cfg = yaml.safe_load(open(CFG))
    cfg["tr_setup"].update(model_name="zarr_se", ckpt_out_base=str(tmp_path / "ckpt"), tensorboard_log_dir=str(tmp_path / "tb"))
    cfg["tr_config"].update(max_epoch=3, max_steps_per_epoch=8, max_val_steps_per_epoch=1, patch_size=[32, 32, 32],
                            engine_optimizer=True)          # the engine's AdamW kernel behind the optimizer hook
    cfg["model_config"] = {"conv_bias": False, "squeeze_excitation": True}
    cfg["dataset_config"].update(synthetic=False, min_labeled_ratio=0.05, min_bbox_percent=0.5, use_cache=True,
                                 cache_folder=str(tmp_path / "cache"),
                                 augment=False,       # 24 steps must show learning: raw patches (the stack: tests/test_augment.py)
                                 volume_paths=[{"input": str(tmp_path / "img.zarr"), "sheet": str(tmp_path / "sheet.zarr"),
                                                "ref_label": "sheet"}])
    p = tmp_path / "cfg.yaml"
    yaml.safe_dump(cfg, open(p, "w"))
    os.chdir(tmp_path)

    class Rec(BaseTrainer):
        losses = []

        def _log(self, *a):
            s = " ".join(str(x) for x in a)
            if s.startswith("[Train]"):
                self.losses.append(float(s.split("sheet: ")[1].split(" ")[0]))

    tr = Rec(str(p), verbose=False)
    from mt3d_amd.dataloading.dataset import ZarrSegmentationDataset3D
    assert isinstance(tr._configure_dataset(), ZarrSegmentationDataset3D)
    from mt3d_amd.training.optim import EngineAdamW
    assert isinstance(tr._get_optimizer(tr._build_model().cuda()), EngineAdamW)
    model = tr.train()
    assert any("squeeze_excitation.fc1.weight" in k for k in model.state_dict())
    assert len(tr.losses) == 3 and tr.losses[-1] < tr.losses[0]
    assert os.path.exists(tmp_path / "cache" / "zarr_se_32_32_32_cache.json")


@pytest.mark.gpu
def test_device_feeder_hands_over_the_loaders_batches_in_order():
    """`DeviceFeeder` (one batch ahead on a copy stream, pinned ring slots released by the copy stream's event) must be invisible:
    the same batches, in order, as float32 device tensors; ring slots are reused (more batches than slots) without corrupting a
    batch that has not been consumed yet; breaking out early leaves nothing hanging."""
    import mt3d_amd  # noqa: F401
    from torch.utils.data import DataLoader, Dataset
    from mt3d_amd.train import DeviceFeeder, PinnedRingCollate

    class Items(Dataset):
        def __len__(self):
            return 23

        def __getitem__(self, i):
            g = torch.Generator().manual_seed(i)
            return {"image": torch.randn((1, 8, 16, 16), generator=g), "sheet": (torch.rand((1, 8, 16, 16), generator=g) > 0.5).to(torch.uint8)}

    ds = Items()
    want = list(DataLoader(ds, batch_size=2, shuffle=False))
    loader = DataLoader(ds, batch_size=2, shuffle=False, collate_fn=PinnedRingCollate(depth=3), pin_memory=False)
    dev = torch.device("cuda", torch.cuda.current_device())
    got = []
    for b in DeviceFeeder(loader, dev):
        assert all(v.is_cuda and v.dtype == torch.float32 for v in b.values())
        got.append({k: v.clone() for k, v in b.items()})
        torch.cuda._sleep(2_000_000)          # the consumer is slow: the feeder is a batch ahead, the ring wraps around meanwhile
    assert len(got) == len(want) == 12
    for a, w in zip(got, want):
        for k in w:
            assert torch.equal(a[k].cpu(), w[k].float()), k
    for i, b in enumerate(DeviceFeeder(loader, dev)):
        if i == 2:
            break
    torch.cuda.synchronize()
