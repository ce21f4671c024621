"""CPU: the numpy restatement of the reference's augmentation stack (dataloading/dataset.py:171-205 -> dataloading/augment.py).

PARITY UNPINNED: albumentations / volumentations are absent from this image, so the members cannot be compared with the real
classes.  What is checked: the structure the reference's source spells out (group probabilities, CoarseDropout3D's arguments, the
(Z, Y)-plane convention of handing a 3-D patch over as `image=`), value range, determinism under a seed, and that each member does
what its docstring says."""
import numpy as np
import pytest
import torch

import mt3d_amd  # noqa: F401
from mt3d_amd.dataloading import augment as A


def _patch(seed=0, shape=(16, 20, 12)):
    return np.random.default_rng(seed).random(shape, dtype=np.float32)


def test_stack_keeps_range_shape_dtype_and_is_seeded():
    x = _patch()
    a = A.augment_image(x, np.random.default_rng(5))
    b = A.augment_image(x, np.random.default_rng(5))
    assert a.shape == x.shape and a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    assert np.array_equal(a, b)
    assert a.min() >= 0.0 and a.max() <= 1.0
    outs = [A.augment_image(x, np.random.default_rng(s)) for s in range(40)]
    assert any(not np.array_equal(o, x) for o in outs) and any(np.array_equal(o, x) for o in outs)
    assert np.array_equal(x, _patch())                                   # the input is never modified in place


def test_group_probabilities_follow_the_reference_source():
    """p = 0.3 / 0.35 / 0.4 per OneOf group, uniform choice inside, 0.5 * 0.5 for the dropout (Compose p x transform default p)"""
    counts = {}
    real = {}

    def spy(name, fn):
        def f(img, rng):
            counts[name] = counts.get(name, 0) + 1
            return fn(img, rng)
        return f
    groups = tuple((p, tuple(spy(m.__name__, m) for m in members)) for p, members in A.GROUPS)
    real["cd"] = A.coarse_dropout_3d
    n_cd = [0]

    def cd(vol, rng, **kw):
        n_cd[0] += 1
        return real["cd"](vol, rng, **kw)
    old = A.GROUPS, A.coarse_dropout_3d
    A.GROUPS, A.coarse_dropout_3d = groups, cd
    try:
        rng = np.random.default_rng(11)
        x = _patch(shape=(8, 8, 4))
        N = 4000
        for _ in range(N):
            A.augment_image(x, rng)
    finally:
        A.GROUPS, A.coarse_dropout_3d = old
    per_group = [sum(counts.get(m.__name__, 0) for m in members) / N for _, members in old[0]]
    for got, (p, _) in zip(per_group, old[0]):
        assert abs(got - p) < 0.03, (per_group, counts)
    for p, members in old[0]:
        for m in members:
            assert abs(counts[m.__name__] / N - p / len(members)) < 0.03, counts
    assert abs(n_cd[0] / N - 0.25) < 0.03


def test_two_d_members_act_in_the_zy_plane_with_one_draw_for_all_x():
    """albumentations sees (Z, Y, X) as height x width x channels: a blur must mix neighbours along z and y, never along x"""
    x = np.zeros((41, 41, 6), np.float32)          # wider than the largest kernel (defocus: 21 x 21), so no mass folds back at the border
    x[20, 20, 2] = 1.0
    for fn in (A.motion_blur, A.defocus, A.advanced_blur):
        y = fn(x.copy(), np.random.default_rng(3))
        assert y[:, :, [0, 1, 3, 4, 5]].max() == 0.0, fn.__name__          # nothing leaks across x
        assert y[:, :, 2].sum() == pytest.approx(1.0, abs=1e-4), fn.__name__   # normalised kernels
        assert (y[:, :, 2] > 0).sum() >= 1
    r = np.random.default_rng(9)
    img = _patch(shape=(16, 16, 5))
    d = A.downscale(img, r)
    blocks = d.reshape(4, 4, 4, 4, 5)
    assert np.array_equal(blocks, np.broadcast_to(blocks[:, :1, :, :1], blocks.shape))     # 4 x 4 nearest-neighbour blocks in (z, y)
    assert not np.array_equal(d[..., 0], d[..., 1])                                        # x untouched
    il = A.illumination(np.full((12, 10, 3), 0.5, np.float32), np.random.default_rng(2))
    assert np.array_equal(il[..., 0], il[..., 2]) and il.std() > 0 and abs(il.mean() - 0.5) <= 0.5 * 0.2 + 1e-6


def test_member_ranges():
    x = np.full((8, 8, 8), 0.5, np.float32)
    for s in range(30):
        r = np.random.default_rng(s)
        m = A.multiplicative_noise(x.copy(), r)
        assert np.unique(m).size == 1 and 0.45 - 1e-6 <= m.flat[0] <= 0.55 + 1e-6
        b = A.random_brightness_contrast(x.copy(), r)
        assert np.unique(b).size == 1 and 0.5 * 0.8 - 0.2 - 1e-6 <= b.flat[0] <= 0.5 * 1.2 + 0.2 + 1e-6
    g = A.gauss_noise(np.full((32, 32, 32), 0.5, np.float32), np.random.default_rng(1))
    interior = g[(g > 0) & (g < 1)]
    assert 0.0 <= g.min() and g.max() <= 1.0 and 0.1 < interior.std() < 0.44


def test_coarse_dropout_boxes_follow_the_reference_arguments():
    vol = _patch(shape=(40, 50, 60)) * 0.4            # nothing equals the fill value beforehand
    for s in range(25):
        out = A.coarse_dropout_3d(vol, np.random.default_rng(s))
        hole = out == 0.5
        assert hole.any() and np.array_equal(out[~hole], vol[~hole])
        lab, n = __import__("scipy.ndimage", fromlist=["label"]).label(hole)
        assert 1 <= n <= 4                                                     # (overlapping boxes merge)
        zs, ys, xs = np.where(hole)
        # the union of <= 4 boxes, each at most 40 % per axis and at least 10 %
        assert hole.sum() <= 4 * int(40 * 0.4) * int(50 * 0.4) * int(60 * 0.4)
        assert hole.sum() >= int(40 * 0.1) * int(50 * 0.1) * int(60 * 0.1)
        assert zs.max() < 40 and ys.max() < 50 and xs.max() < 60


def test_dataset_applies_the_stack_to_the_image_only(tmp_path):
    from types import SimpleNamespace
    from mt3d_amd.dataloading import zarr_lite
    from mt3d_amd.dataloading.dataset import ZarrSegmentationDataset3D
    rng = np.random.default_rng(1)
    img = rng.integers(0, 255, size=(32, 32, 32), dtype=np.uint8)
    lab = (rng.random((32, 32, 32)) > 0.5).astype(np.uint8) * 255
    paths = {}
    for name, arr in (("img", img), ("sheet", lab)):
        paths[name] = str(tmp_path / f"{name}.zarr")
        zarr_lite.write_array(paths[name], arr, (16, 16, 16), compressor="zlib")
    mgr = SimpleNamespace(model_name="m", tasks={"sheet": {"channels": 1}}, train_patch_size=(16, 16, 16), min_labeled_ratio=0.1,
                          min_bbox_percent=0.5, dilate_label=False, use_cache=False, cache_folder=str(tmp_path),
                          volume_paths=[{"input": paths["img"], "sheet": paths["sheet"], "ref_label": "sheet"}],
                          dataset_config={"augment": "restated"})
    ds = ZarrSegmentationDataset3D(mgr)
    mgr.dataset_config = {"augment": False}
    raw = ZarrSegmentationDataset3D(mgr)
    A._rng = np.random.default_rng(3)
    changed = 0
    for i in range(len(ds)):
        a, r = ds[i], raw[i]
        assert a["image"].shape == r["image"].shape == (1, 16, 16, 16) and a["image"].dtype == torch.float32
        assert torch.equal(a["sheet"], r["sheet"])
        assert 0.0 <= float(a["image"].min()) and float(a["image"].max()) <= 1.0
        changed += int(not torch.equal(a["image"], r["image"]))
    assert changed > 0
    A._rng = None
    mgr.dataset_config = {"augment": "sometimes"}
    with pytest.raises(ValueError):
        ZarrSegmentationDataset3D(mgr)
