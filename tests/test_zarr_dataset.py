"""CPU: the minimal zarr v2 reader/writer and the zarr-backed patch dataset (dataloading/dataset.py:18-222 of the reference).
Parity unpinned for the FORMAT side: the `zarr` package is absent, so stores are written by zarr_lite itself (plus one
hand-written store laid out from the v2 spec); the dataset arithmetic is checked against a direct numpy restatement."""
import json
import os
import zlib
from types import SimpleNamespace

import numpy as np
import pytest
import torch

import mt3d_amd  # noqa: F401
from mt3d_amd.dataloading import zarr_lite
from mt3d_amd.dataloading.dataset import ZarrSegmentationDataset3D, find_label_bounding_box, find_valid_patches


@pytest.mark.parametrize("comp,sep", [(None, "."), ("zlib", "."), (None, "/")])
def test_roundtrip_and_slicing(tmp_path, comp, sep):
    rng = np.random.default_rng(0)
    a = rng.integers(0, 65535, size=(21, 17, 30), dtype=np.uint16)
    a[:8, :8, :8] = 0                                           # an all-fill chunk is not stored
    z = zarr_lite.write_array(str(tmp_path / "a.zarr"), a, (8, 8, 8), compressor=comp, dimension_separator=sep)
    assert z.shape == a.shape and z.dtype == a.dtype
    first = "0.0.0" if sep == "." else os.path.join("0", "0", "0")
    assert not os.path.exists(tmp_path / "a.zarr" / first)
    for sl in [np.s_[:, :, :], np.s_[3:19, 0:17, 5:29], np.s_[20:21, 16:17, 29:30], np.s_[4, 2:9, :], np.s_[0:0, :, :],
               np.s_[5:13], np.s_[..., 7]]:
        assert np.array_equal(z[sl], a[sl]), sl
    with pytest.raises(zarr_lite.ZarrLiteError):
        z[::2]


def test_hand_written_store_from_the_spec(tmp_path):
    # laid out by hand from the zarr v2 spec: F-order chunks, '/' separator, fill value 7, zlib, one chunk missing
    p = tmp_path / "h.zarr"
    os.makedirs(p / "0")
    os.makedirs(p / "1")
    meta = {"zarr_format": 2, "shape": [3, 4], "chunks": [2, 4], "dtype": "<i4", "order": "F", "fill_value": 7,
            "filters": None, "compressor": {"id": "zlib", "level": 1}, "dimension_separator": "/"}
    (p / ".zarray").write_text(json.dumps(meta))
    c0 = np.arange(8, dtype="<i4").reshape(2, 4)
    (p / "0" / "0").write_bytes(zlib.compress(c0.tobytes(order="F")))
    z = zarr_lite.open(str(p))
    want = np.full((3, 4), 7, dtype="<i4")
    want[:2] = c0
    assert np.array_equal(z[:, :], want)
    meta["compressor"] = {"id": "blosc", "cname": "lz4"}
    (p / ".zarray").write_text(json.dumps(meta))
    with pytest.raises(zarr_lite.ZarrLiteError):
        zarr_lite.open(str(p))


def _volume(tmp_path):
    rng = np.random.default_rng(1)
    D = 40
    img = rng.integers(0, 255, size=(D, D, D), dtype=np.uint8)
    lab = np.zeros((D, D, D), dtype=np.uint8)
    lab[6:36, 4:38, 5:37] = (rng.random((30, 34, 32)) > 0.6) * 255
    nrm = rng.integers(0, 65535, size=(D, D, D, 3), dtype=np.uint16)
    paths = {}
    for name, arr, ch in [("img", img, (16, 16, 16)), ("sheet", lab, (16, 16, 16)), ("normals", nrm, (16, 16, 16, 3))]:
        paths[name] = str(tmp_path / f"{name}.zarr")
        zarr_lite.write_array(paths[name], arr, ch, compressor="zlib")
    return img, lab, nrm, paths


def test_dataset_items_and_valid_patches(tmp_path):
    img, lab, nrm, paths = _volume(tmp_path)
    tasks = {"sheet": {"channels": 1}, "normals": {"channels": 3}}
    mgr = SimpleNamespace(model_name="m", tasks=tasks, train_patch_size=(16, 16, 16), min_labeled_ratio=0.1,
                          min_bbox_percent=0.9, dilate_label=False, use_cache=True, cache_folder=str(tmp_path / "cache"),
                          volume_paths=[{"input": paths["img"], "sheet": paths["sheet"], "normals": paths["normals"],
                                         "ref_label": "sheet"}])
    ZarrSegmentationDataset3D._warned = False
    with pytest.warns(RuntimeWarning, match="RESTATEMENT"):         # the restated augmentation stack is announced, once
        aug = ZarrSegmentationDataset3D(mgr)
    assert aug.augment
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        ZarrSegmentationDataset3D(mgr)                              # second construction: silent
        ZarrSegmentationDataset3D._warned = False
        mgr.dataset_config = {"augment": "restated"}                # acknowledged in the config: silent
        assert ZarrSegmentationDataset3D(mgr).augment
        mgr.dataset_config = {"augment": False}                     # raw patches: silent
        ds = ZarrSegmentationDataset3D(mgr)
        assert not ds.augment
    nz = np.argwhere(lab > 0)
    bbox = tuple(int(v) for ax in range(3) for v in (nz[:, ax].min(), nz[:, ax].max()))
    assert find_label_bounding_box(zarr_lite.open(paths["sheet"]), (16, 16, 16)) == bbox
    # restatement of the search on the in-memory label
    want = [p["start_pos"] for p in find_valid_patches(lab, (16, 16, 16), 0.9, 0.1)]
    assert len(ds) == len(want) > 0 and [p["start_pos"] for p in ds.all_valid_patches] == want
    assert os.path.exists(ds.cache_file)
    assert len(ZarrSegmentationDataset3D(mgr)) == len(ds)                     # second construction reads the cache
    it = ds[len(ds) // 2]
    z, y, x = ds.all_valid_patches[len(ds) // 2]["start_pos"]
    sl = np.s_[z:z + 16, y:y + 16, x:x + 16]
    assert it["image"].shape == (1, 16, 16, 16) and it["image"].dtype == torch.float32
    assert torch.equal(it["image"][0], torch.from_numpy(img[sl].astype(np.float32) / 255.0))
    assert it["sheet"].shape == (1, 16, 16, 16) and torch.equal(it["sheet"][0], torch.from_numpy(lab[sl].astype(np.float32) / 255.0))
    n_want = (nrm[sl].astype(np.float32) / 32767.5 - 1.0).transpose(3, 0, 1, 2)
    assert it["normals"].shape == (3, 16, 16, 16) and torch.equal(it["normals"], torch.from_numpy(np.ascontiguousarray(n_want)))
    mgr.dilate_label, mgr.use_cache = True, False
    d2 = ZarrSegmentationDataset3D(mgr)[0]["sheet"]
    assert set(np.unique(d2.numpy())) <= {0.0, 1.0} and d2.sum() >= (lab[:16, :16, :16] > 0).sum()
    mgr.volume_paths[0]["input"] = "http://example.invalid/x.zarr"
    with pytest.raises(ValueError):
        ZarrSegmentationDataset3D(mgr)
