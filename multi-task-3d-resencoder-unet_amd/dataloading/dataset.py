"""Patch feeders with the reference's item contract (dataloading/dataset.py:103-222): a dict with
"image" (C, Z, Y, X) float32 and one float32 tensor per task.

`SyntheticPatchDataset` generates SURVEY 8(d)'s synthetic patches (what the headline metric is quoted on).
`ZarrSegmentationDataset3D` reads real zarr v2 volumes through `zarr_lite` (the `zarr` package is absent here): same
constructor, valid-patch search, cache file, dtype scaling and item layout as the reference; its augmentations need
albumentations / volumentations (absent): `dataloading/augment.py` restates the stack in numpy (parity unpinned) and is applied
by default with ONE warning saying so (`dataset_config.augment: "restated"` acknowledges it, `false` feeds raw patches); remote
(http) stores are refused (no network)."""
import json
import os
import warnings
from pathlib import Path

import numpy as np
import torch
from torch.utils.data import Dataset

from . import zarr_lite


class SyntheticPatchDataset(Dataset):
    """SURVEY 8(d)'s synthetic patches.  `dataset_config.synthetic_pool` (default 16) distinct patches are generated once per
    process and handed out round-robin: drawing 2 x 128^3 uniform numbers per item costs ~60 ms of one CPU core, which would make
    a 128^3 training run loader-bound at ~8 patches/s; 0 = draw every item afresh.  With a pool the trainer's train / validation
    split (train.py:99-120 mirror) hands out the SAME pool to both sides: synthetic validation numbers measure throughput and
    plumbing, not generalisation (set `synthetic_pool: 0` for disjoint samples)."""

    def __init__(self, mgr, length=None, seed=1234):
        self.mgr = mgr
        self.patch = tuple(mgr.train_patch_size)
        self.cin = mgr.in_channels
        self.tasks = mgr.tasks
        self.length = int(length if length is not None else mgr.dataset_config.get("synthetic_length", 64))
        self.seed = seed
        self.pool = int(mgr.dataset_config.get("synthetic_pool", 16))
        self._cache = {}

    def __len__(self):
        return self.length

    def __getitem__(self, idx):
        if self.pool > 0:
            key = int(idx) % self.pool
            if key not in self._cache:
                self._cache[key] = self._make(key)
            return dict(self._cache[key])
        return self._make(idx)

    def _make(self, idx):
        g = torch.Generator().manual_seed(self.seed + int(idx))
        item = {"image": torch.rand((self.cin, *self.patch), generator=g)}
        seg = (torch.rand((1, *self.patch), generator=g) > 0.8).float()
        for name, info in self.tasks.items():
            c = info["channels"]
            if info.get("loss_fn", "BCEDiceLoss") == "MaskedCosineLoss":
                v = torch.randn((c, *self.patch), generator=g)
                item[name] = (v / v.norm(dim=0, keepdim=True).clamp(min=1e-8)) * seg
            else:
                item[name] = seg.expand(c, *self.patch).contiguous()
        return item


def find_label_bounding_box(arr, chunk_shape=(192, 192, 192)):
    """helpers.py:71-131: (minz, maxz, miny, maxy, minx, maxx) of the non-zero voxels, read chunk by chunk"""
    D, H, W = arr.shape
    lo, hi = [D, H, W], [-1, -1, -1]
    for z0 in range(0, D, chunk_shape[0]):
        for y0 in range(0, H, chunk_shape[1]):
            for x0 in range(0, W, chunk_shape[2]):
                ch = arr[z0:z0 + chunk_shape[0], y0:y0 + chunk_shape[1], x0:x0 + chunk_shape[2]]
                if ch.any():
                    nz = np.argwhere(ch > 0) + np.array([z0, y0, x0])
                    for d in range(3):
                        lo[d], hi[d] = min(lo[d], int(nz[:, d].min())), max(hi[d], int(nz[:, d].max()))
    return lo[0], hi[0], lo[1], hi[1], lo[2], hi[2]


def find_valid_patches(arr, patch_size, bbox_threshold=0.97, label_threshold=0.10):
    """helpers.py:136-198 (+ _check_patch_chunk :38-68), single process: half-patch stride inside the label bounding box; a
    patch is kept when the bounding box of its labelled voxels covers >= bbox_threshold of the patch and >= label_threshold
    of its voxels are labelled."""
    pZ, pY, pX = patch_size
    minz, maxz, miny, maxy, minx, maxx = find_label_bounding_box(arr)
    out = []
    for z in range(minz, maxz - pZ + 2, max(pZ // 2, 1)):
        for y in range(miny, maxy - pY + 2, max(pY // 2, 1)):
            for x in range(minx, maxx - pX + 2, max(pX // 2, 1)):
                patch = arr[z:z + pZ, y:y + pY, x:x + pX]
                nz = np.argwhere(patch > 0)
                if nz.size == 0:
                    continue
                ext = nz.max(axis=0) - nz.min(axis=0) + 1
                if float(ext[0] * ext[1] * ext[2]) / patch.size < bbox_threshold:
                    continue
                if np.count_nonzero(patch) / patch.size < label_threshold:
                    continue
                out.append({"volume_idx": 0, "start_pos": [int(z), int(y), int(x)]})
    return out


def _ball(radius):
    """skimage.morphology.ball: voxels within `radius` of the centre of a (2r+1)^3 cube"""
    r = np.arange(-radius, radius + 1)
    z, y, x = np.meshgrid(r, r, r, indexing="ij")
    return (z * z + y * y + x * x) <= radius * radius


class ZarrSegmentationDataset3D(Dataset):
    """dataloading/dataset.py:18-222; the augmentation stack (:171-205) through `augment.py` (see the module docstring)."""
    _warned = False

    def __init__(self, mgr):
        self.mgr = mgr
        self.model_name = mgr.model_name
        self.volume_paths = mgr.volume_paths
        self.tasks = mgr.tasks
        self.patch_size = tuple(mgr.train_patch_size)
        self.min_labeled_ratio = mgr.min_labeled_ratio
        self.min_bbox_percent = mgr.min_bbox_percent
        self.dilate_label = mgr.dilate_label
        self.use_cache = mgr.use_cache
        self.cache_folder = mgr.cache_folder
        # the reference's recipe augments every item (dataset.py:171-205: brightness / noise / blur OneOf groups,
        # CoarseDropout3D) with albumentations / volumentations, which are not installed here.  `augment.py` restates that
        # stack; say ONCE per process that it is a restatement instead of silently training a slightly different recipe.
        # dataset_config.augment: true (default) = restated stack + the warning, "restated" = acknowledged, false = raw patches
        mode = getattr(mgr, "dataset_config", {}).get("augment", True)
        if isinstance(mode, str) and mode.lower() not in ("restated", "true", "false"):
            raise ValueError(f"dataset_config.augment: {mode!r} (true, false or \"restated\")")
        self.augment = (mode.lower() != "false") if isinstance(mode, str) else bool(mode)
        if self.augment and not (isinstance(mode, str) and mode.lower() == "restated") and not ZarrSegmentationDataset3D._warned:
            ZarrSegmentationDataset3D._warned = True
            warnings.warn("ZarrSegmentationDataset3D: the reference's augmentation stack (dataloading/dataset.py:171-205) needs "
                          "albumentations / volumentations, which are absent -- patches go through the numpy RESTATEMENT of it "
                          "(dataloading/augment.py: same structure and probabilities, the members' default parameters as "
                          "documented; not compared with albumentations).  Set dataset_config.augment: \"restated\" to "
                          "acknowledge, or false to feed raw patches.", RuntimeWarning, stacklevel=2)
        self.volumes = []
        for vol_idx, info in enumerate(self.volume_paths):
            vd = {"input_path": info["input"], "targets_path": {}, "ref_label_key": info.get("ref_label", "sheet")}
            for task in self.tasks:
                if task not in info:
                    raise ValueError(f"Volume {vol_idx} missing path for '{task}'")
                vd["targets_path"][task] = info[task]
            for pth in [vd["input_path"], *vd["targets_path"].values()]:
                if str(pth).startswith("http"):
                    raise ValueError(f"remote zarr store {pth}: no network in this environment, mirror it locally")
            self.volumes.append(vd)
        ps = self.patch_size
        self.cache_file = Path(f"{self.cache_folder}/{self.model_name}_{ps[0]}_{ps[1]}_{ps[2]}_cache.json")
        self.all_valid_patches = []
        if self.use_cache and self.cache_file.exists():
            with open(self.cache_file) as f:
                self.all_valid_patches = json.load(f)
        else:
            for vol_idx, vd in enumerate(self.volumes):
                ref = zarr_lite.open(vd["targets_path"][vd["ref_label_key"]])
                found = find_valid_patches(ref, ps, self.min_bbox_percent, self.min_labeled_ratio)
                for p in found:
                    p["volume_idx"] = vol_idx
                self.all_valid_patches.extend(found)
            if self.use_cache:
                os.makedirs(os.path.dirname(str(self.cache_file)) or ".", exist_ok=True)
                with open(self.cache_file, "w") as f:
                    json.dump(self.all_valid_patches, f)

    def __len__(self):
        return len(self.all_valid_patches)

    def __getitem__(self, idx):
        info = self.all_valid_patches[idx]
        z0, y0, x0 = info["start_pos"]
        dz, dy, dx = self.patch_size
        sl = np.s_[z0:z0 + dz, y0:y0 + dy, x0:x0 + dx]
        vd = self.volumes[info["volume_idx"]]
        arr = zarr_lite.open(vd["input_path"])
        img = arr[sl]
        og = img.dtype
        img = img.astype(np.float32)
        if og == np.uint8:
            img /= 255.0
        elif og == np.uint16:
            img /= 65535.0
        item = {"image": img}
        for task, path in vd["targets_path"].items():
            t_arr = zarr_lite.open(path)
            t = t_arr[sl].astype(np.float32)
            if task.lower() == "normals":
                t = (t / 32767.5) - 1.0 if t_arr.dtype == np.uint16 else (t * 2.0) - 1.0
                if t.ndim == 4:
                    t = t.transpose(3, 0, 1, 2).copy()
            else:
                if t_arr.dtype == np.uint8:
                    t /= 255.0
                elif t_arr.dtype == np.uint16:
                    t /= 65535.0
                if self.dilate_label:
                    from scipy.ndimage import binary_dilation      # == skimage dilation(t > 0, ball(5)) on a 0/1 volume
                    t = binary_dilation(t > 0, structure=_ball(5)).astype(np.float32)
            item[task] = t
        if self.augment:              # image only; targets untouched (dataset.py:200-205)
            from .augment import augment_image
            item["image"] = augment_image(item["image"])
        if item["image"].ndim == 3:
            item["image"] = item["image"][None, ...]
        item["image"] = torch.from_numpy(np.ascontiguousarray(item["image"]))
        for task in self.tasks:
            t = item[task]
            if t.ndim == 3 and task.lower() != "normals":
                t = t[None, ...]
            item[task] = torch.from_numpy(np.ascontiguousarray(t))
        return item

    def close(self):
        pass
