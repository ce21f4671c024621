"""Patch feeders with the reference's item contract (dataloading/dataset.py:103-222): a dict with
"image" (C, Z, Y, X) float32 and one float32 tensor per task.

`SyntheticPatchDataset` generates SURVEY 8(d)'s synthetic patches (what the headline metric is quoted
on).  The zarr-backed dataset of the reference (valid-patch search, augmentations) is CPU I/O outside the
hot-path scope and needs `zarr`, which this image does not have: asking for it raises."""
import torch
from torch.utils.data import Dataset


class SyntheticPatchDataset(Dataset):
    def __init__(self, mgr, length=None, seed=1234):
        self.mgr = mgr
        self.patch = tuple(mgr.train_patch_size)
        self.cin = mgr.in_channels
        self.tasks = mgr.tasks
        self.length = int(length if length is not None else mgr.dataset_config.get("synthetic_length", 64))
        self.seed = seed

    def __len__(self):
        return self.length

    def __getitem__(self, idx):
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


class ZarrSegmentationDataset3D(Dataset):
    def __init__(self, mgr):
        raise ImportError("ZarrSegmentationDataset3D needs the `zarr` package (absent here) and is outside the hot-path "
                          "scope; set dataset_config.synthetic: true or override BaseTrainer._configure_dataset")
