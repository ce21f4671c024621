"""TEST INFRASTRUCTURE ONLY -- run in the build container (needs /root/reference):

    python oracle/make_golden_losses.py

Runs the REAL reference loss classes (training/losses/losses.py: BCEDiceLoss :307-318, MaskedCosineLoss :187-215) on
small seeded inputs in float64-free plain fp32 CPU torch and writes `tests/golden/losses.npz`: inputs, loss values and
d(loss)/d(prediction), the vectors the fused HIP loss kernels (csrc/rx_loss.hip) are pinned against."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "losses.npz")

CASES = {   # name -> (kind, shape, seed, kwargs, upstream weight)
    "bce_dice_1ch": ("BCEDiceLoss", (2, 1, 12, 16, 16), 11, {"alpha": 0.5, "beta": 0.5}, 1.0),
    "bce_dice_3ch_ragged": ("BCEDiceLoss", (1, 3, 5, 7, 9), 12, {"alpha": 0.3, "beta": 0.7}, 0.25),
    "bce_dice_2d": ("BCEDiceLoss", (2, 2, 24, 20), 13, {"alpha": 1.0, "beta": 2.0}, 1.0),
    "bce_dice_empty_target": ("BCEDiceLoss", (1, 1, 8, 8, 8), 14, {"alpha": 0.5, "beta": 0.5}, 1.0),
    "cosine_3ch": ("MaskedCosineLoss", (2, 3, 10, 12, 14), 21, {}, 1.0),
    "cosine_3ch_ragged": ("MaskedCosineLoss", (1, 3, 5, 7, 9), 22, {}, 0.5),
    "cosine_all_masked": ("MaskedCosineLoss", (1, 3, 4, 4, 4), 23, {}, 1.0),
}


def inputs(kind, shape, seed, name):
    g = torch.Generator().manual_seed(seed)
    pred = torch.randn(shape, generator=g) * 2.0
    if kind == "BCEDiceLoss":
        target = (torch.rand(shape, generator=g) > 0.8).float()
        if "empty" in name:
            target.zero_()
    else:
        v = torch.randn(shape, generator=g)
        v = v / v.norm(dim=1, keepdim=True).clamp(min=1e-8)
        keep = (torch.rand((shape[0], 1, *shape[2:]), generator=g) > 0.6).float()
        target = v * keep
        if "all_masked" in name:
            target.zero_()
    return pred, target


def main():
    _, ref_losses = ref_shim.import_reference()
    arrays = {}
    for name, (kind, shape, seed, kw, weight) in CASES.items():
        pred, target = inputs(kind, shape, seed, name)
        p = pred.clone().requires_grad_(True)
        loss = getattr(ref_losses, kind)(**kw)(p, target)
        (loss * weight).backward()
        arrays[f"{name}.pred"] = pred.numpy()
        arrays[f"{name}.target"] = target.numpy()
        arrays[f"{name}.loss"] = np.float64(loss.item())
        arrays[f"{name}.grad"] = p.grad.numpy()
        print(f"{name:24s} loss={loss.item():.6f}  |grad|={p.grad.norm().item():.4e}")
    np.savez_compressed(OUT, **arrays)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
