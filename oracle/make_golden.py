"""TEST INFRASTRUCTURE ONLY -- run in the build container (needs /root/reference):

    python oracle/make_golden.py [case ...]

Runs the REAL reference `NetworkFromConfig` (through oracle/ref_shim.py) and the REAL reference
losses on seeded synthetic batches and writes `tests/golden/<case>.npz`:
inputs, targets, train-mode logits, eval-mode (activated) outputs, loss, per-parameter init and
gradient checksums (name, numel, sum, l2) over the UNIQUE parameter tensors, the sorted
`state_dict` key list, and a few small gradients in full.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402
from golden_cases import CASES, FULL_GRAD_SUFFIXES  # noqa: E402
from resenc_oracle import synthetic_batch  # noqa: E402  (data generator only)

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def reference_loss(ref_losses, outputs, targets, tasks):
    total = 0.0
    for name, gt in targets.items():
        info = tasks[name]
        cls = getattr(ref_losses, info.get("loss_fn", "BCEDiceLoss"))
        total = total + cls(**info.get("loss_kwargs", {}))(outputs[name], gt) * info.get("weight", 1.0)
    return total


def main():
    torch.set_num_threads(8)
    _, ref_losses = ref_shim.import_reference()
    os.makedirs(OUT, exist_ok=True)
    only = sys.argv[1:]               # optional: regenerate just these cases
    for cname, c in CASES.items():
        if only and cname not in only:
            continue
        mgr = ref_shim.make_mgr(c["patch"], c["tasks"], c["in_channels"], c["batch"], c["autoconfigure"],
                                c["model_config"])
        torch.manual_seed(c["seed"])
        net = ref_shim.build_reference_network(mgr)
        x, targets = synthetic_batch(c["batch"], c["in_channels"], c["patch"], c["tasks"], c["data_seed"])
        net.train()
        # channel dropout (dropout_op_kwargs p > 0): the reference draws its masks from the global CPU generator during this
        # forward; the fixture keeps them (kept (n, c) planes, in execution order) so that the engine can replay the same step
        drop_masks, hooks = [], []
        for mod in net.modules():
            if isinstance(mod, (torch.nn.Dropout3d, torch.nn.Dropout2d)) and mod.p > 0:
                hooks.append(mod.register_forward_hook(
                    lambda m, inp, outp: drop_masks.append((outp.detach().flatten(2).abs().amax(2) > 0).float().numpy())))
        out = net(x)
        for h in hooks:
            h.remove()
        loss = reference_loss(ref_losses, out, targets, c["tasks"])
        loss.backward()
        arrays = {"x": x.numpy(), "loss": np.float64(loss.item())}
        for i, mk in enumerate(drop_masks):
            arrays[f"dropmask.{i:03d}"] = mk
        for k, v in targets.items():
            arrays[f"target.{k}"] = v.numpy()
        for k, v in out.items():
            arrays[f"logits.{k}"] = v.detach().numpy()
        net.eval()
        with torch.no_grad():
            for k, v in net(x).items():
                arrays[f"eval.{k}"] = v.numpy()
        names, init_ck, grad_ck = [], [], []
        seen = set()
        for n, p in net.named_parameters():  # named_parameters() already de-duplicates aliases
            assert id(p) not in seen
            seen.add(id(p))
            names.append(n)
            init_ck.append([p.numel(), p.detach().double().sum().item(), p.detach().double().norm().item()])
            if p.grad is None:
                grad_ck.append([0.0, float("nan"), float("nan")])
            else:
                g = p.grad.double()
                grad_ck.append([1.0, g.sum().item(), g.norm().item()])
                if n.endswith(FULL_GRAD_SUFFIXES + tuple(c.get("full_grads", ()))):
                    arrays[f"grad.{n}"] = p.grad.numpy()
        arrays["param_names"] = np.array(json.dumps(names))
        arrays["init_checksums"] = np.array(init_ck, dtype=np.float64)
        arrays["grad_checksums"] = np.array(grad_ck, dtype=np.float64)
        arrays["state_dict_keys"] = np.array(json.dumps(sorted(net.state_dict().keys())))
        arrays["topology"] = np.array(json.dumps({
            "num_stages": net.num_stages, "features_per_stage": list(net.features_per_stage),
            "n_blocks_per_stage": list(net.n_blocks_per_stage),
            "strides": [list(s) if isinstance(s, (tuple, list)) else s for s in net.strides],
            "kernel_sizes": [list(s) if isinstance(s, (tuple, list)) else s for s in net.kernel_sizes]}))
        path = os.path.join(OUT, f"{cname}.npz")
        np.savez_compressed(path, **arrays)
        print(cname, "loss", loss.item(), "params", len(names), "->", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
