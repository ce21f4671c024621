"""TEST INFRASTRUCTURE ONLY.  How the `data_seed` of every golden case was chosen:

    python oracle/scan_seeds.py [case ...]

The backward of a random-init net is discontinuous in the LeakyReLU / ReLU masks: one pre-activation within fp32
round-off of zero moves whole gradient tensors by 1e-3..1e-2 between two evaluation orders (tests/test_oracle_golden.py::
test_fp32_gradients_are_mask_discontinuous).  A seed has MARGIN when the oracle's fp32 gradients agree with its own fp64
evaluation to ~1e-5 for every parameter AND stay there with a different fp32 summation order (1 thread vs all threads).
The case definitions keep the first such seed; the 1e-3 gradient bar of the parity tests is meaningful only on those."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import resenc_oracle as oracle  # noqa: E402
from golden_cases import CASES, UNPINNED_CASES  # noqa: E402


def rel_l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return ((a - b).norm() / b.norm().clamp(min=1e-30)).item()


def grads(c, mgr, data_seed, dtype, threads):
    torch.set_num_threads(threads)
    torch.manual_seed(c["seed"])
    net = oracle.NetworkFromConfig(mgr).to(dtype)
    x, t = oracle.synthetic_batch(c["batch"], c["in_channels"], c["patch"], c["tasks"], data_seed)
    out = net(x.to(dtype))
    oracle.train_loss(out, {k: v.to(dtype) for k, v in t.items()}, c["tasks"]).backward()
    return {n: p.grad for n, p in net.named_parameters() if p.grad is not None}


def main():
    every = dict(CASES)
    every.update(UNPINNED_CASES)
    names = sys.argv[1:] or list(every)
    nthr = os.cpu_count() or 8
    for name in names:
        c = every[name]
        mgr = oracle.make_mgr(c["patch"], c["tasks"], c["in_channels"], c["batch"], c["autoconfigure"], c["model_config"])
        for ds in [c["data_seed"]] + [s for s in range(1, 40) if s != c["data_seed"]]:
            g64 = grads(c, mgr, ds, torch.float64, nthr)
            worst = 0.0
            for thr in (nthr, 1):
                g32 = grads(c, mgr, ds, torch.float32, thr)
                worst = max(worst, max(rel_l2(g32[n], g64[n]) for n in g64 if g64[n].norm() > 1e-6))
            print(f"{name}: data_seed {ds:3d} worst fp32-vs-fp64 gradient rel-L2 {worst:.2e}", flush=True)
            if worst < 3e-5:
                print(f"{name}: -> data_seed {ds} has mask margin", flush=True)
                break


if __name__ == "__main__":
    main()
