"""Shared test helpers (test infrastructure)."""
import contextlib
import io
import json
import os

import numpy as np
import torch

import resenc_oracle as oracle
from golden_cases import CASES

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, f"{name}.npz"), allow_pickle=False)
    g = {k: z[k] for k in z.files}
    g["param_names"] = json.loads(str(g["param_names"]))
    g["state_dict_keys"] = json.loads(str(g["state_dict_keys"]))
    g["topology"] = json.loads(str(g["topology"]))
    return g


def build_oracle(case_name, dtype=torch.float32):
    c = CASES[case_name]
    mgr = oracle.make_mgr(c["patch"], c["tasks"], c["in_channels"], c["batch"], c["autoconfigure"],
                          c["model_config"])
    torch.manual_seed(c["seed"])
    net = oracle.NetworkFromConfig(mgr)
    return net.to(dtype), c, mgr


def rel_l2(a, b):
    a = torch.as_tensor(a).double().flatten()
    b = torch.as_tensor(b).double().flatten()
    return ((a - b).norm() / b.norm().clamp(min=1e-30)).item()


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


@contextlib.contextmanager
def forced_dropout(masks):
    """replay channel-dropout masks (a golden fixture's `dropmask.NNN`, kept (n, c) planes in execution order) on the engine:
    the i-th dropout layer of a plan keeps masks[i] instead of drawing from torch's generator"""
    from mt3d_amd.engine import plan as plan_mod
    orig = plan_mod.Plan._draw_dropout

    def draw(self, d):
        i = next(j for j, e in enumerate(self._drops) if e is d)
        mk = torch.as_tensor(masks[i])
        d["keep"].fill_(1.0)                               # (padding channels of a padded buffer are zero whatever they keep)
        d["keep"][:, :mk.shape[1]].copy_(mk)
    plan_mod.Plan._draw_dropout = draw
    try:
        yield
    finally:
        plan_mod.Plan._draw_dropout = orig


def golden_dropout_masks(g):
    return [g[k] for k in sorted(g) if k.startswith("dropmask.")]
