"""Builds librxunet.so (C ABI in include/rxunet.h) with hipcc for gfx950, in-tree.

    python multi-task-3d-resencoder-unet_amd/csrc/build.py [--force]

The .so is git-ignored but travels to the GPU box with the repo snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["rx_elementwise.hip", "rx_igemm.hip", "rx_wgrad.hip", "rx_wgrad_halo.hip", "rx_conv_halo.hip", "rx_stem_wgrad.hip", "rx_loss.hip", "rx_se.hip", "rx_dgrad_s2.hip", "rx_prog.hip", "rx_pointwise.hip"]
HEADERS = ["rx_common.h", "rx_prog.h", os.path.join("..", "..", "include", "rxunet.h")]
LIB = os.path.join(HERE, "librxunet.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-Wno-unused-result"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hdrs = [os.path.join(HERE, h) for h in HEADERS] + [os.path.abspath(__file__)]
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(HERE, src.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [os.path.join(HERE, src)] + hdrs):
            cmd = [HIPCC] + FLAGS + ["-c", os.path.join(HERE, src), "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd, cwd=HERE)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    if force or procs or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd, cwd=HERE)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
