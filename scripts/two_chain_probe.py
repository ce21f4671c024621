"""Probe: is ONE dependency chain the limit of a cfg2 step?  Two independent batch-1 replicas, each on its own HIP stream
(enqueued alternately from one thread; launch programs keep the host out of the way), against one batch-2 replica.
If 2 x (batch 1, concurrent) beats 1 x (batch 2), splitting the batch into per-sample chains inside ONE plan would pay."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch  # noqa: E402
import bench  # noqa: E402
import mt3d_amd  # noqa: E402,F401
from mt3d_amd.builders.build_network_from_config import NetworkFromConfig  # noqa: E402
from mt3d_amd.training.losses.losses import LOSS_FN_MAP  # noqa: E402
from mt3d_amd.training.optim import EngineAdamW, clip_and_step  # noqa: E402

WL = sys.argv[1] if len(sys.argv) > 1 else "cfg2"


def make(batch, seed):
    w = dict(bench.WORKLOADS[WL])
    torch.manual_seed(seed)
    net = NetworkFromConfig(bench.make_mgr(w)).cuda()
    net.compute_dtype = torch.bfloat16
    net.train()
    loss_fn = LOSS_FN_MAP["BCEDiceLoss"](0.5, 0.5)
    params = list(net.parameters())
    opt = EngineAdamW(params, model=None, lr=1e-3, weight_decay=0.0)
    x, t = bench.synthetic_batch(w, batch, 1234 + seed, "cuda")

    def step():
        out = net(x)
        loss = loss_fn(out["sheet"], t["sheet"])
        loss.backward()
        clip_and_step(opt, params, 3)
        opt.zero_grad(set_to_none=True)
    return step


def timeit(fn, n=15, warm=6):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


one = make(2, 0)
print(f"1 x batch 2            : {timeit(one):.2f} ms per 2 patches", flush=True)
del one
torch.cuda.empty_cache()
single = make(1, 0)
print(f"1 x batch 1            : {timeit(single):.2f} ms per 1 patch", flush=True)
a, b = single, make(1, 1)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def both():
    with torch.cuda.stream(sa):
        a()
    with torch.cuda.stream(sb):
        b()


print(f"2 x batch 1, 2 streams : {timeit(both):.2f} ms per 2 patches", flush=True)
