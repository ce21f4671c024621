"""CPU, world_size 2, gloo: the bucketed gradient synchroniser (engine/ddp.py) that the N>1 bench path
uses.  The HIP engine itself cannot run here, so a stand-in plan with the same interface
(`params`, `grad_order`) drives the alloc / ready / finish protocol exactly like Plan.run_backward."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class FakePlan:
    def __init__(self, shapes):
        g = torch.Generator().manual_seed(0)
        self.params = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in shapes]
        self.grad_order = list(reversed(range(len(shapes))))     # readiness order != parameter order


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import mt3d_amd  # noqa: F401
    from mt3d_amd.engine.ddp import GradSync, broadcast_parameters
    shapes = [(32, 1, 3, 3, 3), (32,), (64, 32, 3, 3, 3), (64, 32, 1, 1, 1), (1, 32, 1, 1, 1), (1,)]
    plan = FakePlan(shapes)
    # parameters differ per rank until broadcast
    with torch.no_grad():
        for p in plan.params:
            p.add_(rank)
    broadcast_parameters(torch.nn.ParameterList(plan.params), src=0)
    same = all(torch.equal(p, FakePlan(shapes).params[i]) for i, p in enumerate(plan.params))
    sync = GradSync(bucket_bytes=64 * 1024)        # small buckets -> several collectives
    outs = []
    for step in range(2):                           # two backward passes: buffers are fresh each time
        sync.begin(plan)
        grads = {}
        for idx in plan.grad_order:
            g = sync.alloc(idx)
            g.copy_(torch.full(plan.params[idx].shape, float(rank + 1 + step)) * (idx + 1))
            grads[idx] = g
            sync.ready(idx)
        sync.finish()
        outs.append({i: g.clone() for i, g in grads.items()})
    ok = same and sync.stats["buckets"] >= 2
    for step, o in enumerate(outs):
        for idx, g in o.items():
            expect = (sum(r + 1 + step for r in range(world)) / world) * (idx + 1)
            ok = ok and torch.allclose(g, torch.full_like(g, expect))
    ok = ok and outs[0][0].data_ptr() != outs[1][0].data_ptr()
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_bucketed_allreduce_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]
