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
    # accumulation window (DDP.no_sync equivalent): two local micro-batches, then the stepping one -- ONE collective per bucket,
    # result = mean over ranks of the window's sum
    window = []
    for mb, syncing in enumerate([False, False, True]):
        sync.require_sync = syncing
        sync.begin(plan)
        ok = ok and sync.returns_grads == syncing
        grads = {}
        for idx in plan.grad_order:
            g = sync.alloc(idx)
            g.copy_(torch.full(plan.params[idx].shape, float(10 * mb + rank + 1)) * (idx + 1))
            grads[idx] = g
            sync.ready(idx)
        sync.finish()
        window.append((dict(sync.stats), grads))
    ok = ok and window[0][0]["collectives"] == 0 and window[1][0]["collectives"] == 0
    ok = ok and window[2][0]["collectives"] == window[2][0]["buckets"]
    for idx, g in window[2][1].items():
        expect = sum(sum(10 * mb + r + 1 for mb in range(3)) for r in range(world)) / world * (idx + 1)
        ok = ok and torch.allclose(g, torch.full_like(g, expect))
    with sync.no_sync():
        ok = ok and sync.require_sync is False
    ok = ok and sync.require_sync is True
    # an accumulation window that spans TWO plans of one model (the ragged last batch of an epoch runs on a plan of
    # another batch size and is always the stepping micro-batch, train.py:226): the carry follows the parameter, not the plan
    plan_b = FakePlan(shapes)
    plan_b.params = plan.params
    for persistent in (False, True):
        sync.persistent = persistent
        results = None
        for mb, (pl, syncing) in enumerate([(plan, False), (plan, False), (plan_b, True)]):
            sync.require_sync = syncing
            sync.begin(pl)
            grads = {}
            for idx in pl.grad_order:
                g = sync.alloc(idx)
                g.copy_(torch.full(pl.params[idx].shape, float(100 * mb + rank + 1)) * (idx + 1))
                grads[idx] = g
                sync.ready(idx)
            sync.finish()
            results = grads
        for idx, g in results.items():
            expect = sum(sum(100 * mb + r + 1 for mb in range(3)) for r in range(world)) / world * (idx + 1)
            ok = ok and torch.allclose(g, torch.full_like(g, expect))
        ok = ok and not sync._carry
    sync.persistent = False
    # the bench line's `ddp` block: static facts + counters of the last backward, exposed-tail timing
    sync.timing = True
    sync.begin(plan)
    for idx in plan.grad_order:
        sync.alloc(idx).zero_()
        sync.ready(idx)
    sync.finish()
    d = sync.describe()
    tails = sync.tail_ms()
    ok = ok and d["backend"] == "gloo" and d["world"] == world and d["collectives"] == d["buckets"] >= 2 and d["bytes"] > 0
    ok = ok and len(tails) == 1 and tails[0] >= 0.0 and sync.tail_ms() == []
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


def _split_worker(rank, world, port, cfg_path, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import numpy as np
    import mt3d_amd  # noqa: F401
    from mt3d_amd.train import BaseTrainer
    np.random.seed(1000 + rank)                     # every rank's own RNG state, as in a real launch
    tr = BaseTrainer(cfg_path, verbose=False)
    train, val = tr._configure_dataloaders(tr._configure_dataset())
    q.put((rank, sorted(train.sampler.indices), sorted(val.sampler.indices), len(train)))
    dist.destroy_process_group()


def test_trainer_data_split_is_one_partition_for_all_ranks(tmp_path):
    """per-rank unseeded shuffles gave every rank its own train/val partition and unequal loader lengths
    (a hang in all_reduce when one rank runs an extra backward)."""
    import yaml
    cfg = yaml.safe_load(open(os.path.join(ROOT, "tasks", "synthetic_sheet.yaml")))
    cfg["dataset_config"]["synthetic_length"] = 19          # 17 training patches -> 8 per rank, one dropped
    p = tmp_path / "cfg.yaml"
    yaml.safe_dump(cfg, open(p, "w"))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_split_worker, args=(r, 2, port, str(p), q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for pr in procs:
        pr.join(timeout=60)
    (_, tr0, va0, n0), (_, tr1, va1, n1) = res
    assert va0 == va1 and len(va0) == 2
    assert len(tr0) == len(tr1) == 8 and n0 == n1
    assert not set(tr0) & set(tr1) and not (set(tr0) | set(tr1)) & set(va0)
