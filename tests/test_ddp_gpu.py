"""GPU (-m gpu): data parallelism through the REAL engine -- two ranks (gloo rendezvous, both on cuda:0) run
`Plan.run_backward` with `engine/ddp.py::GradSync` attached and must end with gradients that are BIT-IDENTICAL to
the mean of two single-process runs on the two half batches (every kernel of the engine is deterministic: fixed-order
reductions, no float atomics).  Covers what the CPU stand-in of tests/test_ddp_gloo.py cannot: gradients of one bucket
produced on two HIP streams (conv weight gradients on the plan's side stream; SqueezeExcite fc / head gradients on
the main stream), several buckets per backward, and the `no_sync` accumulation window (reference train.py:172,226-230).

The reference has no multi-GPU path (train.py:131, one device), so the checker is the engine's own single-process
result, which tests/test_network_gpu.py pins against the reference's golden vectors."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

TASKS = {"sheet": {"channels": 1, "activation": "none", "weight": 1, "loss_fn": "BCEDiceLoss",
                   "loss_kwargs": {"alpha": 0.5, "beta": 0.5}},
         "normals": {"channels": 3, "activation": "none", "weight": 1, "loss_fn": "MaskedCosineLoss"}}


def _worker(rank, world, port, cfg, q):
    try:
        for p in (ROOT, os.path.join(ROOT, "oracle")):
            if p not in sys.path:
                sys.path.insert(0, p)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        import mt3d_amd  # noqa: F401
        import resenc_oracle as oracle            # only make_mgr (a SimpleNamespace factory): no oracle compute here
        from mt3d_amd.builders.build_network_from_config import NetworkFromConfig
        from mt3d_amd.engine.ddp import GradSync
        from mt3d_amd.training.losses.losses import BCEDiceLoss, MaskedCosineLoss

        B = cfg["batch"]
        tasks = {k: TASKS[k] for k in cfg["tasks"]}
        mgr = oracle.make_mgr(cfg["patch"], tasks, 1, B, cfg.get("autoconfigure", True), cfg["model_config"])
        torch.manual_seed(3)
        net = NetworkFromConfig(mgr).cuda().train()
        gen = torch.Generator().manual_seed(99)
        nmb = cfg["micro_batches"]
        # global batch of every micro-batch: world * B samples; rank r owns [r*B, (r+1)*B)
        xs = [torch.rand((world * B, 1, *cfg["patch"]), generator=gen).cuda() for _ in range(nmb)]
        seg = [(torch.rand((world * B, 1, *cfg["patch"]), generator=gen) > 0.8).float().cuda() for _ in range(nmb)]
        nrm = [torch.nn.functional.normalize(torch.randn((world * B, 3, *cfg["patch"]), generator=gen), dim=1).cuda()
               for _ in range(nmb)]
        losses = {"sheet": BCEDiceLoss(alpha=0.5, beta=0.5), "normals": MaskedCosineLoss()}
        params = [p for p in net.parameters()]
        names = [n for n, _ in net.named_parameters()]
        fwd_same = []

        logits = {}

        def backward(r, m):
            sl = slice(r * B, (r + 1) * B)
            with torch.autocast("cuda", dtype=cfg["dtype"], enabled=cfg["dtype"] is not None):
                out = net(xs[m][sl])
                key = (r, m)
                if key not in logits:
                    logits[key] = {k: v.detach().clone() for k, v in out.items()}
                    fwd_same.append(True)
                else:       # every later forward of the same half batch must reproduce the first one bit for bit
                    fwd_same.append(all(torch.equal(out[k], logits[key][k]) for k in out))
                total = 0.0
                for k in tasks:
                    tgt = seg[m][sl] if k == "sheet" else nrm[m][sl] * seg[m][sl]
                    total = total + losses[k](out[k], tgt)
            total.backward()

        def grab():
            g = [None if p.grad is None else p.grad.clone() for p in params]
            for p in params:
                p.grad = None
            return g

        # ---- single-process runs of every (rank, micro-batch) half batch: no synchroniser attached
        local = {}
        for r in range(world):
            for m in range(nmb):
                backward(r, m)
                local[(r, m)] = grab()

        # ---- the data-parallel run of THIS rank, five times over (a race would not show on every pass)
        sync = GradSync(bucket_bytes=cfg["bucket_bytes"])
        for plan in net._plans.values():
            plan.grad_sync = sync
        expect = []
        for i in range(len(params)):
            if local[(0, 0)][i] is None:
                expect.append(None)
                continue
            # the synchroniser's arithmetic, restated: (carry + own) / world on every rank, then the sum over ranks
            per_rank = []
            for r in range(world):
                carry = None
                for m in range(nmb):          # `flat.add_(carry)`: own gradients + what the window has carried so far
                    carry = local[(r, m)][i] if carry is None else local[(r, m)][i] + carry
                per_rank.append(carry / world)
            e = per_rank[0]
            for r in range(1, world):
                e = e + per_rank[r]
            expect.append(e)
        ok, none_inside, passes, stats = True, True, [], None
        plan0 = next(iter(net._plans.values()))
        pidx = {id(p): i for i, p in enumerate(params)}
        bucket_of = {}
        for bi, b in enumerate(sync._plan_layout(plan0)):      # which bucket each parameter travels in (failure diagnostics)
            for j in b.idxs:
                bucket_of[pidx[id(plan0.params[j])]] = bi
        for rep in range(cfg.get("reps", 7)):      # passes 1-2 eager, 3 records the launch program, 4-5 replay it, 6 eager again (what a
            for plan in net._plans.values():       # profiler or an absent task does), 7 replays: the recorded addresses must still hold
                plan.use_programs = rep != 5
            for m in range(nmb):
                sync.require_sync = (m == nmb - 1)
                backward(rank, m)
                if m < nmb - 1:
                    none_inside = none_inside and all(p.grad is None for p in params)
            torch.cuda.synchronize()
            got = grab()
            stats = dict(sync.stats)
            bad = []
            for i, (g, e) in enumerate(zip(got, expect)):
                if e is None:
                    ok = ok and g is None
                elif g is None or not torch.equal(g, e):
                    ok = False
                    rel = float("inf") if g is None else ((g - e).norm() / e.norm().clamp(min=1e-30)).item()
                    bad.append((names[i], rel, bucket_of.get(i, -1)))
            passes.append(bad[:4] + bad[-4:] + [("n_bad", len(bad), sorted({b[2] for b in bad}))] if bad else [])
        # is a plain single-process backward of this rank's half batch still what it was?  (tells a race inside the plan
        # from one in the synchroniser)
        for plan in net._plans.values():
            plan.grad_sync = None
        backward(rank, nmb - 1)
        again = grab()
        reproducible = all((a is None and b is None) or torch.equal(a, b) for a, b in zip(again, local[(rank, nmb - 1)]))
        n_with_grad = sum(e is not None for e in expect)
        # the two ranks' half batches really differ (otherwise the test could not see a skipped collective)
        differs = any(a is not None and not torch.equal(a, b) for a, b in zip(local[(0, 0)], local[(1, 0)]))
        q.put((rank, bool(ok), dict(stats=stats, passes=passes, differs=differs, none_inside=none_inside,
                                    n_with_grad=n_with_grad, reproducible=reproducible, forward_reproducible=fwd_same)))
        dist.destroy_process_group()
    except Exception as e:      # noqa: BLE001 -- report instead of hanging the parent on q.get
        import traceback
        q.put((rank, False, dict(error=f"{type(e).__name__}: {e}", tb=traceback.format_exc()[-2000:])))


def _run(cfg):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() * 7 + cfg["port_salt"]) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, cfg, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=420) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    return res


BASE = dict(patch=(32, 32, 32), batch=1, tasks=["sheet"], dtype=torch.bfloat16, bucket_bytes=4 << 20,
            micro_batches=1, model_config={})


@pytest.mark.parametrize("name,over", [
    ("plain_bf16", dict(port_salt=1)),
    ("se_bf16", dict(port_salt=2, model_config={"squeeze_excitation": True})),
    ("se_fp32_two_heads_bias", dict(port_salt=3, dtype=None, tasks=["sheet", "normals"], patch=(16, 16, 16), batch=2,
                                    bucket_bytes=1 << 20, model_config={"squeeze_excitation": True, "conv_bias": True})),
    ("se_accumulate_no_sync", dict(port_salt=4, micro_batches=2, model_config={"squeeze_excitation": True})),
    # 24/48/80 features: padded buffers + shadow parameters; the padded gradients are sliced and announced to the synchroniser
    # after the backward list (engine/plan.py::_backward_finish), not from inside it
    ("padded_channels_se_bf16", dict(port_salt=6, patch=(16, 16, 16), batch=2, autoconfigure=False, bucket_bytes=256 << 10, micro_batches=2,
                                     model_config=dict(basic_encoder_block="BasicBlockD", basic_decoder_block="ConvBlock",
                                                       bottleneck_block="BasicBlockD", features_per_stage=[24, 48, 80], num_stages=3,
                                                       n_blocks_per_stage=[1, 2, 2], kernel_sizes=[3, 3, 3],
                                                       n_conv_per_stage_decoder=[1, 1], strides=[1, 2, 2],
                                                       squeeze_excitation=True, conv_bias=True))),
    # BASELINE configs[3] (cfg4) per-rank workload at FULL size: the cfg2 network, 128^3, batch 2 per rank, bf16, the production
    # bucket size (853 MB of gradients in 7 buckets) -- two of the eight ranks (eager, eager, recorded, replayed)
    ("cfg4_rank_workload_full_size", dict(port_salt=5, patch=(128, 128, 128), batch=2, bucket_bytes=128 << 20, reps=4)),
])
def test_two_ranks_real_plan_equal_mean_of_half_batches(name, over):
    cfg = dict(BASE)
    cfg.update(over)
    res = _run(cfg)
    import json
    for rank, ok, info in res:
        print(f"[{name}] rank {rank}: ok={ok} {json.dumps(info)}", flush=True)
    for rank, ok, info in res:
        assert "error" not in info, info
        assert all(info["forward_reproducible"]), ("a forward of the same half batch changed", info["forward_reproducible"])
        assert info["reproducible"], "a single-process backward of the same half batch changed: race inside the plan"
        assert info["differs"], "both ranks saw the same data"
        assert info["stats"]["buckets"] >= 2, info
        assert info["stats"]["collectives"] == info["stats"]["buckets"], info
        assert info["none_inside"], "a no_sync micro-batch handed gradients to autograd"
        assert ok, (name, rank, info)


def _trainer_worker(rank, world, port, cfg_path, workdir, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          LOCAL_RANK="0", RX_DDP_BACKEND="gloo")
        os.chdir(workdir)
        import hashlib
        import mt3d_amd  # noqa: F401
        from mt3d_amd.train import BaseTrainer

        class Rec(BaseTrainer):
            lines = []

            def _log(self, *a):
                s = " ".join(str(x) for x in a)
                if s.startswith("[Train]"):
                    self.lines.append(float(s.split("sheet: ")[1].split(" ")[0]))

        tr = Rec(cfg_path, verbose=False)
        torch.manual_seed(100 + rank)            # ranks start from DIFFERENT weights: broadcast_parameters must fix that
        model = tr.train()
        torch.cuda.synchronize()
        h = hashlib.sha256()
        seen = set()
        for p in model.parameters():
            if id(p) in seen:
                continue
            seen.add(id(p))
            h.update(p.detach().cpu().numpy().tobytes())
        q.put((rank, h.hexdigest(), Rec.lines, tr.last_patches_per_sec))
    except Exception as e:      # noqa: BLE001
        import traceback
        q.put((rank, "error", f"{type(e).__name__}: {e}\n{traceback.format_exc()[-2000:]}", None))


def test_trainer_two_ranks_accumulation_keeps_replicas_identical(tmp_path):
    """`BaseTrainer.train()` under a 2-rank launch (the DDP branch of train.py, never executed with N > 1 in round 1):
    SqueezeExcite on, gradient_accumulation 2 (no_sync window), an odd number of training patches.  After two epochs
    both replicas hold bit-identical parameters (the DDP invariant) and the loss went down."""
    import yaml
    cfg = yaml.safe_load(open(os.path.join(ROOT, "tasks", "synthetic_sheet.yaml")))
    cfg["tr_setup"].update(ckpt_out_base=str(tmp_path / "ckpt"), tensorboard_log_dir=str(tmp_path / "tb"))
    cfg["tr_config"].update(max_epoch=2, max_steps_per_epoch=6, gradient_accumulation=2, batch_size=1)
    cfg["model_config"] = {"conv_bias": False, "squeeze_excitation": True}
    cfg["dataset_config"]["synthetic_length"] = 19
    p = tmp_path / "cfg.yaml"
    yaml.safe_dump(cfg, open(p, "w"))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_trainer_worker, args=(r, 2, port, str(p), str(tmp_path), q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for pr in procs:
        pr.join(timeout=60)
    assert res[0][1] != "error" and res[1][1] != "error", res
    assert res[0][1] == res[1][1], "replicas diverged"
    losses = res[0][2]
    assert len(losses) == 2 and losses[1] < losses[0], losses
    assert res[0][3] > 0


def _rccl_worker(port, q):
    try:
        for p in (ROOT, os.path.join(ROOT, "oracle")):
            if p not in sys.path:
                sys.path.insert(0, p)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        import torch.distributed as dist
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        import mt3d_amd  # noqa: F401
        import resenc_oracle as oracle
        from mt3d_amd.builders.build_network_from_config import NetworkFromConfig
        from mt3d_amd.engine.ddp import GradSync, broadcast_parameters
        from mt3d_amd.training.losses.losses import BCEDiceLoss
        from mt3d_amd.training.optim import EngineAdamW, clip_and_step
        tasks = {"sheet": TASKS["sheet"]}
        mgr = oracle.make_mgr((32, 32, 32), tasks, 1, 2, True, {"squeeze_excitation": True})

        def run(with_sync):
            torch.manual_seed(3)
            net = NetworkFromConfig(mgr).cuda().train()
            if with_sync:
                broadcast_parameters(net)
            params = list(net.parameters())
            opt = EngineAdamW(params, model=None, lr=1e-3, weight_decay=0.0)
            sync = GradSync(bucket_bytes=8 << 20, force_collectives=True) if with_sync else None
            gen = torch.Generator().manual_seed(5)
            x = torch.rand((2, 1, 32, 32, 32), generator=gen).cuda()
            t = (torch.rand((2, 1, 32, 32, 32), generator=gen) > 0.8).float().cuda()
            loss_fn = BCEDiceLoss(alpha=0.5, beta=0.5)
            losses = []
            for _ in range(6):              # eager, eager, recorded, replayed ...
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    out = net(x)
                    if sync is not None:
                        for plan in net._plans.values():
                            plan.grad_sync = sync
                    loss = loss_fn(out["sheet"], t)
                loss.backward()
                clip_and_step(opt, params, 3)
                opt.zero_grad(set_to_none=True)
                losses.append(loss.item())
            return losses, [p.detach().clone() for p in params], (dict(sync.stats) if sync else None)

        l0, p0, _ = run(False)
        l1, p1, stats = run(True)
        same = l0 == l1 and all(torch.equal(a, b) for a, b in zip(p0, p1))
        q.put((bool(same), dict(stats=stats, backend=dist.get_backend(), losses=(l0[-1], l1[-1]))))
        dist.destroy_process_group()
    except Exception as e:      # noqa: BLE001
        import traceback
        q.put((False, dict(error=f"{type(e).__name__}: {e}", tb=traceback.format_exc()[-2000:])))


def test_rccl_backend_rehearsal_in_a_world_of_one():
    """the production backend ("nccl" == RCCL) as far as ONE GPU can exercise it: a world of one rank with the collectives forced
    (`GradSync(force_collectives=True)`): `all_reduce(op=AVG, async_op=True)` on the collective stream, `work.wait()`, persistent
    buckets under recorded launch programs, `broadcast_parameters`, the engine's AdamW reading the bucket views.  An average over one
    rank is the identity, so six training steps must leave losses and parameters bit-identical to the run without a synchroniser.
    (Two or more RCCL ranks need one GPU each: unmeasured on hardware)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(35500 + (os.getpid() % 2000), q))
    p.start()
    ok, info = q.get(timeout=420)
    p.join(timeout=60)
    assert "error" not in info, info
    assert info["backend"] == "nccl" and info["stats"]["buckets"] >= 2 and info["stats"]["collectives"] == info["stats"]["buckets"], info
    assert ok, info


@pytest.mark.gpu
def test_bench_two_rank_rehearsal_emits_a_populated_ddp_block():
    """`bench.py --gpus 2` (launched exactly as the driver launches it, two ranks sharing cuda:0, gloo instead of
    RCCL because one GPU cannot host an RCCL world of two) prints a bench line whose `ddp` block describes the collectives:
    backend, world as the process group sees it, buckets, bytes, one collective per bucket, the exposed tail."""
    import json
    import subprocess
    port = 35500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "3",
           "--workload", "cfg1", "--dist-backend", "gloo", "--bucket-mb", "32", "--no-cpu-baseline", "--no-kernel-timing",
           "--no-h2d", "--no-pmc"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4"))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    d = line["ddp"]
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "dp2" and line["config"]["global_batch"] == 4
    assert d["backend"] == "gloo" and d["world"] == 2 and d["world_env"] == 2 and d["backend_is_rccl"] is False
    assert d["bucket_mb"] == 32 and d["bucket_bytes"] == 32 << 20
    assert d["buckets"] >= 2 and d["collectives"] == d["buckets"], d
    assert d["bytes"] >= 4 * 111_000_000, d          # the 64^3 network: 112.0 M parameters, fp32
    assert d["exposed_tail_ms"]["steps"] == 3 and d["exposed_tail_ms"]["mean"] >= 0.0
    assert line["value"] > 0
