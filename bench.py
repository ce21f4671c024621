#!/usr/bin/env python
"""Headline benchmark: train patches/sec of the ResEncM 3-D residual-encoder U-Net (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One STEP = forward -> per-task loss -> backward -> (RCCL gradient all-reduce, overlapped) ->
clip_grad_norm_(3) -> AdamW step -> zero_grad(set_to_none=True) on one synthetic batch that is already
resident in HBM.  Workload at every N: BASELINE configs[1] = autoconfigured ResEncM,
1-in / 1 seg head, patch 128^3, bf16, batch 2 per GPU (weak scaling: global batch 2N, configs[3] at N=8).
Prints ONE JSON line (rank 0) with `roofline` (dominant MFMA kernel, algorithmic FLOPs / HIP-event time)
and `cpu_baseline` (the CPU oracle = pure-PyTorch restatement of the reference, timed on this host).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT]

MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0, "fp32": 157.3}   # dense, MI355X_MICROARCH.md
HBM_PEAK_GBPS = 8000.0                                                # HBM3E, MI355X_MICROARCH.md
DTYPES = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}

WORKLOADS = {
    # BASELINE.json configs[1] / [3]
    "cfg2": dict(patch=(128, 128, 128), in_channels=1, batch=2, autoconfigure=True, model_config={},
                 tasks={"sheet": {"channels": 1, "activation": "none", "weight": 1, "loss_fn": "BCEDiceLoss",
                                  "loss_kwargs": {"alpha": 0.5, "beta": 0.5}}}),
    # configs[2]: + 3-channel normals regression head (second decoder)
    "cfg3": dict(patch=(128, 128, 128), in_channels=1, batch=1, autoconfigure=True, model_config={},
                 tasks={"sheet": {"channels": 1, "activation": "none", "weight": 1, "loss_fn": "BCEDiceLoss",
                                  "loss_kwargs": {"alpha": 0.5, "beta": 0.5}},
                        "normals": {"channels": 3, "activation": "none", "weight": 1, "loss_fn": "MaskedCosineLoss"}}),
    # configs[4]: manual 6-stage topology capped at 320 features, 2 input channels, 160^3 (run it with --dtype fp16)
    "cfg5": dict(patch=(160, 160, 160), in_channels=2, batch=1, autoconfigure=False,
                 model_config={"basic_encoder_block": "BasicBlockD", "basic_decoder_block": "ConvBlock",
                               "bottleneck_block": "BasicBlockD", "features_per_stage": [32, 64, 128, 256, 320, 320],
                               "num_stages": 6, "n_blocks_per_stage": [1, 3, 4, 6, 6, 6], "kernel_sizes": [3] * 6,
                               "n_conv_per_stage_decoder": [1] * 5, "strides": [1, 2, 2, 2, 2, 2]},
                 tasks={"sheet": {"channels": 1, "activation": "none", "weight": 1, "loss_fn": "BCEDiceLoss",
                                  "loss_kwargs": {"alpha": 0.5, "beta": 0.5}}}),
    # cfg2 with the reference's `squeeze_excitation: true` (tasks/dumb.yaml:53, ink.yaml:53): SE gate in all 26 encoder blocks
    "cfg2se": dict(patch=(128, 128, 128), in_channels=1, batch=2, autoconfigure=True, model_config={"squeeze_excitation": True},
                   tasks={"sheet": {"channels": 1, "activation": "none", "weight": 1, "loss_fn": "BCEDiceLoss",
                                    "loss_kwargs": {"alpha": 0.5, "beta": 0.5}}}),
    # the reference's own task files at their own sizes (values read from tasks/ink.yaml / tasks/dumb.yaml: autoconfigure, conv_bias,
    # squeeze_excitation, batch 3): ink = anisotropic 14 x 256 x 256, 7 stages; dumb = 128^3 with sheet + normals decoders
    "inkyaml": dict(patch=(14, 256, 256), in_channels=1, batch=3, autoconfigure=True,
                    model_config={"conv_bias": True, "squeeze_excitation": True},
                    tasks={"ink": {"channels": 1, "activation": "none", "weight": 1, "loss_fn": "BCEDiceLoss",
                                   "loss_kwargs": {"alpha": 0.5, "beta": 0.5}}}),
    "dumbyaml": dict(patch=(128, 128, 128), in_channels=1, batch=3, autoconfigure=True,
                     model_config={"conv_bias": True, "squeeze_excitation": True},
                     tasks={"sheet": {"channels": 1, "activation": "none", "weight": 1, "loss_fn": "BCEDiceLoss",
                                      "loss_kwargs": {"alpha": 0.5, "beta": 0.5}},
                            "normals": {"channels": 3, "activation": "none", "weight": 1, "loss_fn": "MaskedCosineLoss"}}),
    # configs[0]: 64^3 plumbing case
    "cfg1": dict(patch=(64, 64, 64), in_channels=1, batch=2, autoconfigure=True, model_config={},
                 tasks={"sheet": {"channels": 1, "activation": "none", "weight": 1, "loss_fn": "BCEDiceLoss",
                                  "loss_kwargs": {"alpha": 0.5, "beta": 0.5}}}),
}


def pmc_traffic(kernel_label):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes
    (profiles/*pmc_hbm_traffic.json: FETCH_SIZE x2 x1024 + WRITE_SIZE x1024, see scripts/pmc_traffic.py)."""
    import glob
    def _point(path):           # r01_<letters>_...: a, b, ..., z, aa, ab, ... in measurement order
        parts = os.path.basename(path).split("_")
        tag = parts[1] if len(parts) > 1 else ""
        return (parts[0], len(tag), tag)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_hbm_traffic.json")), key=_point)
    if not files:
        return None
    try:
        data = json.load(open(files[-1]))
    except Exception:
        return None
    base = kernel_label.split(":")[-1]                 # e.g. conv_halo_kernel<64>, wgrad_halo_kernel, igemm_kernel<256,64>
    name = base.split("<")[0]
    args = base.split("<")[1].rstrip(">").split(",") if "<" in base else []
    want = "".join(f"Li{a}E" for a in args)
    for k, v in data.items():
        # (rocprofv3 prints kernels with bool template arguments demangled, and garbles __bf16 into "bool _Accum")
        if name in k and ("DF16b" in k or "_Accum" in k) and want in k:
            return dict(bytes_per_launch=v["read_bytes_per_launch"] + v["write_bytes_per_launch"],
                        read=v["read_bytes_per_launch"], write=v["write_bytes_per_launch"], source=os.path.basename(files[-1]))
    return None


def pmc_traffic_live(workload, dtype, kernel_label, timeout_s=150):
    """HBM bytes per launch of `kernel_label` measured NOW: two child runs of this script under `rocprofv3 --pmc` (FETCH_SIZE, then
    WRITE_SIZE -- separate passes, counters only, as MI355X_MICROARCH.md prescribes), 3 steps each, summarised like
    scripts/pmc_traffic.py (KiB units, FETCH_SIZE doubled).  None when rocprofv3 is missing or a pass fails."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None
    base = kernel_label.split(":")[-1]
    name = base.split("<")[0]
    args = base.split("<")[1].rstrip(">").split(",") if "<" in base else []
    want = "".join(f"Li{a}E" for a in args)
    tot = {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0]}
    for counter in tot:
        with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
            cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", tmp, "-o", "p", "--", sys.executable, os.path.abspath(__file__),
                   "--workload", workload, "--dtype", dtype, "--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--no-kernel-timing",
                   "--no-h2d", "--no-pmc"]
            try:
                subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=dict(os.environ, TMPDIR="/tmp"))
            except subprocess.TimeoutExpired:
                return None
            for f in glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    k = r["Kernel_Name"]
                    if r["Counter_Name"] == counter and name in k and ("DF16b" in k or "_Accum" in k or "DF16_" in k) and want in k:
                        tot[counter][0] += float(r["Counter_Value"])
                        tot[counter][1] += 1
    nf, nw = tot["FETCH_SIZE"][1], tot["WRITE_SIZE"][1]
    if not nf or not nw:
        return None
    rd, wr = 2.0 * tot["FETCH_SIZE"][0] * 1024 / nf, tot["WRITE_SIZE"][0] * 1024 / nw
    return dict(bytes_per_launch=rd + wr, read=rd, write=wr, launches_sampled=nf,
                source="live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child runs of this command (3 steps each)")


def make_mgr(w):
    from types import SimpleNamespace
    return SimpleNamespace(tasks=w["tasks"], train_patch_size=tuple(w["patch"]), train_batch_size=w["batch"],
                           in_channels=w["in_channels"], vram_max=16.0, autoconfigure=w["autoconfigure"],
                           model_config=dict(w["model_config"]), verbose=False)


def synthetic_batch(w, batch, seed, device):
    """image rand in [0,1]; seg target (rand > 0.8); normals: unit vectors zeroed off the sheet"""
    g = torch.Generator().manual_seed(seed)
    x = torch.rand((batch, w["in_channels"], *w["patch"]), generator=g)
    seg = (torch.rand((batch, 1, *w["patch"]), generator=g) > 0.8).float()
    targets = {}
    for name, info in w["tasks"].items():
        if info.get("loss_fn") == "MaskedCosineLoss":
            v = torch.randn((batch, info["channels"], *w["patch"]), generator=g)
            targets[name] = (v / v.norm(dim=1, keepdim=True).clamp(min=1e-8)) * seg
        else:
            targets[name] = seg
    return x.to(device), {k: v.to(device) for k, v in targets.items()}


def host_threads():
    """the CPU share of this process (the GPU box gives 16 cores per GPU; os.cpu_count() reports the whole host)"""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def cpu_baseline_worker(workload, threads, timed=2):
    """runs in a child process: the oracle (pure-PyTorch CPU restatement of the reference path, fp32) -- ONE model and
    optimizer, one untimed warm-up step and then TWO timed full train steps of the workload at batch 1 (BASELINE.md 3:
    >= 1 warm-up + several timed; cfg2 costs ~7 s per step on 16 threads, so the sample stays within ~20-25 s)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import resenc_oracle as oracle
    w = WORKLOADS[workload]
    torch.set_num_threads(threads)

    patch = w["patch"]
    mgr = oracle.make_mgr(patch, w["tasks"], w["in_channels"], 1, w["autoconfigure"], w["model_config"])
    torch.manual_seed(0)
    net = oracle.NetworkFromConfig(mgr)
    opt = torch.optim.AdamW(net.parameters(), lr=1e-3, weight_decay=0.0)
    x, t = oracle.synthetic_batch(1, w["in_channels"], patch, w["tasks"], 1234)

    def one_step():
        t0 = time.perf_counter()
        loss = oracle.train_loss(net(x), t, w["tasks"])
        loss.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 3)
        opt.step()
        opt.zero_grad(set_to_none=True)
        return time.perf_counter() - t0

    warm = one_step()
    times = [one_step() for _ in range(timed)]
    dt = sum(times) / len(times)
    print(json.dumps(dict(value=1.0 / dt, unit="patches/s", cores=threads, kind="port",
                          sample=f"{timed} timed full train steps (fwd+loss+bwd+clip+AdamW) after 1 warm-up ({warm:.1f} s) of {workload} "
                                 f"at batch 1, patch {'x'.join(map(str, patch))}, fp32, torch {torch.__version__} CPU: "
                                 + ", ".join(f"{v:.1f} s" for v in times))),
          flush=True)


def cpu_baseline(workload, timeout_s=300):
    import subprocess
    threads = host_threads()
    print(f"[bench] timing the CPU oracle on {threads} threads (bounded: 1 warm-up + 2 timed steps, <= {timeout_s} s) ...",
          file=sys.stderr, flush=True)
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--workload", workload,
                            "--cpu-threads", str(threads)], capture_output=True, text=True, timeout=timeout_s,
                           env=dict(os.environ, HIP_VISIBLE_DEVICES="", OMP_NUM_THREADS=str(threads)))
        for ln in reversed(r.stdout.strip().splitlines()):
            if ln.startswith("{"):
                return json.loads(ln)
        return dict(value=None, unit="patches/s", cores=threads, kind="port", sample=f"failed: {r.stderr[-300:]}")
    except subprocess.TimeoutExpired:
        return dict(value=None, unit="patches/s", cores=threads, kind="port",
                    sample=f"3 {workload} steps at batch 1 did not finish within {timeout_s} s on {threads} threads")


def clock_probe(dtype, iters=30):
    """The dominant kernel ALONE on one of its cfg2 shapes (64 -> 64 channels, 64^3, batch 2), on random and on all-zero operands:
    MI355X lowers its clock under an MFMA-dense load on random data (MI355X_MICROARCH.md, DVFS give-back), so the same binary is
    20-38 % faster on zeros.  The ratio says how much of `1 - roofline.frac` is power management rather than kernel structure
    (the peak in `roofline` is the 2.4 GHz nameplate figure)."""
    from mt3d_amd.engine import ops, lib as _lib
    n, c, dims = 2, 64, (64, 64, 64)
    flops = 2.0 * n * dims[0] * dims[1] * dims[2] * c * c * 27
    out = {"shape": f"{c}->{c} channels, {dims[0]}^3, batch {n}, 3x3x3 weight gradient", "iters": iters}
    for label, mk in (("random", torch.randn), ("zeros", torch.zeros), ("random_again", torch.randn)):
        x = ops.Act(mk((n, *dims, c), device="cuda").to(dtype))
        dy = ops.Act(mk((n, *dims, c), device="cuda").to(dtype))
        dw = torch.empty((c, c, 3, 3, 3), device="cuda")
        fn = lambda: ops.conv3d_bwd_weight(x, dy, dw, (3, 3, 3), (1, 1, 1))
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        out[label] = {"us_per_launch": us, "tflops": flops / us / 1e6}
        out["kernel"] = _lib.load().rx_last_conv_kernel().decode()
    out["zeros_over_random"] = out["zeros"]["tflops"] / max(out["random"]["tflops"], out["random_again"]["tflops"])
    return out


def through_trainer(workload, batch, dtype, warmup, steps):
    """the same workload through the reference's plug-in surface: `BaseTrainer` (train.py:19-339 mirror) with its hooks, a
    DataLoader over `SyntheticPatchDataset` (pinned memory, worker processes), torch.compile wrapper, autocast, the engine's AdamW
    behind `_get_optimizer`.  Epoch 1 warms up (plan build), epoch 2 is reported (`BaseTrainer.last_patches_per_sec`: wall time of
    the epoch's training loop between two device synchronisations, loss scalars accumulated on the device)."""
    import tempfile
    import yaml
    from mt3d_amd.train import BaseTrainer
    w = WORKLOADS[workload]
    n_steps = max(warmup, 1) + steps
    torch.set_num_threads(host_threads())      # the box's CPU share, not the host's core count (oversubscribed copies crawl)
    with tempfile.TemporaryDirectory() as tmp:
        cfg = {
            "tr_setup": {"model_name": f"bench_{workload}", "autoconfigure": w["autoconfigure"], "tr_val_split": 0.95,
                         "ckpt_out_base": os.path.join(tmp, "ckpt"), "tensorboard_log_dir": os.path.join(tmp, "tb")},
            "tr_config": {"optimizer": "AdamW", "initial_lr": 1e-3, "weight_decay": 0, "gradient_accumulation": 1,
                          "num_dataloader_workers": int(os.environ.get("RX_BENCH_WORKERS", "0")), "patch_size": list(w["patch"]), "batch_size": batch,
                          "max_steps_per_epoch": n_steps, "max_val_steps_per_epoch": 1, "max_epoch": 2,
                          "amp_dtype": dtype, "engine_optimizer": True,
                          "compile": os.environ.get("RX_BENCH_COMPILE", "1") != "0"},
            "model_config": dict(w["model_config"]),
            "dataset_config": {"synthetic": True, "synthetic_length": int(n_steps * batch / 0.95) + 2 * batch,
                               "in_channels": w["in_channels"], "targets": w["tasks"]},
            "inference_config": {},
        }
        path = os.path.join(tmp, "cfg.yaml")
        with open(path, "w") as f:
            yaml.safe_dump(cfg, f)
        cwd = os.getcwd()
        os.chdir(tmp)
        try:
            class Quiet(BaseTrainer):
                rates = []

                def _log(self, *a):
                    if a and str(a[0]).startswith("[Train]"):
                        self.rates.append(self.last_patches_per_sec)
            tr = Quiet(path, verbose=False)
            tr.train()
        finally:
            os.chdir(cwd)
    return dict(value=Quiet.rates[-1], unit="patches/s", ms_per_step=batch / Quiet.rates[-1] * 1e3, steps=n_steps,
                epochs_rates=Quiet.rates,
                note="BaseTrainer.last_patches_per_sec of epoch 2 (DataLoader + pinned H2D + autocast + loss + clip + EngineAdamW)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--dtype", default="bf16", choices=sorted(DTYPES))
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default: the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="take roofline.traffic from the newest profiles/*pmc_hbm_traffic.json "
                                                          "instead of two live rocprofv3 --pmc child runs")
    ap.add_argument("--no-h2d", action="store_true", help="skip the second timed loop that feeds every step from pinned host memory")
    ap.add_argument("--through-trainer", action="store_true",
                    help="also run the workload through the plug-in surface (BaseTrainer + SyntheticPatchDataset + DataLoader) "
                         "and report its patches/s as `trainer`")
    ap.add_argument("--dist-backend", default="nccl", help=argparse.SUPPRESS)   # "gloo": rehearse N>1 on a 1-GPU box
    ap.add_argument("--bucket-mb", type=float, default=128.0,
                    help="gradient bucket size of the all-reduce in MiB (N > 1; engine/ddp.py::GradSync); xGMI is point-to-point, "
                         "so few large buckets keep RCCL in its bandwidth regime")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-threads", type=int, default=0, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_only:
        cpu_baseline_worker(args.workload, args.cpu_threads or host_threads())
        return

    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
    local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.dist_backend)

    import mt3d_amd  # noqa: F401
    from mt3d_amd.builders.build_network_from_config import NetworkFromConfig
    from mt3d_amd.engine import ops
    from mt3d_amd.engine.ddp import GradSync, broadcast_parameters
    from mt3d_amd.engine.streamed_step import StreamedOptimizerStep
    from mt3d_amd.training.losses.losses import LOSS_FN_MAP

    w = dict(WORKLOADS[args.workload])
    batch = args.batch or w["batch"]
    w["batch"] = batch
    torch.manual_seed(0)                               # random-init weights of the named architecture
    net = NetworkFromConfig(make_mgr(w)).to(device)
    net.compute_dtype = DTYPES[args.dtype]
    if world > 1:
        broadcast_parameters(net)
    net.train()
    loss_fns = {k: LOSS_FN_MAP[v.get("loss_fn", "BCEDiceLoss")](**v.get("loss_kwargs", {})) for k, v in w["tasks"].items()}
    params = [p for p in net.parameters()]
    # AdamW (lr 1e-3, wd 0: example.yaml:15-17) + clip_grad_norm_(3), as the reference's loop.
    # Default (RX_ENGINE_ADAMW=2): the engine's AdamW kernel per parameter (`rx_adamw_flat`: update + clip coefficient in ONE
    # pass over p, g, m, v; same arithmetic as torch's, tests/test_optim_gpu.py) -- 0.5 ms per step faster than torch's fused
    # multi-tensor AdamW fed with the clip coefficient (which also writes the scaled gradients back).  The weight re-pack stays on
    # the side stream under the next forward.  =1: re-pack fused into the update too (slower overall); =0: torch.optim.AdamW.
    mode = os.environ.get("RX_ENGINE_ADAMW", "2")
    engine_opt = mode in ("1", "2")
    if engine_opt:
        from mt3d_amd.training.optim import EngineAdamW
        opt = EngineAdamW(params, model=net if mode == "1" else None, lr=1e-3, weight_decay=0.0)
    else:
        opt = torch.optim.AdamW(params, lr=1e-3, weight_decay=0.0, fused=True)
    x, targets = synthetic_batch(w, batch, 1234 + rank, device)
    sync = GradSync(bucket_bytes=int(args.bucket_mb * (1 << 20))) if world > 1 else None
    if sync is not None:
        sync.timing = True          # two event records per backward around finish()'s waits: the exposed tail of the overlap
    # RX_STREAMED_STEP=1: the optimizer update + weight re-pack on the engine's side stream in forward order, overlapped
    # with the next forward (bit-identical parameters).  Off by default: the GPU is throughput-saturated, hiding the
    # HBM-bound update under the forward measured 23.7 vs 23.5 ms per step
    stepper = StreamedOptimizerStep(opt, net) if os.environ.get("RX_STREAMED_STEP", "0") == "1" else None
    from mt3d_amd.training.optim import clip_and_step
    fused_clip = os.environ.get("RX_FUSED_CLIP", "1") != "0"   # same update as clip_grad_norm_(3) + step(), one gradient pass less

    def step():
        out = net(x)
        if sync is not None:
            for plan in net._plans.values():
                plan.grad_sync = sync
        loss = 0.0
        for name, gt in targets.items():
            loss = loss + loss_fns[name](out[name], gt) * w["tasks"][name].get("weight", 1.0)
        loss.backward()
        if stepper is not None and engine_opt:
            opt.clip_grad_norm(3)                 # norm only: the coefficient is applied inside the streamed update kernels
            stepper.step()
        elif stepper is not None:
            torch.nn.utils.clip_grad_norm_(params, 3)
            stepper.step()
        elif fused_clip or engine_opt:
            clip_and_step(opt, params, 3)       # clip(3) + AdamW; the clip scale rides inside the fused update kernel
        else:
            torch.nn.utils.clip_grad_norm_(params, 3)
            opt.step()
        opt.zero_grad(set_to_none=True)
        return loss

    if world == 1 and os.environ.get("RX_GRAPHS", "0") == "1":
        for _ in range(3):            # plan build + two eager passes + HIP graph capture: never inside the timed region
            step()
    for _ in range(args.warmup):
        step()
    if rank == 0:
        print(f"[bench] warm-up done, timing {args.steps} steps ...", file=sys.stderr, flush=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ddp_block = None
    if sync is not None:
        # self-describing multi-GPU run: what the collectives were, and how much of them the backward did NOT hide
        tails = sync.tail_ms()[-args.steps:]
        ddp_block = dict(sync.describe(), world_env=world, bucket_mb=args.bucket_mb,
                         backend_is_rccl=bool(args.dist_backend == "nccl" and torch.version.hip),
                         exposed_tail_ms=dict(mean=sum(tails) / max(len(tails), 1), max=max(tails) if tails else None, steps=len(tails)),
                         note="exposed tail = device time between the last backward kernel on the main stream and the moment every "
                              "bucket's all-reduce has landed (GradSync.finish); buckets are launched from inside the backward, in "
                              "gradient-readiness order, on a side HIP stream")
        sync.timing = False
    # ---- the same K steps with the H2D copy of image + targets INSIDE the step (the reference's step starts at
    # the `.to(device)` of a pinned DataLoader batch, train.py:195-201): two pinned host batches, two device batches, a copy
    # stream that brings batch i+1 while step i computes.  Reported beside `value` (which keeps the batch resident in HBM),
    # never instead of it.
    h2d = None
    if not args.no_h2d:
        host = []
        for j in range(2):
            hx, ht = synthetic_batch(w, batch, 4321 + 2 * rank + j, "cpu")
            host.append((hx.pin_memory(), {k: v.pin_memory() for k, v in ht.items()}))
        dev = [(torch.empty_like(x), {k: torch.empty_like(v) for k, v in targets.items()}) for _ in range(2)]
        copy_stream = torch.cuda.Stream(device=device)
        ready = [torch.cuda.Event() for _ in range(2)]
        consumed = [torch.cuda.Event() for _ in range(2)]

        def prefetch(i):
            b = i % 2
            copy_stream.wait_event(consumed[b])                 # the step that last read device batch b is over
            with torch.cuda.stream(copy_stream):
                dev[b][0].copy_(host[b][0], non_blocking=True)
                for k in targets:
                    dev[b][1][k].copy_(host[b][1][k], non_blocking=True)
                ready[b].record(copy_stream)

        def fed_step(i):
            nonlocal x, targets
            b = i % 2
            torch.cuda.current_stream().wait_event(ready[b])
            prefetch(i + 1)
            x, targets = dev[b]
            loss = step()
            consumed[b].record(torch.cuda.current_stream())
            return loss

        x0, t0_ = x, targets
        for b in range(2):
            consumed[b].record(torch.cuda.current_stream())
        prefetch(0)
        for i in range(2):
            fed_step(i)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        th = time.perf_counter()
        for i in range(2, 2 + args.steps):
            fed_step(i)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dth = time.perf_counter() - th
        x, targets = x0, t0_
        nbytes = x.numel() * 4 + sum(v.numel() * 4 for v in targets.values())
        h2d = dict(value=world * batch * args.steps / dth, unit="patches/s", ms_per_step=dth / args.steps * 1e3,
                   h2d_bytes_per_step=nbytes,
                   note="same loop, every step fed from pinned host memory (image + targets, double-buffered on a copy stream)")

    # Per-kernel durations for the roofline: HIP events around every conv launch, on the stream it is enqueued on, over
    # `prof_steps` further steps of the same loop right after the timed region (two event records per launch cost
    # ~4 % of a step: they stay out of `value`; with RX_GRAPHS=1 events cannot be inserted into the replayed graphs
    # either).  The committed rocprofv3 summary of this command covers the timed steps themselves.
    prof, prof_steps = None, 0
    if not args.no_kernel_timing:
        prof_steps = min(args.steps, 5)
        if rank == 0:
            prof = ops.LaunchProfiler()
            ops.set_profiler(prof)
        for _ in range(prof_steps):
            step()
        torch.cuda.synchronize()
        ops.set_profiler(None)
    # extra, outside the timed region: the same kernels WITHOUT stream overlap (weight gradients back on the main
    # stream) so that per-kernel durations are free of the contention the overlapped schedule creates on purpose
    iso = None
    if not args.no_kernel_timing:
        for plan in net._plans.values():
            plan.overlap_wgrad = False
        if rank == 0:
            iso = ops.LaunchProfiler()
            ops.set_profiler(iso)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        ops.set_profiler(None)
        for plan in net._plans.values():
            plan.overlap_wgrad = True
    if world > 1:
        tmax = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = tmax.item()
    final_loss = float(loss.detach())

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * batch * args.steps / dt
        peak = MFMA_PEAK_TFLOPS[args.dtype]
        roofline, kernels = None, {}
        if prof is not None:
            groups = prof.collect()
            for name, g in groups.items():
                kernels[name] = dict(ms_per_step=g["ms"] / prof_steps, avg_us_per_launch=g["ms"] * 1e3 / max(g["launches"], 1),
                                     launches_per_step=g["launches"] / prof_steps,
                                     tflops=g["flops"] / (g["ms"] * 1e-3) / 1e12 if g["ms"] > 0 else 0.0)
            if groups:
                dom = max(groups, key=lambda k: groups[k]["ms"])
                g = groups[dom]
                ach = g["flops"] / (g["ms"] * 1e-3) / 1e12
                tr = None
                if world == 1 and not args.no_pmc:
                    print("[bench] measuring HBM traffic of the dominant kernel (2 rocprofv3 --pmc child runs) ...", file=sys.stderr, flush=True)
                    try:
                        tr = pmc_traffic_live(args.workload, args.dtype, dom)
                    except Exception as e:      # noqa: BLE001 -- a profiler problem must not cost the bench line
                        print(f"[bench] live PMC pass failed: {e}", file=sys.stderr, flush=True)
                if tr is None and args.dtype == "bf16":
                    tr = pmc_traffic(dom)
                roofline = dict(bound="mfma", kernel=dom, achieved=ach, peak=peak, unit="TFLOP/s", frac=ach / peak,
                                traffic=tr["bytes_per_launch"] if tr else None, traffic_detail=tr,
                                avg_us_per_launch=g["ms"] * 1e3 / max(g["launches"], 1),
                                flops_per_launch=g["flops"] / max(g["launches"], 1),
                                note=f"HIP events over {prof_steps} further steps of the same loop, right after the timed region")
        hbm = None
        if prof is not None and prof.hbm:
            # HBM-bound launches: ALGORITHMIC bytes (each operand tensor once) / HIP-event time on their stream,
            # in the overlapped step (contended with the other stream), as fraction of the 8 TB/s HBM3E peak
            hbm = {}
            for name, g in sorted(prof.hbm.items(), key=lambda kv: -kv[1]["ms"]):
                gbps = g["bytes"] / (g["ms"] * 1e-3) / 1e9 if g["ms"] > 0 else 0.0
                hbm[name] = dict(ms_per_step=g["ms"] / prof_steps, launches_per_step=g["calls"] / prof_steps,
                                 GB_per_step=g["bytes"] / prof_steps / 1e9, achieved_GBps=gbps, frac_of_peak=gbps / HBM_PEAK_GBPS)
            hbm["_total"] = dict(ms_per_step=sum(v["ms_per_step"] for v in hbm.values()),
                                 GB_per_step=sum(v["GB_per_step"] for v in hbm.values()), peak_GBps=HBM_PEAK_GBPS)
        line = {
            "metric": "train patches/sec (b,c,z,y,x) ResEncM 1x128^3", "value": value, "unit": "patches/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{args.workload}: ResEncM ({net.num_stages} stages, feats {list(net.features_per_stage)}, "
                                   f"blocks {list(net.n_blocks_per_stage)}), "
                                   f"{w['in_channels']}-in, heads {list(w['tasks'])}, patch {'x'.join(map(str, w['patch']))}, "
                                   f"batch {batch}/GPU, full train step (fwd+loss+bwd+allreduce+clip+AdamW)",
                       "global_batch": world * batch, "parallelism": f"dp{world}"},
            "final_loss": final_loss,
            "hip_graphs": bool(world == 1 and os.environ.get("RX_GRAPHS", "0") == "1"),
            "launch_programs": bool(any(st.get("prog") is not None for plan in net._plans.values() for st in plan._pstate.values())),
            "roofline": roofline,
            "roofline_isolated": None,
            "kernels": kernels,
            "hbm": hbm,
            "with_h2d": h2d,
            "ddp": ddp_block,
        }
        if iso is not None:
            gi = iso.collect()
            if gi:
                isolated = {k: dict(avg_us_per_launch=v["ms"] * 1e3 / max(v["launches"], 1), tflops=v["flops"] / (v["ms"] * 1e-3) / 1e12)
                            for k, v in gi.items() if v["ms"] > 0}
                dom_i = roofline["kernel"] if roofline and roofline["kernel"] in isolated else max(isolated, key=lambda k: gi[k]["ms"])
                if iso.hbm:
                    line["hbm_isolated"] = {k: dict(achieved_GBps=v["bytes"] / (v["ms"] * 1e-3) / 1e9,
                                                    frac_of_peak=v["bytes"] / (v["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS)
                                            for k, v in iso.hbm.items() if v["ms"] > 0}
                line["roofline_isolated"] = dict(kernel=dom_i, achieved=isolated[dom_i]["tflops"], peak=peak, unit="TFLOP/s",
                                                 frac=isolated[dom_i]["tflops"] / peak,
                                                 avg_us_per_launch=isolated[dom_i]["avg_us_per_launch"],
                                                 note="same kernels, weight-gradient stream overlap disabled (2 extra steps)",
                                                 all=isolated)
        if world == 1 and not args.no_kernel_timing and args.dtype != "fp32":
            try:
                line["clock_probe"] = clock_probe(DTYPES[args.dtype])
            except Exception as e:      # noqa: BLE001 -- a diagnostic must not cost the bench line
                print(f"[bench] clock probe failed: {e}", file=sys.stderr, flush=True)
        if args.through_trainer and world == 1:
            line["trainer"] = through_trainer(args.workload, batch, args.dtype, args.warmup, args.steps)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.workload)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
