"""`BaseTrainer` -- the reference's trainer plug-in surface (train.py:19-339) around the HIP engine.

Same constructor and overridable hooks (`_build_model`, `_configure_dataset`, `_build_loss`, `_get_optimizer`,
`_get_scheduler`, `_get_scaler`, `_configure_dataloaders`, `train`), same loop semantics (gradient
accumulation flush condition incl. the reference's `len(train_dataloader)` quirk, `clip_grad_norm_(…, 3)`,
per-epoch checkpoint dict {'model','optimizer','scheduler','epoch'} pruned to 10, validation each epoch,
CosineAnnealingLR stepped per epoch, final `<model_name>_final.pth`).  Differences, all documented in DESIGN.md:
  * `model = torch.compile(model)` is kept (train.py:133) so that checkpoints carry the reference's `_orig_mod.` keys
    (train.py:250); the engine's forward is marked `torch.compiler.disable` -- it is one opaque autograd boundary over
    ctypes launches, there is nothing for a tracer to lower.  `tr_config.compile: false` skips the wrapper; checkpoints
    with or without the prefix are accepted on load;
  * `tr_config.amp_dtype` ("bf16" default | "fp16" | "fp32") picks the autocast dtype; a GradScaler is only
    enabled for fp16;
  * launched under `torch.distributed.run` it becomes data parallel (RCCL all-reduce overlapped with backward,
    engine/ddp.py) -- the reference is single-GPU;
  * the CLI passes arguments by keyword (the reference swaps `verbose` / `debug_dataloader` positionally);
  * TensorBoard / debug GIFs are optional extras and skipped when their packages are missing.
"""
import os
import time
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist
from torch.optim import SGD, AdamW
from torch.optim.lr_scheduler import CosineAnnealingLR
from torch.utils.data import DataLoader, SubsetRandomSampler

from .builders.build_network_from_config import NetworkFromConfig
from .configuration.config_manager import ConfigManager
from .dataloading.dataset import SyntheticPatchDataset, ZarrSegmentationDataset3D
from .engine.ddp import GradSync, broadcast_parameters
from .engine.streamed_step import StreamedOptimizerStep
from .training.losses.losses import LOSS_FN_MAP
from .training.optim import clip_and_step

_AMP = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": None}


class PinnedRingCollate:
    """`collate_fn` of the in-process loader (num_workers == 0): stacks the items of a batch straight into PERSISTENT pinned host
    buffers (a ring of `depth` batches) instead of `default_collate` + `pin_memory=True`.  At 128^3 the default path costs ~27 ms
    of host time per batch on the GPU box (a fresh 16 MB tensor per key: page faults on first touch, then a second copy into a
    freshly pinned one) -- more than the 18 ms the GPU needs for the step.  A slot is reused only after the H2D copies that
    read it have completed (`copied()` records an event on the copying stream)."""

    def __init__(self, depth=4):
        self.depth = depth
        self.slots = [None] * depth
        self.events = [None] * depth
        self.turn = 0
        self.last = None

    def __call__(self, items):
        slot = self.turn % self.depth
        self.turn += 1
        ev = self.events[slot]
        if ev is not None:
            ev.synchronize()
            self.events[slot] = None
        bufs = self.slots[slot] or {}
        out = {}
        for k, first in items[0].items():
            first = torch.as_tensor(first)
            shape = (len(items), *first.shape)
            b = bufs.get(k)
            if b is None or tuple(b.shape) != shape or b.dtype != first.dtype:
                b = torch.empty(shape, dtype=first.dtype, pin_memory=torch.cuda.is_available())
            for i, it in enumerate(items):          # plain memcpy per item (torch.stack(out=pinned) ran at 1.6 GB/s on the GPU box)
                b[i].copy_(torch.as_tensor(it[k]))
            out[k] = b
        self.slots[slot] = out
        self.last = slot
        return out

    def copied(self):
        """call after the `.to(device, non_blocking=True)` copies of the batch returned last were enqueued"""
        if self.last is not None and torch.cuda.is_available():
            ev = torch.cuda.Event()
            ev.record()
            self.events[self.last] = ev


class DeviceFeeder:
    """Iterates a DataLoader one batch AHEAD of the consumer: the host-to-device copies of batch i+1 are enqueued on a copy stream
    while step i computes, and the compute stream only waits for their event.  On the main stream (the reference's
    `batch["image"].to(device)`, train.py:199-203) the 33.5 MB of a cfg2 batch sit between two steps: 0.7 ms of a 17 ms step with
    the device idle (`scripts/step_timeline.py` on a `--through-trainer` trace: first kernel 707 us after the previous step's last)."""

    def __init__(self, loader, device):
        self.loader, self.device = loader, device
        self.stream = torch.cuda.Stream(device)
        ring = getattr(loader, "collate_fn", None)
        self.ring = ring if isinstance(ring, PinnedRingCollate) else None

    def __len__(self):
        return len(self.loader)

    def _stage(self, it):
        try:
            batch = next(it)
        except StopIteration:
            return None
        with torch.cuda.stream(self.stream):
            dev = {k: v.to(self.device, dtype=torch.float32, non_blocking=True) for k, v in batch.items()}
            if self.ring is not None:
                self.ring.copied()          # (records on the copy stream: the pinned slot is free once THESE copies are done)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return dev, ev

    def __iter__(self):
        it = iter(self.loader)
        nxt = self._stage(it)
        while nxt is not None:
            dev, ev = nxt
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            for t in dev.values():
                t.record_stream(cur)        # allocated on the copy stream, consumed on the compute stream
            yield dev                       # the consumer enqueues its step ...
            nxt = self._stage(it)           # ... and only then does the host collate and copy the next batch, under that step


class BaseTrainer:
    def __init__(self, config_file: str, verbose: bool = True, debug_dataloader: bool = False):
        self.mgr = ConfigManager(config_file, verbose=verbose)
        self.verbose = verbose
        self.debug_dataloader = debug_dataloader
        self.world = int(os.environ.get("WORLD_SIZE", 1))
        self.rank = int(os.environ.get("RANK", 0))
        self.local_rank = int(os.environ.get("LOCAL_RANK", 0))
        self.last_patches_per_sec = None

    # ---- hooks ----------------------------------------------------------------------------------
    def _build_model(self):
        return NetworkFromConfig(self.mgr)

    def _configure_dataset(self):
        if self.mgr.dataset_config.get("synthetic", False) or not self.mgr.volume_paths:
            return SyntheticPatchDataset(self.mgr)
        return ZarrSegmentationDataset3D(self.mgr)

    def _build_loss(self):
        fns = {}
        for name, info in self.mgr.tasks.items():
            key = info.get("loss_fn", "BCEDiceLoss")
            if key not in LOSS_FN_MAP:
                raise ValueError(f"Loss function {key} not found in LOSS_FN_MAP. Add it to the mapping and try again.")
            fns[name] = LOSS_FN_MAP[key](**info.get("loss_kwargs", {}))
        return fns

    def _get_optimizer(self, model):
        if self.mgr.optimizer == "SGD":
            return SGD(model.parameters(), lr=self.mgr.initial_lr, momentum=0.9, nesterov=True,
                       weight_decay=self.mgr.weight_decay)
        if bool(getattr(self.mgr, "tr_configs", {}).get("engine_optimizer", False)) and self._amp_name() != "fp16":
            # opt-in (`tr_config.engine_optimizer: true`): the engine's AdamW kernel (update + clip coefficient in one pass per
            # parameter, same arithmetic as torch's; training/optim/engine_adamw.py) -- 0.5 ms per cfg2 step faster
            from .training.optim import EngineAdamW
            return EngineAdamW(model.parameters(), model=None, lr=self.mgr.initial_lr, weight_decay=self.mgr.weight_decay)
        return AdamW(model.parameters(), lr=self.mgr.initial_lr, weight_decay=self.mgr.weight_decay)

    def _get_scheduler(self, optimizer):
        return CosineAnnealingLR(optimizer, T_max=self.mgr.max_epoch, eta_min=0)

    def _get_scaler(self):
        return torch.amp.GradScaler("cuda", enabled=self._amp_name() == "fp16")

    def _configure_dataloaders(self, dataset):
        n = len(dataset)
        idx = list(range(n))
        np.random.shuffle(idx)
        if self.world > 1 and dist.is_initialized():
            # ONE permutation for the whole job (rank 0's): per-rank shuffles would give every rank its own train / val
            # partition -- overlapping training slices, one rank's validation samples in another rank's training set
            perm = torch.tensor(idx, dtype=torch.int64)
            if dist.get_backend() == "nccl":
                perm = perm.cuda(self.local_rank)
            dist.broadcast(perm, src=0)
            idx = perm.cpu().tolist()
        split = int(np.floor(self.mgr.tr_val_split * n))
        tr, va = idx[:split], idx[split:] or idx[-1:]
        if self.world > 1:
            # each rank trains on its own slice; all slices have the SAME length (DistributedSampler drop_last semantics), so
            # every rank runs the same number of backward passes / collectives and flushes on the same iterations
            per_rank = len(tr) // self.world
            if per_rank == 0:
                raise ValueError(f"{len(tr)} training patches cannot be split over {self.world} ranks")
            tr = tr[:per_rank * self.world][self.rank::self.world]
        workers = self.mgr.train_num_dataloader_workers
        # in-process loading: batches are assembled directly in persistent pinned buffers (PinnedRingCollate); with worker
        # processes the default collate + the loader's pin-memory thread, workers kept alive across epochs
        ring = PinnedRingCollate() if workers == 0 else None
        extra = dict(collate_fn=ring, pin_memory=False) if ring is not None else dict(pin_memory=True, persistent_workers=True)
        train = DataLoader(dataset, batch_size=self.mgr.train_batch_size, sampler=SubsetRandomSampler(tr),
                           num_workers=workers, **extra)
        val = DataLoader(dataset, batch_size=1, sampler=SubsetRandomSampler(va), pin_memory=True,
                         num_workers=workers, **({"persistent_workers": True} if workers else {}))
        return train, val

    # ---- helpers ----------------------------------------------------------------------------------
    def _amp_name(self):
        return str(self.mgr.tr_configs.get("amp_dtype", "bf16")).lower()

    @staticmethod
    def _strip_compile_prefix(sd):
        return {(k[len("_orig_mod."):] if k.startswith("_orig_mod.") else k): v for k, v in sd.items()}

    def _log(self, *a):
        if self.rank == 0:
            print(*a, flush=True)

    # ---- training loop ----------------------------------------------------------------------------
    def train(self):
        if self.world > 1 and not dist.is_initialized():
            torch.cuda.set_device(self.local_rank)
            backend = os.environ.get("RX_DDP_BACKEND", "nccl")      # "nccl" IS RCCL on ROCm; "gloo" for rehearsals on one GPU
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank))
            else:
                dist.init_process_group(backend)
        device = torch.device("cuda", self.local_rank if self.world > 1 else torch.cuda.current_device())
        model = self._build_model()
        optimizer = self._get_optimizer(model)
        loss_fns = self._build_loss()
        dataset = self._configure_dataset()
        scheduler = self._get_scheduler(optimizer)
        scaler = self._get_scaler()
        model = model.to(device)
        engine_model = model                     # the NetworkFromConfig itself (plans, streamed step), whatever wraps it
        if bool(self.mgr.tr_configs.get("compile", True)):
            model = torch.compile(model)         # reference train.py:133; state_dict keys gain `_orig_mod.` (train.py:250)
        amp_dtype = _AMP[self._amp_name()]
        sync = None
        if self.world > 1:
            broadcast_parameters(model)
            sync = GradSync()
        train_loader, val_loader = self._configure_dataloaders(dataset)
        if self.debug_dataloader:
            self._log("debug_dataloader: shapes of one item:", {k: tuple(v.shape) for k, v in dataset[0].items()})
            return

        start_epoch = 0
        ckpt_dir = Path(self.mgr.ckpt_out_base)
        if self.rank == 0:
            os.makedirs(ckpt_dir, exist_ok=True)
        if self.mgr.checkpoint_path is not None and Path(self.mgr.checkpoint_path).exists():
            self._log(f"Loading checkpoint from {self.mgr.checkpoint_path}")
            ck = torch.load(self.mgr.checkpoint_path, map_location=device, weights_only=True)
            engine_model.load_state_dict(self._strip_compile_prefix(ck["model"]))
            if not self.mgr.load_weights_only:
                optimizer.load_state_dict(ck["optimizer"])
                scheduler.load_state_dict(ck["scheduler"])
                start_epoch = ck["epoch"] + 1
            else:
                scheduler = self._get_scheduler(optimizer)

        writer = None
        if self.rank == 0:
            try:
                from torch.utils.tensorboard import SummaryWriter
                writer = SummaryWriter(log_dir=self.mgr.tensorboard_log_dir)
            except Exception:
                writer = None
        accum = self.mgr.gradient_accumulation
        params = [p for p in model.parameters()]
        # opt-in (RX_STREAMED_STEP=1): optimizer update + weight re-pack on the engine's side stream, overlapped with the
        # next forward (falls back to a plain optimizer.step() for anything but a fused Adam/AdamW; never with a GradScaler)
        stepper = (StreamedOptimizerStep(optimizer, engine_model)
                   if isinstance(engine_model, NetworkFromConfig) and os.environ.get("RX_STREAMED_STEP", "0") == "1" else None)

        def forward_loss(batch, train_mode):
            staged = batch["image"].is_cuda          # a DeviceFeeder batch: already on the device, its pinned slot already released
            x = batch["image"].to(device, dtype=torch.float32, non_blocking=True)
            targets = {k: v.to(device, dtype=torch.float32, non_blocking=True) for k, v in batch.items() if k != "image"}
            ring = getattr(train_loader, "collate_fn", None)
            if train_mode and not staged and isinstance(ring, PinnedRingCollate):
                ring.copied()               # the pinned slot may be refilled once these copies are done
            with torch.autocast("cuda", dtype=amp_dtype, enabled=amp_dtype is not None):
                out = model(x)
                if sync is not None:
                    for plan in getattr(engine_model, "_plans", {}).values():
                        plan.grad_sync = sync
                total, per = 0.0, {}
                for name, gt in targets.items():
                    l = loss_fns[name](out[name], gt)
                    if train_mode:
                        l = l * self.mgr.tasks[name].get("weight", 1.0)
                    total = total + l
                    per[name] = l.detach()
            return total, per, x.shape[0]

        for epoch in range(start_epoch, self.mgr.max_epoch):
            model.train()
            running = {t: 0.0 for t in self.mgr.tasks}
            steps, patches = 0, 0
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            feeder = (DeviceFeeder(train_loader, device)
                      if device.type == "cuda" and os.environ.get("RX_DEVICE_FEEDER", "1") != "0" else train_loader)
            for i, batch in enumerate(feeder):
                if i >= self.mgr.max_steps_per_epoch:
                    break
                total, per, bsz = forward_loss(batch, True)
                flush = (i + 1) % accum == 0 or (i + 1) == len(train_loader)
                if sync is not None:        # DDP.no_sync equivalent: only the stepping micro-batch all-reduces
                    sync.require_sync = flush
                scaler.scale(total / accum).backward()
                if flush:
                    if stepper is not None and not scaler.is_enabled():
                        torch.nn.utils.clip_grad_norm_(params, 3)
                        stepper.step()
                    else:       # clip(3) + step; the clip scale rides inside a fused Adam/AdamW update (training/optim)
                        clip_and_step(optimizer, params, 3, scaler)
                    scaler.update()
                    optimizer.zero_grad(set_to_none=True)
                for k, v in per.items():
                    running[k] = running[k] + v         # device scalars: no host sync inside the epoch (the reference's
                steps += 1                              # `.item()` per task and step, train.py:215-218, stalls the queue)
                patches += bsz
            if stepper is not None:
                stepper.synchronize()
            torch.cuda.synchronize(device)
            dt = time.perf_counter() - t0
            running = {k: float(v) for k, v in running.items()}
            self.last_patches_per_sec = patches * self.world / max(dt, 1e-9)
            desc = " | ".join(f"{k}: {running[k] / max(steps, 1):.4f}" for k in running)
            self._log(f"[Train] Epoch {epoch + 1} => {desc} | {self.last_patches_per_sec:.2f} patches/s")
            if writer is not None:
                for k in running:
                    writer.add_scalar(f"train/{k}_loss", running[k] / max(steps, 1), epoch)

            if self.rank == 0:
                torch.save({"model": model.state_dict(), "optimizer": optimizer.state_dict(),
                            "scheduler": scheduler.state_dict(), "epoch": epoch},
                           f"{ckpt_dir}/{self.mgr.model_name}_{epoch + 1}.pth")
                ckpts = sorted(ckpt_dir.glob(f"{self.mgr.model_name}_*.pth"), key=lambda p: p.stat().st_mtime)
                while len(ckpts) > 10:
                    ckpts.pop(0).unlink()

            model.eval()
            with torch.no_grad():
                vrun, vsteps = {t: 0.0 for t in self.mgr.tasks}, 0
                for i, batch in enumerate(val_loader):
                    if i >= self.mgr.max_val_steps_per_epoch:
                        break
                    _, per, _ = forward_loss(batch, False)
                    for k, v in per.items():
                        vrun[k] = vrun[k] + v
                    vsteps += 1
                vrun = {k: float(v) for k, v in vrun.items()}
                for k in vrun:
                    self._log(f"Task '{k}', epoch {epoch + 1} avg val loss: {vrun[k] / max(vsteps, 1):.4f}")
            scheduler.step()

        self._log("Training Finished!")
        if self.rank == 0:
            torch.save(model.state_dict(), f"{self.mgr.model_name}_final.pth")
        return model


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser(description="Train script for the multi-task 3-D ResEnc U-Net (HIP engine).")
    ap.add_argument("--config_path", type=str, required=True)
    ap.add_argument("--debug_dataloader", action="store_true")
    ap.add_argument("--verbose", action="store_true")
    a = ap.parse_args()
    BaseTrainer(a.config_path, verbose=a.verbose, debug_dataloader=a.debug_dataloader).train()
