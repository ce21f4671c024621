"""Batch-dimension data parallelism: one process per GPU, gradients averaged with bucketed
all-reduce (torch.distributed backend "nccl" == RCCL over xGMI on ROCm) launched from INSIDE the
backward launch list as soon as a bucket's last gradient kernel is enqueued, on a side HIP stream,
so the collective overlaps the remaining (FLOP-heavy, shallow-stage) backward convolutions.

The reference has no multi-GPU path; this is a new-build requirement.  Design points:
  * the weight-gradient kernels write straight into views of the flat bucket (no pack copy);
  * bucket order == the plan's static gradient-readiness order (decoder heads -> decoder stages ->
    encoder stage 5 .. 0): the deep stages hold ~80 % of the bytes and finish first;
  * xGMI is point-to-point (7 links x ~153 GB/s): few, large buckets (default 128 MiB) keep RCCL's
    ring/tree in its bandwidth regime; parameters without a gradient (the unused deep-supervision
    heads) are not engine inputs and therefore never enter a bucket;
  * InstanceNorm statistics are per sample -> no statistic synchronisation exists;
  * STREAMS: a bucket's gradients are produced on more than one HIP stream (conv / transposed-conv weight gradients on
    the plan's side stream; SqueezeExcite fc gradients, head gradients and the zero fill of an unused task on the main
    stream).  `ready()` notes the stream it is called on (the producer's), and the collective's stream waits for an event
    recorded on EVERY stream that produced into the bucket -- not only on the stream of whoever closed it;
  * gradient accumulation (`require_sync = False` / `no_sync()`, the equivalent of DDP.no_sync for the reference's
    `gradient_accumulation > 1`, train.py:172,226-230): a micro-batch that does not step keeps its bucket LOCAL (carried,
    summed on the device) and hands autograd no gradient; the stepping micro-batch adds the carry to its own bucket before
    the one collective, so `.grad` ends up as mean_ranks(sum_microbatches g).
Works unchanged with the gloo backend on CPU tensors (used by the world_size-2 tests).
"""
import contextlib
import weakref
from typing import Dict, List, Optional

import torch
import torch.distributed as dist


class _Bucket:
    __slots__ = ("idxs", "offsets", "numel", "flat", "pflat", "pending", "work", "streams", "t_closed")

    def __init__(self):
        self.idxs, self.offsets, self.numel = [], {}, 0
        self.flat, self.pending, self.work = None, 0, None
        self.pflat = None          # the bucket's PERSISTENT storage (a plan that replays a recorded backward writes fixed addresses)
        self.streams = {}          # stream id -> torch.cuda.Stream of every producer of this bucket (this backward)
        self.t_closed = None       # (optional, GradSync.timing) event recorded on the collective's stream behind the collective


class GradSync:
    def __init__(self, process_group=None, bucket_bytes: int = 128 << 20, average: bool = True, force_collectives: bool = False):
        self.group = process_group
        self.bucket_bytes = int(bucket_bytes)
        self.tail_bytes = min(8 << 20, self.bucket_bytes // 4)      # size of the last bucket (see _plan_layout)
        self.average = average
        self.force_collectives = force_collectives      # issue the collectives even in a world of one (backend rehearsal on one GPU)
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # RCCL averages inside the collective (ncclAvg; probed on this image with scripts/nccl_avg_probe.py): no
        # pre-scale pass over the 853 MB of gradients.  gloo has no AVG -> pre-scale by 1/world there.
        self._native_avg = dist.is_initialized() and dist.get_backend(process_group) == "nccl"
        # static bucket layout per plan, keyed by the plan OBJECT (weakly: `id(plan)` can be recycled once a model drops its
        # plans in `_apply`)
        self._layout: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()
        # Gradient-accumulation carry, keyed by the PARAMETER (not by plan or bucket): within one accumulation window the
        # micro-batches may run on different plans of one model (a ragged last batch of an epoch has another batch size), and
        # the stepping micro-batch must reduce every carried gradient whatever plan produced it.
        self._carry: Dict[torch.nn.Parameter, torch.Tensor] = {}
        self._cur: Optional[List[_Bucket]] = None
        self._of: Dict[int, _Bucket] = {}
        self._params = None
        self._side = None
        self.persistent = False         # set by a plan that replays a recorded backward: bucket storage must not move
        self.require_sync = True        # False: accumulate locally, no collective (see no_sync)
        self._syncing = True            # value of require_sync latched by begin() for the running backward
        self.stats = dict(buckets=0, bytes=0, collectives=0)
        # timing = True: every finish() brackets its waits (collectives still in flight when the backward's last kernel is
        # enqueued = the EXPOSED tail of the overlap) with events on the current stream; tail_ms() reads them (bench.py's `ddp` block)
        self.timing = False
        self._tails: List[tuple] = []

    def describe(self):
        """what the bench line's `ddp` block reports about this synchroniser (static facts + the last backward's counters)"""
        backend = dist.get_backend(self.group) if dist.is_initialized() else None
        return dict(backend=backend, world=self.world, bucket_bytes=self.bucket_bytes, tail_bytes=self.tail_bytes,
                    reduce_op=("avg" if (self.average and self._native_avg) else ("sum/world" if self.average else "sum")),
                    buckets=self.stats["buckets"], bytes=self.stats["bytes"], collectives=self.stats["collectives"])

    def tail_ms(self):
        """exposed tail of every timed backward so far (ms), oldest first; clears the list"""
        out = []
        for a, b in self._tails:
            if isinstance(a, float):
                out.append((b - a) * 1e3)
            else:
                b.synchronize()
                out.append(a.elapsed_time(b))
        self._tails = []
        return out

    @contextlib.contextmanager
    def no_sync(self):
        """backward passes inside this context keep their gradients local (summed into a carry); the first backward
        outside it reduces carry + its own gradients in one collective per bucket"""
        old, self.require_sync = self.require_sync, False
        try:
            yield
        finally:
            self.require_sync = old

    @property
    def returns_grads(self):
        """does the running backward hand gradients to autograd?  (a local micro-batch keeps them in the carry)"""
        return self._syncing

    # ---- static layout per plan -----------------------------------------------------------------
    def _plan_layout(self, plan):
        key = plan
        if key in self._layout:
            return self._layout[key]
        buckets, cur = [], _Bucket()
        for idx in plan.grad_order:
            n = plan.params[idx].numel()
            if cur.numel and (cur.numel + n) * 4 > self.bucket_bytes:
                buckets.append(cur)
                cur = _Bucket()
            cur.offsets[idx] = cur.numel
            cur.idxs.append(idx)
            cur.numel += (n + 63) // 64 * 64          # keep every view 256-byte aligned
        if cur.numel:
            buckets.append(cur)
        # The collective of the bucket that holds the LAST-ready gradient is the only one nothing can hide.  In this network the
        # late gradients are the small ones (stem and the full-resolution stages: a few MB), so they get a bucket of their own
        # (<= tail_bytes) and the rest of what used to be the last bucket (39 MB at cfg2) goes out ~1.5 ms earlier.
        last = buckets[-1] if buckets else None
        if last is not None and len(last.idxs) > 1 and last.numel * 4 > 2 * self.tail_bytes:
            k, tail = len(last.idxs), 0
            while k > 1:
                n = (plan.params[last.idxs[k - 1]].numel() + 63) // 64 * 64
                if (tail + n) * 4 > self.tail_bytes:
                    break
                tail += n
                k -= 1
            if 0 < k < len(last.idxs):
                head, end = _Bucket(), _Bucket()
                for j, idx in enumerate(last.idxs):
                    b = head if j < k else end
                    b.offsets[idx] = b.numel
                    b.idxs.append(idx)
                    b.numel += (plan.params[idx].numel() + 63) // 64 * 64
                buckets[-1:] = [head, end]
        self._layout[key] = buckets
        return buckets

    # ---- per-backward protocol (called by Plan.run_backward) -----------------------------------
    def begin(self, plan):
        self._cur = self._plan_layout(plan)
        self._params = plan.params
        self._of = {}
        self._syncing = bool(self.require_sync)
        dev = plan.params[0].device
        for b in self._cur:
            if self.persistent:
                # created once and never replaced: an eager pass in between (profiler, a task left out of the loss) uses fresh
                # storage below and must not move what the recorded program writes to
                if b.pflat is None or b.pflat.device != dev:
                    b.pflat = torch.empty(b.numel, dtype=torch.float32, device=dev)
                b.flat = b.pflat
            else:
                b.flat = torch.empty(b.numel, dtype=torch.float32, device=dev)
            b.pending = len(b.idxs)
            b.work = None
            b.streams = {}
            for i in b.idxs:
                self._of[i] = b
        if dev.type == "cuda" and self._side is None:
            self._side = torch.cuda.Stream(device=dev)
        self.stats = dict(buckets=len(self._cur), bytes=sum(b.numel for b in self._cur) * 4, collectives=0)

    def _view(self, b, idx):
        p = self._params[idx]
        o = b.offsets[idx]
        return b.flat[o:o + p.numel()].view(p.shape)

    def alloc(self, idx):
        return self._view(self._of[idx], idx)

    def ready(self, idx):
        """called right after the LAST kernel writing gradient `idx` was enqueued, on the stream it was enqueued on"""
        b = self._of[idx]
        if b.flat.is_cuda:
            st = torch.cuda.current_stream()
            b.streams[st.cuda_stream] = st
        b.pending -= 1
        if b.pending == 0:
            self._launch(b)

    def _reduce(self, b):
        """(on the collective's stream / the host for CPU tensors) carry + own gradients, then the collective"""
        views = [self._view(b, i) for i in b.idxs]
        held = [(v, self._carry.pop(self._params[i], None)) for v, i in zip(views, b.idxs)]
        held = [(v, c) for v, c in held if c is not None]
        if held:
            torch._foreach_add_([v for v, _ in held], [c for _, c in held])
        if not self._syncing:
            # stays local; the stepping micro-batch picks it up -- on whatever plan it runs (a copy when the bucket storage is
            # reused by the next backward)
            keep = b.flat.clone() if self.persistent else b.flat
            for i in b.idxs:
                p = self._params[i]
                o = b.offsets[i]
                self._carry[p] = keep[o:o + p.numel()].view(p.shape)
            return
        if self.world == 1 and not self.force_collectives:
            return
        self.stats["collectives"] += 1
        if self.average and self._native_avg and b.flat.is_cuda:
            b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
        else:
            if self.average:
                b.flat.div_(self.world)        # pre-scale: sum of (g / world) == mean, no overflow step
            b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _launch(self, b):
        if (self.world == 1 and self._syncing and not self.force_collectives
                and not any(self._params[i] in self._carry for i in b.idxs)):
            return
        if b.flat.is_cuda:
            # every stream that produced into this bucket: the kernels enqueued there so far include all of the bucket's
            for st in b.streams.values():
                ev = torch.cuda.Event()
                ev.record(st)
                self._side.wait_event(ev)
            with torch.cuda.stream(self._side):
                self._reduce(b)
            b.flat.record_stream(self._side)
        else:
            self._reduce(b)

    def finish(self):
        cuda = bool(self._cur) and self._cur[0].flat.is_cuda
        t0 = None
        if self.timing:
            if cuda:
                t0 = torch.cuda.Event(enable_timing=True)
                t0.record()
            else:
                import time
                t0 = time.perf_counter()
        for b in self._cur or []:
            if b.pending != 0:
                raise RuntimeError("gradient bucket not completed: a parameter of the plan produced no gradient")
            if b.work is not None:
                b.work.wait()                          # makes the CURRENT stream wait for the collective
        if cuda and self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)
        if t0 is not None:
            if cuda:
                t1 = torch.cuda.Event(enable_timing=True)
                t1.record()
            else:
                import time
                t1 = time.perf_counter()
            self._tails.append((t0, t1))
        self._cur = None

    def drop_carry(self):
        """forget accumulated local gradients (e.g. `optimizer.zero_grad()` in the middle of an accumulation window)"""
        self._carry.clear()


def broadcast_parameters(module, src=0, group=None):
    """make every rank start from rank `src`'s weights (unique tensors only: the state_dict aliases
    of the reference's module tree point at the same storage)"""
    seen = set()
    for p in module.parameters():
        if id(p) in seen:
            continue
        seen.add(id(p))
        dist.broadcast(p.data, src=src, group=group)
