"""Optimizer step hidden under the NEXT forward pass.

`optimizer.step()` of the train loop (train.py:229) is 2.2 ms of pure HBM traffic at cfg2 (213 M parameters x 7 fp32
accesses) at the very end of a step, with nothing to overlap it.  The engine never reads a convolution weight directly:
every conv consumes a packed compute-dtype copy that is refreshed on the side HIP stream and awaited through a
per-parameter event (engine/plan.py::refresh_packs).  So the update itself can move to that stream:

    StreamedOptimizerStep(optimizer, model).step()

enqueues, on the side stream and in FORWARD order, chunks of [fused Adam/AdamW update of the chunk's parameters ->
re-pack of those parameters -> event], and returns at once.  The next forward starts immediately; each conv waits only
for its own parameter's event, the few parameters the kernels read raw (stem, biases, heads -- updated first) are guarded
by one event at the start of the forward.  Same kernels and per-tensor arithmetic as `optimizer.step()` (torch's fused
multi-tensor Adam is element-wise per tensor), hence bit-identical parameters (tests/test_network_gpu.py).

Contract: between `step()` and the next forward of the model, code that touches the parameters or their gradients on
another stream must call `synchronize()` first (state_dict / checkpointing / evaluation through a different plan do it
through `Plan.run_forward`, which orders itself after the side stream for raw parameters; `zero_grad` is safe -- the
gradient storage is kept alive for the side stream).  Anything but a fused torch Adam/AdamW without amsgrad / maximize /
capturable falls back to a plain `optimizer.step()`.
"""
from typing import List

import torch


class StreamedOptimizerStep:
    def __init__(self, optimizer, model, chunk_bytes: int = 96 << 20):
        self.opt = optimizer
        self.model = model
        self.chunk_bytes = int(chunk_bytes)
        self._warm = False
        self._clip = None

    # ------------------------------------------------------------------------------------------------------------
    def _engine_opt(self):
        from ..training.optim.engine_adamw import EngineAdamW
        return isinstance(self.opt, EngineAdamW) and self.opt.model is None

    def _supported(self):
        if self._engine_opt():        # the engine's AdamW (flat mode): clip coefficient inside the update, no state warm-up needed
            return True
        if not isinstance(self.opt, (torch.optim.Adam, torch.optim.AdamW)):
            return False
        for g in self.opt.param_groups:
            if not g.get("fused") or g.get("amsgrad") or g.get("maximize") or g.get("capturable") or g.get("differentiable"):
                return False
        return getattr(self.opt, "grad_scale", None) is None and getattr(self.opt, "found_inf", None) is None

    def _train_plans(self):
        plans = [p for p in getattr(self.model, "_plans", {}).values() if p.needs_grad and p.device.type == "cuda"]
        if any(p._shadows for p in plans):
            raise NotImplementedError("streamed optimizer step: plans with padded channel extents pack from shadow tensors "
                                      "(feature counts that are not multiples of 32); use the plain step")
        return plans

    def synchronize(self):
        for plan in self._train_plans():
            if plan._side is not None:
                torch.cuda.current_stream().wait_stream(plan._side)

    # ------------------------------------------------------------------------------------------------------------
    def step(self):
        plans = self._train_plans()
        engine = self._engine_opt()
        if engine and (len(plans) != 1 or plans[0]._side is None):
            self.opt.step()
            return
        if engine:
            self._warm = True
        if not self._supported() or len(plans) != 1 or plans[0]._side is None or not self._warm:
            # first step (optimizer state not initialised yet), unsupported optimizer, or no single training plan
            self.opt.step()
            self._warm = True
            return
        plan = plans[0]
        plan.use_programs = False       # a recorded forward does not wait for packs that THIS step leaves in flight
        side, main = plan._side, torch.cuda.current_stream()
        group_of = {}
        for g in self.opt.param_groups:
            for p in g["params"]:
                group_of[id(p)] = g
        with_grad = [p for g in self.opt.param_groups for p in g["params"] if p.grad is not None]
        self._clip = self.opt.take_clip() if engine else None
        if not engine and any(p not in self.opt.state or "exp_avg" not in self.opt.state[p] for p in with_grad):
            self.opt.step()                               # a parameter got its first gradient: let torch create its state
            return
        packed = {id(e["param"]): e for e in plan.packs}
        raw = [p for p in with_grad if id(p) not in packed]
        ordered = [e["param"] for e in plan.packs if e["param"].grad is not None]     # plan.packs is in first-use order
        chunks: List[List[torch.Tensor]] = []
        cur, size = [], 0
        for p in ordered:
            cur.append(p)
            size += p.numel() * 4
            if size >= self.chunk_bytes:
                chunks.append(cur)
                cur, size = [], 0
        if cur:
            chunks.append(cur)

        side.wait_stream(main)                            # gradients (and their clipping) are complete
        with torch.cuda.stream(side):
            self._adam(raw, group_of)
            ev0 = torch.cuda.Event()
            ev0.record(side)
            self.model._raw_param_event = ev0             # every plan of the model waits for it once (Plan._forward_body)
            for chunk in chunks:
                self._adam(chunk, group_of)
                plan.repack([packed[id(p)] for p in chunk])
        if self._clip is not None:
            self._clip.record_stream(side)
            self._clip = None
        for p in with_grad:
            p.grad.record_stream(side)                    # zero_grad(set_to_none=True) may free it while the side stream reads
        self.opt._opt_called = True                       # what torch's lr schedulers look at

    def _adam(self, params, group_of):
        if not params:
            return
        if self._engine_opt():
            self.opt.update_subset(params, self._clip)    # one table-kernel launch per <= 48 tensors (rx_adamw_flat_multi)
            return
        from torch.optim.adam import adam
        by_group = {}
        for p in params:
            by_group.setdefault(id(group_of[id(p)]), (group_of[id(p)], []))[1].append(p)
        for g, ps in by_group.values():
            st = [self.opt.state[p] for p in ps]
            beta1, beta2 = g["betas"]
            adam(ps, [p.grad for p in ps], [s["exp_avg"] for s in st], [s["exp_avg_sq"] for s in st], [], [s["step"] for s in st],
                 amsgrad=False, has_complex=False, beta1=beta1, beta2=beta2, lr=g["lr"], weight_decay=g["weight_decay"],
                 eps=g["eps"], maximize=False, foreach=False, capturable=False, differentiable=False, fused=True,
                 grad_scale=None, found_inf=None, decoupled_weight_decay=g.get("decoupled_weight_decay", isinstance(self.opt, torch.optim.AdamW)))
