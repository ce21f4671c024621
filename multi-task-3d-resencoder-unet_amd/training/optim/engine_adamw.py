"""AdamW for the HIP engine: optimizer update, gradient clipping and weight re-pack in one pass.

`torch.optim.AdamW` (what train.py:69-86 builds) followed by the engine's per-step weight re-pack touches every conv
weight three times per step (clip scale pass, Adam pass, pack pass): 3.7 ms of kernel time at cfg2.  `EngineAdamW` keeps
torch's `Optimizer` interface (param_groups, state_dict with `step` / `exp_avg` / `exp_avg_sq`, lr schedulers) and
torch's AdamW arithmetic (decoupled weight decay, bias correction), but its `step()`

  * multiplies the gradient by a DEVICE-scalar clip coefficient on the fly (`clip_grad_norm(max_norm)` below computes the
    norm exactly like `torch.nn.utils.clip_grad_norm_` and leaves the gradients untouched),
  * updates a conv / convT weight and rewrites BOTH packed compute-dtype copies of the model's training plan in the same
    kernel (`rx_adamw_pack`), so the next forward finds fresh packs and re-packs nothing,
  * updates the un-packed parameters (stem, biases, heads) with `rx_adamw_flat`.

Arithmetic is fp32 like torch's fused kernel; results agree with `torch.optim.AdamW(fused=True)` to fp32 round-off
(tests/test_optim_gpu.py), not bit for bit (different FMA contraction).  CPU parameters are refused: this optimizer only
exists for the engine."""
import ctypes
import os
from ctypes import c_void_p

import torch

from ...engine import lib as _l
from ...engine.lib import check, load, stream_ptr


def _p(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


class EngineAdamW(torch.optim.Optimizer):
    def __init__(self, params, model=None, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.model = model            # NetworkFromConfig whose training plan's packs are rewritten (optional)
        self._clip = None             # device scalar set by clip_grad_norm(), consumed by the next step()
        self._norm_ws = None          # scratch of rx_grad_norm_clip

    # ---- gradient clipping without touching the gradients -------------------------------------------------------
    @torch.no_grad()
    def clip_grad_norm(self, max_norm, norm_type=2.0):
        """same value as torch.nn.utils.clip_grad_norm_ (returned); the scale is applied inside the next step()"""
        grads = [p.grad for g in self.param_groups for p in g["params"] if p.grad is not None]
        if not grads:
            return torch.zeros(())
        if (float(norm_type) == 2.0 and os.environ.get("RX_ENGINE_GRAD_NORM", "1") != "0"
                and all(g.is_cuda and g.dtype == torch.float32 and g.is_contiguous() for g in grads)):
            # two launches (table kernel + one-workgroup finalize) instead of torch's ~20: norm AND coefficient on the device
            n = len(grads)
            VP, LP = ctypes.c_void_p * n, ctypes.c_long * n
            numel = LP(*[g.numel() for g in grads])
            need = load().rx_grad_norm_clip_partials(n, numel)
            if self._norm_ws is None or self._norm_ws.numel() < need or self._norm_ws.device != grads[0].device:
                self._norm_ws = torch.empty(int(need), dtype=torch.float32, device=grads[0].device)
            out = torch.empty(2, dtype=torch.float32, device=grads[0].device)
            from ...engine import ops as _ops
            _ops.timed_bytes("grad_norm_clip", 4 * sum(g.numel() for g in grads), lambda: check(
                load().rx_grad_norm_clip(n, VP(*[g.data_ptr() for g in grads]), numel, float(max_norm), _p(self._norm_ws),
                                         self._norm_ws.numel(), _p(out), stream_ptr()), "rx_grad_norm_clip"))
            self._clip = out[1:2]
            return out[0]
        norms = torch._foreach_norm(grads, norm_type)
        total = torch.linalg.vector_norm(torch.stack(norms), norm_type)
        self._clip = torch.clamp(max_norm / (total + 1e-6), max=1.0).to(torch.float32).reshape(1)
        return total

    def _plan(self):
        plans = [p for p in getattr(self.model, "_plans", {}).values() if p.needs_grad and p.device.type == "cuda"] \
            if self.model is not None else []
        # (a plan with padded channel extents packs from zero-padded SHADOW tensors: the fused update-and-pack kernel would write
        # un-padded layouts -- such plans take the plain update and re-pack in their next forward)
        return plans[0] if len(plans) == 1 and not plans[0]._shadows else None

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        plan = self._plan()
        packed = {id(e["param"]): e for e in plan.packs} if plan is not None else {}
        if plan is not None and plan._side is not None:
            torch.cuda.current_stream().wait_stream(plan._side)     # nobody may still read the packs / gradients
        lib = load()
        clip = self._clip
        self._clip = None
        for group in self.param_groups:
            b1, b2 = group["betas"]
            flat = []                      # un-packed parameters of this group: one C call for all of them (rx_adamw_flat_multi)
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise _l.RxError("EngineAdamW updates contiguous fp32 parameters on the HIP device only")
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] = int(st["step"]) + 1
                ent = packed.get(id(p))
                if ent is not None:
                    kind = 0 if ent["kind"] == "conv" else 1
                    A, B = p.shape[0], p.shape[1]
                    taps = p[0, 0].numel()
                    check(lib.rx_adamw_pack(_l.DTYPE_CODE[plan.dtype], _p(p), _p(g), _p(st["exp_avg"]), _p(st["exp_avg_sq"]), _p(clip),
                                            group["lr"], b1, b2, group["eps"], group["weight_decay"], st["step"], kind, A, B, taps,
                                            _p(ent["w_fwd"]), _p(ent["w_bwd"]), stream_ptr()), "rx_adamw_pack")
                    ent["version"], ent["ptr"], ent["event"] = p._version, p.data_ptr(), None
                    ent["epoch"] = getattr(plan.net, "_weights_epoch", 0)
                else:
                    flat.append((p, g, st))
            self._flat_update(group, flat, clip)
        return loss

    def _flat_update(self, group, items, clip):
        """items: (param, grad, state) of ONE param group, state["step"] already advanced -> rx_adamw_flat_multi on the current stream
        (tensors that share the step count go in one call: normally all of them)"""
        b1, b2 = group["betas"]
        by_step = {}
        for p, g, st in items:
            by_step.setdefault(st["step"], []).append((p, g, st))
        for step_no, its in by_step.items():
            n = len(its)
            VP, LP = ctypes.c_void_p * n, ctypes.c_long * n
            from ...engine import ops as _ops
            # AdamW touches p, g, m, v (read) and p, m, v (write): 28 bytes per parameter
            _ops.timed_bytes("adamw", 28 * sum(p.numel() for p, _, _ in its), lambda: check(
                load().rx_adamw_flat_multi(n, VP(*[p.data_ptr() for p, _, _ in its]), VP(*[g.data_ptr() for _, g, _ in its]),
                                           VP(*[st["exp_avg"].data_ptr() for _, _, st in its]),
                                           VP(*[st["exp_avg_sq"].data_ptr() for _, _, st in its]),
                                           LP(*[p.numel() for p, _, _ in its]), _p(clip), group["lr"], b1, b2, group["eps"],
                                           group["weight_decay"], step_no, stream_ptr()), "rx_adamw_flat_multi"))

    # ---- pieces of step() for engine/streamed_step.py: the same update for a SUBSET of the parameters, on the current stream ----
    def take_clip(self):
        """the device scalar left by clip_grad_norm() (None if it was not called); consumed"""
        clip, self._clip = self._clip, None
        return clip

    @torch.no_grad()
    def update_subset(self, params, clip):
        group_of = {id(p): g for g in self.param_groups for p in g["params"]}
        per_group = {}
        for p in params:
            if p.grad is None:
                continue
            if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                raise _l.RxError("EngineAdamW updates contiguous fp32 parameters on the HIP device only")
            g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
            st = self.state[p]
            if not st:
                st["step"] = 0
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            st["step"] = int(st["step"]) + 1
            grp = group_of[id(p)]
            per_group.setdefault(id(grp), (grp, []))[1].append((p, g, st))
        for grp, items in per_group.values():
            self._flat_update(grp, items, clip)
