from .engine_adamw import EngineAdamW  # noqa: F401


def clip_and_step(optimizer, params, max_norm, scaler=None):
    """`clip_grad_norm_(params, max_norm)` followed by `optimizer.step()` (train.py:227-229) without the separate
    gradient-scaling pass: for torch's FUSED Adam / AdamW the clip coefficient is handed to the update kernel through its
    `grad_scale` input (the hook GradScaler uses; the kernel divides every gradient by it on the fly and stores the result
    back into `.grad`), so the gradients are read once by the norm and once by the update -- 1.7 GB less HBM traffic
    per cfg2 step.  Same arithmetic as the two calls (g * c vs g / (1/c): one rounding), the same total norm is returned.
    Anything else (an enabled GradScaler, a non-fused optimizer, EngineAdamW) takes the plain two-call path."""
    import torch
    params = [p for p in params if p.grad is not None]
    fused = (isinstance(optimizer, (torch.optim.AdamW, torch.optim.Adam)) and optimizer.defaults.get("fused")
             and (scaler is None or not scaler.is_enabled()) and all(p.is_cuda for p in params))
    if isinstance(optimizer, EngineAdamW):
        total = optimizer.clip_grad_norm(max_norm)
        optimizer.step()
        return total
    if not fused or not params:
        total = torch.nn.utils.clip_grad_norm_(params, max_norm)
        if scaler is not None:
            scaler.step(optimizer)
        else:
            optimizer.step()
        return total
    with torch.no_grad():
        norms = torch._foreach_norm([p.grad for p in params], 2.0)
        total = torch.linalg.vector_norm(torch.stack(norms), 2.0)
        coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
        optimizer.grad_scale = (1.0 / coef).to(torch.float32).reshape(())
        optimizer.found_inf = torch.zeros((), dtype=torch.float32, device=total.device)
    try:
        optimizer.step()
    finally:
        del optimizer.grad_scale, optimizer.found_inf
    return total
