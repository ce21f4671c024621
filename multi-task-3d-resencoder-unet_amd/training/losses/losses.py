"""Task losses consumed by the train step (reference: training/losses/losses.py).  Same class names, constructor
arguments and arithmetic as the reference so `_build_loss` (train.py:43-66) maps YAML names 1:1.

`BCEDiceLoss` and `MaskedCosineLoss` -- the two losses of the BASELINE configs -- run as single-pass HIP kernels
(csrc/rx_loss.hip) whenever they are handed HIP tensors: one read of logits + target forward, one
read + one write backward, loss value and upstream gradient kept as device scalars (the torch formulation makes 5-8
passes per direction).  On a HIP device a missing librxunet.so is an error, not a fallback; tensors that live on the
CPU (the host-side unit tests of the trainer plumbing) take the torch formulation, which is also what the GPU tests
compare the kernels against.  The remaining losses of the reference's map are plain torch modules."""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _hip_eligible(pred, target):
    return pred.is_cuda and target.is_cuda and pred.shape == target.shape and pred.dim() >= 3


def _as_f32c(t):
    t = t if t.dtype == torch.float32 else t.float()
    return t if t.is_contiguous() else t.contiguous()


class _BCEDiceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, alpha, beta, smoothing):
        from ...engine import ops
        x, t = _as_f32c(logits.detach()), _as_f32c(target.detach())
        loss, coef = ops.bce_dice_loss_fwd(x, t, alpha, beta, smoothing, 1e-6)
        ctx.save_for_backward(x, t, coef)
        ctx.hp = (alpha, beta, smoothing, logits.dtype)
        return loss

    @staticmethod
    def backward(ctx, g):
        from ...engine import ops
        x, t, coef = ctx.saved_tensors
        alpha, beta, smoothing, dtype = ctx.hp
        g = _as_f32c(g)
        d = ops.bce_dice_loss_bwd(x, t, coef, g, alpha, beta, smoothing)
        return (d if dtype == torch.float32 else d.to(dtype)), None, None, None, None


class _MaskedCosineFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target):
        from ...engine import ops
        x, t = _as_f32c(pred.detach()), _as_f32c(target.detach())
        loss, coef = ops.masked_cosine_loss_fwd(x, t)
        ctx.save_for_backward(x, t, coef)
        ctx.dtype = pred.dtype
        return loss

    @staticmethod
    def backward(ctx, g):
        from ...engine import ops
        x, t, coef = ctx.saved_tensors
        d = ops.masked_cosine_loss_bwd(x, t, coef, _as_f32c(g))
        return (d if ctx.dtype == torch.float32 else d.to(ctx.dtype)), None


def flatten(tensor):
    """(N, C, *spatial) -> (C, N * prod(spatial))   (losses.py:321-333)"""
    c = tensor.size(1)
    return tensor.transpose(0, 1).reshape(c, -1)


def compute_per_channel_dice(input, target, epsilon=1e-6, weight=None):
    """V-Net dice per channel: 2 * sum(p t) / (sum(p^2) + sum(t^2))   (losses.py:17-43)"""
    assert input.size() == target.size(), "'input' and 'target' must have the same shape"
    p, t = flatten(input), flatten(target).float()
    inter = (p * t).sum(-1)
    if weight is not None:
        inter = weight * inter
    den = (p * p).sum(-1) + (t * t).sum(-1)
    return 2 * (inter / den.clamp(min=epsilon))


class DiceLoss(nn.Module):
    """1 - mean_c dice_c on sigmoid / softmax / raw inputs   (losses.py:95-138)"""

    def __init__(self, weight=None, normalization="sigmoid"):
        super().__init__()
        assert normalization in ("sigmoid", "softmax", "none")
        self.register_buffer("weight", weight)
        self.normalization = normalization

    def forward(self, input, target):
        if self.normalization == "sigmoid":
            input = torch.sigmoid(input)
        elif self.normalization == "softmax":
            input = torch.softmax(input, dim=1)
        return 1.0 - compute_per_channel_dice(input, target, weight=self.weight).mean()


class BCEWithLogitsLossLabelSmoothing(nn.Module):
    """targets y -> y (1 - 2 s) + s, then BCE-with-logits   (losses.py:217-238)"""

    def __init__(self, smoothing=0.1, reduction="mean"):
        super().__init__()
        self.smoothing, self.reduction = smoothing, reduction

    def forward(self, logits, targets):
        with torch.no_grad():
            smoothed = targets * (1.0 - 2.0 * self.smoothing) + self.smoothing
        return F.binary_cross_entropy_with_logits(logits, smoothed, reduction=self.reduction)


class BCEWithLogitsLossZSmooth(nn.Module):
    """label smoothing that grows linearly with the distance from the central Z slice (losses.py:240-304)"""

    def __init__(self, center_smoothing=0.1, edge_smoothing=0.4, reduction="mean"):
        super().__init__()
        self.center_smoothing, self.edge_smoothing, self.reduction = center_smoothing, edge_smoothing, reduction

    def forward(self, logits, targets):
        assert logits.shape == targets.shape, "Logits and targets must match in shape."
        d = logits.shape[2]
        z = torch.arange(d, device=logits.device, dtype=logits.dtype)
        ratio = (z - (d - 1) / 2.0).abs() / (d // 2)
        alpha = (self.center_smoothing + (self.edge_smoothing - self.center_smoothing) * ratio).view(1, 1, d, 1, 1)
        return F.binary_cross_entropy_with_logits(logits, targets * (1.0 - 2.0 * alpha) + alpha,
                                                  reduction=self.reduction)


class BCEDiceLoss(nn.Module):
    """alpha * smoothed BCE + beta * Dice   (losses.py:307-318)"""

    def __init__(self, alpha, beta):
        super().__init__()
        self.alpha, self.beta = alpha, beta
        self.bce = BCEWithLogitsLossLabelSmoothing(smoothing=0.1, reduction="mean")
        self.dice = DiceLoss()

    def forward(self, input, target):
        if _hip_eligible(input, target):
            return _BCEDiceFn.apply(input, target, float(self.alpha), float(self.beta), float(self.bce.smoothing))
        return self.alpha * self.bce(input, target) + self.beta * self.dice(input, target)


class MaskedCosineLoss(nn.Module):
    """1 - mean cosine similarity over voxels whose target normal is non-zero   (losses.py:187-215)"""

    def forward(self, pred, target):
        if _hip_eligible(pred, target) and pred.shape[1] <= 8:
            return _MaskedCosineFn.apply(pred, target)
        mask = (torch.norm(target, dim=1) > 1e-6).float()
        unit = pred / torch.norm(pred, dim=1, keepdim=True).clamp(min=1e-8)
        cos = F.cosine_similarity(unit, target, dim=1, eps=1e-8)
        return 1.0 - (cos * mask).sum() / (mask.sum() + 1e-8)


LOSS_FN_MAP = {
    "BCEDiceLoss": BCEDiceLoss,
    "BCEWithLogitsLossLabelSmoothing": BCEWithLogitsLossLabelSmoothing,
    "BCEWithLogitsLossZSmooth": BCEWithLogitsLossZSmooth,
    "BCEWithLogitsLoss": nn.BCEWithLogitsLoss,
    "BCELoss": nn.BCELoss,
    "CrossEntropyLoss": nn.CrossEntropyLoss,
    "MSELoss": nn.MSELoss,
    "MaskedCosineLoss": MaskedCosineLoss,
}
