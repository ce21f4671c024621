"""ORACLE -- TEST INFRASTRUCTURE ONLY.

CPU (torch fp32/fp64) restatement of the reference's hot path: the multi-task 3-D residual-encoder
U-Net `NetworkFromConfig` forward (autograd supplies the backward), plus the two losses the
BASELINE train step uses.  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s
`cpu_baseline` leg may import this file; the product package
(`multi-task-3d-resencoder-unet_amd/`) never does and fails loudly without its HIP library.

Parity status: PINNED.  `tests/test_oracle_vs_reference.py` (runs only where /root/reference
exists) checks key-for-key `state_dict` equality and forward/backward equality against the real
reference imported through `oracle/ref_shim.py`; `tests/golden/*.npz` (made by
`oracle/make_golden.py` from the real reference) pin it everywhere else.  PARITY UNPINNED for two
options: `squeeze_excitation=True` and `stochastic_depth_p>0`.  Their arithmetic lives in the
un-vendored third-party `dynamic_network_architectures` (MIC-DKFZ; version unpinned by the reference,
package absent from this image, no reference test or fixture covers it), so `SqueezeExcite` /
`DropPath` below restate that package's published source (a copy of timm's squeeze_excite.py /
drop.py with a `conv_op` argument) anchored on the reference's call sites (resblocks.py:79-87,
109-112); nothing here could be checked against a run of the real classes.

Each function cites the reference file:line (relative to /root/reference) it follows.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Dict, List, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# topology planner  (builders/utils.py:334-402, :428-445)
# --------------------------------------------------------------------------------------
def plan_pooling(patch_size: Sequence[int], min_feature_map_size: int = 4,
                 spacing: Sequence[float] | None = None, max_numpool: int = 999999):
    """nnU-Net style: halve every axis that is still >= 2*min_feature_map_size (and whose
    spacing is within 2x of the finest poolable axis) until none qualifies.
    Returns (num_pool_per_axis, strides_per_stage, kernel_sizes_per_stage)."""
    dim = len(patch_size)
    spacing = [1.0] * dim if spacing is None else [float(s) for s in spacing]
    size = [int(s) for s in patch_size]
    strides = [tuple([1] * dim)]
    kernels: List[tuple] = []
    npool = [0] * dim
    ks = [1] * dim
    while True:
        axes = [a for a in range(dim) if size[a] >= 2 * min_feature_map_size]
        if not axes:
            break
        finest = min(spacing[a] for a in axes)
        axes = [a for a in axes if spacing[a] / finest < 2 and npool[a] < max_numpool]
        if not axes:
            break
        for a in range(dim):
            if ks[a] != 3 and spacing[a] / min(spacing) < 2:
                ks[a] = 3
        step = [1] * dim
        for a in axes:
            step[a] = 2
            npool[a] += 1
            spacing[a] *= 2
            size[a] = int(math.ceil(size[a] / 2))
        strides.append(tuple(step))
        kernels.append(tuple(ks))
    kernels.append(tuple([3] * dim))  # bottleneck
    return npool, tuple(strides), tuple(kernels)


def blocks_per_stage(n_stages: int) -> List[int]:
    """builders/utils.py:428-445"""
    table = [1, 3, 4]
    return [table[i] if i < 3 else 6 for i in range(n_stages)]


def _as_list(v, dim):
    return list(v) if isinstance(v, (list, tuple)) else [v] * dim


# --------------------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------------------
class _Ops:
    """conv/pool/norm/dropout classes by dimensionality (build_network_from_config.py:188-205)."""

    def __init__(self, dim: int):
        self.dim = dim
        self.conv = {2: nn.Conv2d, 3: nn.Conv3d}[dim]
        self.convT = {2: nn.ConvTranspose2d, 3: nn.ConvTranspose3d}[dim]
        self.pool = {2: nn.AvgPool2d, 3: nn.AvgPool3d}[dim]
        self.norm = {2: nn.InstanceNorm2d, 3: nn.InstanceNorm3d}[dim]
        self.drop = {2: nn.Dropout2d, 3: nn.Dropout3d}[dim]


class ConvDropoutNormReLU(nn.Module):
    """conv(k, stride, pad=(k-1)//2, bias) -> dropout -> instnorm -> nonlin, any of the last
    three optional.  Registers the conv as `.conv` AND as `.all_modules.0`
    (builders/simple_conv_blocks.py:43-52,69)."""

    def __init__(self, ops: _Ops, cin, cout, kernel, stride, bias, norm_kw, drop_kw, nonlin, nonlin_kw,
                 use_norm=True, nonlin_first=False):
        super().__init__()
        kernel = _as_list(kernel, ops.dim)
        stride = _as_list(stride, ops.dim)
        self.stride = stride
        seq = []
        self.conv = ops.conv(cin, cout, kernel, stride, padding=[(k - 1) // 2 for k in kernel],
                             dilation=1, bias=bias)
        seq.append(self.conv)
        if drop_kw is not None:
            self.dropout = ops.drop(**drop_kw)
            seq.append(self.dropout)
        if use_norm:
            self.norm = ops.norm(cout, **norm_kw)
            seq.append(self.norm)
        if nonlin is not None:
            self.nonlin = nonlin(**nonlin_kw)
            seq.append(self.nonlin)
        if nonlin_first and use_norm and nonlin is not None:
            seq[-1], seq[-2] = seq[-2], seq[-1]
        self.all_modules = nn.Sequential(*seq)

    def forward(self, x):
        return self.all_modules(x)


class StackedConvBlocks(nn.Module):
    """builders/simple_conv_blocks.py:82-148"""

    def __init__(self, n, ops, cin, cout, kernel, stride, bias, norm_kw, drop_kw, nonlin, nonlin_kw,
                 nonlin_first=False):
        super().__init__()
        couts = list(cout) if isinstance(cout, (list, tuple)) else [cout] * n
        mods = []
        for i in range(n):
            mods.append(ConvDropoutNormReLU(ops, cin if i == 0 else couts[i - 1], couts[i], kernel,
                                            stride if i == 0 else 1, bias, norm_kw, drop_kw, nonlin,
                                            nonlin_kw, nonlin_first=nonlin_first))
        self.convs = nn.Sequential(*mods)
        self.output_channels = couts[-1]

    def forward(self, x):
        return self.convs(x)


def make_divisible(v, divisor=8, min_value=None, round_limit=0.9):
    """timm.layers.helpers.make_divisible, as used by DNA's SqueezeExcite (round_limit=0. there)."""
    min_value = min_value or divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < round_limit * v:
        new_v += divisor
    return new_v


class SqueezeExcite(nn.Module):
    """PARITY UNPINNED.  dynamic_network_architectures.building_blocks.regularization.SqueezeExcite as
    constructed at builders/resblocks.py:84-87 (`SqueezeExcite(C, conv_op, rd_ratio=1/16, rd_divisor=8)`):
    fc1 / fc2 are 1x1 convs WITH bias, ReLU between, sigmoid gate, and -- 2-D heritage kept verbatim by
    the published source -- the squeeze is `x.mean((2, 3), keepdim=True)`: a 5-D activation is pooled over
    (z, y) only and the gate varies along x."""

    def __init__(self, channels, ops, rd_ratio=1. / 16, rd_divisor=8):
        super().__init__()
        rd = make_divisible(channels * rd_ratio, rd_divisor, round_limit=0.)
        self.fc1 = ops.conv(channels, rd, kernel_size=1, bias=True)
        self.fc2 = ops.conv(rd, channels, kernel_size=1, bias=True)

    def forward(self, x):
        x_se = x.mean((2, 3), keepdim=True)
        x_se = torch.relu(self.fc1(x_se))
        return x * torch.sigmoid(self.fc2(x_se))


class DropPath(nn.Module):
    """PARITY UNPINNED.  DNA / timm DropPath (builders/resblocks.py:79-81): in training a per-sample
    bernoulli(keep_prob) mask divided by keep_prob multiplies the residual branch; identity in eval.
    `forced_scale` (tests only) replaces the random draw by given per-sample factors."""

    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = drop_prob
        self.forced_scale = None

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        shape = (x.shape[0],) + (1,) * (x.ndim - 1)
        if self.forced_scale is not None:
            return x * self.forced_scale.to(x.dtype).reshape(shape)
        r = x.new_empty(shape).bernoulli_(keep)
        if keep > 0.0:
            r.div_(keep)
        return x * r


class BasicBlockD(nn.Module):
    """ResNet-D basic block (builders/resblocks.py:15-114).
    out = nonlin( IN(conv2( nonlin(IN(drop(conv1 x))) )) + skip(x) );
    skip = identity | AvgPool(s,s) | 1x1 conv->IN | AvgPool -> 1x1 conv -> IN."""

    def __init__(self, ops, cin, cout, kernel, stride, bias, norm_kw, drop_kw, nonlin, nonlin_kw,
                 stochastic_depth_p=0.0, squeeze_excitation=False, rd_ratio=1. / 16):
        super().__init__()
        stride = _as_list(stride, ops.dim)
        self.conv1 = ConvDropoutNormReLU(ops, cin, cout, kernel, stride, bias, norm_kw, drop_kw,
                                         nonlin, nonlin_kw)
        self.conv2 = ConvDropoutNormReLU(ops, cout, cout, kernel, 1, bias, norm_kw, None, None, None)
        self.nonlin2 = nonlin(**nonlin_kw)
        self.apply_stochastic_depth = stochastic_depth_p != 0.0        # resblocks.py:78-81
        if self.apply_stochastic_depth:
            self.drop_path = DropPath(stochastic_depth_p)
        self.apply_se = bool(squeeze_excitation)                       # resblocks.py:83-87
        if self.apply_se:
            self.squeeze_excitation = SqueezeExcite(cout, ops, rd_ratio=rd_ratio, rd_divisor=8)
        has_stride = any(s != 1 for s in stride)
        if has_stride or cin != cout:
            seq = []
            if has_stride:
                seq.append(ops.pool(stride, stride))
            if cin != cout:
                seq.append(ConvDropoutNormReLU(ops, cin, cout, 1, 1, False, norm_kw, None, None, None))
            self.skip = nn.Sequential(*seq)
        else:
            self.skip = lambda t: t

    def forward(self, x):
        res = self.skip(x)
        out = self.conv2(self.conv1(x))
        if self.apply_stochastic_depth:
            out = self.drop_path(out)
        if self.apply_se:
            out = self.squeeze_excitation(out)
        return self.nonlin2(out + res)


class BottleneckD(nn.Module):
    """builders/resblocks.py:135-239: 1x1 -> kxk(stride) -> 1x1, ResNet-D skip."""

    def __init__(self, ops, cin, cmid, cout, kernel, stride, bias, norm_kw, drop_kw, nonlin, nonlin_kw,
                 stochastic_depth_p=0.0, squeeze_excitation=False, rd_ratio=1. / 16):
        super().__init__()
        stride = _as_list(stride, ops.dim)
        self.conv1 = ConvDropoutNormReLU(ops, cin, cmid, 1, 1, bias, norm_kw, None, nonlin, nonlin_kw)
        self.conv2 = ConvDropoutNormReLU(ops, cmid, cmid, kernel, stride, bias, norm_kw, drop_kw,
                                         nonlin, nonlin_kw)
        self.conv3 = ConvDropoutNormReLU(ops, cmid, cout, 1, 1, bias, norm_kw, None, None, None)
        self.nonlin3 = nonlin(**nonlin_kw)
        self.apply_stochastic_depth = stochastic_depth_p != 0.0        # resblocks.py:203-206
        if self.apply_stochastic_depth:
            self.drop_path = DropPath(stochastic_depth_p)
        self.apply_se = bool(squeeze_excitation)                       # resblocks.py:208-212
        if self.apply_se:
            self.squeeze_excitation = SqueezeExcite(cout, ops, rd_ratio=rd_ratio, rd_divisor=8)
        has_stride = any(s != 1 for s in stride)
        if has_stride or cin != cout:
            seq = []
            if has_stride:
                seq.append(ops.pool(stride, stride))
            if cin != cout:
                seq.append(ConvDropoutNormReLU(ops, cin, cout, 1, 1, False, norm_kw, None, None, None))
            self.skip = nn.Sequential(*seq)
        else:
            self.skip = lambda t: t

    def forward(self, x):
        res = self.skip(x)
        out = self.conv3(self.conv2(self.conv1(x)))
        if self.apply_stochastic_depth:
            out = self.drop_path(out)
        if self.apply_se:
            out = self.squeeze_excitation(out)
        return self.nonlin3(out + res)


class StackedResidualBlocks(nn.Module):
    """builders/resblocks.py:262-343: n blocks, only the first strided / channel-changing."""

    def __init__(self, n, ops, cin, cout, kernel, stride, bias, norm_kw, drop_kw, nonlin, nonlin_kw,
                 bottleneck=False, bottleneck_channels=None, stochastic_depth_p=0.0,
                 squeeze_excitation=False):
        super().__init__()
        mods = []
        for i in range(n):
            ci, st = (cin, stride) if i == 0 else (cout, 1)
            if bottleneck:
                mods.append(BottleneckD(ops, ci, bottleneck_channels, cout, kernel, st, bias, norm_kw,
                                        drop_kw, nonlin, nonlin_kw, stochastic_depth_p, squeeze_excitation))
            else:
                mods.append(BasicBlockD(ops, ci, cout, kernel, st, bias, norm_kw, drop_kw, nonlin,
                                        nonlin_kw, stochastic_depth_p, squeeze_excitation))
        self.blocks = nn.Sequential(*mods)
        self.output_channels = cout

    def forward(self, x):
        return self.blocks(x)


class Encoder(nn.Module):
    """builders/encoder.py:27-158.  `basic_block`/`bottleneck_block` strings are compared with
    `==` (the reference's `is` comparison only works for interned literals; documented fix)."""

    def __init__(self, ops, cin, basic_block, n_stages, features, n_blocks, bottleneck_block,
                 kernel_sizes, strides, bias, norm_kw, drop_kw, nonlin, nonlin_kw, return_skips, do_stem,
                 stem_channels, bottleneck_channels, stochastic_depth_p, squeeze_excitation):
        super().__init__()
        kernel_sizes = _as_list(kernel_sizes, n_stages) if isinstance(kernel_sizes, int) else list(kernel_sizes)
        features = _as_list(features, n_stages) if isinstance(features, int) else list(features)
        n_blocks = _as_list(n_blocks, n_stages) if isinstance(n_blocks, int) else list(n_blocks)
        strides = _as_list(strides, n_stages) if isinstance(strides, int) else list(strides)
        if bottleneck_channels is None or isinstance(bottleneck_channels, int):
            bottleneck_channels = [bottleneck_channels] * n_stages
        residual = basic_block in ("BasicBlockD", "BottleneckBlockD")
        use_bottleneck = (bottleneck_block == "BottleneckBlockD") and basic_block != "BasicBlockD"
        if basic_block == "BottleneckBlockD" and not use_bottleneck:
            # encoder.py:74-79 leaves `block` unbound for this combination
            raise UnboundLocalError("basic_block='BottleneckBlockD' needs bottleneck_block='BottleneckBlockD'")
        if do_stem:
            stem_channels = features[0] if stem_channels is None else stem_channels
            self.stem = StackedConvBlocks(1, ops, cin, stem_channels, kernel_sizes[0], 1, bias, norm_kw,
                                          drop_kw, nonlin, nonlin_kw)
            cin = stem_channels
        else:
            self.stem = None
        stages = []
        for s in range(n_stages):
            if residual:
                stages.append(StackedResidualBlocks(
                    n_blocks[s], ops, cin, features[s], kernel_sizes[s], strides[s], bias, norm_kw, drop_kw,
                    nonlin, nonlin_kw, bottleneck=use_bottleneck, bottleneck_channels=bottleneck_channels[s],
                    stochastic_depth_p=stochastic_depth_p, squeeze_excitation=squeeze_excitation))
            else:
                stages.append(nn.Sequential(StackedConvBlocks(
                    n_blocks[s], ops, cin, features[s], kernel_sizes[s], strides[s], bias, norm_kw, drop_kw,
                    nonlin, nonlin_kw)))
            cin = features[s]
        self.stages = nn.Sequential(*stages)
        self.output_channels = features
        self.strides = [_as_list(s, ops.dim) for s in strides]
        self.kernel_sizes = kernel_sizes
        self.return_skips = return_skips
        self.conv_bias = bias

    def forward(self, x):
        if self.stem is not None:
            x = self.stem(x)
        skips = []
        for st in self.stages:
            x = st(x)
            skips.append(x)
        return skips if self.return_skips else skips[-1]


class Decoder(nn.Module):
    """builders/decoder.py:16-162.  Per stage: ConvTranspose(k=s=stride) -> cat((up, skip), 1)
    -> conv stage; all n_stages-1 1x1 heads are built, only the last is used."""

    def __init__(self, ops, encoder: Encoder, basic_block, num_classes, n_conv_per_stage, norm_kw, drop_kw,
                 nonlin, nonlin_kw):
        super().__init__()
        self.encoder = encoder  # registered on purpose: aliases the encoder's keys (decoder.py:49)
        n_enc = len(encoder.output_channels)
        n_conv = _as_list(n_conv_per_stage, n_enc - 1) if isinstance(n_conv_per_stage, int) else list(n_conv_per_stage)
        assert len(n_conv) == n_enc - 1
        bias = encoder.conv_bias
        stages, ups, heads = [], [], []
        for s in range(1, n_enc):
            c_below = encoder.output_channels[-s]
            c_skip = encoder.output_channels[-(s + 1)]
            st = encoder.strides[-s]
            ups.append(ops.convT(c_below, c_skip, st, st, bias=bias))
            if basic_block == "ResidualBlock":
                stages.append(StackedResidualBlocks(n_conv[s - 1], ops, 2 * c_skip, c_skip,
                                                    encoder.kernel_sizes[-(s + 1)], 1, bias, norm_kw, drop_kw,
                                                    nonlin, nonlin_kw))
            elif basic_block == "ConvBlock":
                stages.append(StackedConvBlocks(n_conv[s - 1], ops, 2 * c_skip, c_skip,
                                                encoder.kernel_sizes[-(s + 1)], 1, bias, norm_kw, drop_kw,
                                                nonlin, nonlin_kw))
            else:
                raise UnboundLocalError("basic_decoder_block must be 'ConvBlock' or 'ResidualBlock'")
            heads.append(ops.conv(c_skip, num_classes, 1, 1, 0, bias=True))
        self.stages = nn.ModuleList(stages)
        self.transpconvs = nn.ModuleList(ups)
        self.seg_layers = nn.ModuleList(heads)

    def forward(self, skips):
        low = skips[-1]
        for s in range(len(self.stages)):
            x = self.transpconvs[s](low)
            x = torch.cat((x, skips[-(s + 2)]), 1)
            x = self.stages[s](x)
            low = x
        return self.seg_layers[-1](low)


def _activation(name: str):
    """build_network_from_config.py:6-18"""
    name = name.lower()
    if name == "none":
        return None
    if name == "sigmoid":
        return nn.Sigmoid()
    if name == "softmax":
        return nn.Softmax(dim=1)
    raise ValueError(f"Unknown activation type: {name}")


_MANUAL_KEYS = ("basic_encoder_block", "basic_decoder_block", "bottleneck_block", "features_per_stage",
                "num_stages", "n_blocks_per_stage", "kernel_sizes", "n_conv_per_stage_decoder", "strides")


class NetworkFromConfig(nn.Module):
    """build_network_from_config.py:20-326 (prints omitted)."""

    def __init__(self, mgr):
        super().__init__()
        self.tasks = mgr.tasks
        self.patch_size = tuple(mgr.train_patch_size)
        mc = mgr.model_config
        if mgr.autoconfigure:
            self.basic_encoder_block, self.basic_decoder_block, self.bottleneck_block = \
                "BasicBlockD", "ConvBlock", "BasicBlockD"
            _, strides, kernels = plan_pooling(self.patch_size, 4, (1.0,) * len(self.patch_size), 999999)
            self.num_stages = len(strides)
            self.features_per_stage = [min(32 * 2 ** i, 512) for i in range(self.num_stages)]
            self.n_blocks_per_stage = blocks_per_stage(self.num_stages)
            self.n_conv_per_stage_decoder = [1] * (self.num_stages - 1)
            self.strides, self.kernel_sizes = strides, kernels
        else:
            for k in _MANUAL_KEYS:
                if k not in mc:
                    raise ValueError(f"autoconfigure=False, but '{k}' was not provided in the config!")
            self.basic_encoder_block = mc["basic_encoder_block"]
            self.basic_decoder_block = mc["basic_decoder_block"]
            self.bottleneck_block = mc["bottleneck_block"]
            self.features_per_stage = mc["features_per_stage"]
            self.num_stages = mc["num_stages"]
            self.n_blocks_per_stage = mc["n_blocks_per_stage"]
            self.kernel_sizes = mc["kernel_sizes"]
            self.n_conv_per_stage_decoder = mc["n_conv_per_stage_decoder"]
            self.strides = mc["strides"]
        if len(self.patch_size) not in (2, 3):
            raise ValueError("Patch size must have either 2 or 3 dimensions!")
        ops = _Ops(len(self.patch_size))
        self.op_dims = ops.dim
        norm_kw = mc.get("norm_op_kwargs", {"affine": False, "eps": 1e-5})
        drop_kw = mc.get("dropout_op_kwargs", {"p": 0.0})
        nonlin_name = mc.get("nonlin", "nn.LeakyReLU")
        if nonlin_name == "nn.LeakyReLU":
            nonlin, nonlin_kw = nn.LeakyReLU, {"negative_slope": 1e-2, "inplace": True}
        elif nonlin_name == "nn.ReLU":
            nonlin, nonlin_kw = nn.ReLU, {"inplace": True}
        else:
            raise TypeError(f"unsupported nonlin {nonlin_name!r}")
        bneck = mc.get("bottleneck_channels", None)
        if self.bottleneck_block == "BottleneckBlockD":
            if bneck is None:
                bneck = [f // 4 for f in self.features_per_stage]
            elif isinstance(bneck, int):
                bneck = [bneck] * len(self.features_per_stage)
        else:
            bneck = None
        self.shared_encoder = Encoder(
            ops, mgr.in_channels, self.basic_encoder_block, self.num_stages, self.features_per_stage,
            self.n_blocks_per_stage, self.bottleneck_block, self.kernel_sizes, self.strides,
            mc.get("conv_bias", False), norm_kw, drop_kw, nonlin, nonlin_kw, mc.get("return_skips", True),
            mc.get("do_stem", True), self.features_per_stage[0], bneck, mc.get("stochastic_depth_p", 0.0),
            mc.get("squeeze_excitation", False))
        self.task_decoders = nn.ModuleDict()
        self.task_activations = nn.ModuleDict()
        for name, info in self.tasks.items():
            self.task_decoders[name] = Decoder(ops, self.shared_encoder, self.basic_decoder_block,
                                               info["channels"], self.n_conv_per_stage_decoder, norm_kw,
                                               drop_kw, nonlin, nonlin_kw)
            self.task_activations[name] = _activation(info.get("activation", "none"))

    def forward(self, x):
        skips = self.shared_encoder(x)
        out = {}
        for name, dec in self.task_decoders.items():
            y = dec(skips)
            act = self.task_activations[name]
            if act is not None and not self.training:
                y = act(y)
            out[name] = y
        return out


# --------------------------------------------------------------------------------------
# losses used by the BASELINE train step (training/losses/losses.py)
# --------------------------------------------------------------------------------------
def bce_dice_loss(logits, target, alpha=0.5, beta=0.5, smoothing=0.1, eps=1e-6):
    """BCEDiceLoss (losses.py:307-318) = alpha * BCE-with-logits on label-smoothed targets
    (:217-238, y*(1-2s)+s, mean) + beta * (1 - mean_c Dice_c), Dice_c on sigmoid probabilities with
    the V-Net denominator sum(p^2)+sum(t^2) over (N, spatial) (:17-43,128-138,321-333)."""
    t_s = target * (1.0 - 2.0 * smoothing) + smoothing
    bce = F.binary_cross_entropy_with_logits(logits, t_s)
    p = torch.sigmoid(logits)
    c = logits.shape[1]
    pf = p.transpose(0, 1).reshape(c, -1)
    tf = target.float().transpose(0, 1).reshape(c, -1)
    inter = (pf * tf).sum(-1)
    den = (pf * pf).sum(-1) + (tf * tf).sum(-1)
    dice = 2 * inter / den.clamp(min=eps)
    return alpha * bce + beta * (1.0 - dice.mean())


def masked_cosine_loss(pred, target):
    """MaskedCosineLoss (losses.py:187-215)."""
    mask = (torch.norm(target, dim=1) > 1e-6).float()
    pred_unit = pred / torch.norm(pred, dim=1, keepdim=True).clamp(min=1e-8)
    cos = F.cosine_similarity(pred_unit, target, dim=1, eps=1e-8)
    return 1.0 - (cos * mask).sum() / (mask.sum() + 1e-8)


LOSSES = {"BCEDiceLoss": bce_dice_loss, "MaskedCosineLoss": masked_cosine_loss}


def make_mgr(patch_size, tasks, in_channels=1, batch_size=2, autoconfigure=True, model_config=None):
    return SimpleNamespace(tasks=tasks, train_patch_size=tuple(patch_size), train_batch_size=batch_size,
                           in_channels=in_channels, vram_max=16.0, autoconfigure=autoconfigure,
                           model_config=dict(model_config or {}))


def synthetic_batch(batch, in_channels, patch, tasks: Dict[str, dict], seed=1234):
    """SURVEY 8(d) synthetic inputs: image rand in [0,1]; seg target (rand > 0.8); normals target
    unit vectors zeroed where the seg target is 0."""
    g = torch.Generator().manual_seed(seed)
    x = torch.rand((batch, in_channels, *patch), generator=g)
    seg = (torch.rand((batch, 1, *patch), generator=g) > 0.8).float()
    targets = {}
    for name, info in tasks.items():
        c = info["channels"]
        if info.get("loss_fn", "BCEDiceLoss") == "MaskedCosineLoss":
            v = torch.randn((batch, c, *patch), generator=g)
            v = v / v.norm(dim=1, keepdim=True).clamp(min=1e-8)
            targets[name] = v * seg
        else:
            targets[name] = seg.expand(batch, c, *patch).contiguous() if c > 1 else seg
    return x, targets


def train_loss(outputs, targets, tasks):
    """train.py:206-218: sum over tasks of loss_fn(pred, gt) * weight."""
    total = 0.0
    for name, gt in targets.items():
        info = tasks[name]
        fn = LOSSES[info.get("loss_fn", "BCEDiceLoss")]
        kw = info.get("loss_kwargs", {"alpha": 0.5, "beta": 0.5} if fn is bce_dice_loss else {})
        total = total + fn(outputs[name], gt, **kw) * info.get("weight", 1.0)
    return total
