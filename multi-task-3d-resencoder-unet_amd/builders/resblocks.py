"""Residual block containers (reference: builders/resblocks.py:15-133 BasicBlockD, :135-259
BottleneckD, :262-353 StackedResidualBlocks).  Unlike the reference this file does not import the
third-party `dynamic_network_architectures`: the three helpers it took from there have twins in
this package.  `SqueezeExcite` / `DropPath` exist only in that package (absent from the reference tree and from
this image): the containers below follow its published source (timm's squeeze_excite.py / drop.py with a
`conv_op` argument) and the reference's call sites (resblocks.py:79-87, 109-112) -- PARITY UNPINNED, see
DESIGN.md; their arithmetic runs in csrc/rx_se.hip."""
import numpy as np
from torch import nn

from .simple_conv_blocks import ConvDropoutNormReLU, EngineOnly
from .utils import get_matching_pool_op, maybe_convert_scalar_to_list


def make_divisible(v, divisor=8, min_value=None, round_limit=0.9):
    """timm's helper behind SqueezeExcite's reduction width (called with round_limit=0.)."""
    min_value = min_value or divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < round_limit * v:
        new_v += divisor
    return new_v


class SqueezeExcite(EngineOnly):
    """Parameter container of the channel gate `x * sigmoid(fc2(relu(fc1(x.mean((2, 3), keepdim=True)))))`
    (1x1 convs with bias; state_dict keys `fc1.*`, `fc2.*`).  PARITY UNPINNED."""

    def __init__(self, channels, conv_op, rd_ratio=1. / 16, rd_channels=None, rd_divisor=8):
        super().__init__()
        if not rd_channels:
            rd_channels = make_divisible(channels * rd_ratio, rd_divisor, round_limit=0.)
        self.channels, self.rd_channels = channels, rd_channels
        self.fc1 = conv_op(channels, rd_channels, kernel_size=1, bias=True)
        self.fc2 = conv_op(rd_channels, channels, kernel_size=1, bias=True)


class DropPath(EngineOnly):
    """Stochastic depth on the residual branch: in training the branch of sample n is multiplied by
    bernoulli(1 - drop_prob) / (1 - drop_prob); identity in eval.  No parameters.  PARITY UNPINNED."""

    def __init__(self, drop_prob=0.0, scale_by_keep=True):
        super().__init__()
        self.drop_prob = float(drop_prob)
        self.scale_by_keep = scale_by_keep


def _make_skip(conv_op, cin, cout, stride, norm_op, norm_op_kwargs):
    """ResNet-D skip: AvgPool(stride) when strided, then 1x1 conv -> norm when channels change."""
    has_stride = any(s != 1 for s in stride)
    if not has_stride and cin == cout:
        return None
    ops = []
    if has_stride:
        ops.append(get_matching_pool_op(conv_op=conv_op, adaptive=False, pool_type="avg")(stride, stride))
    if cin != cout:
        ops.append(ConvDropoutNormReLU(conv_op, cin, cout, 1, 1, False, norm_op, norm_op_kwargs, None, None, None,
                                       None))
    return nn.Sequential(*ops)


class _ResidualBase(EngineOnly):
    def _regularizers(self, conv_op, cout, stochastic_depth_p, squeeze_excitation, rd_ratio):
        """same attribute names and creation order as resblocks.py:78-87 (seeded init draws fc1, fc2 before the skip)"""
        self.apply_stochastic_depth = stochastic_depth_p != 0.0
        if self.apply_stochastic_depth:
            self.drop_path = DropPath(drop_prob=stochastic_depth_p)
        self.apply_se = bool(squeeze_excitation)
        if self.apply_se:
            self.squeeze_excitation = SqueezeExcite(cout, conv_op, rd_ratio=rd_ratio, rd_divisor=8)

    def _finish(self, conv_op, cin, cout, stride, norm_op, norm_op_kwargs):
        skip = _make_skip(conv_op, cin, cout, stride, norm_op, norm_op_kwargs)
        if skip is None:
            self.skip = lambda x: x  # plain attribute like the reference: contributes no state_dict keys
            self.skip_ops = []
        else:
            self.skip = skip
            self.skip_ops = list(skip)


class BasicBlockD(_ResidualBase):
    def __init__(self, conv_op, input_channels, output_channels, kernel_size, stride, conv_bias=False, norm_op=None,
                 norm_op_kwargs=None, dropout_op=None, dropout_op_kwargs=None, nonlin=None, nonlin_kwargs=None,
                 stochastic_depth_p=0.0, squeeze_excitation=False, squeeze_excitation_reduction_ratio=1. / 16):
        super().__init__()
        self.input_channels, self.output_channels = input_channels, output_channels
        self.stride = maybe_convert_scalar_to_list(conv_op, stride)
        kernel_size = maybe_convert_scalar_to_list(conv_op, kernel_size)
        norm_op_kwargs = norm_op_kwargs or {}
        nonlin_kwargs = nonlin_kwargs or {}
        self.conv1 = ConvDropoutNormReLU(conv_op, input_channels, output_channels, kernel_size, self.stride, conv_bias,
                                         norm_op, norm_op_kwargs, dropout_op, dropout_op_kwargs, nonlin, nonlin_kwargs)
        self.conv2 = ConvDropoutNormReLU(conv_op, output_channels, output_channels, kernel_size, 1, conv_bias, norm_op,
                                         norm_op_kwargs, None, None, None, None)
        self.nonlin2 = nonlin(**nonlin_kwargs) if nonlin is not None else None
        self._regularizers(conv_op, output_channels, stochastic_depth_p, squeeze_excitation,
                           squeeze_excitation_reduction_ratio)
        self._finish(conv_op, input_channels, output_channels, self.stride, norm_op, norm_op_kwargs)

    def main_path(self):
        return [self.conv1, self.conv2]

    def final_nonlin(self):
        return self.nonlin2

    def compute_conv_feature_map_size(self, input_size):
        after = [i // j for i, j in zip(input_size, self.stride)]
        one = np.prod([self.output_channels, *after], dtype=np.int64)
        return 2 * one + (one if self.skip_ops else 0)


class BottleneckD(_ResidualBase):
    def __init__(self, conv_op, input_channels, bottleneck_channels, output_channels, kernel_size, stride,
                 conv_bias=False, norm_op=None, norm_op_kwargs=None, dropout_op=None, dropout_op_kwargs=None,
                 nonlin=None, nonlin_kwargs=None, stochastic_depth_p=0.0, squeeze_excitation=False,
                 squeeze_excitation_reduction_ratio=1. / 16):
        super().__init__()
        self.input_channels, self.output_channels = input_channels, output_channels
        self.bottleneck_channels = bottleneck_channels
        self.stride = maybe_convert_scalar_to_list(conv_op, stride)
        kernel_size = maybe_convert_scalar_to_list(conv_op, kernel_size)
        norm_op_kwargs = norm_op_kwargs or {}
        nonlin_kwargs = nonlin_kwargs or {}
        self.conv1 = ConvDropoutNormReLU(conv_op, input_channels, bottleneck_channels, 1, 1, conv_bias, norm_op,
                                         norm_op_kwargs, None, None, nonlin, nonlin_kwargs)
        self.conv2 = ConvDropoutNormReLU(conv_op, bottleneck_channels, bottleneck_channels, kernel_size, self.stride,
                                         conv_bias, norm_op, norm_op_kwargs, dropout_op, dropout_op_kwargs, nonlin,
                                         nonlin_kwargs)
        self.conv3 = ConvDropoutNormReLU(conv_op, bottleneck_channels, output_channels, 1, 1, conv_bias, norm_op,
                                         norm_op_kwargs, None, None, None, None)
        self.nonlin3 = nonlin(**nonlin_kwargs) if nonlin is not None else None
        self._regularizers(conv_op, output_channels, stochastic_depth_p, squeeze_excitation,
                           squeeze_excitation_reduction_ratio)
        self._finish(conv_op, input_channels, output_channels, self.stride, norm_op, norm_op_kwargs)

    def main_path(self):
        return [self.conv1, self.conv2, self.conv3]

    def final_nonlin(self):
        return self.nonlin3

    def compute_conv_feature_map_size(self, input_size):
        after = [i // j for i, j in zip(input_size, self.stride)]
        out = np.prod([self.bottleneck_channels, *input_size], dtype=np.int64)
        out += np.prod([self.bottleneck_channels, *after], dtype=np.int64)
        one = np.prod([self.output_channels, *after], dtype=np.int64)
        return out + one + (one if self.skip_ops else 0)


class StackedResidualBlocks(EngineOnly):
    def __init__(self, n_blocks, conv_op, input_channels, output_channels, kernel_size, initial_stride,
                 conv_bias=False, norm_op=None, norm_op_kwargs=None, dropout_op=None, dropout_op_kwargs=None,
                 nonlin=None, nonlin_kwargs=None, block=BasicBlockD, bottleneck_channels=None,
                 stochastic_depth_p=0.0, squeeze_excitation=False, squeeze_excitation_reduction_ratio=1. / 16):
        super().__init__()
        assert n_blocks > 0, "n_blocks must be > 0"
        assert block in (BasicBlockD, BottleneckD), "block must be BasicBlockD or BottleneckD"
        if not isinstance(output_channels, (tuple, list)):
            output_channels = [output_channels] * n_blocks
        if not isinstance(bottleneck_channels, (tuple, list)):
            bottleneck_channels = [bottleneck_channels] * n_blocks
        mods = []
        for n in range(n_blocks):
            cin = input_channels if n == 0 else output_channels[n - 1]
            st = initial_stride if n == 0 else 1
            common = (conv_bias, norm_op, norm_op_kwargs, dropout_op, dropout_op_kwargs, nonlin, nonlin_kwargs,
                      stochastic_depth_p, squeeze_excitation, squeeze_excitation_reduction_ratio)
            if block is BasicBlockD:
                mods.append(BasicBlockD(conv_op, cin, output_channels[n], kernel_size, st, *common))
            else:
                mods.append(BottleneckD(conv_op, cin, bottleneck_channels[n], output_channels[n], kernel_size, st,
                                        *common))
        self.blocks = nn.Sequential(*mods)
        self.initial_stride = maybe_convert_scalar_to_list(conv_op, initial_stride)
        self.output_channels = output_channels[-1]

    def compute_conv_feature_map_size(self, input_size):
        out = self.blocks[0].compute_conv_feature_map_size(input_size)
        after = [i // j for i, j in zip(input_size, self.initial_stride)]
        for b in self.blocks[1:]:
            out += b.compute_conv_feature_map_size(after)
        return out
