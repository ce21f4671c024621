"""Residual block containers (reference: builders/resblocks.py:15-133 BasicBlockD, :135-259
BottleneckD, :262-353 StackedResidualBlocks).  Unlike the reference this file does not import the
third-party `dynamic_network_architectures`: the three helpers it took from there have twins in
this package; SqueezeExcite / DropPath exist only in that package (parity unpinned) and
are rejected here."""
import numpy as np
from torch import nn

from .simple_conv_blocks import ConvDropoutNormReLU, EngineOnly
from .utils import get_matching_pool_op, maybe_convert_scalar_to_list


def _reject_unpinned(stochastic_depth_p, squeeze_excitation):
    if stochastic_depth_p != 0.0:
        raise NotImplementedError("stochastic_depth_p > 0 (DropPath) is not available: its arithmetic lives in the "
                                  "un-vendored dynamic_network_architectures package (parity unpinned)")
    if squeeze_excitation:
        raise NotImplementedError("squeeze_excitation=True (SqueezeExcite) is not available: its arithmetic lives in "
                                  "the un-vendored dynamic_network_architectures package (parity unpinned)")


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
        _reject_unpinned(stochastic_depth_p, squeeze_excitation)
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
        self.apply_stochastic_depth = False
        self.apply_se = False
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
        _reject_unpinned(stochastic_depth_p, squeeze_excitation)
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
        self.apply_stochastic_depth = False
        self.apply_se = False
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
