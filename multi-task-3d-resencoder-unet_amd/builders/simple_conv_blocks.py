"""Parameter containers with the reference's constructor signatures and `state_dict` key names
(reference: builders/simple_conv_blocks.py:13-79 ConvDropoutNormReLU, :82-148 StackedConvBlocks).

These modules own the torch parameters (same init order -> same seeded init as the reference) and
describe the op sequence to the HIP engine (`engine/plan.py`).  They do NOT compute: the product has
no PyTorch fallback, calling them directly raises."""
import numpy as np
from torch import nn

from .utils import maybe_convert_scalar_to_list


class EngineOnly(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError(
            f"{type(self).__name__} is a parameter container of the HIP engine: run the whole "
            "NetworkFromConfig (gfx950 kernels); there is no PyTorch fallback path")


class ConvDropoutNormReLU(EngineOnly):
    def __init__(self, conv_op, input_channels, output_channels, kernel_size, stride, conv_bias=False,
                 norm_op=None, norm_op_kwargs=None, dropout_op=None, dropout_op_kwargs=None, nonlin=None,
                 nonlin_kwargs=None, nonlin_first=False):
        super().__init__()
        self.input_channels, self.output_channels = input_channels, output_channels
        self.stride = maybe_convert_scalar_to_list(conv_op, stride)
        kernel_size = maybe_convert_scalar_to_list(conv_op, kernel_size)
        self.kernel_size = list(kernel_size)
        seq = []
        self.conv = conv_op(input_channels, output_channels, kernel_size, self.stride,
                            padding=[(k - 1) // 2 for k in kernel_size], dilation=1, bias=conv_bias)
        seq.append(self.conv)
        if dropout_op is not None:
            self.dropout = dropout_op(**(dropout_op_kwargs or {}))
            seq.append(self.dropout)
        if norm_op is not None:
            self.norm = norm_op(output_channels, **(norm_op_kwargs or {}))
            seq.append(self.norm)
        if nonlin is not None:
            self.nonlin = nonlin(**(nonlin_kwargs or {}))
            seq.append(self.nonlin)
        self.nonlin_first = bool(nonlin_first and norm_op is not None and nonlin is not None)
        if self.nonlin_first:
            seq[-1], seq[-2] = seq[-2], seq[-1]
        # the conv is registered twice on purpose (`conv.*` and `all_modules.0.*` keys)
        self.all_modules = nn.Sequential(*seq)

    # what the engine needs to know
    def spec(self):
        drop = getattr(self, "dropout", None)
        norm = getattr(self, "norm", None)
        act = getattr(self, "nonlin", None)
        return dict(conv=self.conv, kernel=list(self.kernel_size), stride=list(self.stride),
                    dropout_p=float(drop.p) if drop is not None else 0.0, norm=norm, nonlin=act,
                    nonlin_first=self.nonlin_first)

    def compute_conv_feature_map_size(self, input_size):
        assert len(input_size) == len(self.stride)
        return np.prod([self.output_channels, *[i // j for i, j in zip(input_size, self.stride)]], dtype=np.int64)


class StackedConvBlocks(EngineOnly):
    def __init__(self, num_convs, conv_op, input_channels, output_channels, kernel_size, initial_stride,
                 conv_bias=False, norm_op=None, norm_op_kwargs=None, dropout_op=None, dropout_op_kwargs=None,
                 nonlin=None, nonlin_kwargs=None, nonlin_first=False):
        super().__init__()
        if not isinstance(output_channels, (tuple, list)):
            output_channels = [output_channels] * num_convs
        blocks = []
        for i in range(num_convs):
            blocks.append(ConvDropoutNormReLU(
                conv_op, input_channels if i == 0 else output_channels[i - 1], output_channels[i], kernel_size,
                initial_stride if i == 0 else 1, conv_bias, norm_op, norm_op_kwargs, dropout_op,
                dropout_op_kwargs, nonlin, nonlin_kwargs, nonlin_first))
        self.convs = nn.Sequential(*blocks)
        self.output_channels = output_channels[-1]
        self.initial_stride = maybe_convert_scalar_to_list(conv_op, initial_stride)

    def compute_conv_feature_map_size(self, input_size):
        out = self.convs[0].compute_conv_feature_map_size(input_size)
        after = [i // j for i, j in zip(input_size, self.initial_stride)]
        for b in self.convs[1:]:
            out += b.compute_conv_feature_map_size(after)
        return out
