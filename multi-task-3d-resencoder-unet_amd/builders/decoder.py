"""Per-task decoder container (reference: builders/decoder.py:16-193).

Per stage: ConvTranspose(kernel = stride) of the stage below -> concat (upsampled first, skip
second) -> conv stage -> 1x1x1 head.  All n_stages-1 heads are built so checkpoints load
(decoder.py:100,131), only the last is ever used (:151-152) -- the unused ones therefore never get a
gradient.  The shared encoder is registered as a sub-module exactly like the reference
(decoder.py:49), which is why `state_dict` carries `task_decoders.<task>.encoder.*` aliases."""
import numpy as np
from torch import nn

from .resblocks import StackedResidualBlocks
from .simple_conv_blocks import EngineOnly, StackedConvBlocks
from .utils import get_matching_convtransp


class Decoder(EngineOnly):
    def __init__(self, encoder, basic_block, num_classes, n_conv_per_stage, deep_supervision, nonlin_first=False,
                 norm_op=None, norm_op_kwargs=None, dropout_op=None, dropout_op_kwargs=None, nonlin=None,
                 nonlin_kwargs=None, conv_bias=None):
        super().__init__()
        if deep_supervision:
            raise NotImplementedError("deep_supervision=True is never requested by NetworkFromConfig "
                                      "(build_network_from_config.py:274)")
        self.deep_supervision = deep_supervision
        self.encoder = encoder
        self.num_classes = num_classes
        n_enc = len(encoder.output_channels)
        if isinstance(n_conv_per_stage, int):
            n_conv_per_stage = [n_conv_per_stage] * (n_enc - 1)
        assert len(n_conv_per_stage) == n_enc - 1, \
            "n_conv_per_stage must have as many entries as we have resolution stages - 1 (n_stages in encoder - 1), " \
            "here: %d" % n_enc
        transpconv_op = get_matching_convtransp(conv_op=encoder.conv_op)
        conv_bias = encoder.conv_bias if conv_bias is None else conv_bias
        norm_op = encoder.norm_op if norm_op is None else norm_op
        norm_op_kwargs = encoder.norm_op_kwargs if norm_op_kwargs is None else norm_op_kwargs
        dropout_op = encoder.dropout_op if dropout_op is None else dropout_op
        dropout_op_kwargs = encoder.dropout_op_kwargs if dropout_op_kwargs is None else dropout_op_kwargs
        nonlin = encoder.nonlin if nonlin is None else nonlin
        nonlin_kwargs = encoder.nonlin_kwargs if nonlin_kwargs is None else nonlin_kwargs
        if basic_block not in ("ConvBlock", "ResidualBlock"):
            raise UnboundLocalError("basic_decoder_block must be 'ConvBlock' or 'ResidualBlock' "
                                    "(anything else leaves `stages` unbound in the reference, decoder.py:68-135)")
        stages, ups, heads = [], [], []
        for s in range(1, n_enc):
            below = encoder.output_channels[-s]
            skip = encoder.output_channels[-(s + 1)]
            stride = encoder.strides[-s]
            # the reference passes encoder.conv_bias on the ResidualBlock path and conv_bias on the ConvBlock path
            ups.append(transpconv_op(below, skip, stride, stride,
                                     bias=encoder.conv_bias if basic_block == "ResidualBlock" else conv_bias))
            if basic_block == "ResidualBlock":
                stages.append(StackedResidualBlocks(
                    n_blocks=n_conv_per_stage[s - 1], conv_op=encoder.conv_op, input_channels=2 * skip,
                    output_channels=skip, kernel_size=encoder.kernel_sizes[-(s + 1)], initial_stride=1,
                    conv_bias=conv_bias, norm_op=norm_op, norm_op_kwargs=norm_op_kwargs, dropout_op=dropout_op,
                    dropout_op_kwargs=dropout_op_kwargs, nonlin=nonlin, nonlin_kwargs=nonlin_kwargs))
            else:
                stages.append(StackedConvBlocks(
                    n_conv_per_stage[s - 1], encoder.conv_op, 2 * skip, skip, encoder.kernel_sizes[-(s + 1)], 1,
                    conv_bias, norm_op, norm_op_kwargs, dropout_op, dropout_op_kwargs, nonlin, nonlin_kwargs,
                    nonlin_first))
            heads.append(encoder.conv_op(skip, num_classes, 1, 1, 0, bias=True))
        self.stages = nn.ModuleList(stages)
        self.transpconvs = nn.ModuleList(ups)
        self.seg_layers = nn.ModuleList(heads)

    def forward(self, skips):
        """`model.task_decoders[task](skips)` as upstream (decoder.py:137-162): the task's raw logits from a list of encoder
        outputs (lowest resolution last), computed by the decoder part of the owning network's plan on the HIP kernels, WITHOUT
        autograd; a container that is not part of a NetworkFromConfig has nothing to run on."""
        owner = self._owner() if getattr(self, "_owner", None) is not None else None
        if owner is None or self._task not in owner.task_decoders or owner.task_decoders[self._task] is not self:
            return super().forward(skips)
        return owner.decode(self._task, skips)

    def compute_conv_feature_map_size(self, input_size):
        skip_sizes = []
        for s in range(len(self.encoder.strides) - 1):
            skip_sizes.append([i // j for i, j in zip(input_size, self.encoder.strides[s])])
            input_size = skip_sizes[-1]
        assert len(skip_sizes) == len(self.stages)
        out = np.int64(0)
        for s in range(len(self.stages)):
            out += self.stages[s].compute_conv_feature_map_size(skip_sizes[-(s + 1)])
            out += np.prod([self.encoder.output_channels[-(s + 2)], *skip_sizes[-(s + 1)]], dtype=np.int64)
            if s == len(self.stages) - 1:
                out += np.prod([self.num_classes, *skip_sizes[-(s + 1)]], dtype=np.int64)
        return out
