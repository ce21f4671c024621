"""TEST INFRASTRUCTURE ONLY.  The golden cases shared by `oracle/make_golden.py` (which runs the
REAL reference in the build container) and the tests that replay them against the oracle and the
HIP engine.  Weights are never stored: they are re-created with `torch.manual_seed(seed)` followed
by construction in the reference's order, and pinned by per-parameter checksums in the fixture."""

TASKS_2HEAD = {
    "sheet": {"channels": 1, "activation": "none", "weight": 1, "loss_fn": "BCEDiceLoss",
              "loss_kwargs": {"alpha": 0.5, "beta": 0.5}},
    "normals": {"channels": 3, "activation": "none", "weight": 1, "loss_fn": "MaskedCosineLoss"},
}
TASKS_SIGMOID = {
    "ink": {"channels": 1, "activation": "sigmoid", "weight": 1, "loss_fn": "BCEDiceLoss",
            "loss_kwargs": {"alpha": 0.5, "beta": 0.5}},
}
TASKS_SOFTMAX2 = {
    "seg": {"channels": 2, "activation": "softmax", "weight": 0.7, "loss_fn": "BCEDiceLoss",
            "loss_kwargs": {"alpha": 0.5, "beta": 0.5}},
}

TASKS_SIGMOID_SHEET = {
    "sheet": {"channels": 1, "activation": "sigmoid", "weight": 1, "loss_fn": "BCEDiceLoss",
              "loss_kwargs": {"alpha": 0.5, "beta": 0.5}},
}
TASKS_SOFTMAX2_W1 = {
    "seg": {"channels": 2, "activation": "softmax", "weight": 1, "loss_fn": "BCEDiceLoss",
            "loss_kwargs": {"alpha": 0.5, "beta": 0.5}},
}


def _manual(**kw):
    """manual topology (autoconfigure False); block names are Python LITERALS of this file (interned), which is what the
    reference's `is` comparisons (encoder.py:72-79) need -- strings parsed from YAML never select these paths upstream"""
    mc = {"basic_encoder_block": "BasicBlockD", "basic_decoder_block": "ConvBlock", "bottleneck_block": "BasicBlockD",
          "features_per_stage": [32, 64, 128], "num_stages": 3, "n_blocks_per_stage": [1, 2, 2],
          "kernel_sizes": [3, 3, 3], "n_conv_per_stage_decoder": [1, 1], "strides": [1, 2, 2]}
    mc.update(kw)
    return mc


CASES = {
    # isotropic autoconfig, two task heads (cfg3-style, shrunk): 3 stages [32,64,128], blocks [1,3,4]
    "auto16_2head": dict(patch=(16, 16, 16), batch=2, in_channels=1, tasks=TASKS_2HEAD,
                         autoconfigure=True, model_config={}, seed=0, data_seed=1234, train=True),
    # anisotropic autoconfig (per-axis strides (1,2,2) appear), conv_bias on, eval-mode sigmoid
    "auto_aniso_bias": dict(patch=(8, 32, 32), batch=1, in_channels=1, tasks=TASKS_SIGMOID,
                            autoconfigure=True, model_config={"conv_bias": True}, seed=1, data_seed=24,
                            train=True),
    # manual topology, 2 input channels, 320-style cap shrunk, softmax 2-class head, weight 0.7
    "manual_2in": dict(patch=(16, 16, 16), batch=1, in_channels=2, tasks=TASKS_SOFTMAX2,
                       autoconfigure=False,
                       model_config={"basic_encoder_block": "BasicBlockD", "basic_decoder_block": "ConvBlock",
                                     "bottleneck_block": "BasicBlockD", "features_per_stage": [32, 64, 64],
                                     "num_stages": 3, "n_blocks_per_stage": [1, 2, 2],
                                     "kernel_sizes": [3, 3, 3], "n_conv_per_stage_decoder": [1, 1],
                                     "strides": [1, 2, 2]},
                       seed=2, data_seed=11, train=True),
    # ---- round 2: the non-default topologies the reference can build -------------------------------
    # BottleneckD encoder (resblocks.py:135-259): 1x1x1 -> 3x3x3(stride) -> 1x1x1, projection skips
    "bottleneck_enc": dict(patch=(16, 16, 16), batch=2, in_channels=1, tasks=TASKS_SIGMOID_SHEET, autoconfigure=False,
                           model_config=_manual(basic_encoder_block="BottleneckBlockD", bottleneck_block="BottleneckBlockD",
                                                bottleneck_channels=[32, 32, 64]),
                           seed=11, data_seed=5, train=True,
                           full_grads=("blocks.0.conv3.conv.weight", "stages.1.blocks.1.conv1.conv.weight",
                                       "stages.0.blocks.0.conv2.conv.weight")),
    # ResidualBlock decoder (decoder.py:68-100): StackedResidualBlocks on the concat, softmax 2-class head
    "resdec_softmax": dict(patch=(16, 16, 16), batch=2, in_channels=1, tasks=TASKS_SOFTMAX2_W1, autoconfigure=False,
                           model_config=_manual(basic_decoder_block="ResidualBlock"),
                           seed=11, data_seed=5, train=True,
                           full_grads=("seg.stages.1.blocks.0.skip.0.conv.weight", "seg.stages.1.blocks.0.conv2.conv.weight",
                                       "seg.transpconvs.1.weight")),
    # plain-conv encoder (encoder.py:122-130, selected by the "ResidualBlock" quirk) + nn.ReLU + two convs in decoder stage 0
    "plain_relu_2conv": dict(patch=(16, 16, 16), batch=2, in_channels=2, tasks=TASKS_SIGMOID_SHEET, autoconfigure=False,
                             model_config=_manual(basic_encoder_block="ResidualBlock", nonlin="nn.ReLU",
                                                  n_conv_per_stage_decoder=[2, 1]),
                             seed=11, data_seed=1, train=True,
                             full_grads=("shared_encoder.stages.0.0.convs.0.conv.weight", "sheet.transpconvs.1.weight")),
    # 2-D patch (op_dims == 2, build_network_from_config.py:188-205): Conv2d / InstanceNorm2d / AvgPool2d
    "two_d": dict(patch=(32, 32), batch=2, in_channels=1, tasks=TASKS_SIGMOID_SHEET, autoconfigure=False,
                  model_config=_manual(), seed=11, data_seed=5, train=True,
                  full_grads=("stages.0.blocks.0.conv1.conv.weight", "sheet.transpconvs.1.weight",
                              "sheet.stages.1.convs.0.conv.weight")),
    # per-stage anisotropic kernels and strides given by hand
    "aniso_kernels": dict(patch=(8, 16, 16), batch=2, in_channels=1, tasks=TASKS_SIGMOID_SHEET, autoconfigure=False,
                          model_config=_manual(kernel_sizes=[[1, 3, 3], [3, 3, 3], [3, 3, 3]],
                                               strides=[[1, 1, 1], [1, 2, 2], [2, 2, 2]]),
                          seed=11, data_seed=1, train=True,
                          full_grads=("stages.0.blocks.0.conv1.conv.weight", "sheet.transpconvs.1.weight",
                                      "sheet.stages.1.convs.0.conv.weight")),
    # do_stem: false (encoder.py:81-89): the first residual block reads the image itself (conv1 1->32 and the 1x1x1 projection of
    # its skip path are both first-layer convolutions), 2 input channels
    "no_stem": dict(patch=(16, 16, 16), batch=2, in_channels=2, tasks=TASKS_SIGMOID_SHEET, autoconfigure=False,
                    model_config=_manual(do_stem=False), seed=11, data_seed=1, train=True,
                    full_grads=("stages.0.blocks.0.conv1.conv.weight", "stages.0.blocks.0.skip.0.conv.weight",
                                "sheet.transpconvs.1.weight")),
    # the same on the plain-conv encoder: stage 0's first ConvDropoutNormReLU is the first layer
    "no_stem_plain": dict(patch=(16, 16, 16), batch=2, in_channels=1, tasks=TASKS_SIGMOID_SHEET, autoconfigure=False,
                          model_config=_manual(do_stem=False, basic_encoder_block="ResidualBlock"), seed=11, data_seed=1, train=True,
                          full_grads=("shared_encoder.stages.0.0.convs.0.conv.weight", "sheet.transpconvs.1.weight")),
    # channel dropout (dropout_op_kwargs p > 0, build_network_from_config.py:169-170; nn.Dropout3d between conv and norm in the
    # stem, every block's conv1 and the decoder convs): the fixture carries the masks the reference drew (dropmask.NNN)
    "dropout": dict(patch=(16, 16, 16), batch=2, in_channels=1, tasks=TASKS_SIGMOID_SHEET, autoconfigure=False,
                    model_config=_manual(dropout_op_kwargs={"p": 0.25}), seed=11, data_seed=1, train=True,
                    full_grads=("stages.0.blocks.0.conv1.conv.weight", "stages.1.blocks.1.conv1.conv.weight",
                                "sheet.stages.1.convs.0.conv.weight")),
    # feature counts that are not a multiple of the kernels' 32-channel K tile (features_per_stage is free in the reference,
    # build_network_from_config.py:85-148): padded buffers + zero-padded shadow parameters inside the engine; conv_bias on,
    # 2-class softmax head, ResidualBlock decoder (its skip projection reads the padded concat too)
    "odd_channels": dict(patch=(16, 16, 16), batch=2, in_channels=1, tasks=TASKS_SOFTMAX2_W1, autoconfigure=False,
                         model_config=_manual(features_per_stage=[24, 48, 80], basic_decoder_block="ResidualBlock", conv_bias=True),
                         seed=11, data_seed=1, train=True,
                         full_grads=("stages.0.blocks.0.conv1.conv.weight", "stages.1.blocks.1.conv2.conv.weight",
                                     "seg.stages.1.blocks.0.conv1.conv.weight", "seg.transpconvs.0.weight",
                                     "seg.stages.1.blocks.0.skip.0.conv.weight")),
    # the round-2 widenings composed: 2-D patch, no stem, feature counts off the K tile, conv_bias, channel dropout, BottleneckD
    # encoder with its default C/4 bottlenecks (6, 12, 20 channels), ResidualBlock decoder, 3-class softmax head, 2 input channels
    "combo_2d": dict(patch=(32, 32), batch=2, in_channels=2,
                     tasks={"seg": {"channels": 3, "activation": "softmax", "weight": 1, "loss_fn": "BCEDiceLoss",
                                    "loss_kwargs": {"alpha": 0.5, "beta": 0.5}}},
                     autoconfigure=False,
                     model_config=_manual(do_stem=False, features_per_stage=[24, 48, 80], conv_bias=True,
                                          dropout_op_kwargs={"p": 0.2}, basic_encoder_block="BottleneckBlockD",
                                          bottleneck_block="BottleneckBlockD", basic_decoder_block="ResidualBlock"),
                     seed=11, data_seed=1, train=True,
                     full_grads=("stages.0.blocks.0.conv1.conv.weight", "stages.0.blocks.0.skip.0.conv.weight",
                                 "stages.1.blocks.1.conv2.conv.weight", "seg.transpconvs.1.weight",
                                 "seg.stages.1.blocks.0.conv1.conv.weight")),
    # ---- round 3: the conv geometries and widths the engine used to refuse (VERDICT r2 "What's missing" #2, #3) ----------------
    # 5- / 7-wide kernels and strides 3 / 4, per stage, by hand (build_network_from_config.py:85-148 -> Conv(k, stride, pad=(k-1)//2)):
    # 24^3 -> stride 3 -> 8^3 -> stride 2 -> 4^3; AvgPool(3) / AvgPool(2) skip paths, ConvTranspose k = s = 2 and 3 in the decoder
    "big_kernels": dict(patch=(24, 24, 24), batch=2, in_channels=1, tasks=TASKS_SIGMOID_SHEET, autoconfigure=False,
                        model_config=_manual(features_per_stage=[32, 32, 64], kernel_sizes=[[5, 5, 5], [3, 3, 3], [7, 7, 7]],
                                             strides=[[1, 1, 1], [3, 3, 3], [2, 2, 2]]),
                        seed=11, data_seed=1, train=True,
                        full_grads=("stages.1.blocks.0.conv1.conv.weight", "sheet.transpconvs.1.weight", "sheet.transpconvs.0.weight")),
    # stride 4 with a (1, 4, 4) anisotropic step and mixed kernel sizes per axis, plain-conv encoder, ResidualBlock decoder
    "stride4_mixed": dict(patch=(8, 32, 32), batch=2, in_channels=2, tasks=TASKS_SOFTMAX2_W1, autoconfigure=False,
                          model_config=_manual(features_per_stage=[32, 64], num_stages=2, n_blocks_per_stage=[1, 2],
                                               n_conv_per_stage_decoder=[1], kernel_sizes=[[3, 5, 5], [1, 7, 3]],
                                               strides=[[1, 1, 1], [1, 4, 4]], basic_decoder_block="ResidualBlock"),
                          seed=11, data_seed=1, train=True,
                          full_grads=("stages.1.blocks.0.conv1.conv.weight", "seg.transpconvs.0.weight")),
    # 20 input channels (the first-layer kernels read <= 16) into a 96-channel stem (their weight gradient holds <= 64)
    "wide_in_stem": dict(patch=(16, 16, 16), batch=2, in_channels=20, tasks=TASKS_SIGMOID_SHEET, autoconfigure=False,
                         model_config=_manual(stem_channels=96), seed=11, data_seed=1, train=True,
                         full_grads=("stages.0.blocks.0.conv1.conv.weight", "stages.0.blocks.0.skip.0.conv.weight",
                                     "sheet.transpconvs.1.weight")),
}

# Cases WITHOUT a reference fixture -- PARITY UNPINNED: SqueezeExcite / DropPath live in the un-vendored
# dynamic_network_architectures package; the checker for these is the oracle's restatement of their published source.  Data
# seeds curated like the golden ones (oracle/scan_seeds.py), so the live-oracle comparison can carry the 1e-3 bar too.
UNPINNED_CASES = {
    "squeeze_excite": dict(patch=(16, 16, 16), batch=2, in_channels=1, tasks=TASKS_SIGMOID_SHEET, autoconfigure=False,
                           model_config=_manual(squeeze_excitation=True), seed=11, data_seed=5),
    "squeeze_excite_bottleneck": dict(patch=(16, 16, 16), batch=2, in_channels=1, tasks=TASKS_SOFTMAX2_W1, autoconfigure=False,
                                      model_config=_manual(basic_encoder_block="BottleneckBlockD",
                                                           bottleneck_block="BottleneckBlockD",
                                                           bottleneck_channels=[32, 32, 64], squeeze_excitation=True),
                                      seed=11, data_seed=3),
    "squeeze_excite_odd_channels": dict(patch=(16, 16, 16), batch=2, in_channels=1, tasks=TASKS_SIGMOID_SHEET, autoconfigure=False,
                                        model_config=_manual(features_per_stage=[24, 48, 80], squeeze_excitation=True), seed=11,
                                        data_seed=1),
    "squeeze_excite_2d": dict(patch=(32, 32), batch=2, in_channels=1, tasks=TASKS_SIGMOID_SHEET, autoconfigure=False,
                              model_config=_manual(squeeze_excitation=True), seed=11, data_seed=5),
}

# NOTE on data seeds: the network's backward is discontinuous in the LeakyReLU masks.  Gradients late in
# training-free random nets are tiny and sparse, so ONE near-zero pre-activation whose sign differs between two
# fp32 evaluation orders moves a gradient tensor by 2e-3..2e-2 -- the reference's own fp32 CPU path differs from
# its fp64 evaluation by that much on about half of all seeds (tests/test_oracle_golden.py::
# test_fp32_gradients_are_mask_discontinuous).  The seeds below are ones where every mask has margin, so the
# 1e-3 bar is meaningful (`python oracle/scan_seeds.py` is the selection procedure: fp32 vs fp64 of the same path
# at two thread counts, first seed below 3e-5).
# parameters whose full gradient is stored in the fixture (small tensors); everything else is pinned
# through (sum, l2) checksums
FULL_GRAD_SUFFIXES = ("stem.convs.0.conv.weight", "seg_layers.1.weight", "seg_layers.1.bias",
                      "seg_layers.2.weight", "seg_layers.2.bias",
                      "stages.1.blocks.0.skip.1.conv.weight", "transpconvs.0.bias")
