"""`NetworkFromConfig(mgr)` -- the drop-in boundary (reference:
builders/build_network_from_config.py:20-326).

Same constructor contract (reads `mgr.tasks`, `train_patch_size`, `train_batch_size`, `in_channels`,
`vram_max`, `autoconfigure`, `model_config`), same module tree / parameter names / seeded init /
`state_dict` keys, same `forward(x) -> {task: tensor}` with the eval-only activation -- but
`forward` runs on the hand-written gfx950 kernels of librxunet.so through ONE autograd boundary
(`engine/plan.py`).  There is no PyTorch fallback: CPU tensors, a missing library or a non-gfx950
device raise.

Compute dtype (the reference gets it from the caller's `torch.amp.autocast`, train.py:203): fp32
outside autocast (parity mode, v_mfma_f32_32x32x2_f32), the autocast dtype inside it (bf16 / fp16
MFMA); `net.compute_dtype = torch.bfloat16` (or `model_config["compute_dtype"]`) pins it.
Logits are returned in fp32 in every mode.
"""
import torch
import torch.nn as nn

from ..engine import lib as _l
from ..engine.plan import Plan
from .decoder import Decoder
from .encoder import Encoder
from .utils import get_n_blocks_per_stage, get_pool_and_conv_props


def get_activation_module(activation_str: str):
    a = activation_str.lower()
    if a == "none":
        return None
    if a == "sigmoid":
        return nn.Sigmoid()
    if a == "softmax":
        return nn.Softmax(dim=1)
    raise ValueError(f"Unknown activation type: {activation_str}")


_MANUAL_KEYS = ("basic_encoder_block", "basic_decoder_block", "bottleneck_block", "features_per_stage", "num_stages",
                "n_blocks_per_stage", "kernel_sizes", "n_conv_per_stage_decoder", "strides")
_DTYPES = {"fp32": torch.float32, "float32": torch.float32, "bf16": torch.bfloat16, "bfloat16": torch.bfloat16,
           "fp16": torch.float16, "float16": torch.float16}


class _EngineFn(torch.autograd.Function):
    """the single autograd boundary: HIP forward list / HIP backward list of a `Plan`"""

    @staticmethod
    def forward(ctx, plan, x, names, *params):
        outs = plan.run_forward(x, apply_act=False)
        ctx.plan, ctx.names, ctx.generation = plan, names, plan.generation
        ctx.set_materialize_grads(False)
        return tuple(outs[n].clone() for n in names)

    @staticmethod
    def backward(ctx, *gouts):
        plan = ctx.plan
        if plan.generation != ctx.generation:
            raise RuntimeError("the engine keeps ONE set of activation buffers per input shape: call backward() "
                               "before the next forward() of the same shape (train.py:204-224 does)")
        grads = plan.run_backward(dict(zip(ctx.names, gouts)))
        return (None, None, None, *grads)


class NetworkFromConfig(nn.Module):
    def __init__(self, mgr):
        super().__init__()
        self.mgr = mgr
        self.tasks = mgr.tasks
        self.patch_size = mgr.train_patch_size
        self.batch_size = mgr.train_batch_size
        self.in_channels = mgr.in_channels
        self.vram_target = mgr.vram_max
        self.autoconfigure = mgr.autoconfigure
        model_config = mgr.model_config
        self.model_name = model_config.get("model_name", "Model")
        self.use_timm = False

        if mgr.autoconfigure:
            self.basic_encoder_block = "BasicBlockD"
            self.basic_decoder_block = "ConvBlock"
            self.bottleneck_block = "BasicBlockD"
            num_pool_per_axis, pool_op_kernel_sizes, conv_kernel_sizes, final_patch_size, _ = \
                get_pool_and_conv_props((1.0,) * len(mgr.train_patch_size), mgr.train_patch_size, 4, 999999)
            self.num_stages = len(pool_op_kernel_sizes)
            self.features_per_stage = [min(32 * 2 ** i, 512) for i in range(self.num_stages)]
            self.num_pool_per_axis = num_pool_per_axis
            self.pool_op_kernel_sizes = pool_op_kernel_sizes
            self.kernel_sizes = conv_kernel_sizes
            self.n_blocks_per_stage = get_n_blocks_per_stage(self.num_stages)
            self.n_conv_per_stage_decoder = [1] * (self.num_stages - 1)
            self.strides = pool_op_kernel_sizes
            self.final_patch_size = final_patch_size
        else:
            self.use_timm = model_config.get("use_timm_encoder", False)
            for key in _MANUAL_KEYS:
                if key not in model_config:
                    raise ValueError(f"autoconfigure=False, but '{key}' was not provided in the config!")
            self.basic_encoder_block = model_config["basic_encoder_block"]
            self.basic_decoder_block = model_config["basic_decoder_block"]
            self.bottleneck_block = model_config["bottleneck_block"]
            self.features_per_stage = model_config["features_per_stage"]
            self.num_stages = model_config["num_stages"]
            self.n_blocks_per_stage = model_config["n_blocks_per_stage"]
            self.kernel_sizes = model_config["kernel_sizes"]
            self.n_conv_per_stage_decoder = model_config["n_conv_per_stage_decoder"]
            self.strides = model_config["strides"]

        # read-then-overridden placeholders, as in the reference (:166-205)
        self.conv_op_kwargs = model_config.get("conv_op_kwargs", {"bias": False})
        self.dropout_op_kwargs = model_config.get("dropout_op_kwargs", {"p": 0.0})
        self.norm_op_kwargs = model_config.get("norm_op_kwargs", {"affine": False, "eps": 1e-5})
        self.conv_bias = model_config.get("conv_bias", False)
        self.nonlin = model_config.get("nonlin", "nn.LeakyReLU")
        self.nonlin_kwargs = model_config.get("nonlin_kwargs", {"inplace": True})
        self.return_skips = model_config.get("return_skips", True)
        self.do_stem = model_config.get("do_stem", True)
        self.stem_channels = model_config.get("stem_channels", None)
        self.bottleneck_channels = model_config.get("bottleneck_channels", None)
        self.stochastic_depth_p = model_config.get("stochastic_depth_p", 0.0)
        self.squeeze_excitation = model_config.get("squeeze_excitation", False)
        self.squeeze_excitation_reduction_ratio = 1.0 / 16.0 if self.squeeze_excitation else None
        self.stem_n_channels = self.features_per_stage[0]

        if len(self.patch_size) == 2:
            self.op_dims = 2
            self.conv_op, self.pool_op, self.norm_op, self.dropout_op = \
                nn.Conv2d, nn.AvgPool2d, nn.InstanceNorm2d, nn.Dropout2d
        elif len(self.patch_size) == 3:
            self.op_dims = 3
            self.conv_op, self.pool_op, self.norm_op, self.dropout_op = \
                nn.Conv3d, nn.AvgPool3d, nn.InstanceNorm3d, nn.Dropout3d
        else:
            raise ValueError("Patch size must have either 2 or 3 dimensions!")

        if self.nonlin == "nn.LeakyReLU":
            self.nonlin, self.nonlin_kwargs = nn.LeakyReLU, {"negative_slope": 1e-2, "inplace": True}
        elif self.nonlin == "nn.ReLU":
            self.nonlin, self.nonlin_kwargs = nn.ReLU, {"inplace": True}
        else:
            raise TypeError(f"nonlin {self.nonlin!r}: only 'nn.LeakyReLU' and 'nn.ReLU' are understood "
                            "(the reference would fail calling the string)")

        if self.bottleneck_block == "BottleneckBlockD":
            if self.bottleneck_channels is None:
                self.bottleneck_channels = [f // 4 for f in self.features_per_stage]
            elif isinstance(self.bottleneck_channels, int):
                self.bottleneck_channels = [self.bottleneck_channels] * len(self.features_per_stage)
        else:
            self.bottleneck_channels = None

        self.shared_encoder = Encoder(
            input_channels=self.in_channels, basic_block=self.basic_encoder_block, n_stages=self.num_stages,
            features_per_stage=self.features_per_stage, n_blocks_per_stage=self.n_blocks_per_stage,
            bottleneck_block=self.bottleneck_block, conv_op=self.conv_op, kernel_sizes=self.kernel_sizes,
            conv_bias=self.conv_bias, norm_op=self.norm_op, norm_op_kwargs=self.norm_op_kwargs,
            dropout_op=self.dropout_op, dropout_op_kwargs=self.dropout_op_kwargs, nonlin=self.nonlin,
            nonlin_kwargs=self.nonlin_kwargs, strides=self.strides, return_skips=self.return_skips,
            do_stem=self.do_stem, stem_channels=self.stem_n_channels, bottleneck_channels=self.bottleneck_channels,
            stochastic_depth_p=self.stochastic_depth_p, squeeze_excitation=self.squeeze_excitation,
            squeeze_excitation_reduction_ratio=self.squeeze_excitation_reduction_ratio)

        self.task_decoders = nn.ModuleDict()
        self.task_activations = nn.ModuleDict()
        for task_name, task_info in self.tasks.items():
            self.task_decoders[task_name] = Decoder(
                encoder=self.shared_encoder, basic_block=self.basic_decoder_block, num_classes=task_info["channels"],
                n_conv_per_stage=self.n_conv_per_stage_decoder, deep_supervision=False)
            self.task_activations[task_name] = get_activation_module(task_info.get("activation", "none"))

        cd = model_config.get("compute_dtype", None)
        self.compute_dtype = _DTYPES[cd] if isinstance(cd, str) else cd   # None -> follow autocast
        self._bind_submodules()
        self._plans = {}
        self._weights_epoch = 0     # advanced by every backward: fused optimizers do not bump Tensor._version (engine/plan.py)
        if getattr(mgr, "verbose", False):
            print(f"--- NetworkFromConfig (rxunet HIP engine): stages={self.num_stages} "
                  f"features={self.features_per_stage} blocks={self.n_blocks_per_stage} strides={self.strides} "
                  f"tasks={list(self.tasks)} ---")

    # ---- engine plumbing --------------------------------------------------------------------
    def _bind_submodules(self):
        """`model.shared_encoder(x)` / `model.task_decoders[t](skips)` run on this network's plans: the containers hold a weak
        reference to their network (not a submodule link)."""
        import weakref
        object.__setattr__(self.shared_encoder, "_owner", weakref.ref(self))
        for task_name in self.task_decoders.keys():
            object.__setattr__(self.task_decoders[task_name], "_owner", weakref.ref(self))
            object.__setattr__(self.task_decoders[task_name], "_task", task_name)

    def __getstate__(self):          # copy.deepcopy / pickle: plans are launch lists over THIS object's tensors -- rebuilt on demand
        st = dict(self.__dict__)
        st["_plans"] = {}
        return st

    def __setstate__(self, state):
        super().__setstate__(state)
        self._plans = {}
        self._bind_submodules()      # a copy's containers must run on the copy, not on the original

    def _resolve_dtype(self):
        if self.compute_dtype is not None:
            return self.compute_dtype
        if torch.is_autocast_enabled("cuda"):
            return torch.get_autocast_dtype("cuda")
        return torch.float32

    def plan_for(self, shape, dtype, device, needs_grad):
        key = (tuple(shape), dtype, str(device), bool(needs_grad))
        plan = self._plans.get(key)
        if plan is None:
            plan = Plan(self, shape, dtype, device, needs_grad)
            self._plans[key] = plan
        return plan

    def _apply(self, fn, *a, **k):   # .to()/.cuda()/.float() move parameters: cached plans hold stale pointers
        for plan in self._plans.values():     # (plans are reference cycles of closures: hand their library resources back now)
            plan.release()
        self._plans = {}
        return super()._apply(fn, *a, **k)

    @torch.compiler.disable(recursive=True)
    def forward(self, x):
        """`torch.compile(model)` (reference train.py:133) survives: the forward is one opaque engine call (ctypes launches
        on raw pointers, nothing a tracer could lower), so dynamo is told to run it as it is; the compiled wrapper still
        trains and its `state_dict()` carries the reference's `_orig_mod.` keys (train.py:250)."""
        self._check_input(x)
        dtype = self._resolve_dtype()
        needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        plan = self.plan_for(x.shape, dtype, x.device, needs_grad)
        names = tuple(self.task_decoders.keys())
        if needs_grad:
            outs = dict(zip(names, _EngineFn.apply(plan, x, names, *plan.params)))
            if not self.training:
                for n in names:     # rare: eval mode with autograd on -- activation on the logits
                    act = self.task_activations[n] if n in self.task_activations else None
                    if act is not None:
                        outs[n] = act(outs[n])
            return outs
        outs = plan.run_forward(x, apply_act=not self.training)
        return {n: outs[n].clone() for n in names}

    @torch.compiler.disable(recursive=True)
    def encode(self, x):
        """the shared encoder alone (reference encoder.py:148-158, `model.shared_encoder(x)`): per-stage outputs as NCDHW fp32
        tensors, computed by the encoder part of an inference plan (no autograd)."""
        self._check_input(x)
        with torch.no_grad():
            plan = self.plan_for(x.shape, self._resolve_dtype(), x.device, False)
            return plan.run_encoder(x)

    @torch.compiler.disable(recursive=True)
    def decode(self, task_name, skips):
        """one task decoder alone on a list of encoder outputs (reference decoder.py:137-162, `model.task_decoders[t](skips)`):
        raw logits (the task activation is `NetworkFromConfig.forward`'s business upstream too), computed by the decoder part of
        an inference plan (no autograd).  The input patch shape is the first stage's output times the first stage's stride."""
        skips = list(skips)
        s0 = skips[0]
        if not s0.is_cuda:
            raise _l.RxError("NetworkFromConfig runs only on an MI355X (gfx950) HIP device: got CPU tensors")
        st0 = list(self.strides[0]) if isinstance(self.strides[0], (list, tuple)) else [self.strides[0]] * self.op_dims
        shape = (s0.shape[0], self.in_channels, *[int(v) * int(k) for v, k in zip(s0.shape[2:], st0)])
        names = list(self.task_decoders.keys())
        with torch.no_grad():
            plan = self.plan_for(torch.Size(shape), self._resolve_dtype(), s0.device, False)
            return plan.run_decoder(names.index(task_name), task_name, skips)

    def _check_input(self, x):
        if not x.is_cuda:
            raise _l.RxError("NetworkFromConfig runs only on an MI355X (gfx950) HIP device: got a CPU tensor and "
                             "there is no CPU/PyTorch fallback (oracle/ holds the CPU checker used by the tests)")
        if x.dim() != self.op_dims + 2 or x.shape[1] != self.in_channels:
            raise ValueError(f"expected input (B, {self.in_channels}, *{self.op_dims} spatial dims), got {tuple(x.shape)}")

    @torch.compiler.disable(recursive=True)
    def forward_logits(self, x):
        """inference-only forward that returns RAW logits in whatever mode the module is in (`forward` applies the task
        activation in eval mode, build_network_from_config.py:320-323).  For callers that activate themselves -- the
        reference's inference.py:121-133 does, after `model.eval()` (:112) -- without switching the module to train mode,
        which would also switch stochastic depth on."""
        self._check_input(x)
        with torch.no_grad():
            plan = self.plan_for(x.shape, self._resolve_dtype(), x.device, False)
            outs = plan.run_forward(x, apply_act=False)
            return {n: outs[n].clone() for n in self.task_decoders.keys()}
