"""Static execution plan of the whole network on the HIP kernels.

A `Plan` is built once per (input shape, compute dtype, needs-grad): it walks the parameter-container
tree (`builders/*`), allocates every activation / statistics / gradient buffer up front
(channels-last, compute dtype; sized for 288 GB of HBM -- nothing is recomputed or freed inside a
step) and records two flat launch lists, forward and backward.  A step is then nothing but
C-ABI calls on torch's current stream: no torch op touches an activation.

Layout decisions:
  * `torch.cat((up, skip), 1)` (decoder.py:147) never runs: the transposed conv writes channels
    [0, C) of the concat buffer, and the encoder stage that produces the skip writes its output
    straight into channels [C, 2C) of the FIRST task decoder's buffer (further decoders get a
    strided copy).  Gradients mirror this.
  * the backward list runs the decoders first (decoder 0 first, because its concat-gradient write
    initialises the buffer the other decoders and the encoder accumulate into), then the encoder
    in reverse.
  * per-parameter packed copies ([tap][Co][Ci] and [tap][Ci][Co], compute dtype) are refreshed only
    when the parameter's version counter moved.
"""
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional

import os

import torch
from torch import nn

from . import lib as _l
from . import ops
from .ops import Act


class UnsupportedConfig(NotImplementedError):
    pass


_SIDE_STREAMS: Dict[str, "torch.cuda.Stream"] = {}


def side_stream(device):
    """ONE side HIP stream per device, shared by every plan: weight gradients, weight packing and the streamed optimizer
    step are FIFO-ordered against each other across plans (a training plan's parameter update vs. an evaluation plan's
    re-pack of the same parameters)."""
    key = str(device)
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return st


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


@dataclass(eq=False)
class AT:
    """an activation tensor of the plan and (training plans) the buffer of its gradient"""
    act: Act
    name: str
    gact: Optional[Act] = None
    written: bool = False      # plan-build-time tracking: has some backward step initialised gact?
    needs_grad: bool = True


@dataclass(eq=False)
class Rec:
    kind: str
    a: dict = field(default_factory=dict)


def _slope_of(nonlin):
    if nonlin is None:
        return 1.0
    if isinstance(nonlin, nn.LeakyReLU):
        return float(nonlin.negative_slope)
    if isinstance(nonlin, nn.ReLU):
        return 0.0
    raise UnsupportedConfig(f"nonlinearity {type(nonlin).__name__} has no HIP kernel (LeakyReLU / ReLU only)")


def _check_norm(norm):
    if norm is None:
        raise UnsupportedConfig("conv blocks without a norm layer are never produced by NetworkFromConfig")
    if not isinstance(norm, (nn.InstanceNorm3d, nn.InstanceNorm2d)) or norm.affine or norm.track_running_stats:
        raise UnsupportedConfig("only InstanceNorm(affine=False, track_running_stats=False) has a HIP kernel")
    return float(norm.eps)


class Plan:
    def __init__(self, net, in_shape, dtype, device, needs_grad):
        device = torch.device(device)
        if device.type != "meta":        # a 'meta' plan only validates structure (CPU tests); it cannot run
            _l.require_device()
        self.net = net
        self.dtype = dtype
        self.device = device
        self.needs_grad = needs_grad
        self.in_shape = tuple(in_shape)
        self.generation = 0
        self.two_d = net.op_dims == 2
        B, Cin = in_shape[0], in_shape[1]
        spatial = tuple(in_shape[2:])
        if self.two_d:
            spatial = (1,) + spatial         # a 2-D net is the 3-D engine with a unit Z axis
        self.B, self.Cin, self.spatial = B, Cin, spatial

        self.params: List[nn.Parameter] = []
        self._pidx: Dict[int, int] = {}
        self.packs: List[dict] = []          # {param, kind, w_fwd, w_bwd, version}
        self.enc_tape: List[Rec] = []
        self.dec_tapes: List[List[Rec]] = []
        self.fwd: List[Callable] = []
        self.bwd: List[Callable] = []
        self.dy_pool: Dict[tuple, Act] = {}
        self._gbase: Dict[int, torch.Tensor] = {}
        self.outputs: Dict[str, torch.Tensor] = {}
        self.bytes_alloc = 0
        self._x = None                        # the caller's input tensor of the current step
        self._dlogits: Dict[str, torch.Tensor] = {}
        self._grads: List[Optional[torch.Tensor]] = []
        self.grad_order: List[int] = []       # parameter indices in the order their gradients become ready
        self.grad_sync = None                 # optional engine.ddp.GradSync
        import os
        # weight gradients on a side HIP stream (off the dependency chain); RX_OVERLAP_WGRAD=0 keeps one stream
        self.overlap_wgrad = os.environ.get("RX_OVERLAP_WGRAD", "1") != "0"
        # dy slots per shape (RX_DY_RING): with 2 the main stream stalls whenever the side stream is more than one layer behind
        self.dy_ring = max(2, int(os.environ.get("RX_DY_RING", "4")))
        self._side = None
        self._ws2 = None
        self._dy_turn: Dict[tuple, int] = {}
        # Cross-stream ordering goes through the library's NUMBERED events (rx_event_record / rx_stream_wait) in every mode, so
        # that a recorded launch program (below) contains exactly what an eager pass issues.  _dy_free: dy buffers whose last
        # side-stream reader has recorded its event and that the main stream must wait for before writing them again.
        self._dy_free: Dict[int, int] = {}
        self._dy_ev: Dict[int, int] = {}          # id(dy tensor) -> event slot
        self._ev_fork = None                      # main -> side
        self._ev_join = None                      # side -> main
        self._pack_groups = None                  # [(entries, event slot)] in first-use order (static per plan)
        self._gates: List[dict] = []              # SqueezeExcite / DropPath state of every gated block, forward order
        self._drops: List[dict] = []              # channel-dropout layers (dropout_op_kwargs p > 0), forward order
        # Channel counts that are not a multiple of the kernels' K tile (features_per_stage is free in the reference,
        # build_network_from_config.py:85-148): every activation buffer is PADDED to the next multiple, and each parameter that
        # touches a padded extent gets a zero-padded fp32 SHADOW that the kernels read instead (refreshed from the real
        # parameter in _forward_pre; zero weights / bias keep the padding channels exactly 0 through conv, InstanceNorm,
        # LeakyReLU, pool, transposed conv and head); its gradient is produced padded and sliced back in _backward_finish.
        # No kernel knows about it.  (Cost: the padded FLOPs and ~3 small torch copies per padded parameter and step.)
        self._shadows: List[dict] = []
        self._shadow_of: Dict[int, dict] = {}
        self._pad_idx: Dict[int, dict] = {}       # parameter index -> shadow entry
        self._pad_done: List[int] = []
        # Launch programs (default on; RX_PROGRAMS=0 = one ctypes call per launch): the forward / backward lists are static, so
        # after two eager passes each is RECORDED once by the library while it executes (rx_prog_begin/end) and every later
        # step replays it with one C call -- the same launches on the same two streams, without ~700 host-language calls
        # (8 ms of host time per cfg2 step; the 64^3 configuration was host-bound).
        self.use_programs = os.environ.get("RX_PROGRAMS", "1") == "1"
        self._pstate: Dict[tuple, dict] = {}
        self._gflat = None                        # persistent gradient storage (program mode: recorded pointers must not move)
        self._gviews: Dict[int, torch.Tensor] = {}
        self._marks: List[tuple] = []             # (command index, parameter index, produced on the side stream?) while recording
        self._recording = None
        self._on_side = False
        self._use_gstore = False
        self._gflat_range = (0, 0)
        self._pack_slots: List[int] = []
        self._unstable_params = False
        self._ws_refs: List[object] = []
        self._raw_seen = None                 # last `net._raw_param_event` (engine/streamed_step.py) this plan waited for
        # HIP graphs (opt-in: RX_GRAPHS=1 or plan.use_graphs = True): the forward / backward launch lists are static
        # (fixed buffers, fixed shapes), so after two eager passes (lazy allocations, kernel attribute calls) each list
        # is captured once -- side stream, events and weight packing included -- and every later step is a single
        # hipGraphLaunch instead of ~700 host-side launches (8 ms of host time per cfg2 step -> < 1 ms).  Off by
        # default: a cfg2 step is GPU-bound (the host runs ahead of the queue), so replaying graphs measured the SAME
        # step time, and a capture must not race with other threads' HIP calls (e.g. a pin-memory loader thread).
        self.use_graphs = os.environ.get("RX_GRAPHS", "0") == "1"
        self._gstate: Dict[tuple, dict] = {}
        self._x_static = None
        self._dl_static: Dict[str, torch.Tensor] = {}
        self._image_t = None                      # channels-last copy of an image with > 16 channels (see _image_act)
        self._pack_delay_at = None                # forward-list index behind the full-resolution encoder stage (deferred weight packs)
        self._pack_deferred = None
        self._build()

    def release(self):
        """drop the recorded programs and return this plan's numbered events to the library (65536 slots per process; a model that
        rebuilds plans -- `.to()`, many input shapes in inference -- used to leak ~60 per plan).  Idempotent; called
        by `__del__` and by the model when it discards its plans."""
        try:
            self._pstate.clear()                  # programs first: they mention the slots
            slots = set(self._dy_ev.values()) | set(self._pack_slots)
            for s in (self._ev_fork, self._ev_join):
                if s is not None:
                    slots.add(s)
            self._dy_ev, self._pack_slots, self._ev_fork, self._ev_join = {}, [], None, None
            self._dy_free = {}
            for ent in self.packs:
                ent["event"] = None
            for s in slots:
                ops.event_free(s)
        except Exception:       # interpreter shutdown: the library may be gone
            pass

    def __del__(self):
        self.release()

    # ------------------------------------------------------------------ helpers
    def _k3(self, v):
        v = list(v)
        return [1] + v if self.two_d else v

    def _check_extent(self, numel, name):
        # the kernels address a SAMPLE with 32-bit element (some: byte) offsets and step from sample to sample with 64-bit strides
        # (`ptr + (long)n * sample_stride`; the LDS-DMA variants, whose buffer descriptors span the whole tensor, step aside above
        # 2^31 bytes per tensor): the cap is 2^31 bytes PER SAMPLE of a tensor.  Exercised on hardware: 256^3 x 32 channels bf16 at
        # batch 2 (2^31 bytes per tensor) and batch 3 (3.2 GB per tensor), scripts/big_patch_check.py
        per_sample = numel // max(self.B, 1)
        if per_sample * self.dtype.itemsize > (1 << 31):
            raise UnsupportedConfig(f"activation tensor {name} would have {per_sample} elements per sample "
                                    f"({per_sample * self.dtype.itemsize} bytes > 2^31): lower the patch size")

    def _new(self, dims, c, name, ld=None, needs_grad=True):
        self._check_extent(self.B * dims[0] * dims[1] * dims[2] * (ld or c), name)
        t = torch.empty((self.B, *dims, ld or c), dtype=self.dtype, device=self.device)
        self.bytes_alloc += t.numel() * t.element_size()
        return AT(Act(t, 0, c), name, needs_grad=needs_grad)

    def _new_cat(self, dims, c, name, allow_planar=True):
        """concat buffer (upsampled | skip) of a decoder stage.  Interleaved (2c channels per voxel, the halves are channel
        slices) in general; PLANAR -- two dense c-channel tensors, rx_act.cs -- for the 32-channel full-resolution level, where
        a 64-byte slice of a 128-byte voxel costs every kernel that streams a half 1.5-2x (scripts/probes/strided_probe.py)
        and the three conv kernels that read / write the whole concat can address its planes separately."""
        vox = dims[0] * dims[1] * dims[2]
        min_vox = int(os.environ.get("RX_PLANAR_MIN_VOXELS", str(64 ** 3)))
        planar = (allow_planar and c == 32 and self.dtype != torch.float32 and not self.two_d and dims[2] % 16 == 0 and vox >= min_vox
                  and self.B * (dims[0] // 4) * (dims[1] // 4) * (dims[2] // 16) >= 256
                  and os.environ.get("RX_PLANAR_CAT", "1") != "0")
        if not planar:
            return self._new(dims, 2 * c, name, ld=2 * c)
        self._check_extent(self.B * vox * 32, name)       # per PLANE: the two planes are separate dense tensors (rx_act.cs is 64-bit)
        root = torch.empty((2, self.B, *dims, 32), dtype=self.dtype, device=self.device)
        self.bytes_alloc += root.numel() * root.element_size()
        return AT(Act.planar(root), name)

    def _param(self, p):
        if id(p) not in self._pidx:
            self._pidx[id(p)] = len(self.params)
            self.params.append(p)
        return self._pidx[id(p)]

    def _pack(self, p, kind):
        if p.dim() != (4 if self.two_d else 5):
            raise UnsupportedConfig("unexpected weight rank")
        a, b = p.shape[0], p.shape[1]
        taps = p[0, 0].numel()
        co, ci = (a, b) if kind == "conv" else (b, a)
        ent = dict(param=p, kind=kind, version=-1,
                   w_fwd=torch.empty((taps, co, ci), dtype=self.dtype, device=self.device),
                   w_bwd=torch.empty((taps, ci, co), dtype=self.dtype, device=self.device) if self.needs_grad else None)
        self.bytes_alloc += 2 * ent["w_fwd"].numel() * ent["w_fwd"].element_size()
        self.packs.append(ent)
        return ent

    def _cp(self, c):
        """channel count of the BUFFER that holds c channels (next multiple of the kernels' K tile)"""
        return (c + 31) // 32 * 32      # (output channels: 32 in every compute type; the fp32 kernels' K tile is 16)

    def _shadow(self, p, shape, pairs):
        """the tensor the kernels read for parameter p: p itself when nothing is padded, else a persistent zero tensor of `shape`
        whose index ranges `pairs` = [(dst index, src index)] mirror p (see self._shadows)"""
        if p is None or tuple(shape) == tuple(p.shape):
            return p
        ent = self._shadow_of.get(id(p))
        if ent is None:
            sh = torch.zeros(tuple(shape), dtype=p.dtype, device=self.device)
            self.bytes_alloc += sh.numel() * sh.element_size()
            ent = dict(param=p, sh=sh, pairs=list(pairs), seen=None, gpad=None)
            self._shadows.append(ent)
            self._shadow_of[id(p)] = ent
            self._pad_idx[self._param(p)] = ent
        elif tuple(ent["sh"].shape) != tuple(shape):
            raise UnsupportedConfig("one parameter read with two different padded shapes")
        return ent["sh"]

    def _refresh_shadows(self):
        epoch = getattr(self.net, "_weights_epoch", 0)
        for e in self._shadows:
            q = e["param"]
            tag = (q._version, q.data_ptr(), epoch)
            if e["seen"] != tag:
                src = q.detach()
                for dst, sidx in e["pairs"]:
                    e["sh"][dst].copy_(src[sidx])
                e["seen"] = tag

    @staticmethod
    def _out_dims(dims, kernel, stride):
        return tuple((d + 2 * ((k - 1) // 2) - k) // s + 1 for d, k, s in zip(dims, kernel, stride))

    # ------------------------------------------------------------------ emitters (forward tape)
    def _emit_cdnr(self, tape, m, x: AT, out: Optional[AT] = None, residual: Optional[AT] = None,
                   final_slope: Optional[float] = None, first_of_net=False, se=None, drop_p=0.0):
        """conv -> (dropout p=0) -> InstanceNorm -> nonlin of one ConvDropoutNormReLU container.
        `residual`/`final_slope` fuse the block epilogue `nonlin(out + residual)` into the same
        elementwise pass.  Returns the activated output."""
        sp = m.spec()
        p_drop = float(sp["dropout_p"])
        if p_drop != 0.0:
            if not isinstance(m.dropout, (nn.Dropout3d, nn.Dropout2d)) or not (0.0 < p_drop < 1.0):
                raise UnsupportedConfig("only channel dropout (nn.Dropout3d / nn.Dropout2d, build_network_from_config.py:196-205) "
                                        "with 0 < p < 1 runs on the HIP path")
            if residual is not None or se is not None or drop_p > 0.0:
                raise UnsupportedConfig("dropout inside a block epilogue is never produced by NetworkFromConfig")
        if sp["nonlin_first"]:
            raise UnsupportedConfig("nonlin_first=True is never selected by NetworkFromConfig")
        eps = _check_norm(sp["norm"])
        slope = _slope_of(sp["nonlin"]) if final_slope is None else final_slope
        conv = sp["conv"]
        kernel, stride = self._k3(sp["kernel"]), self._k3(sp["stride"])
        # any kernel size 1..7 and stride 1..4 per axis (build_network_from_config.py:85-148 hands a manual model_config's values
        # straight to Conv(k, stride, pad=(k-1)//2)): everything beyond 1 / 3 and 1 / 2 runs on the tap-table gather kernels
        for k in kernel:
            if not 1 <= k <= 7:
                raise UnsupportedConfig(f"kernel size {k} has no HIP kernel (1..7 per axis)")
        for s in stride:
            if not 1 <= s <= 4:
                raise UnsupportedConfig(f"stride {s} has no HIP kernel (1..4 per axis)")
        cin, cout = conv.in_channels, conv.out_channels
        cout_p = self._cp(cout)
        first_of_net = first_of_net or x is None      # x is None: the layer reads the NCDHW image (the stem; without one, the first block)
        if first_of_net and (any(s != 1 for s in stride) or any(k not in (1, 3) for k in kernel) or cin > 16 or cout_p > 64
                             or (cin > 4 and cout_p * cin * kernel[0] * kernel[1] * kernel[2] * 4 > 160 * 1024)):
            # the first-layer kernels read the NCDHW image at stride 1, with 1- / 3-wide kernels, <= 16 input and <= 64 output channels
            # (weights LDS-resident): anything else (do_stem=False with strides[0] != 1, in_channels 17.., stem_channels > 64, 5- / 7-wide
            # first kernels) converts the image to the channels-last layout (padded to a multiple of 32 channels) and runs the
            # ordinary convolution kernels on it
            x = self._image_act(tape)
            first_of_net = False
        in_dims = x.act.dims[1:] if not first_of_net else self.spatial
        odims = self._out_dims(in_dims, kernel, stride)
        y = self._new(odims, cout_p, f"y:{len(tape)}")
        stats = torch.empty((self.B, cout_p, 2), dtype=torch.float32, device=self.device)
        if out is None:
            out = self._new(odims, cout_p, f"a:{len(tape)}")
        elif out.act.c != cout_p:
            raise RuntimeError("plan bug: destination of a layer has the wrong channel extent")
        widx = self._param(conv.weight)
        bidx = self._param(conv.bias) if conv.bias is not None else None
        kshape = tuple(conv.weight.shape[2:])
        b_k = self._shadow(conv.bias, (cout_p,), [((slice(0, cout),), (slice(None),))])
        if first_of_net:
            w_k = self._shadow(conv.weight, (cout_p, cin, *kshape), [((slice(0, cout),), (slice(None),))])
            tape.append(Rec("stem", dict(y=y, w=w_k, b=b_k, widx=widx, bidx=bidx, kernel=kernel)))
        else:
            # input-channel ranges of the weight -> where they sit in the (padded) input buffer; a decoder conv reads the concat
            # [upsampled C | skip C], whose halves are padded separately
            segs = getattr(x, "segs", None) or [(0, cin, 0)]
            if sum(L for _, L, _ in segs) != cin:
                raise RuntimeError("plan bug: input-channel segments do not add up")
            w_k = self._shadow(conv.weight, (cout_p, x.act.c, *kshape),
                               [((slice(0, cout), slice(d, d + L)), (slice(None), slice(s0, s0 + L))) for s0, L, d in segs])
            pk = self._pack(w_k, "conv")
            tape.append(Rec("conv", dict(x=x, y=y, pk=pk, b=b_k, widx=widx, bidx=bidx, kernel=kernel,
                                         stride=stride)))
        gate = None
        if se is not None or drop_p > 0.0:
            gate = self._gate_buffers(odims, cout_p, se, drop_p)
        drop = None
        if p_drop != 0.0:
            # channel dropout in front of the InstanceNorm = the norm with eps * (1-p)^2 for the kept (n, c) planes and rstd = 0 for
            # the dropped ones (rx_instnorm_stats_mask): no pass over y, forward or backward.  keep: this step's draw (torch RNG, on
            # the device, forward order); eps_now / active follow net.training (_forward_pre).
            drop = dict(p=p_drop, keep=torch.ones((self.B, cout_p), dtype=torch.float32, device=self.device), eps=eps,
                        eps_now=eps, active=False)
            self._drops.append(drop)
        tape.append(Rec("inact", dict(y=y, stats=stats, eps=eps, res=residual, out=out, slope=slope, gate=gate, drop=drop)))
        return out

    def _image_act(self, tape):
        if getattr(self, "_image_at", None) is not None:
            return self._image_at
        cin = self.Cin
        cp = self._cp(cin)
        y = self._new(self.spatial, cp, "image", needs_grad=False)
        if cin > 16:
            # more input channels than the first-layer kernels read: a boundary conversion by torch (permute + cast into the persistent
            # channels-last buffer, whose padding channels stay zero), issued next to the static input copy in run_forward -- outside
            # any recorded program or captured graph
            if self.device.type != "meta":
                y.act.t.zero_()
            self._image_t = y
            self._image_at = y
            return y
        eye = torch.zeros((cp, cin, 1, 1, 1), dtype=torch.float32, device=self.device)
        if self.device.type != "meta":
            eye[torch.arange(cin), torch.arange(cin)] = 1.0
        tape.append(Rec("image", dict(y=y, w=eye)))
        self._image_at = y
        return y

    def _draw_dropout(self, d):
        """this step's kept (n, c) planes of one dropout layer: 1 / 0, in place (bernoulli(1 - p) as torch's feature dropout)"""
        d["keep"].bernoulli_(1.0 - d["p"])

    def _gate_buffers(self, odims, c, se, drop_p):
        """SqueezeExcite / DropPath state of one residual block (csrc/rx_se.hip): the fc parameters are read raw (fp32),
        the small per-(n, line, c) tensors live for the whole step."""
        keep_x = 0 if (se is not None and self.two_d) else 1      # x.mean((2, 3)) of a 4-D tensor pools every spatial axis
        L = odims[2] if keep_x else 1
        f32 = dict(dtype=torch.float32, device=self.device)
        g = dict(keep_x=keep_x, drop_p=float(drop_p), se=None, scale=None,
                 mult=torch.empty((self.B, L, c), **f32), dadd=torch.empty((self.B, L, c), **f32),
                 m12=torch.empty((self.B, c, 2), **f32), pooled=None, hidden=None, gate=None)
        if drop_p > 0.0:
            g["scale"] = torch.ones(self.B, **f32)
        if se is not None:
            rd = se.fc1.out_channels
            if rd > 64:
                raise UnsupportedConfig(f"SqueezeExcite with {rd} reduction channels (the gate kernel holds <= 64)")
            c_real = se.fc2.out_channels
            ones = tuple(se.fc1.weight.shape[2:])
            g["se"] = dict(rd=rd, keep_x=keep_x,
                           w1=self._shadow(se.fc1.weight, (rd, c, *ones), [((slice(None), slice(0, c_real)), (slice(None), slice(None)))]),
                           b1=se.fc1.bias,
                           w2=self._shadow(se.fc2.weight, (c, rd, *ones), [((slice(0, c_real),), (slice(None),))]),
                           b2=self._shadow(se.fc2.bias, (c,), [((slice(0, c_real),), (slice(None),))]),
                           idx=[self._param(se.fc1.weight), self._param(se.fc1.bias), self._param(se.fc2.weight),
                                self._param(se.fc2.bias)])
            g["pooled"] = torch.empty((self.B, L, c), **f32)
            g["hidden"] = torch.empty((self.B, L, rd), **f32)
            g["gate"] = torch.empty((self.B, L, c), **f32)
        if self.device.type == "cuda":
            need = 4 * (self.B * 256 * 2 * odims[2] * c + self.B * odims[2] * (3 * c + 64)) + 1024
            ops.workspace(need, self.device)
        g["scale_now"] = None
        self._gates.append(g)
        return g

    @staticmethod
    def _se_args(g):
        se = g["se"]
        if se is None:
            return None
        return dict(w1=se["w1"], b1=se["b1"], w2=se["w2"], b2=se["b2"], rd=se["rd"], keep_x=se["keep_x"])

    @staticmethod
    def _eps_now(ia):
        d = ia.get("drop")
        return d["eps_now"] if d is not None else ia["eps"]

    @staticmethod
    def _mask_dropped(ia):
        d = ia.get("drop")
        if d is not None and d["active"]:
            ops.instnorm_stats_mask(ia["stats"], d["keep"])

    def _draw_path_scale(self, g):
        """DropPath: per-sample bernoulli(keep)/keep in training, None (identity) in eval"""
        if g["scale"] is None or not self.net.training:
            return None
        keep = 1.0 - g["drop_p"]
        g["scale"].bernoulli_(keep)
        if keep > 0.0:
            g["scale"].div_(keep)
        return g["scale"]

    def _emit_block(self, tape, blk, x: AT, out: Optional[AT] = None):
        """BasicBlockD / BottleneckD: skip path, main path, fused `nonlin(IN(conv_k(..)) + skip)`."""
        r = x
        if x is None and (not blk.skip_ops or isinstance(blk.skip_ops[0], (nn.AvgPool3d, nn.AvgPool2d))):
            # do_stem=False and the first block adds (or pools) the RAW image on its skip path: bring the image into the engine's
            # channels-last layout with the first-layer kernel and an identity 1x1x1 weight (a constant, not a parameter)
            r = self._image_act(tape)
        for op in blk.skip_ops:
            if isinstance(op, (nn.AvgPool3d, nn.AvgPool2d)):
                st = self._k3(op.stride if isinstance(op.stride, (list, tuple)) else [op.stride] * (2 if self.two_d else 3))
                odims = tuple(d // s for d, s in zip(r.act.dims[1:], st))
                p = self._new(odims, r.act.c, f"pool:{len(tape)}", needs_grad=r.needs_grad)
                tape.append(Rec("pool", dict(x=r, y=p, stride=st)))
                r = p
            else:
                r = self._emit_cdnr(tape, op, r)        # 1x1 conv -> IN (no nonlin: slope 1)
        path = blk.main_path()
        h = x
        for m in path[:-1]:
            h = self._emit_cdnr(tape, m, h)
        se = blk.squeeze_excitation if getattr(blk, "apply_se", False) else None
        drop_p = blk.drop_path.drop_prob if getattr(blk, "apply_stochastic_depth", False) else 0.0
        return self._emit_cdnr(tape, path[-1], h, out=out, residual=r, final_slope=_slope_of(blk.final_nonlin()),
                               se=se, drop_p=drop_p)

    # ------------------------------------------------------------------ build
    def _build(self):
        net = self.net
        enc = net.shared_encoder
        tasks = list(net.task_decoders.keys())
        n_st = len(enc.stages)
        feats = list(enc.output_channels)
        featsp = [self._cp(f) for f in feats]

        # ---- spatial size per encoder stage
        dims = [None] * n_st
        cur = self.spatial
        for s in range(n_st):
            st = self._k3(enc.strides[s])
            kz = self._k3(enc.kernel_sizes[s] if isinstance(enc.kernel_sizes[s], (list, tuple)) else
                          [enc.kernel_sizes[s]] * (2 if self.two_d else 3))
            cur = self._out_dims(cur, kz, st)
            dims[s] = cur

        # ---- concat buffers of every decoder (decoder d, decoder-stage j <-> encoder stage n_st-2-j)
        cats: List[List[AT]] = []
        for d in range(len(tasks)):
            row = []
            for j in range(n_st - 1):
                es = n_st - 2 - j
                # the planar layout is addressed as a whole only by the 3x3x3 stride-1 halo kernels: the stage that reads the concat
                # must be a plain conv stack starting with such a conv (a ResidualBlock stage also reads it through a 1x1x1 projection;
                # found by tests/test_fuzz_gpu.py)
                st_mod = net.task_decoders[tasks[d]].stages[j]
                first = None if hasattr(st_mod, "blocks") else st_mod.convs[0]
                halo_ok = first is not None and list(self._k3(first.spec()["kernel"])) == [3, 3, 3]
                cat = self._new_cat(dims[es], featsp[es], f"cat{d}.{j}", allow_planar=halo_ok)
                cat.segs = [(0, feats[es], 0), (feats[es], feats[es], featsp[es])]     # (weight range start, length, buffer start)
                row.append(cat)
            cats.append(row)

        def skip_home(es):
            """where encoder stage `es` writes its output: second half of decoder 0's concat buffer"""
            if es == n_st - 1 or not tasks:
                return None
            cat = cats[0][n_st - 2 - es]
            c = featsp[es]
            return AT(cat.act.slice(c, c), f"skip{es}")

        # ---- encoder
        tape = self.enc_tape
        x = None
        # do_stem=False (encoder.py:81-89): the first block of stage 0 reads the NCDHW image itself -- its first main-path conv and
        # its skip projection both run on the first-layer kernels (h stays None until then)
        h = None
        for m in (list(enc.stem.convs) if enc.stem is not None else []):
            h = self._emit_cdnr(tape, m, h)
        skips: List[AT] = []
        for s in range(n_st):
            stage = enc.stages[s]
            home = skip_home(s)
            if enc.is_residual:
                blocks = list(stage.blocks)
                for bi, blk in enumerate(blocks):
                    h = self._emit_block(tape, blk, h, out=home if bi == len(blocks) - 1 else None)
            else:
                mods = list(stage[0].convs)
                for mi, m in enumerate(mods):
                    h = self._emit_cdnr(tape, m, h, out=home if mi == len(mods) - 1 else None)
            skips.append(h)
        self.enc_skips = [(at, feats[i]) for i, at in enumerate(skips)]      # (activation, real channel count) per stage

        # ---- decoders
        self.head_recs = []
        for d, name in enumerate(tasks):
            dec = net.task_decoders[name]
            tape = []
            self.dec_tapes.append(tape)
            low = skips[-1]
            for j in range(n_st - 1):
                es = n_st - 2 - j
                c = featsp[es]
                cat = cats[d][j]
                up = AT(cat.act.slice(0, c), f"up{d}.{j}")
                tconv = dec.transpconvs[j]
                stride = self._k3(tconv.stride)
                if list(tconv.kernel_size) != list(tconv.stride):
                    raise UnsupportedConfig("ConvTranspose with kernel != stride")
                ti, to = tconv.in_channels, tconv.out_channels
                if self._cp(ti) != low.act.c or self._cp(to) != c:
                    raise RuntimeError("plan bug: transposed-conv channel extents")
                w_k = self._shadow(tconv.weight, (low.act.c, c, *tconv.weight.shape[2:]),
                                   [((slice(0, ti), slice(0, to)), (slice(None), slice(None)))])
                b_k = self._shadow(tconv.bias, (c,), [((slice(0, to),), (slice(None),))])
                pk = self._pack(w_k, "convT")
                widx = self._param(tconv.weight)
                bidx = self._param(tconv.bias) if tconv.bias is not None else None
                tape.append(Rec("convT", dict(x=low, y=up, pk=pk, b=b_k, widx=widx, bidx=bidx, stride=stride,
                                              cat=cat)))
                if d > 0:
                    dst = AT(cat.act.slice(c, c), f"skipcopy{d}.{j}")
                    tape.append(Rec("copy", dict(x=skips[es], y=dst, cat=cat)))
                st_mod = dec.stages[j]
                h = cat
                if hasattr(st_mod, "blocks"):
                    for blk in st_mod.blocks:
                        h = self._emit_block(tape, blk, h)
                else:
                    for m in st_mod.convs:
                        h = self._emit_cdnr(tape, m, h)
                low = h
            head = dec.seg_layers[-1]
            k = head.out_channels
            if k > 1024:
                raise UnsupportedConfig("task heads with more than 1024 channels have no HIP kernel")
            out = torch.empty((self.B, k, *low.act.dims[1:]), dtype=torch.float32, device=self.device)
            self.outputs[name] = out
            act_mod = net.task_activations[name] if name in net.task_activations else None
            act_code = _l.RX_ACT_NONE
            if isinstance(act_mod, nn.Sigmoid):
                act_code = _l.RX_ACT_SIGMOID
            elif isinstance(act_mod, nn.Softmax):
                act_code = _l.RX_ACT_SOFTMAX
            hw_k = self._shadow(head.weight, (k, low.act.c, *head.weight.shape[2:]),
                                [((slice(None), slice(0, head.in_channels)), (slice(None), slice(None)))])
            tape.append(Rec("head", dict(x=low, w=hw_k, b=head.bias, widx=self._param(head.weight),
                                         bidx=self._param(head.bias), k=k, name=name, out=out, act=act_code)))

        self._gen_forward()
        if self.needs_grad:
            self._gen_backward()

    # ------------------------------------------------------------------ launch lists
    def _gen_forward(self):
        P = self
        f = self.fwd
        self.n_fwd_enc = None                 # number of forward steps that belong to the encoder tape (run_encoder)
        self.fwd_dec_start = []               # first forward step of each decoder tape (run_decoder)
        for tape in [self.enc_tape] + self.dec_tapes:
            if tape is not self.enc_tape and self.n_fwd_enc is None:
                self.n_fwd_enc = len(f)
            if tape is not self.enc_tape:
                self.fwd_dec_start.append(len(f))
            # conv -> InstanceNorm pairs whose statistics can come out of the conv epilogue (rx_conv3d_fwd_stats): 3x3x3
            # stride-1 layers in a 16-bit compute type, above the size the single-launch InstanceNorm kernel takes
            for i, rec in enumerate(tape[:-1]):
                nxt = tape[i + 1]
                if (rec.kind == "conv" and nxt.kind == "inact" and nxt.a["y"] is rec.a["y"] and self.dtype != torch.float32
                        and list(rec.a["kernel"]) == [3, 3, 3] and list(rec.a["stride"]) == [1, 1, 1]
                        and rec.a["y"].act.voxels > 512 and rec.a["y"].act.dims[3] >= 16):
                    rec.a["stats_to"] = nxt.a
                    nxt.a["stats_done"] = True
                if (rec.kind == "stem" and nxt.kind == "inact" and nxt.a["y"] is rec.a["y"] and self.dtype != torch.float32
                        and rec.a["y"].act.voxels > 512 and os.environ.get("RX_FUSED_STEM_STATS", "1") != "0"):      # rx_stem_conv_fwd_stats
                    rec.a["stats_to"] = nxt.a
                    nxt.a["stats_done"] = True
            # block output -> AvgPool of the next block's skip path: one pass (rx_instnorm_act_pool_fwd) above the size the
            # single-launch InstanceNorm kernel takes
            producers = {id(r.a["out"]): r for r in tape if r.kind == "inact" and r.a["gate"] is None}
            for rec in tape:
                if (rec.kind == "pool" and id(rec.a["x"]) in producers and rec.a["x"].act.voxels > 512
                        and os.environ.get("RX_FUSED_POOL", "1") != "0"):
                    src = producers[id(rec.a["x"])]
                    if "pool_to" not in src.a:
                        src.a["pool_to"] = rec.a
                        rec.a["fused"] = True
            # the layer under a task head: InstanceNorm apply + LeakyReLU + the head's 1x1x1 conv in one pass (the activated
            # output is written for the backward but not re-read by a separate head kernel)
            if self.dtype != torch.float32 and os.environ.get("RX_FUSED_HEAD_FWD", "1") != "0":
                for rec in tape:
                    if rec.kind != "head" or rec.a["k"] > 4:
                        continue
                    prods = [r for r in tape if r.kind == "inact" and r.a["out"] is rec.a["x"]]
                    if len(prods) != 1:
                        continue
                    pa = prods[0].a
                    c = pa["out"].act.c
                    if (pa["gate"] is None and pa["res"] is None and "pool_to" not in pa and not pa.get("norm_done")
                            and pa["out"].act.voxels > 512 and c % 8 == 0 and 64 % (c // 8) == 0):
                        pa["head_to"] = rec.a
                        rec.a["fwd_fused"] = True
            for rec in tape:
                a = rec.a
                if (tape is self.enc_tape and self._pack_delay_at is None and rec.kind in ("conv", "pool")
                        and rec.a["y"].act.voxels < self.spatial[0] * self.spatial[1] * self.spatial[2]):
                    self._pack_delay_at = len(f)          # the first layer below full resolution
                if rec.kind == "stem":
                    def sstep(a=a):
                        st = a.get("stats_to")
                        if st is not None:
                            ops.stem_conv_fwd_stats(P._x, a["w"], a["b"], a["y"].act, a["kernel"], st["stats"], P._eps_now(st))
                            P._mask_dropped(st)
                        else:
                            ops.stem_conv_fwd(P._x, a["w"], a["b"], a["y"].act, a["kernel"])
                    f.append(sstep)
                elif rec.kind == "image":        # NCDHW fp32 image -> channels-last compute type (identity 1x1x1 first-layer conv)
                    f.append(lambda a=a: ops.stem_conv_fwd(P._x, a["w"], None, a["y"].act, [1, 1, 1]))
                elif rec.kind == "conv":
                    def cstep(a=a):
                        P._await_pack(a["pk"])
                        st = a.get("stats_to")
                        if st is not None:
                            ops.conv3d_fwd_stats(a["x"].act, a["pk"]["w_fwd"], a["b"], a["y"].act, a["kernel"], a["stride"],
                                                 st["stats"], P._eps_now(st))
                            P._mask_dropped(st)
                        else:
                            ops.conv3d_fwd(a["x"].act, a["pk"]["w_fwd"], a["b"], a["y"].act, a["kernel"], a["stride"])
                    f.append(cstep)
                elif rec.kind == "convT":
                    def tstep(a=a):
                        P._await_pack(a["pk"])
                        ops.convT3d_fwd(a["x"].act, a["pk"]["w_fwd"], a["b"], a["y"].act, a["stride"])
                    f.append(tstep)
                elif rec.kind == "inact" and a["gate"] is not None:
                    def gstep(a=a):
                        g = a["gate"]
                        res = a["res"].act if a["res"] is not None else None
                        # g["scale_now"]: this step's DropPath factors, drawn at the start of the forward (_forward_body)
                        if g["se"] is None and g["scale_now"] is None:        # DropPath in eval: the plain block
                            ops.instnorm_fwd(a["y"].act, a["stats"], a["out"].act, a["slope"], res, a["eps"])
                            return
                        if not a.get("stats_done"):
                            ops.instnorm_stats(a["y"].act, a["stats"], a["eps"])
                        ops.se_gate_fwd(a["y"].act, a["stats"], P._se_args(g), g["pooled"], g["hidden"], g["gate"], g["mult"],
                                        g["scale_now"])
                        ops.instnorm_gate_act_fwd(a["y"].act, a["stats"], g["mult"], g["keep_x"], a["out"].act, a["slope"], res)
                    f.append(gstep)
                elif rec.kind == "inact":
                    def step(a=a):
                        res = a["res"].act if a["res"] is not None else None
                        head = a.get("head_to")
                        if head is not None:
                            if not a.get("stats_done"):
                                ops.instnorm_stats(a["y"].act, a["stats"], P._eps_now(a))
                                P._mask_dropped(a)
                            # the activated output is stored only if somebody will read it: a training plan whose backward
                            # rebuilds the head's gradients from y (a["head_src"], set by _gen_backward) and an inference plan do not
                            keep = P.needs_grad and not a.get("head_dw_fused")
                            ops.instnorm_act_head_fwd(a["y"].act, a["stats"], a["out"].act if keep else None, head["w"].view(head["k"], -1),
                                                      head["b"], head["out"], head["act"] if P._apply_act else _l.RX_ACT_NONE, a["slope"])
                            return
                        pool = a.get("pool_to")
                        if pool is not None:
                            if not a.get("stats_done"):
                                ops.instnorm_stats(a["y"].act, a["stats"], P._eps_now(a))
                                P._mask_dropped(a)
                            ops.instnorm_act_pool_fwd(a["y"].act, a["stats"], a["out"].act, pool["y"].act, pool["stride"], a["slope"], res)
                        elif a.get("stats_done"):     # the producing conv left (mean, rstd) behind
                            ops.instnorm_act_fwd(a["y"].act, a["stats"], a["out"].act, a["slope"], res)
                        elif a["drop"] is not None and a["drop"]["active"]:     # statistics, dropped planes, apply
                            ops.instnorm_stats(a["y"].act, a["stats"], P._eps_now(a))
                            P._mask_dropped(a)
                            ops.instnorm_act_fwd(a["y"].act, a["stats"], a["out"].act, a["slope"], res)
                        else:
                            ops.instnorm_fwd(a["y"].act, a["stats"], a["out"].act, a["slope"], res, a["eps"])
                    f.append(step)
                elif rec.kind == "pool":
                    if not a.get("fused"):
                        f.append(lambda a=a: ops.avgpool_fwd(a["x"].act, a["y"].act, a["stride"]))
                elif rec.kind == "copy":
                    f.append(lambda a=a: ops.avgpool_fwd(a["x"].act, a["y"].act, (1, 1, 1)))
                elif rec.kind == "head":
                    if a.get("fwd_fused"):          # computed by the InstanceNorm step of the layer below
                        continue

                    def step(a=a):
                        w2 = a["w"].view(a["k"], -1)
                        ops.head_fwd(a["x"].act, w2, a["b"], a["out"], a["act"] if P._apply_act else _l.RX_ACT_NONE)
                    f.append(step)

    def _dy_for(self, y: AT):
        """scratch buffer for dL/dy of a conv output.  Two per shape, used alternately: the weight-gradient kernel
        of layer L (side stream) may still be reading its dy while layer L-1's backward already writes the next."""
        key = (y.act.dims, y.act.c)
        turn = self._dy_turn.get(key, 0)
        self._dy_turn[key] = (turn + 1) % self.dy_ring
        slot = key + (turn,)
        if slot not in self.dy_pool:
            t = torch.empty((*y.act.dims, y.act.c), dtype=self.dtype, device=self.device)
            self.bytes_alloc += t.numel() * t.element_size()
            self.dy_pool[slot] = Act(t)
        return self.dy_pool[slot]

    def _gen_backward(self):
        P = self
        b = self.bwd
        cat_written = {}

        def mark_cat(cat):
            cat_written[cat.act.key] = True

        def view_written(at):
            return at.written or cat_written.get(at.act.key, False)

        def new_grad(idx):
            # a FRESH tensor every backward (autograd may keep / accumulate into what we return); a gradient
            # synchroniser may hand out views of its flat buckets instead (engine/ddp.py)
            sync = P.grad_sync
            ent = P._pad_idx.get(idx)
            if ent is not None:                 # padded parameter: the kernels write the padded gradient, _backward_finish slices it
                if ent["gpad"] is None:
                    ent["gpad"] = torch.empty(tuple(ent["sh"].shape), dtype=torch.float32, device=P.device)
                P._pad_done.append(idx)
                return ent["gpad"]
            if sync is not None:
                g = sync.alloc(idx)
            elif P._use_gstore:             # program mode: the same storage every backward (see _grad_store)
                g = P._gviews[idx]
            else:
                g = torch.empty_like(P.params[idx], memory_format=torch.contiguous_format)
            P._grads[idx] = g
            return g

        # TIMING ABLATIONS ONLY (wrong results): honoured only together with RX_ABLATION=1, anything else is an error
        skip = set(filter(None, os.environ.get("RX_SKIP", "").split(",")))
        if skip and os.environ.get("RX_ABLATION", "0") != "1":
            raise _l.RxError("RX_SKIP drops launches from the backward pass (timing ablations, wrong gradients): "
                             "set RX_ABLATION=1 as well to confirm, or unset RX_SKIP")

        def done(idx):
            if idx in P._pad_idx:               # completed (and announced to a gradient synchroniser) in _backward_finish
                return
            if P._recording is not None:        # while a program is recorded: where in the command list the gradient is complete
                P._marks.append((len(P._recording), idx, P._on_side))
            if P.grad_sync is not None:
                P.grad_sync.ready(idx)

        # events guarding the dy slots of every shape: the side stream's last reader of a slot must be done
        # before the main stream writes that slot again
        dy_free = self._dy_free

        def side_run(fn, dy_act):
            """run `fn` (weight-gradient launches reading dy_act) on the side stream, ordered after everything the
            main stream has enqueued so far"""
            if not (P.overlap_wgrad and P._side is not None):
                fn(ops.workspace())
                return
            ops.event_record(P._ev_fork)              # main (current) stream
            ops.stream_wait(P._ev_fork, P._side)
            key = id(dy_act.t)
            slot = P._dy_ev.get(key)
            if slot is None:
                slot = P._dy_ev[key] = ops.event_new()
            P._on_side = True
            try:
                with torch.cuda.stream(P._side):
                    fn(P._ws2)
                    ops.event_record(slot)
            finally:
                P._on_side = False
            dy_free[key] = slot

        def before_dy_write(dy_act):
            slot = dy_free.pop(id(dy_act.t), None)
            if slot is not None:
                ops.stream_wait(slot)

        order = self.grad_order

        # who wrote an activation's gradient LAST (plan-build order = run order): when the output gradient of an InstanceNorm
        # layer WITHOUT residual is completed by a 3x3x3 backward-data launch, that launch can accumulate the layer's two
        # backward sums in its epilogue (rx_conv3d_bwd_data_instats) and the reduce pass over (g, y) disappears
        def wrote(at, kind, info=None):
            at._gw = (kind, info)

        for tape in self.dec_tapes + [self.enc_tape]:
            for rec in reversed(tape):
                a = rec.a
                if rec.kind == "head":
                    x = a["x"]
                    assert not view_written(x)
                    gx = self._grad_buf(x)
                    x.written = True
                    hinfo = {"fused": False}      # set by the InstanceNorm step below when it rebuilds the head's data gradient itself
                    wrote(x, "head", hinfo)
                    a["hinfo"] = hinfo

                    def step(a=a, gx=gx, hinfo=hinfo):
                        dl = P._dlogits.get(a["name"])
                        if dl is None:      # this task did not take part in the loss
                            gx.t.zero_()
                            if P.grad_sync is not None:     # collectives need every rank to fill every bucket
                                for i in (a["widx"], a["bidx"]):
                                    new_grad(i).zero_()
                                    done(i)
                            return
                        if hinfo.get("dw_fused"):     # dw / db come out of the InstanceNorm backward's reduce pass (next step)
                            return
                        dw, db = new_grad(a["widx"]), new_grad(a["bidx"])
                        ops.head_bwd(dl, a["x"].act, a["w"].view(a["k"], -1), None if hinfo["fused"] else gx, dw, db)
                        done(a["widx"])
                        done(a["bidx"])
                    order += [a["widx"], a["bidx"]]
                    b.append(step)
                elif rec.kind == "inact":
                    out, y, res = a["out"], a["y"], a["res"]
                    gout = self._grad_buf(out) if out.gact is None else out.gact
                    if not view_written(out):
                        raise RuntimeError(f"plan bug: gradient of {out.name} is consumed before it is produced")
                    dy = self._dy_for(y)
                    y.gact = dy
                    gres, acc = None, False
                    if res is not None and res.needs_grad:        # (the converted image of a stem-less net needs no gradient)
                        gres = self._grad_buf(res)
                        acc = view_written(res)
                        res.written = True
                        wrote(res, "gres")
                    gw = getattr(out, "_gw", (None, None))
                    # the layer under a task head (no residual): its output gradient is rank K -- rebuilt from the logit gradient
                    # inside both InstanceNorm passes instead of written by the head and read back twice
                    if (gw[0] == "head" and a["gate"] is None and res is None and self.dtype != torch.float32
                            and out.act.voxels > 512 and os.environ.get("RX_FUSED_HEAD_BWD", "1") != "0"):
                        heads = [r.a for r in tape if r.kind == "head" and r.a["x"] is out]
                        if len(heads) == 1 and heads[0]["k"] <= 4:
                            a["head_src"] = heads[0]
                            gw[1]["fused"] = True
                            # ... and the head's own dw / db from the same reduce pass (the activation is recomputed from y): no
                            # head_bwd launch, and the forward does not store this layer's activated output at all.  Needs the
                            # forward's fused head (the un-fused head kernel reads the stored output) and an un-padded head weight.
                            if (heads[0].get("fwd_fused") and heads[0]["w"] is self.params[heads[0]["widx"]]
                                    and os.environ.get("RX_FUSED_HEAD_DW", "1") != "0"):
                                a["head_dw_fused"] = True
                                gw[1]["dw_fused"] = True
                    if (gw[0] == "conv" and a["gate"] is None and res is None and self.dtype != torch.float32
                            and out.act.full_buffer and out.act.root is None and out.act.c % 32 == 0 and out.act.voxels > 512
                            and out.act.dims[3] >= 16 and os.environ.get("RX_FUSED_BWD_STATS", "1") != "0"):
                        a["m12"] = torch.empty((self.B, out.act.c, 2), dtype=torch.float32, device=self.device)
                        a["m12_valid"] = False
                        gw[1]["inact"] = a          # that conv's backward-data step now also fills a["m12"]
                    def gistep(a=a, gout=gout, dy=dy, gres=gres, acc=acc):
                        g = a["gate"]
                        before_dy_write(dy)
                        if g["se"] is None and g.get("scale_now") is None:
                            ops.instnorm_act_bwd(gout, a["y"].act, a["stats"],
                                                 a["out"].act if (a["slope"] != 1.0 and a["res"] is not None) else None, dy,
                                                 a["slope"], gres, acc)
                            return
                        grads = [new_grad(i) for i in g["se"]["idx"]] if g["se"] is not None else [None] * 4
                        ops.se_gate_bwd(gout, a["y"].act, a["stats"], a["out"].act, a["slope"], P._se_args(g), g["pooled"],
                                        g["hidden"], g["gate"], g["mult"], g["dadd"], g["m12"], *grads,
                                        path_scale=g.get("scale_now"))
                        ops.instnorm_gate_act_bwd(gout, a["y"].act, a["stats"], a["out"].act, a["slope"], g["mult"], g["dadd"],
                                                  g["m12"], g["keep_x"], dy, gres, acc)
                        if g["se"] is not None:
                            for i in g["se"]["idx"]:
                                done(i)
                    if a["gate"] is not None:
                        if a["gate"]["se"] is not None:
                            order += a["gate"]["se"]["idx"]
                        b.append(gistep)
                        continue

                    # residual-block epilogue whose residual gradient starts here: the masked gradient g' is written ONCE (by the
                    # reduce pass, into gres) and the apply pass reads (g', y) only -- rx_instnorm_act_bwd_res, 7 tensor passes
                    # instead of 8; if the AvgPool of the next stage's skip path was the last writer of gout, its gradient is
                    # added on the fly and that pass (1R 1W over the full-resolution gradient) is gone too
                    pend = getattr(out, "_pool_pending", None)
                    out._pool_pending = None
                    fuse_res = (gres is not None and not acc and res is not None and y.act.voxels > 512
                                and os.environ.get("RX_FUSED_RES_BWD", "1") != "0")
                    if pend is not None and not fuse_res:        # the deferred pool gradient runs as its own pass after all
                        b.append(lambda pend=pend: ops.avgpool_bwd(pend["gy"], pend["gx"], pend["stride"], True))
                        pend = None
                    if fuse_res:
                        def rstep(a=a, gout=gout, dy=dy, gres=gres, pend=pend):
                            before_dy_write(dy)
                            ops.instnorm_act_bwd_res(gout, a["y"].act, a["stats"], a["out"].act, dy, gres, a["slope"],
                                                     pool_dy=pend["gy"] if pend is not None else None,
                                                     pool_stride=pend["stride"] if pend is not None else (1, 1, 1))
                        b.append(rstep)
                        continue

                    def istep(a=a, gout=gout, dy=dy, gres=gres, acc=acc):
                        before_dy_write(dy)
                        # the saved output is only needed for the mask of residual blocks (sign(out) != sign(xhat) there)
                        mask_out = a["out"].act if (a["slope"] != 1.0 and a["res"] is not None) else None
                        hs = a.get("head_src")
                        dl = P._dlogits.get(hs["name"]) if hs is not None else None   # None: task outside the loss, gout was zeroed
                        if dl is not None and a.get("head_dw_fused"):
                            dw, db = new_grad(hs["widx"]), new_grad(hs["bidx"])
                            ops.instnorm_act_bwd_head(dl, hs["w"].view(hs["k"], -1), a["y"].act, a["stats"], dy, a["slope"],
                                                      dw=dw.view(hs["k"], -1), db=db)
                            done(hs["widx"])
                            done(hs["bidx"])
                        elif dl is not None:
                            ops.instnorm_act_bwd_head(dl, hs["w"].view(hs["k"], -1), a["y"].act, a["stats"], dy, a["slope"])
                        elif a.get("m12_valid"):      # the two means came out of the backward-data kernel that completed gout
                            a["m12_valid"] = False
                            ops.instnorm_act_bwd_apply(gout, a["y"].act, a["stats"], mask_out, dy, a["m12"], a["slope"], gres, acc)
                        elif "inbwd_reduce" in skip and a["y"].act.voxels > 512:
                            if "m12x" not in a:
                                a["m12x"] = torch.zeros((self.B, a["y"].act.c, 2), dtype=torch.float32, device=self.device)
                            ops.instnorm_act_bwd_apply(gout, a["y"].act, a["stats"], mask_out, dy, a["m12x"], a["slope"], gres, acc)
                        else:
                            ops.instnorm_act_bwd(gout, a["y"].act, a["stats"], mask_out, dy, a["slope"], gres, acc)
                    b.append(istep)
                elif rec.kind in ("conv", "stem"):
                    y = a["y"]
                    dy = y.gact

                    def wstep(a=a, dy=dy, kind=rec.kind):
                        dw = new_grad(a["widx"])
                        db = new_grad(a["bidx"]) if a["bidx"] is not None else None

                        def launches(ws):
                            if "wgrad" in skip and kind != "stem":
                                pass
                            elif kind == "stem":
                                ops.stem_conv_bwd_weight(P._x, dy, dw, a["kernel"], ws)
                            else:
                                ops.conv3d_bwd_weight(a["x"].act, dy, dw, a["kernel"], a["stride"], ws)
                            done(a["widx"])
                            if db is not None:
                                ops.channel_sum(dy, db, ws)
                                done(a["bidx"])
                        side_run(launches, dy)
                    order += [a["widx"]] + ([a["bidx"]] if a["bidx"] is not None else [])
                    b.append(wstep)
                    if rec.kind == "conv" and a["x"].needs_grad:
                        x = a["x"]
                        gx = self._grad_buf(x)
                        acc = view_written(x)
                        x.written = True
                        if x.act.full_buffer:
                            mark_cat(x)     # a full-buffer write initialises every channel view of it
                        bsinfo = {"inact": None}
                        wrote(x, "conv", bsinfo)

                        def dstep(a=a, dy=dy, gx=gx, acc=acc, bsinfo=bsinfo):
                            if "dgrad" in skip:
                                return
                            ia = bsinfo["inact"]
                            if ia is None:
                                ops.conv3d_bwd_data(dy, a["pk"]["w_bwd"], gx, a["kernel"], a["stride"], acc)
                            else:
                                ia["m12_valid"] = ops.conv3d_bwd_data_instats(dy, a["pk"]["w_bwd"], gx, a["kernel"], a["stride"], acc,
                                                                              ia["y"].act, ia["stats"], ia["slope"], ia["m12"])
                        b.append(dstep)
                elif rec.kind == "convT":
                    x, y = a["x"], a["y"]
                    if not view_written(y):
                        raise RuntimeError("plan bug: concat gradient not initialised before the transposed conv")
                    gy = y.act.like(a["cat"].gact.root if a["cat"].gact.root is not None else a["cat"].gact.t)
                    gx = self._grad_buf(x)
                    acc = view_written(x)
                    x.written = True
                    wrote(x, "convT")

                    def step(a=a, gy=gy, gx=gx, acc=acc):
                        dw = new_grad(a["widx"])
                        db = new_grad(a["bidx"]) if a["bidx"] is not None else None

                        def launches(ws):
                            ops.convT3d_bwd_weight(a["x"].act, gy, dw, a["stride"], ws)
                            done(a["widx"])
                            if db is not None:
                                ops.channel_sum(gy, db, ws)
                                done(a["bidx"])
                        side_run(launches, gy)      # gy is a gradient buffer that is not rewritten in this backward
                        ops.convT3d_bwd_data(gy, a["pk"]["w_bwd"], gx, a["stride"], acc)
                    order += [a["widx"]] + ([a["bidx"]] if a["bidx"] is not None else [])
                    b.append(step)
                elif rec.kind == "pool":
                    x, y = a["x"], a["y"]
                    if not x.needs_grad:
                        continue
                    gx = self._grad_buf(x)
                    acc = view_written(x)
                    x.written = True
                    wrote(x, "pool")
                    assert y.gact is not None, "plan bug: pooled tensor has no gradient"
                    # the pool opens a stage's skip path; x is the previous stage's last block output, whose InstanceNorm backward
                    # is the very next step: hand the pool gradient to it instead of accumulating it in a pass of its own (above)
                    ti = next(i for i, r in enumerate(tape) if r is rec)
                    prev = tape[ti - 1] if ti > 0 else None
                    if (acc and prev is not None and prev.kind == "inact" and prev.a["out"] is x and prev.a["res"] is not None
                            and prev.a["gate"] is None and x.act.voxels > 512 and os.environ.get("RX_FUSED_POOL_BWD", "1") != "0"):
                        x._pool_pending = dict(gy=y.gact, gx=gx, stride=a["stride"])
                        continue
                    b.append(lambda a=a, gy=y.gact, gx=gx, acc=acc: ops.avgpool_bwd(gy, gx, a["stride"], acc))
                elif rec.kind == "copy":
                    x, y = a["x"], a["y"]
                    gy = y.act.like(a["cat"].gact.root if a["cat"].gact.root is not None else a["cat"].gact.t)
                    gx = self._grad_buf(x)
                    acc = view_written(x)
                    x.written = True
                    wrote(x, "copy")
                    b.append(lambda gy=gy, gx=gx, acc=acc: ops.avgpool_bwd(gy, gx, (1, 1, 1), acc))

    def _grad_buf(self, at: AT):
        """gradient buffer of an activation; channel views of one buffer share one gradient buffer"""
        if at.gact is not None:
            return at.gact
        base = at.act.root if at.act.root is not None else at.act.t
        if at.act.key not in self._gbase:
            t = torch.empty_like(base)
            self.bytes_alloc += t.numel() * t.element_size()
            self._gbase[at.act.key] = t
        at.gact = at.act.like(self._gbase[at.act.key])
        return at.gact

    # ------------------------------------------------------------------ run
    def _ensure_side(self):
        if self.device.type == "cuda" and self.overlap_wgrad and self._side is None:
            self._side = side_stream(self.device)
            self._ws2 = torch.empty(ops.workspace().numel(), dtype=torch.uint8, device=self.device)
        if self._ev_fork is None and self.device.type == "cuda":
            self._ev_fork, self._ev_join = ops.event_new(), ops.event_new()

    def _await_pack(self, ent):
        if ent.get("deferred") and self._pack_deferred is not None:
            self._issue_deferred_packs()
        slot = ent.get("event")
        if slot is not None:
            ops.stream_wait(slot)
            for e in ent["group"]:          # one event per pack group: the first consumer's wait covers them all (stream order)
                e["event"] = None

    def _packs_stale(self, force=False):
        """Staleness: tensor version / address AND the model's weight epoch.  torch's FUSED optimizers (fused=True
        Adam/AdamW/SGD: `_fused_adamw_` ...) update parameters WITHOUT bumping `Tensor._version`, so a version check alone
        would keep the initial packed weights for a whole training run.  Every backward pass through the engine
        therefore starts a new weight epoch (an optimizer step is what normally follows it) and all plans of the model
        re-pack on their next forward; pure inference (no backward) keeps its packs."""
        epoch = getattr(self.net, "_weights_epoch", 0)
        return [e for e in self.packs
                if force or e.get("epoch") != epoch
                or not (e["version"] == e["param"]._version and e.get("ptr") == e["param"].data_ptr())]

    def refresh_packs(self, force=False):
        """re-pack every parameter whose version moved (all of them with `force`: inside a captured graph).  The packs are pure HBM
        traffic (1.7 GB at cfg2) while the first stages of the forward pass are MFMA/LDS bound: they run on the side stream in
        first-use order, in a few table launches, and each consumer conv waits for its group's event."""
        stale = self._packs_stale(force)
        if not stale:
            return
        side = None
        if self.device.type == "cuda" and self.overlap_wgrad:
            self._ensure_side()
            side = self._side
            ops.event_record(self._ev_fork)                    # the optimizer's writes are on the main stream
            ops.stream_wait(self._ev_fork, side)
        self._pack_entries(stale, side, defer=True)

    def _issue_deferred_packs(self):
        """(from inside the forward list) the pack groups that refresh_packs held back: ordered after the optimizer's writes by the
        fork event recorded there (the side stream executes in order), and before their first consumer by position in the list"""
        d, self._pack_deferred = self._pack_deferred, None
        if d is not None:
            groups, side = d
            ops.event_record(self._ev_fork)                    # main stream: everything up to here
            ops.stream_wait(self._ev_fork, side)
            self._pack_groups_now(groups, side, 1)

    def _pack_entries(self, entries, side, defer=False):
        """re-pack `entries` (in first-use order) on `side` (or the current stream).  Table launches (rx_pack_multi, 40 tensors
        each) in GROUPS of ~RX_PACK_GROUP_MB of parameters with one numbered event per group: the first convs of the forward
        wait for the first (small) group only, the 512-channel stages' packing runs under the stages before them.  Was: one
        launch + one event per tensor, 66 launches of 17 us on average per cfg2 step."""
        limit = int(float(os.environ.get("RX_PACK_GROUP_MB", "96")) * (1 << 20))
        groups, cur, cur_bytes = [], [], 0
        for ent in entries:
            nb = ent["param"].numel() * 4
            if cur and cur_bytes + nb > limit and len(groups) < 3:     # at most 4 groups: [first ~96 MB] [next] [next] [rest]
                groups.append(cur)
                cur, cur_bytes = [], 0
            cur.append(ent)
            cur_bytes += nb
        if cur:
            groups.append(cur)
        # RX_PACK_DELAY (default on): only the first group is packed now; the others (the 512-channel stages: 80 % of the bytes) are
        # issued from inside the forward list, behind the full-resolution stage (see _forward_body) -- the packing then competes
        # with the 64^3 / 32^3 convolutions instead of the HBM-bound full-resolution InstanceNorm passes
        self._pack_deferred = None
        if (defer and side is not None and len(groups) > 1 and self._pack_delay_at is not None
                and os.environ.get("RX_PACK_DELAY", "1") != "0"):
            self._pack_deferred = (groups[1:], side)
            for grp in groups[1:]:
                for ent in grp:
                    ent["deferred"] = True        # a consumer that comes earlier than expected issues them itself (_await_pack)
            groups = groups[:1]
        self._pack_groups_now(groups, side, 0)

    def _pack_groups_now(self, groups, side, gi0):
        ctx = torch.cuda.stream(side) if side is not None else _NullCtx()
        with ctx:
            for gi, grp in enumerate(groups, gi0):
                items = []
                for ent in grp:
                    w = ent["param"].detach()
                    if w.dtype != torch.float32 or not w.is_contiguous():
                        w = w.float().contiguous()
                        self._unstable_params = True        # a temporary: its address is not replayable
                    if self.two_d:
                        w = w.unsqueeze(2)
                    items.append((w, 0 if ent["kind"] == "conv" else 1, ent["w_fwd"], ent["w_bwd"]))
                ops.pack_weights_multi(items, self.dtype)
                slot = None
                if side is not None:
                    while len(self._pack_slots) <= gi:
                        self._pack_slots.append(ops.event_new())
                    slot = self._pack_slots[gi]
                    ops.event_record(slot)
                for ent in grp:
                    p = ent["param"]
                    ent["deferred"] = False
                    ent["event"], ent["group"] = slot, grp
                    ent["version"], ent["ptr"] = p._version, p.data_ptr()
                    ent["epoch"] = getattr(self.net, "_weights_epoch", 0)

    def repack(self, entries):
        """re-pack `entries` on the side stream NOW (called from inside a `torch.cuda.stream(side)` block by the streamed
        optimizer step, right after the update of those parameters) and mark them fresh"""
        self._pack_entries(entries, self._side)

    # ---- launch programs -------------------------------------------------------------------------------------
    def _programs_on(self):
        return (self.use_programs and not self.use_graphs and self.device.type == "cuda" and ops._PROF is None
                and not getattr(self, "_unstable_params", False))

    def _streams(self):
        main = torch.cuda.current_stream()
        return [main] + ([self._side] if self._side is not None else [])

    def _programmed(self, key, body, segments=None):
        """run `body` eagerly twice (lazy allocations, workspace growth, first-use attributes), then once more under the
        library's recorder, and replay the recorded program ever after.  `segments(prog, streams)` (optional) replays it
        piecewise (the gradient synchroniser's bucket launches sit between segments).  Returns (state, replayed?)."""
        st = self._pstate.setdefault(key, {"calls": 0})
        ptrs = self._param_ptrs()
        if st.get("prog") is not None and st["ptrs"] != ptrs:       # parameters were re-allocated: start over
            st.clear()
            st["calls"] = 0
        if st.get("prog") is not None:
            streams = self._streams()
            if segments is not None:
                segments(st, streams)
            else:
                st["prog"].run(streams)
            return st, True
        if st["calls"] < 2:
            st["calls"] += 1
            st["result"] = body()
            return st, False
        prog = ops.Program()
        self._ws_refs += [ops.workspace(), self._ws2]     # the recorded scratch pointers stay alive with the plan
        self._marks = []
        self._recording = prog
        self._on_side = False
        prog.begin(self._streams())
        try:
            st["result"] = body()
        finally:
            self._recording = None
            prog.end()
        st["prog"], st["ptrs"], st["marks"] = prog, ptrs, list(self._marks)
        return st, False

    def _grad_store(self):
        """one flat fp32 buffer for every parameter gradient, in readiness order: a recorded backward writes the SAME addresses
        every step.  autograd gets fresh tensor objects over it and normally adopts them as `.grad`; a `.grad` the caller kept
        (gradient accumulation) is moved out of the way before the next backward overwrites the storage."""
        if self._gflat is None:
            offs, n = {}, 0
            for idx in self.grad_order:
                if idx not in offs:
                    offs[idx] = n
                    n += (self.params[idx].numel() + 63) // 64 * 64
            self._gflat = torch.empty(max(n, 1), dtype=torch.float32, device=self.device)
            self.bytes_alloc += self._gflat.numel() * 4
            lo, hi = self._gflat.data_ptr(), self._gflat.data_ptr() + self._gflat.numel() * 4
            self._gflat_range = (lo, hi)
            for idx, o in offs.items():
                p = self.params[idx]
                self._gviews[idx] = self._gflat[o:o + p.numel()].view(p.shape)
        lo, hi = self._gflat_range
        for p in self.params:                      # gradient accumulation: a kept .grad must not alias what we overwrite now
            gr = p.grad
            if gr is not None and lo <= gr.data_ptr() < hi:
                p.grad = gr.clone()

    # ---- HIP graph plumbing ---------------------------------------------------------------------------------
    def _graphs_on(self):
        # (needs_grad plans only: an inference plan re-packs nothing and has no backward list worth capturing)
        return (self.use_graphs and self.device.type == "cuda" and self.needs_grad
                and self.grad_sync is None and ops._PROF is None and not self._drops)

    def _param_ptrs(self):
        return tuple(p.data_ptr() for p in self.params)

    def _graphed(self, key, body):
        """run `body` eagerly twice, then capture it once and replay it ever after.  Returns the state dict."""
        st = self._gstate.setdefault(key, {"calls": 0})
        ptrs = self._param_ptrs()
        if st.get("graph") is not None and st["ptrs"] != ptrs:      # parameters were re-allocated: start over
            st.clear()
            st["calls"] = 0
        if st.get("graph") is not None:
            st["graph"].replay()
        elif st["calls"] < 2:
            st["calls"] += 1
            st["result"] = body()
            st["eager"] = True
            return st
        else:
            torch.cuda.synchronize(self.device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                st["result"] = body()
            st["graph"], st["ptrs"] = g, ptrs
            g.replay()
        st["eager"] = False
        return st

    def _forward_pre(self):
        """host-side work of a forward that is not a library call (never part of a recorded program)"""
        ev = getattr(self.net, "_raw_param_event", None)   # stem / bias / head parameters are read raw by the kernels:
        if ev is not None and ev is not self._raw_seen:     # a streamed optimizer step updates them on the side stream
            torch.cuda.current_stream().wait_event(ev)
            self._raw_seen = ev
        for g in self._gates:                               # DropPath: this step's per-sample factors (torch RNG)
            g["scale_now"] = self._draw_path_scale(g)
        if self._shadows:                                   # padded channel extents: zero-padded copies of the parameters
            self._refresh_shadows()
        training = bool(self.net.training)
        for d in self._drops:                               # channel dropout: this step's kept planes, in forward order
            d["active"] = training
            d["eps_now"] = d["eps"] * (1.0 - d["p"]) ** 2 if training else d["eps"]
            if training:
                self._draw_dropout(d)

    def _forward_body(self, force_packs):
        self.refresh_packs(force=force_packs)
        for i, step in enumerate(self.fwd):
            if i == self._pack_delay_at:
                self._issue_deferred_packs()
            step()
        self._issue_deferred_packs()
        for ent in self.packs:          # parameters of unused branches: never leave a pack in flight
            self._await_pack(ent)

    def _mark_packs_fresh(self):
        epoch = getattr(self.net, "_weights_epoch", 0)
        for ent in self.packs:
            ent["version"], ent["ptr"], ent["event"] = ent["param"]._version, ent["param"].data_ptr(), None
            ent["epoch"] = epoch

    def run_forward(self, x, apply_act):
        if tuple(x.shape) != self.in_shape:
            raise ValueError(f"plan built for input {self.in_shape}, got {tuple(x.shape)}")
        x = x.detach()
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.float().contiguous()
        self._apply_act = apply_act
        self._forward_pre()
        if self._graphs_on():
            if self._x_static is None:
                self._x_static = torch.empty(self.in_shape, dtype=torch.float32, device=self.device)
            self._x_static.copy_(x)                 # the graph reads a fixed address
            self._x = self._x_static.unsqueeze(2) if self.two_d else self._x_static
            self._fill_image_t()
            st = self._gstate.get(("f", apply_act))
            capturing = st is not None and st.get("graph") is None and st["calls"] >= 2
            self._graphed(("f", apply_act), lambda: self._forward_body(force_packs=capturing))
            st = self._gstate[("f", apply_act)]
            if not st["eager"]:                       # the graph re-packed every parameter
                self._mark_packs_fresh()
        elif self._programs_on():
            self._ensure_side()
            if self._x_static is None:
                self._x_static = torch.empty(self.in_shape, dtype=torch.float32, device=self.device)
            self._x_static.copy_(x)                 # the program reads a fixed address
            self._x = self._x_static.unsqueeze(2) if self.two_d else self._x_static
            self._fill_image_t()
            stale = bool(self._packs_stale())
            dropping = any(g["scale_now"] is not None for g in self._gates) or any(d["active"] for d in self._drops)
            # one program per launch-list VARIANT: with / without the weight re-pack, with / without DropPath factors
            key = ("f", apply_act, stale, dropping, self.overlap_wgrad)
            _, replayed = self._programmed(key, lambda: self._forward_body(force_packs=stale))
            if replayed and stale:
                self._mark_packs_fresh()
        else:
            self._x = x.unsqueeze(2) if self.two_d else x
            self._fill_image_t()
            self._forward_body(force_packs=False)
        self.generation += 1
        outs = {}
        for k, v in self.outputs.items():
            outs[k] = v.squeeze(2) if self.two_d else v
        return outs

    def _fill_image_t(self):
        if self._image_t is not None:
            self._image_t.act.t[..., :self.Cin].copy_(self._x.permute(0, 2, 3, 4, 1))

    def run_encoder(self, x):
        """the encoder part of the forward list alone (eager launches): the per-stage outputs (`Encoder.forward`'s skips) as NCDHW
        fp32 tensors.  Stage outputs that live inside a decoder's concat buffer are read through their channel view."""
        if tuple(x.shape) != self.in_shape:
            raise ValueError(f"plan built for input {self.in_shape}, got {tuple(x.shape)}")
        x = x.detach()
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.float().contiguous()
        self._apply_act = False
        self._forward_pre()
        self._x = x.unsqueeze(2) if self.two_d else x
        self._fill_image_t()
        self.refresh_packs(force=False)
        n = self.n_fwd_enc if self.n_fwd_enc is not None else len(self.fwd)
        for step in self.fwd[:n]:
            step()
        self._issue_deferred_packs()
        for ent in self.packs:
            self._await_pack(ent)
        outs = []
        for at, c in self.enc_skips:
            t = at.act.to_ncdhw()[:, :c].float()
            outs.append(t.squeeze(2) if self.two_d else t)
        self.generation += 1
        return outs

    def run_decoder(self, d, name, skips):
        """one task decoder alone (eager launches) on caller-supplied encoder outputs: `skips` = the list `run_encoder` returns (NCDHW,
        one tensor per stage, real channel counts).  They are written into the plan's stage-output buffers (channels-last compute
        type, padded channels zero), then the decoder's part of the forward list runs; returns the raw logits (decoder.py:137-162)."""
        if len(skips) != len(self.enc_skips):
            raise ValueError(f"expected {len(self.enc_skips)} encoder outputs, got {len(skips)}")
        self._apply_act = False
        self._forward_pre()
        self.refresh_packs(force=False)
        self._issue_deferred_packs()
        for ent in self.packs:
            self._await_pack(ent)
        for (at, c), t in zip(self.enc_skips, skips):
            t = t.detach()
            if self.two_d:
                t = t.unsqueeze(2)
            want = (self.B, c, *at.act.dims[1:4])
            if tuple(t.shape) != want:
                raise ValueError(f"encoder output of shape {tuple(t.shape)} where the plan holds {want}")
            act = at.act                     # (a channel view of its buffer: stage outputs may live inside a concat)
            act.t[..., act.c0:act.c0 + c].copy_(t.permute(0, 2, 3, 4, 1))
            if act.c > c:
                act.t[..., act.c0 + c:act.c0 + act.c].zero_()
        a = self.fwd_dec_start[d]
        b = self.fwd_dec_start[d + 1] if d + 1 < len(self.fwd_dec_start) else len(self.fwd)
        for step in self.fwd[a:b]:
            step()
        self.generation += 1
        v = self.outputs[name]
        return (v.squeeze(2) if self.two_d else v).clone()

    def _backward_body(self):
        self._dy_free.clear()       # the previous backward ended with the side stream joined: nothing is still read
        for step in self.bwd:
            step()
        if self._side is not None:
            ops.event_record(self._ev_join, self._side)     # every weight gradient is complete
            ops.stream_wait(self._ev_join)

    def _backward_finish(self):
        grads = self._grads
        if self._pad_idx:
            # padded parameters: slice the real gradient out of the padded one (main stream, after the join of the side stream)
            sync = self.grad_sync
            for idx in sorted(set(self._pad_done)):
                ent = self._pad_idx[idx]
                q = ent["param"]
                if sync is not None:
                    real = sync.alloc(idx)
                elif self._use_gstore:
                    real = self._gviews[idx]
                else:
                    real = torch.empty_like(q, memory_format=torch.contiguous_format)
                for dst, sidx in ent["pairs"]:
                    real[sidx].copy_(ent["gpad"][dst])
                grads[idx] = real
                if sync is not None:
                    sync.ready(idx)
            self._pad_done = []
        if self.grad_sync is not None:
            self.grad_sync.finish()
            if not self.grad_sync.returns_grads:    # a local micro-batch of an accumulation window: the synchroniser carries
                grads = [None] * len(self.params)   # the gradients until the stepping micro-batch (engine/ddp.py::no_sync)
        self._grads = []
        return grads

    def _replay_backward_segments(self, st, streams):
        """replay a recorded backward with the gradient synchroniser's hooks at the recorded positions: `ready(idx)` right
        after the command that completed gradient idx, announced from the stream that produced it"""
        prog, sync, pos = st["prog"], self.grad_sync, 0
        for cmd, idx, on_side in st["marks"]:
            if cmd > pos:
                prog.run(streams, pos, cmd)
                pos = cmd
            self._grads[idx] = sync.alloc(idx)
            if on_side and self._side is not None:
                with torch.cuda.stream(self._side):
                    sync.ready(idx)
            else:
                sync.ready(idx)
        prog.run(streams, pos, -1)

    def run_backward(self, dlogits: Dict[str, Optional[torch.Tensor]]):
        graphs = self._graphs_on()
        # programs: every task must take part in the loss (the zero-fill path of an absent task is torch ops, not library calls)
        programs = (not graphs) and self._programs_on() and all(dlogits.get(k) is not None for k in self.outputs)
        self._dlogits = {}
        for k, g in dlogits.items():
            if g is None:
                continue
            g = g.detach()
            if g.dtype != torch.float32 or not g.is_contiguous():
                g = g.float().contiguous()
            if graphs or programs:
                buf = self._dl_static.get(k)
                if buf is None or buf.shape != g.shape:
                    buf = self._dl_static[k] = torch.empty_like(g)
                buf.copy_(g)
                g = buf
            self._dlogits[k] = g.unsqueeze(2) if self.two_d else g
        self._ensure_side()
        # a backward is what an optimizer step follows: new weight epoch for every plan of this model (see refresh_packs)
        self.net._weights_epoch = getattr(self.net, "_weights_epoch", 0) + 1
        self._grads = [None] * len(self.params)
        if programs:
            sync = self.grad_sync
            self._use_gstore = sync is None
            if sync is None:
                self._grad_store()
            else:
                sync.persistent = True          # bucket storage is reused step after step (recorded pointers)
                sync.begin(self)
            key = ("b", tuple(sorted(self._dlogits)), id(sync) if sync is not None else 0, self.overlap_wgrad)
            st, replayed = self._programmed(key, self._backward_body,
                                            segments=self._replay_backward_segments if sync is not None else None)
            if replayed:
                self._pad_done = [idx for idx in set(self.grad_order) if idx in self._pad_idx]
            if replayed and sync is None:
                for idx in set(self.grad_order):
                    self._grads[idx] = self._gviews[idx]
            grads = self._backward_finish()
            # fresh tensor objects over the persistent storage (autograd adopts a gradient only if nobody else holds it)
            return [t.detach() if t is not None else None for t in grads]
        self._use_gstore = False
        if self.grad_sync is not None:
            self.grad_sync.persistent = False
            self.grad_sync.begin(self)
        if not graphs:
            self._backward_body()
            return self._backward_finish()
        key = ("b", tuple(sorted(self._dlogits)))
        st = self._gstate.get(key)
        if st is not None and st.get("graph") is not None:
            # the graph rewrites the SAME gradient storage every replay.  autograd normally steals the tensors we
            # return as .grad; if the caller kept such a .grad (gradient accumulation) move it out of the way first.
            owned = st["owned"]
            for p in self.params:
                gr = p.grad
                if gr is not None and gr.data_ptr() in owned:
                    p.grad = gr.clone()

        def body():
            self._backward_body()
            return self._backward_finish()
        st = self._graphed(key, body)
        grads = st["result"]
        if st["eager"]:
            return grads
        if "owned" not in st:
            st["owned"] = {t.data_ptr() for t in grads if t is not None}
        # fresh tensor objects over the graph-owned storage (autograd steals a gradient only if nobody else holds it)
        return [t.detach() if t is not None else None for t in grads]
