"""ctypes binding of librxunet.so (C ABI: include/rxunet.h).  Fails loudly when the library is
missing -- there is no fallback path."""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_long, c_size_t, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# RX_LIBRARY: an alternative build of the SAME library (kernel A/B runs on one box; device clocks differ box to box)
LIB_PATH = os.environ.get("RX_LIBRARY") or os.path.join(os.path.dirname(_HERE), "csrc", "librxunet.so")

RX_F32, RX_BF16, RX_F16 = 0, 1, 2
RX_ACT_NONE, RX_ACT_SIGMOID, RX_ACT_SOFTMAX = 0, 1, 2
DTYPE_CODE = {torch.float32: RX_F32, torch.bfloat16: RX_BF16, torch.float16: RX_F16}


class RxError(RuntimeError):
    pass


class RxAct(ctypes.Structure):
    """mirror of `rx_act` (include/rxunet.h)"""
    _fields_ = [("ptr", c_void_p), ("n", c_int32), ("z", c_int32), ("y", c_int32), ("x", c_int32),
                ("c", c_int32), ("ld", c_int32), ("cs", ctypes.c_int64)]


class RxSeParams(ctypes.Structure):
    """mirror of `rx_se_params` (include/rxunet.h)"""
    _fields_ = [("w1", c_void_p), ("b1", c_void_p), ("w2", c_void_p), ("b2", c_void_p), ("rd", c_int32),
                ("keep_x", c_int32)]


I3 = c_int32 * 3
_P = POINTER(RxAct)
_PSE = POINTER(RxSeParams)

_SIGNATURES = {
    "rx_abi_version": (c_int, []),
    "rx_last_error": (c_char_p, []),
    "rx_device_arch_ok": (c_int, []),
    "rx_last_conv_kernel": (c_char_p, []),
    "rx_pack_conv_weight": (c_int, [c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "rx_pack_convT_weight": (c_int, [c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "rx_pack_multi": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rx_conv_workspace_hint": (c_size_t, []),
    "rx_conv3d_fwd": (c_int, [c_int, _P, c_void_p, c_void_p, _P, I3, I3, c_void_p, c_size_t, c_void_p]),
    "rx_conv3d_fwd_stats": (c_int, [c_int, _P, c_void_p, c_void_p, _P, I3, I3, c_float, c_void_p, c_void_p, c_size_t, c_void_p]),
    "rx_conv3d_bwd_data": (c_int, [c_int, _P, c_void_p, _P, I3, I3, c_int, c_void_p, c_size_t, c_void_p]),
    "rx_conv3d_bwd_data_instats": (c_int, [c_int, _P, c_void_p, _P, I3, I3, c_int, _P, c_void_p, c_float, c_void_p, POINTER(c_int),
                                           c_void_p, c_size_t, c_void_p]),
    "rx_instnorm_act_bwd_apply": (c_int, [c_int, _P, _P, c_void_p, _P, c_float, c_void_p, _P, _P, c_int, c_void_p]),
    "rx_conv3d_bwd_weight_workspace": (c_size_t, [_P, _P, I3]),
    "rx_conv3d_bwd_weight": (c_int, [c_int, _P, _P, c_void_p, I3, I3, c_void_p, c_size_t, c_void_p]),
    "rx_convT3d_fwd": (c_int, [c_int, _P, c_void_p, c_void_p, _P, I3, c_void_p, c_size_t, c_void_p]),
    "rx_convT3d_bwd_data": (c_int, [c_int, _P, c_void_p, _P, I3, c_int, c_void_p, c_size_t, c_void_p]),
    "rx_convT3d_bwd_weight_workspace": (c_size_t, [_P, _P, I3]),
    "rx_convT3d_bwd_weight": (c_int, [c_int, _P, _P, c_void_p, I3, c_void_p, c_size_t, c_void_p]),
    "rx_instnorm_stats_workspace": (c_size_t, [_P]),
    "rx_instnorm_stats": (c_int, [c_int, _P, c_float, c_void_p, c_void_p, c_size_t, c_void_p]),
    "rx_instnorm_act_fwd": (c_int, [c_int, _P, c_void_p, _P, _P, c_float, c_void_p]),
    "rx_instnorm_stats_mask": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "rx_instnorm_fwd": (c_int, [c_int, _P, c_float, c_void_p, _P, _P, c_float, c_void_p, c_size_t, c_void_p]),
    "rx_instnorm_act_bwd": (c_int, [c_int, _P, _P, c_void_p, _P, c_float, _P, _P, c_int, c_void_p, c_size_t,
                                    c_void_p]),
    "rx_instnorm_act_bwd_res": (c_int, [c_int, _P, _P, c_void_p, _P, c_float, _P, I3, _P, _P, c_void_p, c_size_t, c_void_p]),
    "rx_se_workspace": (c_size_t, [_P]),
    "rx_se_gate_fwd": (c_int, [c_int, _P, c_void_p, c_void_p, _PSE, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                               c_void_p]),
    "rx_instnorm_gate_act_fwd": (c_int, [c_int, _P, c_void_p, c_void_p, c_int, _P, _P, c_float, c_void_p]),
    "rx_se_gate_bwd": (c_int, [c_int, _P, _P, c_void_p, _P, c_float, c_void_p, _PSE, c_void_p, c_void_p, c_void_p, c_void_p,
                               c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "rx_instnorm_gate_act_bwd": (c_int, [c_int, _P, _P, c_void_p, _P, c_float, c_void_p, c_void_p, c_void_p, c_int, _P, _P,
                                         c_int, c_void_p]),
    "rx_avgpool_fwd": (c_int, [c_int, _P, _P, I3, c_void_p]),
    "rx_instnorm_act_pool_fwd": (c_int, [c_int, _P, c_void_p, _P, _P, _P, I3, c_float, c_void_p]),
    "rx_avgpool_bwd": (c_int, [c_int, _P, _P, I3, c_int, c_void_p]),
    "rx_stem_conv_fwd": (c_int, [c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, _P, I3,
                                 c_void_p]),
    "rx_stem_conv_fwd_stats": (c_int, [c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, _P, I3,
                                       c_float, c_void_p, c_void_p, c_size_t, c_void_p]),
    "rx_stem_conv_bwd_weight_workspace": (c_size_t, [c_int, c_int, c_int]),
    "rx_stem_conv_bwd_weight": (c_int, [c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, _P, c_void_p, I3,
                                        c_void_p, c_size_t, c_void_p]),
    "rx_head_fwd": (c_int, [c_int, _P, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "rx_head_bwd_workspace": (c_size_t, [_P, c_int]),
    "rx_instnorm_act_head_fwd": (c_int, [c_int, _P, c_void_p, _P, c_float, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "rx_instnorm_act_bwd_head": (c_int, [c_int, c_void_p, c_int, c_void_p, _P, c_void_p, c_float, _P, c_void_p, c_void_p, c_void_p, c_size_t,
                                         c_void_p]),
    "rx_grad_norm_clip_partials": (ctypes.c_long, [c_int, c_void_p]),
    "rx_grad_norm_clip": (c_int, [c_int, c_void_p, c_void_p, c_float, c_void_p, ctypes.c_long, c_void_p, c_void_p]),
    "rx_head_bwd": (c_int, [c_int, c_void_p, _P, c_void_p, c_int, _P, c_void_p, c_void_p, c_void_p, c_size_t,
                            c_void_p]),
    "rx_channel_sum_workspace": (c_size_t, [_P]),
    "rx_channel_sum": (c_int, [c_int, _P, c_void_p, c_void_p, c_size_t, c_void_p]),
    "rx_loss_workspace": (c_size_t, [c_int, c_int, c_long]),
    "rx_bce_dice_loss_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_long, c_float, c_float, c_float, c_float, c_void_p,
                                     c_void_p, c_void_p, c_size_t, c_void_p]),
    "rx_bce_dice_loss_bwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_long, c_float, c_float, c_float, c_void_p, c_void_p,
                                     c_void_p, c_void_p]),
    "rx_masked_cosine_loss_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_long, c_void_p, c_void_p, c_void_p, c_size_t,
                                          c_void_p]),
    "rx_masked_cosine_loss_bwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_long, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rx_adamw_pack": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_double, c_double, c_double, c_double,
                              c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "rx_adamw_flat_multi": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_double, c_double,
                                    c_double, c_double, c_int, c_void_p]),
    "rx_prog_create": (c_void_p, []),
    "rx_prog_destroy": (None, [c_void_p]),
    "rx_prog_begin": (c_int, [c_void_p, c_void_p, c_int]),
    "rx_prog_end": (c_int, [c_void_p]),
    "rx_prog_len": (c_int, [c_void_p]),
    "rx_prog_cmd_name": (c_char_p, [c_void_p, c_int]),
    "rx_prog_cmd_kernel": (c_char_p, [c_void_p, c_int]),
    "rx_prog_cmd_stream": (c_int, [c_void_p, c_int]),
    "rx_prog_run": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p]),
    "rx_event_new": (c_int, []),
    "rx_event_free": (c_int, [c_int]),
    "rx_event_slots_in_use": (c_int, []),
    "rx_event_record": (c_int, [c_int, c_void_p]),
    "rx_stream_wait": (c_int, [c_int, c_void_p]),
    "rx_adamw_flat": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_double, c_double, c_double, c_double, c_int,
                              c_long, c_void_p]),
}

_lib = None


def load():
    """Returns the loaded library; raises RxError (never falls back) if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RxError(f"{LIB_PATH} not found: build it with `python __graft_entry__.py` "
                      "(hipcc --offload-arch=gfx950); this engine has no PyTorch/CPU fallback")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here == ABI drift, also loud
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def exported_symbols():
    return sorted(_SIGNATURES)


def check(rc, what=""):
    if rc != 0:
        msg = load().rx_last_error().decode("utf-8", "replace")
        raise RxError(f"{what} failed with status {rc}: {msg}")


def require_device():
    """The compute entry points only exist for gfx950; anything else is an error, not a fallback."""
    if not torch.cuda.is_available():
        raise RxError("no HIP device visible: the rxunet engine runs only on an MI355X (gfx950)")
    if not load().rx_device_arch_ok():
        raise RxError("current HIP device is not gfx950")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def stream_ptr():
    """hipStream_t of torch's current stream.  The raw getter skips building a torch.cuda.Stream object (~5 us, on
    ~2000 enqueues per train step)."""
    if _raw_stream is not None and _cur_device is not None:
        return c_void_p(_raw_stream(_cur_device()))
    return c_void_p(torch.cuda.current_stream().cuda_stream)
