"""ctypes binding of libsedmi.so (the C ABI declared in include/sedmi.h).

There is no CPU fallback: if the shared library is missing or an entry point
fails, the caller gets an exception.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SEDMI_LIB_PATH") or os.path.join(_HERE, "libsedmi.so")

SEDMI_MEL_LOG = 1
SEDMI_MEL_NORMALISE = 2
CRNN_LAYERS = 7
PRECISION_F32 = 0
PRECISION_SPLIT_F16 = 1
SEDMI_ERECURRENCE = -5
RECURRENCE_AUTO = 0
RECURRENCE_PER_STEP = 1


class SedmiError(RuntimeError):
    pass


class SedmiRecurrenceError(SedmiError):
    """The single-launch GRU recurrence of a ShakeRCNN forward gave up (SEDMI_ERECURRENCE): that forward's
    logits are NaN."""


class CrnnWeights(C.Structure):
    _FP = C.POINTER(C.c_float)
    _fields_ = [
        ("n_class", C.c_int), ("n_freq", C.c_int),
        ("filters", C.c_int * CRNN_LAYERS), ("pool_t", C.c_int * CRNN_LAYERS),
        ("pool_f", C.c_int * CRNN_LAYERS), ("hidden", C.c_int), ("bn_eps", C.c_float),
        ("conv_w", _FP * CRNN_LAYERS), ("conv_b", _FP * CRNN_LAYERS),
        ("bn_w", _FP * CRNN_LAYERS), ("bn_b", _FP * CRNN_LAYERS),
        ("bn_mean", _FP * CRNN_LAYERS), ("bn_var", _FP * CRNN_LAYERS),
        ("glu_w", _FP * CRNN_LAYERS), ("glu_b", _FP * CRNN_LAYERS),
        ("gru_w_ih", (_FP * 2) * 2), ("gru_w_hh", (_FP * 2) * 2),
        ("gru_b_ih", (_FP * 2) * 2), ("gru_b_hh", (_FP * 2) * 2),
        ("dense_w", _FP), ("dense_b", _FP),
    ]


_FP3 = C.POINTER(C.c_float) * 3


class ShakeCnnWeights(C.Structure):
    _fields_ = [(n, _FP3) for n in ("conv_w", "conv_b", "bn1_w", "bn1_b", "bn1_mean", "bn1_var", "br1_w", "br1_b",
                                   "br2_w", "br2_b", "bn2_w", "bn2_b", "bn2_mean", "bn2_var")]


class ShakeRcnnWeights(C.Structure):
    _FP = C.POINTER(C.c_float)
    _fields_ = [("n_class", C.c_int), ("cnn", ShakeCnnWeights),
                ("gru_w_ih", _FP * 2), ("gru_w_hh", _FP * 2), ("gru_b_ih", _FP * 2), ("gru_b_hh", _FP * 2),
                ("ln_w", _FP), ("ln_b", _FP), ("fc_w", _FP), ("fc_b", _FP)]


SHAKE_MAX_LAYERS = 12


class ShakeEncoderLayerWeights(C.Structure):
    _fields_ = [(n, C.POINTER(C.c_float)) for n in ("in_proj_w", "in_proj_b", "out_proj_w", "out_proj_b", "lin1_w",
                                                    "lin1_b", "lin2_w", "lin2_b", "norm1_w", "norm1_b", "norm2_w",
                                                    "norm2_b")]


class ShakeTransformerWeights(C.Structure):
    _FP = C.POINTER(C.c_float)
    _fields_ = [("n_class", C.c_int), ("n_layers", C.c_int), ("cnn", ShakeCnnWeights), ("fc1_w", _FP), ("fc1_b", _FP),
                ("layers", ShakeEncoderLayerWeights * SHAKE_MAX_LAYERS), ("fc2_w", _FP), ("fc2_b", _FP)]


class BasicRcnnWeights(C.Structure):
    _FP = C.POINTER(C.c_float)
    _fields_ = [("n_class", C.c_int), ("frequency_dim", C.c_int), ("conv_w", _FP3), ("conv_b", _FP3),
                ("gru_w_ih", _FP), ("gru_w_hh", _FP), ("gru_b_ih", _FP), ("gru_b_hh", _FP), ("fc_w", _FP), ("fc_b", _FP)]


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_int64), ("ms", C.c_double),
                ("flops", C.c_double), ("bytes", C.c_double)]


# name -> (restype, argtypes); must list every symbol include/sedmi.h declares
_VP, _I64, _SZ, _F, _I = C.c_void_p, C.c_int64, C.c_size_t, C.c_float, C.c_int
SIGNATURES = {
    "sedmi_version": (_I, []),
    "sedmi_last_error": (C.c_char_p, []),
    "sedmi_device_count": (_I, []),
    "sedmi_mel_plan_create": (_I, [C.POINTER(_VP), _I, _I, _I, _I, _I]),
    "sedmi_mel_plan_destroy": (_I, [_VP]),
    "sedmi_mel_num_frames": (_I64, [_VP, _I64]),
    "sedmi_mel_filterbank": (_I, [_VP, _VP]),
    "sedmi_melspec_f32": (_I, [_VP, _VP, _I64, _I64, _I, _VP, _VP, _VP, _VP]),
    "sedmi_melspec_i16": (_I, [_VP, _VP, _I64, _I64, _I, _VP, _VP, _VP, _VP]),
    "sedmi_mel_stats_f64": (_I, [_VP, _I64, _I, _I64, _VP, _VP, _VP]),
    "sedmi_mel_apply_f32": (_I, [_VP, _I64, _I, _I64, _I, _VP, _VP, _VP]),
    "sedmi_crnn_create": (_I, [C.POINTER(_VP), C.POINTER(CrnnWeights)]),
    "sedmi_crnn_destroy": (_I, [_VP]),
    "sedmi_crnn_set_precision": (_I, [_VP, _I]),
    "sedmi_crnn_out_frames": (_I64, [_VP, _I64]),
    "sedmi_crnn_workspace_bytes": (_SZ, [_VP, _I64, _I64]),
    "sedmi_crnn_forward_f32": (_I, [_VP, _VP, _I64, _I64, _VP, _VP, _SZ, _VP]),
    "sedmi_crnn_cnn_f32": (_I, [_VP, _VP, _I64, _I64, _VP, _VP, _SZ, _VP]),
    "sedmi_crnn_gru_f32": (_I, [_VP, _VP, _I64, _I64, _VP, _VP, _SZ, _VP]),
    "sedmi_crnn_head_f32": (_I, [_VP, _VP, _I64, _VP, _VP]),
    "sedmi_shake_rcnn_create": (_I, [C.POINTER(_VP), C.POINTER(ShakeRcnnWeights)]),
    "sedmi_shake_rcnn_destroy": (_I, [_VP]),
    "sedmi_shake_rcnn_workspace_bytes": (_SZ, [_VP, _I64, _I64]),
    "sedmi_shake_rcnn_forward_f32": (_I, [_VP, _VP, _I64, _I64, C.POINTER(_F), _VP, _VP, _SZ, _VP]),
    "sedmi_shake_rcnn_set_recurrence": (_I, [_VP, _I]),
    "sedmi_shake_rcnn_check": (_I, [_VP, _VP, _SZ, _I64, _I64, _VP]),
    "sedmi_shake_cnn_f32": (_I, [_VP, _VP, _I64, _I64, C.POINTER(_F), _VP, _VP, _SZ, _VP]),
    "sedmi_shake_transformer_create": (_I, [C.POINTER(_VP), C.POINTER(ShakeTransformerWeights)]),
    "sedmi_shake_transformer_destroy": (_I, [_VP]),
    "sedmi_shake_transformer_workspace_bytes": (_SZ, [_VP, _I64, _I64]),
    "sedmi_shake_transformer_forward_f32": (_I, [_VP, _VP, _I64, _I64, C.POINTER(_F), _VP, _VP, _SZ, _VP]),
    "sedmi_basic_rcnn_create": (_I, [C.POINTER(_VP), C.POINTER(BasicRcnnWeights)]),
    "sedmi_basic_rcnn_destroy": (_I, [_VP]),
    "sedmi_basic_rcnn_workspace_bytes": (_SZ, [_VP, _I64, _I64]),
    "sedmi_basic_rcnn_forward_f32": (_I, [_VP, _VP, _I64, _I64, _VP, _VP, _SZ, _VP]),
    "sedmi_threshold_i64": (_I, [_VP, _I64, _F, _I, _VP, _VP]),
    "sedmi_decode_workspace_bytes": (_SZ, [_I64, _I]),
    "sedmi_decode_events": (_I, [_VP, _I64, _I, _VP, _VP, _I64, _F, _I, _I, _VP, _VP, _VP, _VP,
                                 _I64, _VP, _VP, _SZ, _VP]),
    "sedmi_decode_events_static": (_I, [_VP, _I64, _I, _VP, _VP, _I64, _F, _I, _I, _VP, _VP, _VP, _VP,
                                        _I64, _VP, _VP]),
    "sedmi_confusion_counts": (_I, [_VP, _VP, _I64, _I, _F, _I, _VP, _VP]),
    "sedmi_profile_enable": (_I, [_I]),
    "sedmi_profile_collect": (_I, [C.POINTER(KernelStat), _I]),
}

_lib = None


def lib() -> C.CDLL:
    """The loaded library; raises if it was never built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SedmiError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C sound_event_detection_amd/csrc`). There is no CPU fallback.")
        # torch first: its bundled HIP runtime must be the one this process uses, or the tensors'
        # device pointers and streams would belong to a different runtime instance than the kernels
        import torch  # noqa: F401
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)   # AttributeError if the symbol is missing
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(rc: int, what: str = "") -> int:
    if rc < 0:
        msg = lib().sedmi_last_error().decode("utf-8", "replace")
        cls = SedmiRecurrenceError if rc == SEDMI_ERECURRENCE else SedmiError
        raise cls(f"{what or 'sedmi call'} failed ({rc}): {msg}")
    return rc


def ptr(t) -> C.c_void_p:
    """Device (or host) address of a torch tensor / None."""
    return C.c_void_p(0 if t is None else t.data_ptr())


def stream_ptr(device) -> C.c_void_p:
    """The hipStream_t torch is currently enqueueing on for ``device``."""
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def profile_enable(on: bool) -> None:
    check(lib().sedmi_profile_enable(int(bool(on))))


def profile_collect():
    """[{name, launches, ms, flops, bytes}] summed per kernel since enable; synchronises."""
    buf = (KernelStat * 64)()
    n = check(lib().sedmi_profile_collect(buf, 64))
    return [dict(name=buf[i].name.decode(), launches=int(buf[i].launches), ms=float(buf[i].ms),
                 flops=float(buf[i].flops), bytes=float(buf[i].bytes)) for i in range(n)]
