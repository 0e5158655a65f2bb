"""ctypes binding of libscade_hip.so (include/scade_hip.h).

There is deliberately NO fallback: if the library is missing or a call fails the
product path raises.  PyTorch only provides device memory and the stream.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_long, c_void_p
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SCADE_LIB") or os.path.join(_HERE, "lib", "libscade_hip.so")  # SCADE_LIB: kernel experiments

_P = c_void_p
_I = c_int

# name -> (restype, argtypes); must list every symbol declared in include/scade_hip.h
SIGNATURES = {
    "scade_version": (c_int, []),
    "scade_last_error": (c_char_p, []),
    "scade_mlp_packed_floats": (c_long, []),
    "scade_mlp_lds_bytes": (c_int, []),
    "scade_mlp_pack": (c_int, [_P, _P, _P]),
    "scade_mlp_pack_step": (c_int, [_I, _P, _I, _P, _P, _P]),
    "scade_mlp_pack_step_f16x3": (c_int, [_I, _P, _P, _P, _P, _P]),
    "scade_mlp_fwd": (c_int, [_P, _I, _P, _P, _I, _P, _I, _I, _P, _P, _P]),
    "scade_mlp_acts_floats": (c_long, [c_long]),
    "scade_mlp_packed_t_floats": (c_long, []),
    "scade_mlp_pack_t": (c_int, [_P, _P, _P]),
    "scade_mlp_bwd_workspace_floats": (c_long, [_I]),
    "scade_mlp_bwd_chunks": (c_int, [_I]),
    "scade_mlp_bwd": (c_int, [_P, _P, _P, _P, _I, _P, _P, _P]),
    "scade_mlp_bwd2_workspace_floats": (c_long, [_I, _I]),
    "scade_mlp_bwd2": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P]),
    "scade_mlp_bwd2_phases": (c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "scade_mlp_packed_f16_bytes": (c_long, []),
    "scade_mlp_pack_f16": (c_int, [_P, _P, _P]),
    "scade_mlp_fwd_f16": (c_int, [_P, _I, _P, _P, _I, _P, _I, _I, _P, _P, _P]),
    "scade_mlp_packed_lp_bytes": (c_long, []),
    "scade_mlp_pack_lp": (c_int, [_P, _P, _I, _P]),
    "scade_mlp_fwd_lp": (c_int, [_P, _I, _I, _P, _P, _I, _P, _I, _I, _P, _P, _P]),
    "scade_mlp_acts_lp_bytes": (c_long, [c_long]),
    "scade_mlp_packed_t_lp_bytes": (c_long, []),
    "scade_mlp_pack_t_lp": (c_int, [_P, _P, _I, _P]),
    "scade_mlp_bwd_lp_workspace_bytes": (c_long, [_I]),
    "scade_mlp_bwd_lp": (c_int, [_P, _P, _I, _P, _P, _I, _P, _P, _P]),
    "scade_mlp_lp_point_tiles": (c_int, [_I]),
    "scade_mlp_bwd_lp2_workspace_bytes": (c_long, [_I, _I]),
    "scade_mlp_bwd_lp2": (c_int, [_P, _I, _P, _P, _P, _P, _P, _P]),
    "scade_mlp_bwd_lp2_phases": (c_int, [_P, _I, _P, _P, _P, _P, _P, _I, _P]),
    "scade_mlp_packed_t_f16_bytes": (c_long, []),
    "scade_mlp_pack_t_f16": (c_int, [_P, _P, _P]),
    "scade_mlp_bwd_f16": (c_int, [_P, _P, _P, _P, _I, _I, _P, _P, _P]),
    "scade_mlp_bwd_f16_2": (c_int, [_P, _P, _P, _P, _P, _I, _P, _P, _P]),
    "scade_embed": (c_int, [_P, _I, _I, _I, _P, _P]),
    "scade_ray_points": (c_int, [_P, _I, _P, _P, _I, _I, _I, _P, _P, _P]),
    "scade_ray_points_draw": (c_int, [_P, _I, _P, _I, _I, _I, ctypes.c_ulonglong, ctypes.c_ulonglong, _P, _I, _P, _P, _P, _P,
                                      _P]),
    "scade_perturb_z": (c_int, [_P, _P, _I, _I, _P, _P]),
    "scade_composite_fwd": (c_int, [_P, _P, _P, _I, _P, _I, _I, _P, _P, _P, _P, _P, _P]),
    "scade_composite_bwd": (c_int, [_P, _P, _P, _I, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "scade_sample_pdf_fwd": (c_int, [_P, _I, _I, _P, _I, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "scade_sample_pdf_bwd": (c_int, [_P, _I, _I, _P, _I, _P, _I, _P, _I, _I, _I, _P, _P]),
    "scade_merge_sorted": (c_int, [_P, _I, _P, _I, _P, _I, _I, _P, _P, _P]),
    "scade_ray_tail": (c_int, [_P, _P, _P, _I, _P, _I, _I, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "scade_ray_tail_bwd": (c_int, [_P, _P, _P, _I, _P, _I, _I, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "scade_carve_workspace_floats": (c_long, [_I, _I, _I, _I]),
    "scade_carve_fwd": (c_int, [_P, _P, _P, c_float, _I, _I, _I, _I, _P, _P, _P]),
    "scade_carve_bwd": (c_int, [_P, _P, _P, c_float, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "scade_carve_joint_colmean": (c_int, [_P, _P, _P, c_float, _I, _I, _I, _P, _P]),
    "scade_carve_joint_min": (c_int, [_P, _I, _I, _P, _P]),
    "scade_carve_knp_fwd": (c_int, [_P, _P, _P, c_float, _I, _I, _I, _I, _P, _P, _P]),
    "scade_carve_knp_bwd": (c_int, [_P, _P, _P, c_float, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "scade_carve_knp_joint_colmean": (c_int, [_P, _P, _P, c_float, _I, _I, _I, _P, _P]),
    "scade_mse_fwd": (c_int, [_P, _P, _P, _I, _I, _P, _P]),
    "scade_train_loss_fwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _I, _I, c_float, c_float, c_float,
                                     _I, _I, _I, _P, _P, _P]),
    "scade_train_loss_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _I, _I, c_float, c_float, c_float,
                                     _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "scade_train_loss_fb": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _I, _I, c_float, c_float, c_float,
                                    _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "scade_gen_rays": (c_int, [_P, _I, _I, _I, _P, _P, _I, c_float, c_float, _P, _P, _I, _I, _I, _P, _P, _P,
                                _P, _P, _P, _P]),
    "scade_gather_batch": (c_int, [_P, _I, _I, _I, _P, _P, _I, c_float, c_float, _P, _P, _I, _I, _I, _P, _P, _P, _P,
                                   _P, ctypes.c_longlong, _P, _P]),
    "scade_adam_step": (c_int, [_P, _P, _P, _P, c_long, c_float, c_float, c_float, c_float, _I, c_float, _P]),
    "scade_adam_step_dev": (c_int, [_P, _P, _P, _P, c_long, _P, _P]),
    "scade_adam_step2": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "scade_mse_bwd": (c_int, [_P, _P, _P, _I, _I, _P, _P, _P]),
    "scade_ray_tail_train_gmax": (c_int, [_P, _P, _P, _I, _I, _I, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P,
                                          _P, _P, _P, _P, _P, _P, _I, _P, _I, _I, c_float, c_float, c_float, _I, _P, _P, _P, _P, _I,
                                          _P, _P, _P, _I, _P, _P, _P, _P, _P]),
    "scade_stage_inputs_points": (c_int, [_P, _P, _P, _I, _P, ctypes.c_longlong, _P, _P, _I, _P, _I, _I, _I,
                                          ctypes.c_ulonglong, ctypes.c_ulonglong, _I, _P, _P, _P, _P,
                                          _I, _I, _P, _P, _P, _P, _P]),
    "scade_gather_batch_points": (c_int, [_P, _I, _I, _I, _P, _P, _I, c_float, c_float, _P, _P, _I, _I, _I, _P, _P, _P, _P,
                                          _P, ctypes.c_longlong, _P, _P, _I, _I, ctypes.c_ulonglong, ctypes.c_ulonglong,
                                          _I, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P]),
    "scade_mlp_bwd2_deferred": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P]),
    "scade_mlp_bwd_lp2_deferred": (c_int, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "scade_mlp_bwd_f16_2_deferred": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P]),
    "scade_step_finish": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "scade_mlp_wgrad_lp_plan": (c_int, [_P, _I, _P, _P, _P, _P, _P, _P]),
    "scade_stage_inputs": (c_int, [_P, _P, _P, _I, _P, ctypes.c_longlong, _P, _P]),
    "scade_ray_tail_train": (c_int, [_P, _P, _P, _I, _P, _I, _I, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P,
                                     _P, _P, _P, _P, _P, _P, _I, _P, _I, _I, c_float, c_float, c_float, _I, _P, _P, _P, _P, _I,
                                     _P, _P, _P, _P, _I, _P, _P]),
}

_lib: Optional[ctypes.CDLL] = None


def load() -> ctypes.CDLL:
    """dlopen the in-tree library and type every entry point.  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"scade_amd: {LIB_PATH} not found. Build it with `python -m scade_amd.build` "
            "(hipcc, gfx950). There is no CPU/PyTorch fallback for this path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so is stale
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error() -> str:
    msg = load().scade_last_error()
    return msg.decode() if msg else ""


def ptr(t: Optional[torch.Tensor]):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else c_void_p(t.data_ptr())


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """hipStream_t of torch's current stream on the current device (the raw getter costs ~1 us; the
    Stream-object route ~10 us, which adds up over the ~20 launches of a render step)."""
    if _RAW_STREAM is not None:
        return c_void_p(_RAW_STREAM(torch.cuda.current_device()))
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def check_current_device(t: torch.Tensor, what: str) -> None:
    """Kernels are enqueued on the CURRENT device's current stream: a tensor that lives on another
    GPU of the process would be dereferenced on the wrong device.  Checked once per operator entry."""
    if t.is_cuda and t.device.index != torch.cuda.current_device():
        raise RuntimeError(f"{what}: tensor is on {t.device} but the current device is "
                           f"cuda:{torch.cuda.current_device()}; select it first "
                           "(torch.cuda.set_device / `with torch.cuda.device(...)`)")


def call(name: str, *args) -> None:
    rc = getattr(load(), name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed (code {rc}): {last_error()}")


def check(t: torch.Tensor, what: str, dtype=torch.float32) -> torch.Tensor:
    """Product-path guard: HIP device tensor of the right dtype."""
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{what}: expected a torch.Tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise RuntimeError(
            f"{what}: tensor is on {t.device}; scade_amd runs on the MI355X HIP device only "
            "(no CPU fallback)")
    if t.dtype != dtype:
        raise TypeError(f"{what}: expected {dtype}, got {t.dtype}")
    return t
