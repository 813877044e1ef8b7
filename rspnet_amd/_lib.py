"""ctypes binding of librspnet_hip.so (include/rspnet_hip.h).  No torch types cross this boundary.

This is the binding a maintainer of the reference would add (INTEGRATION.md shows the same stub): the
reference has no FFI of its own — its ops are ATen calls — so the Python side passes raw device pointers,
sizes and the current HIP stream.  Importing this module never falls back to anything: if the shared
library is missing, ``load()`` raises and every op in ``rspnet_amd.ops`` fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RSPNET_HIP_LIB") or os.path.join(_HERE, "librspnet_hip.so")   # override: tuning builds only

c_f32p = C.c_void_p
c_i32p = C.c_void_p
c_stream = C.c_void_p


class ConvDesc(C.Structure):
    """rsp_conv3d_desc"""
    _fields_ = [(n, C.c_int32) for n in (
        "N", "Di", "Hi", "Wi", "Cin", "Do", "Ho", "Wo", "Cout", "kT", "kH", "kW", "sT", "sH", "sW",
        "pT", "pH", "pW", "in_ld", "out_ld")]


class PoolDesc(C.Structure):
    """rsp_pool3d_desc"""
    _fields_ = [(n, C.c_int32) for n in (
        "N", "Di", "Hi", "Wi", "C", "Do", "Ho", "Wo", "kT", "kH", "kW", "sT", "sH", "sW", "pT", "pH", "pW",
        "in_ld", "out_ld", "res_ld")]


class PackJob(C.Structure):
    """rsp_pack_job (112 bytes)"""
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("total", C.c_int64)] + [(n, C.c_int32) for n in (
        "kind", "Cout_src", "Cin_src", "kT", "kH", "kW", "transpose", "O", "C", "Kld", "nTd", "nTh", "nTw", "k0d", "k0h", "k0w",
        "kstepd", "ksteph", "kstepw", "ntaps", "blocks", "reserved")]


class BnEmaJob(C.Structure):
    """rsp_bn_ema_job (32 bytes)"""
    _fields_ = [("running_mean", C.c_void_p), ("running_var", C.c_void_p), ("batch_stats", C.c_void_p), ("C", C.c_int32),
                ("momentum", C.c_float)]


class AugmentClipDesc(C.Structure):
    """rsp_augment_clip_desc (88 bytes)"""
    _fields_ = [("src", C.c_void_p), ("frame_pitch", C.c_int64), ("row_pitch", C.c_int32), ("h", C.c_int32), ("w", C.c_int32),
                ("gray", C.c_int32), ("flip", C.c_int32), ("n_ops", C.c_int32), ("op", C.c_int32 * 4),
                ("factor", C.c_float * 4), ("one_minus", C.c_float * 4)]


_PD = C.POINTER(ConvDesc)
_PP = C.POINTER(PoolDesc)
_sz = C.c_size_t
_i32 = C.c_int32
_i64 = C.c_int64
_f = C.c_float
_p = C.c_void_p

# name -> (restype, argtypes); mirrors include/rspnet_hip.h one to one (tests/test_abi.py checks both ways)
SIGNATURES = {
    "rsp_strerror": (C.c_char_p, [C.c_int]),
    "rsp_last_error": (C.c_char_p, []),
    "rsp_last_conv_kernel": (C.c_char_p, []),
    "rsp_version": (C.c_int, []),
    "rsp_conv3d_packed_fwd_elems": (_sz, [_PD]),
    "rsp_conv3d_pack_fwd": (C.c_int, [_PD, _p, _p, _p]),
    "rsp_conv3d_stat_tiles": (_i32, [_PD]),
    "rsp_conv3d_fwd_workspace": (_sz, [_PD]),
    "rsp_conv3d_fwd": (C.c_int, [_PD, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "rsp_conv3d_dgrad_workspace": (_sz, [_PD]),
    "rsp_conv3d_dgrad": (C.c_int, [_PD, _p, _p, _p, _p, _sz, _p]),
    "rsp_conv3d_packed_dgrad_elems": (_sz, [_PD]),
    "rsp_conv3d_dgrad_packed": (C.c_int, [_PD, _p, _p, _p, _p, _sz, _p]),
    "rsp_conv3d_pack_jobs": (_i32, [_PD, _i32, _i32, _i32, _p, _p, _p, _i32]),
    "rsp_pack_run": (C.c_int, [_p, _i32, _i32, _p]),
    "rsp_conv3d_pack_forget": (None, [_p]),
    "rsp_conv3d_wgrad_workspace": (_sz, [_PD]),
    "rsp_conv3d_wgrad": (C.c_int, [_PD, _p, _p, _p, _p, _p, _sz, _p]),
    "rsp_conv3d_wgrad_v": (C.c_int, [_PD, _p, _p, _p, _i32, _i32, _p, _sz, _p]),
    "rsp_conv3d_rowgeom_bytes": (_sz, [_PD]),
    "rsp_conv3d_rowgeom": (C.c_int, [_PD, _p, _p]),
    "rsp_conv3d_wgrad_t": (C.c_int, [_PD, _p, _p, _p, _p, _i32, _i32, _p, _p, _sz, _p]),
    "rsp_conv3d_kernel_name": (C.c_char_p, [_PD, C.c_int]),
    "rsp_conv3d_executed_fraction": (C.c_double, [_PD, C.c_int]),
    "rsp_conv3d_set_option": (C.c_int, [C.c_char_p, _i32]),
    "rsp_fastdiv_check": (C.c_int, [C.c_int, C.c_int]),
    "rsp_bn_finalize_workspace": (_sz, [_i32, _i32]),
    "rsp_bn_finalize": (C.c_int, [_p, _i32, _i32, _i32, _i64, _p, _p, _p, _f, _f, _p, _p, _p, _p, _p, _sz, _p]),
    "rsp_bn_finalize_v": (C.c_int, [_p, _i32, _i32, _i32, _i32, _i64, _p, _p, _p, _f, _f, _p, _p, _p, _p, _p, _sz, _p]),
    "rsp_bn_finalize_x": (C.c_int, [_p, _i32, _i32, _i32, _i32, _i64, _p, _p, _p, _f, _f, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "rsp_bn_running_update": (C.c_int, [_p, _i32, _i32, _p]),
    "rsp_bn_stat_tiles": (_i32, [_i64]),
    "rsp_bn_stats": (C.c_int, [_p, _i64, _i32, _i32, _p, _p]),
    "rsp_bn_act_pool_fwd": (C.c_int, [_PP, _p, _p, _p, C.c_int, _p, _p]),
    "rsp_bn_act_pool_gate_fwd": (C.c_int, [_PP, _p, _p, _p, C.c_int, _p, _p, _p]),
    "rsp_bn_act_maxpool_applicable": (C.c_int, [_PP]),
    "rsp_bn_act_maxpool_fwd": (C.c_int, [_PP, _p, _p, C.c_int, _p, _p, _p]),
    "rsp_bn_act_maxpool_gate_fwd": (C.c_int, [_PP, _p, _p, C.c_int, _p, _p, _p, _p]),
    "rsp_bn_bwd_workspace": (_sz, [_PP]),
    "rsp_bn_act_pool_bwd": (C.c_int, [_PP, _p, _p, _p, _p, _p, _p, C.c_int, _p, _p, _p, _p, _p, _sz, _p]),
    "rsp_bn_act_pool_bwd_v": (C.c_int, [_PP, _p, _p, _p, _p, _p, _p, C.c_int, _p, _p, _p, _p, _i32, _p, _sz, _p]),
    "rsp_bn_act_pool_bwd_g": (C.c_int, [_PP, _p, _p, _p, _p, _p, _p, C.c_int, _p, _p, _p, _p, _i32, _p, _p, _p, _sz, _p]),
    "rsp_maxpool3d_fwd": (C.c_int, [_PP, _p, _p, _p, _p]),
    "rsp_maxpool3d_bwd": (C.c_int, [_PP, _p, _p, _p, _p]),
    "rsp_gate_fwd_workspace": (_sz, [_i32, _i32, _i32]),
    "rsp_gate_fwd": (C.c_int, [_p, _i32, _i32, _i32, _i32, _p, _p, _p, _i32, _p, _p, _p, _sz, _p]),
    "rsp_bn_gate_sums": (C.c_int, [_p, _i32, _i32, _i32, _i32, _p, C.c_int, _p, _i32, _p, _p, _p, _p, _p, _sz, _p]),
    "rsp_gate_apply": (C.c_int, [_p, _i32, _i32, _i32, _i32, _p, _p, _i32, _p]),
    "rsp_gate_bwd_params": (C.c_int, [_p, _p, C.c_int, _p, _i32, _i32, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "rsp_gate_bwd_workspace": (_sz, [_i32, _i32, _i32]),
    "rsp_gate_bwd": (C.c_int, [_p, _p, _i32, _i32, _i32, _i32, _i32, _p, _p, _p, _p, _i32, _p, _p, _p, _sz, _p]),
    "rsp_head_fwd": (C.c_int, [_p, _i32, _i32, _i32, _i32, _p, _p, _p, _p, _i32, _p, _p, _p, _p, _p]),
    "rsp_head_bwd_workspace": (_sz, [_i32, _i32]),
    "rsp_head_bwd": (C.c_int, [_p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "rsp_spatial_mean_fwd": (C.c_int, [_p, _i32, _i32, _i32, _i32, _p, _p]),
    "rsp_spatial_mean_bwd": (C.c_int, [_p, _i32, _i32, _i32, _i32, _p, _p]),
    "rsp_linear_fwd": (C.c_int, [_p, _i32, _i32, _p, _p, _i32, C.c_int, _p, _p]),
    "rsp_linear_bwd_workspace": (_sz, [_i32, _i32]),
    "rsp_linear_bwd": (C.c_int, [_p, _p, _p, _p, _i32, _i32, _i32, C.c_int, _p, _p, _p, _p, _sz, _p]),
    "rsp_l2norm_fwd": (C.c_int, [_p, _i32, _i32, _p, _p]),
    "rsp_l2norm_bwd": (C.c_int, [_p, _p, _i32, _i32, _p, _p]),
    "rsp_logits_fwd": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _f, _p, _p, _p, _p, _p]),
    "rsp_logits_bwd_workspace": (_sz, [_i32, _i32, _i32]),
    "rsp_logits_bwd": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _f, _p, _p, _p, _sz, _p]),
    "rsp_loss_fwd_bwd": (C.c_int, [_p, _p, _p, _p, _i32, _i32, _f, _f, _f, _p, _p, _p, _p, _p, _p, _p]),
    "rsp_queue_enqueue": (C.c_int, [_p, _i32, _i32, _i32, _p, _i32, _p]),
    "rsp_queue_enqueue_dev": (C.c_int, [_p, _i32, _i32, _p, _p, _i32, _p]),
    "rsp_clip_gather": (C.c_int, [_p, _i32, _i32, _i32, _i32, _i32, _p, _p, _i32, _i32, _i32, _p, _p]),
    "rsp_clip_gather_multi": (C.c_int, [_i32, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p]),
    "rsp_momentum_update": (C.c_int, [_p, _p, _i64, _f, _p]),
    "rsp_sgd_step": (C.c_int, [_p, _p, _p, _i64, _f, _f, _f, _f, C.c_int, _p]),
    "rsp_rows_gather": (C.c_int, [_p, _p, _i32, _i32, _p, _p]),
    "rsp_eltwise": (C.c_int, [_i32, _p, _p, _p, _i64, _p]),
    "rsp_augment_workspace": (_sz, [_i32, _i32, _i32]),
    "rsp_augment_batch": (C.c_int, [_p, _i32, _i32, _i32, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), _p, _i64,
                                    _p, _sz, _p]),
}

_lib: Optional[C.CDLL] = None


class RspError(RuntimeError):
    pass


def load(init_gpu: bool = False) -> C.CDLL:
    """Load the HIP library or raise — there is no fallback path.  init_gpu=True (what HipOps passes) additionally brings the
    HIP runtime up through torch; host-only users (descriptor / workspace / kernel-name queries, rsp_fastdiv_check, the ABI
    tests, build()) get the symbols without the calling process ever initialising the GPU."""
    global _lib
    if init_gpu:
        _init_gpu()
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RspError(
            f"{LIB_PATH} not found: build it with rspnet_amd/csrc/build.sh (or __graft_entry__.build()). "
            "rspnet_amd has no CPU or eager fallback.")
    # Map libamdhip64 through torch first (torch's copy must be the one the process ends up with), WITHOUT initialising the
    # runtime: importing torch loads it, the runtime itself comes up lazily at the first HIP call.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


_gpu_ready = False


def _init_gpu():
    """Let PyTorch bring up the HIP runtime before the first launch from this library: with the runtime initialised later by
    torch — library launches first — every launch failed with "no ROCm-capable device is detected" on the GPU box (observed
    with build() + smoke() in one process)."""
    global _gpu_ready
    if _gpu_ready:
        return
    import torch
    if torch.cuda.is_available():      # (without a device every op still raises: HipOps only accepts HIP device tensors)
        torch.cuda.init()
        _gpu_ready = True


def check(rc: int, what: str):
    if rc != 0:
        lib = load()
        raise RspError(f"{what} failed: {lib.rsp_strerror(rc).decode()} ({lib.rsp_last_error().decode()})")


def source_hash() -> str:
    """sha256 over the kernel sources the library is built from (rspnet_amd/csrc/*.hip, *.h and include/*.h, names and
    contents): ties committed profile summaries (profiles/traffic.json) to the build they were measured on."""
    import glob
    import hashlib
    here = os.path.dirname(os.path.abspath(__file__))
    files = sorted(glob.glob(os.path.join(here, "csrc", "*.hip")) + glob.glob(os.path.join(here, "csrc", "*.h"))
                   + glob.glob(os.path.join(os.path.dirname(here), "include", "*.h")))
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()
