"""ctypes binding of libsvs_hip.so -- the thin boundary between the Python host code and the
hand-written gfx950 kernels (declared in include/svs_hip.h).

There is NO fallback: if the library is missing or a call fails, the caller gets an exception.
"""
from __future__ import annotations

import ctypes as C
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
# SVS_LIB_PATH: load another build of the same library (same-device A/B runs of tools/ and bench.py)
LIB_PATH = os.environ.get("SVS_LIB_PATH") or os.path.join(HERE, "libsvs_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(HERE), "include", "svs_hip.h")

_lib = None

P = C.c_void_p
I = C.c_int
L = C.c_int64
F = C.c_float
D = C.c_double
Z = C.c_size_t
U32 = C.c_uint32
U64 = C.c_uint64

_SIGS = {
    # name: (restype, [argtypes])
    "svs_version": (I, []),
    "svs_last_error_string": (C.c_char_p, []),
    "svs_tuning_set": (I, [C.c_char_p, C.c_long]),
    "svs_fill_uniform": (I, [P, L, U32, U64, F, F, P]),
    "svs_fill_tiles": (I, [P, P, I, I, I, L, P]),
    "svs_crop_tiles": (I, [P, P, P, P, P, P, I, I, I, P, P, P]),
    "svs_phase_angle": (I, [P, P, L, P]),
    "svs_dropout_mask": (I, [P, I, I, I, U32, I, I, P]),
    "svs_dropout_masks_all": (I, [P, I, U32, I, I, P]),
    "svs_pack_weight_gather": (I, [P, P, I, I, P]),
    "svs_pack_weight_parity": (I, [P, P, I, I, P]),
    "svs_bn_fold": (I, [P, P, P, P, P, F, P, P, I, P]),
    "svs_enc_block_workspace_bytes": (Z, [I, I, I, I, I]),
    "svs_enc_block_fwd": (I, [P, L, I, I, I, I, P, P, P, P, F, P, L, I, I, P, Z, P]),
    "svs_dec_block_workspace_bytes": (Z, [I, I, I, I, I, I, I]),
    "svs_dec_block_fwd": (I, [P, L, I, I, I, I, P, P, P, P, F, P, L, I, I, I, I, P, Z, P]),
    "svs_out_block_fwd": (I, [P, L, I, I, I, I, P, P, P, I, I, I, P]),
    "svs_enc_block_bwd_data": (I, [P, L, I, I, I, I, P, P, L, I, I, I, I, P, Z, P]),
    "svs_dec_block_bwd_data": (I, [P, L, I, I, I, I, P, P, L, I, I, I, I, P, Z, P]),
    "svs_block_bwd_weight_workspace_bytes": (Z, [I, I, I, I, I]),
    "svs_enc_block_bwd_weight": (I, [P, L, I, I, I, I, P, L, I, I, I, P, P, P, Z, P]),
    "svs_dec_block_bwd_weight": (I, [P, L, I, I, I, I, P, L, I, I, I, P, P, P, Z, P]),
    "svs_describe_plan": (I, [I, I, I, I, I, I, I, I, C.c_char_p, Z]),
    "svs_bn_workspace_bytes": (Z, [L, I]),
    "svs_bn_stats": (I, [P, L, L, I, P, Z, P]),
    "svs_bn_finalize": (I, [P, L, I, F, F, P, P, P, P, P, P]),
    "svs_bn_act_apply": (I, [P, L, L, I, L, P, P, P, P, F, P, P, L, P]),
    "svs_bn_bwd": (I, [P, L, P, L, L, I, L, P, P, P, P, F, P, P, P, P, P, Z, P]),
    "svs_l1_mask_loss_workspace_bytes": (Z, [L]),
    "svs_l1_mask_loss_fwd_bwd": (I, [P, P, P, L, F, P, P, P, Z, P]),
    "svs_adam_step": (I, [P, P, P, P, L, D, D, D, D, I, F, P]),
    "svs_apply_mask": (I, [P, P, P, L, I, P]),
    "svs_unet_param_offset": (L, [I]),
    "svs_unet_buffer_offset": (L, [I, I]),
    "svs_unet_prepared_bytes": (Z, []),
    "svs_unet_prepare_eval": (I, [P, P, P, P]),
    "svs_unet_eval_workspace_bytes": (Z, [I, I, I]),
    "svs_unet_forward_eval": (I, [P, P, P, I, I, I, P, Z, P]),
    "svs_unet_prepared_bf16_bytes": (Z, []),
    "svs_unet_prepare_eval_bf16": (I, [P, P, P]),
    "svs_unet_eval_bf16_workspace_bytes": (Z, [I, I, I]),
    "svs_unet_forward_eval_bf16": (I, [P, P, P, I, I, I, P, Z, P]),
    "svs_unet_train_workspace_bytes": (Z, [I, I, I]),
    "svs_unet_train_fwd_bwd": (I, [P, P, P, P, P, P, P, I, I, I, F, P, P, P, Z, P]),
    "svs_unet_train_forward": (I, [P, P, P, P, P, I, I, I, P, P, Z, P]),
    "svs_unet_train_backward": (I, [P, P, P, P, P, P, I, I, I, P, Z, P]),
    "svs_unet_train_fwd_loss": (I, [P, P, P, P, P, P, I, I, I, F, P, P, P, Z, P]),
    "svs_unet_train_bwd_part": (I, [P, P, P, P, I, I, I, I, P, Z, P]),
    "svs_unet_train_bwd_sync": (I, [P]),
    "svs_unet_train_mr_workspace_bytes": (Z, [I, I, I]),
    "svs_unet_train_fwd_loss_mr": (I, [P, P, P, P, P, P, P, P, I, I, I, I, F, F, P, P, P, Z, P, Z, P]),
    "svs_unet_ws_offset": (L, [C.c_char_p, I, I, I, I]),
    "svs_stft_frames": (I, [L, I]),
    "svs_stft_fwd": (I, [P, L, I, I, P, P, P]),
    "svs_istft_workspace_bytes": (Z, [I, I, I]),
    "svs_istft": (I, [P, P, I, I, I, I, P, P, Z, P]),
    "svs_stft_tiles": (I, [P, L, I, I, I, P, L, I, I, I, I, P, I, P, P]),
    "svs_stft_groups": (I, [I]),
    "svs_istft_tiles": (I, [P, L, I, I, I, P, I, P, I, I, I, I, I, P, P, P]),
    "svs_istft_groups": (I, [I, I, I]),
    "svs_transpose_c64": (I, [P, P, I, I, P]),
    "svs_istft_bwd_mask": (I, [P, P, P, P, P, F, I, I, I, I, P]),
    "svs_mrstft_workspace_bytes": (Z, [I, L]),
    "svs_mrstft_loss_fwd_bwd": (I, [P, P, I, L, F, P, P, P, Z, P]),
    "svs_absmax": (I, [P, L, P, P, Z, P]),
    "svs_max": (I, [P, L, P, P]),
    "svs_scale_by_inv": (I, [P, L, P, F, P]),
}


class SvsError(RuntimeError):
    pass


def header_symbols():
    """Every function name declared in include/svs_hip.h."""
    text = open(HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(svs_[a-z0-9_]+)\s*\(", text)))


def lib():
    """The loaded library (loads on first use; raises if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SvsError(f"{LIB_PATH} not found: build it with `python -m svs_unet_pytorch_amd.build` "
                           "(there is no CPU or PyTorch fallback for the HIP path)")
        # torch first: its bundled libamdhip64 must be THE HIP runtime of the process (the streams we are
        # handed belong to it).  Loaded the other way round, the system runtime and torch's both get
        # mapped and the second one finds no device.
        import torch  # noqa: F401
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().svs_last_error_string()
        raise SvsError(f"{what or 'libsvs_hip'} failed with code {rc}: {msg.decode() if msg else ''}")


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream_ptr(device=None):
    """The current HIP stream of `device` (default: the current device) as an integer handle."""
    import torch
    return torch.cuda.current_stream(device).cuda_stream


def tuning(name: str, value: int = -1):
    """Planner override for tests / sweeps (include/svs_hip.h: svs_tuning_set); value -1 restores the default."""
    check(lib().svs_tuning_set(name.encode(), int(value)), "svs_tuning_set")
