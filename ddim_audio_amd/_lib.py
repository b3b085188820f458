"""ctypes binding of libddimx.so (C ABI: include/ddimx.h).  No fallback: if the library is missing
or a call fails, a RuntimeError is raised (the reference's ``main.py:212-223`` logs exceptions)."""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int64, c_longlong, c_ulonglong, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DDIMX_LIB", os.path.join(_HERE, "libddimx.so"))

DDIMX_F32, DDIMX_BF16 = 0, 1
MAX_LEVELS = 8


class DdimxConfig(Structure):
    _fields_ = [
        ("in_channels", c_int), ("f_size", c_int), ("n_levels", c_int),
        ("ch", c_int * MAX_LEVELS), ("res", c_int * MAX_LEVELS), ("krn", c_int * MAX_LEVELS),
        ("n_timesteps", c_int), ("fnet_hidden", c_int), ("fnet_layers", c_int), ("fnet_inter", c_int),
        ("fnet_ln_eps", c_float), ("act_dtype", c_int), ("fnet_dtype", c_int),
    ]


class DdimxTables(Structure):
    _fields_ = [("posenc", c_void_p), ("dft_hidden", c_void_p), ("dft_seq", c_void_p), ("temb_table", c_void_p)]

    def __init__(self, posenc=None, dft_hidden=None, dft_seq=None, temb_table=None):
        super().__init__(posenc, dft_hidden, dft_seq, temb_table)


_SIGS = {
    "ddimx_abi_version": (c_int, []),
    "ddimx_last_error": (c_char_p, []),
    "ddimx_create": (c_int, [POINTER(DdimxConfig), POINTER(c_void_p)]),
    "ddimx_destroy": (c_int, [c_void_p]),
    "ddimx_set_dropout_counter": (c_int, [c_void_p, c_void_p]),
    "ddimx_num_params": (c_int, [c_void_p]),
    "ddimx_param_info": (c_int, [c_void_p, c_int, POINTER(c_char_p), POINTER(c_longlong)]),
    "ddimx_packed_bytes": (c_longlong, [c_void_p]),
    "ddimx_workspace_bytes": (c_longlong, [c_void_p, c_int, c_int]),
    "ddimx_pack_weights": (c_int, [c_void_p, POINTER(c_void_p), c_int, c_void_p, c_void_p]),
    "ddimx_pack_fnet_inference": (c_int, [c_void_p, c_void_p, c_void_p]),
    "ddimx_unet_fwd": (c_int, [c_void_p, c_void_p, POINTER(DdimxTables), c_void_p, c_longlong, c_void_p, c_void_p,
                               c_void_p, c_int, c_int, c_void_p]),
    "ddimx_unet_fwd_forked": (c_int, [c_void_p, c_void_p, POINTER(DdimxTables), c_void_p, c_longlong, c_void_p, c_void_p,
                                      c_void_p, c_int, c_int, c_void_p, c_void_p, POINTER(c_void_p), c_int, ctypes.c_uint]),
    "ddimx_packed_bwd_bytes": (c_longlong, [c_void_p]),
    "ddimx_pack_weights_bwd": (c_int, [c_void_p, POINTER(c_void_p), c_int, c_void_p, c_void_p, c_void_p]),
    "ddimx_train_tape_bytes": (c_longlong, [c_void_p, c_int, c_int]),
    "ddimx_train_workspace_bytes": (c_longlong, [c_void_p, c_int, c_int]),
    "ddimx_grad_floats": (c_longlong, [c_void_p]),
    "ddimx_grad_offset": (c_longlong, [c_void_p, c_int]),
    "ddimx_unet_fwd_train": (c_int, [c_void_p, c_void_p, POINTER(DdimxTables), c_void_p, c_longlong, c_void_p, c_longlong, c_void_p,
                                     c_void_p, c_void_p, c_int, c_int, c_float, c_ulonglong, c_void_p]),
    "ddimx_unet_bwd": (c_int, [c_void_p, c_void_p, c_void_p, POINTER(DdimxTables), c_void_p, c_longlong, c_void_p, c_longlong,
                               c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_ulonglong, c_void_p]),
    "ddimx_grad_buckets": (c_int, [c_void_p, POINTER(c_longlong)]),
    "ddimx_unet_bwd_staged": (c_int, [c_void_p, c_void_p, c_void_p, POINTER(DdimxTables), c_void_p, c_longlong, c_void_p, c_longlong,
                                      c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_ulonglong, POINTER(c_void_p), c_int,
                                      c_void_p]),
    "ddimx_bwd_side_events": (c_int, [c_void_p]),
    "ddimx_unet_bwd_forked": (c_int, [c_void_p, c_void_p, c_void_p, POINTER(DdimxTables), c_void_p, c_longlong, c_void_p, c_longlong,
                                      c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_ulonglong, POINTER(c_void_p), c_int,
                                      c_void_p, c_void_p, POINTER(c_void_p), c_int]),
    "ddimx_sqerr_loss_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_longlong, c_void_p]),
    "ddimx_sqerr_loss_bwd_mean": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_longlong, c_void_p]),
    "ddimx_to_nhwc": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "ddimx_from_nhwc": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "ddimx_pack_conv": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "ddimx_pack_convT": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "ddimx_op_workspace_bytes": (c_longlong, [c_int, c_int, c_int, c_int, c_int]),
    "ddimx_resblock_fwd": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_int] + [c_void_p] * 8 +
                           [c_void_p, c_int, c_int, c_int, c_void_p]),
    "ddimx_rb_tape_floats": (c_longlong, [c_int, c_int]),
    "ddimx_resblock_fwd_train": (c_int, [c_int, c_int] + [c_void_p] * 3 + [c_int] + [c_void_p] * 12 + [c_int] * 3 + [c_void_p]),
    "ddimx_pack_conv_dgrad": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "ddimx_resblock_bwd_workspace_bytes": (c_longlong, [c_int] * 5),
    "ddimx_resblock_bwd": (c_int, [c_int, c_int] + [c_void_p] * 20 + [c_int, c_void_p] + [c_int] * 3 + [c_void_p]),
    "ddimx_conv3x3_fwd": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int,
                                  c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "ddimx_debug_set_stamps": (c_int, [c_void_p]),
    "ddimx_pack_conv_frag": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "ddimx_pack_conv_frag_k": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "ddimx_downsample_wreg_fwd": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "ddimx_pack_frag_from_taps": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "ddimx_upsample_add_wreg_fwd": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                            c_void_p]),
    "ddimx_conv3x3_pipe_fwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int,
                                       c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "ddimx_conv3x3_pipe_stats_floats": (c_longlong, [c_int, c_int, c_int, c_int]),
    "ddimx_conv3x3_wreg_fwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int,
                                       c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "ddimx_conv3x3_stats_floats": (c_longlong, [c_int, c_int, c_int, c_int, c_int]),
    "ddimx_debug_conv3x3_stamps": (c_int, [c_int, c_int] + [c_void_p] * 8 + [c_int, c_int, c_int, c_void_p]),
    "ddimx_resid_gn_fwd": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                   c_void_p]),
    "ddimx_downsample_fwd": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                     c_void_p]),
    "ddimx_upsample_add_fwd": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                       c_int, c_int, c_void_p]),
    "ddimx_temb_fwd": (c_int, [c_void_p] * 11 + [c_int, c_int, c_int, c_int, c_void_p]),
    "ddimx_conv_in_stats_floats": (c_longlong, [c_int] * 4),
    "ddimx_conv_in_fwd": (c_int, [c_int] + [c_void_p] * 5 + [c_int] * 5 + [c_void_p]),
    "ddimx_conv_out_fwd": (c_int, [c_int] + [c_void_p] * 5 + [c_int] * 5 + [c_void_p]),
    "ddimx_fnet_fwd": (c_int, [c_void_p, c_void_p, POINTER(DdimxTables), c_void_p, c_longlong, c_void_p, c_void_p, c_int, c_int,
                               c_void_p]),
    "ddimx_fnet_fwd_train": (c_int, [c_void_p, c_void_p, POINTER(DdimxTables), c_void_p, c_longlong, c_void_p, c_longlong, c_void_p,
                                     c_void_p, c_int, c_int, c_float, c_ulonglong, c_void_p]),
    "ddimx_fnet_bwd": (c_int, [c_void_p, c_void_p, c_void_p, POINTER(DdimxTables), c_void_p, c_longlong, c_void_p, c_longlong,
                               c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_ulonglong, c_void_p]),
    "ddimx_downup_bwd_workspace_bytes": (c_longlong, [c_int] * 6),
    "ddimx_downsample_bwd": (c_int, [c_int] * 3 + [c_void_p] * 8 + [c_int] * 3 + [c_void_p]),
    "ddimx_upsample_add_bwd": (c_int, [c_int] * 3 + [c_void_p] * 7 + [c_int] * 3 + [c_void_p]),
    "ddimx_edge_bwd_workspace_floats": (c_longlong, [c_int] * 6),
    "ddimx_conv_in_bwd": (c_int, [c_int] + [c_void_p] * 5 + [c_int] * 5 + [c_void_p]),
    "ddimx_conv_out_bwd": (c_int, [c_int] + [c_void_p] * 8 + [c_int] * 5 + [c_void_p]),
    "ddimx_temb_fwd_train": (c_int, [c_void_p] * 11 + [c_int] * 4 + [c_void_p]),
    "ddimx_temb_bwd": (c_int, [c_void_p] * 15 + [c_int] * 4 + [c_void_p]),
    "ddimx_fnet_mix_supported": (c_int, [c_int, c_int]),
    "ddimx_fnet_mix": (c_int, [c_void_p] * 6 + [c_int, c_int, c_int, c_int, c_void_p]),
    "ddimx_step_begin": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "ddimx_step_begin_ex": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "ddimx_ddim_update": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_longlong, c_void_p]),
    "ddimx_ddpm_update": (c_int, [c_void_p] * 7 + [c_longlong, c_void_p]),
    "ddimx_step_end": (c_int, [c_void_p, c_void_p]),
    "ddimx_qsample": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_longlong, c_void_p]),
    "ddimx_sqerr_loss": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_longlong, c_void_p]),
    "ddimx_ema_block_elems": (c_int, []),
    "ddimx_ema_update_multi": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float, c_void_p]),
    "ddimx_grad_norm_multi": (c_int, [c_void_p] * 4 + [c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "ddimx_scale_multi": (c_int, [c_void_p] * 4 + [c_int, c_void_p, c_void_p]),
    "ddimx_adam_multi": (c_int, [c_void_p] * 7 + [c_int, c_void_p, c_float, c_float, c_float, c_float, c_float, c_int, c_int, c_void_p]),
    "ddimx_adam_multi_dyn": (c_int, [c_void_p] * 7 + [c_int, c_void_p, c_void_p, c_float, c_float, c_float, c_float, c_int, c_void_p]),
}

EXPORTS = tuple(_SIGS)
_lib = None


def load():
    """Load libddimx.so once; raise loudly if it is missing (there is no CPU or eager fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -m ddim_audio_amd.build` "
                "(hipcc, gfx950). The HIP library is the only compute path of ddim_audio_amd.")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        if lib.ddimx_abi_version() != 2:
            raise RuntimeError("libddimx ABI version mismatch")
        _lib = lib
    return _lib


def check(rc):
    if rc != 0:
        raise RuntimeError("libddimx: " + load().ddimx_last_error().decode(errors="replace"))


def ptr(t):
    """Device pointer of a tensor (or None)."""
    return None if t is None else c_void_p(t.data_ptr())


def stream():
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)
