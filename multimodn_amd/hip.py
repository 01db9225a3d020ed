"""ctypes binding of libmmn_hip.so (include/mmn_hip.h).  No torch types cross this boundary:
device pointers are passed as integers, the stream as a void*.

The library is built in-tree by `__graft_entry__.build()` / `python -m multimodn_amd.build`.
There is no CPU fallback: if the shared object is missing or does not export the full ABI,
`load()` raises and every training entry point fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

MAX_ENCODERS = 16
MAX_DECODERS = 8
MAX_LAYERS = 8
MAX_DIM = 128
ADAM_MAX_SEG = 512
ERR_UNSUPPORTED = -2
ERR_PEER = -6
VERSION = 113
MAX_DEC_HIDDEN = 3
ENC_MLP, ENC_MIMIC = 0, 1
MODEL_GENERIC_TIER = 1      # mmn_model.flags (include/mmn_hip.h)

ACT_IDENTITY, ACT_RELU, ACT_SIGMOID = 0, 1, 2

LIB_NAME = "libmmn_hip.so"
LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), LIB_NAME)

#: every symbol include/mmn_hip.h declares
ABI_SYMBOLS = (
    "mmn_version", "mmn_source_hash", "mmn_error_string", "mmn_last_hip_error", "mmn_stats_floats", "mmn_epoch_doubles",
    "mmn_workspace_bytes", "mmn_plan_create", "mmn_plan_destroy", "mmn_nan_flags", "mmn_prepare",
    "mmn_nan_scan", "mmn_chain_kernel_name", "mmn_chain_fwd",
    "mmn_chain_bwd", "mmn_chain_fwd_bwd", "mmn_wgrad", "mmn_reduce", "mmn_epoch_accumulate", "mmn_train_step",
    "mmn_eval_step", "mmn_adam_blocks", "mmn_adam_step", "mmn_adam_step_accumulate", "mmn_train_step_adam", "mmn_reduce_adam", "mmn_regroup_rows", "mmn_regroup", "mmn_epoch_reset", "mmn_epoch_read", "mmn_debug_buffer",
    "mmn_dropout_floats", "mmn_draw_dropout", "mmn_dropout_reset", "mmn_dropout_adopt",
    "mmn_nan_flags_set", "mmn_pack_invalidate", "mmn_pack_refresh", "mmn_train_step_ex", "mmn_epoch_write", "mmn_adam_fusable", "mmn_regroup_ex", "mmn_dp_rescale", "mmn_eval_step_ex",
    "mmn_dp_xbuf_bytes", "mmn_dp_xbuf_alloc", "mmn_dp_xbuf_open", "mmn_dp_xbuf_close", "mmn_dp_oneshot_attach",
    "mmn_dp_oneshot_error", "mmn_adam_step_accumulate_oneshot", "mmn_regroup_multi", "mmn_wgrad_reduce",
    "mmn_epoch_small_rows", "mmn_train_epoch_small", "mmn_dp_oneshot_detach", "mmn_per_sample_supported", "mmn_dp_oneshot_diag",
)


class Linear(C.Structure):
    _fields_ = [("w", C.c_void_p), ("b", C.c_void_p), ("gw", C.c_void_p), ("gb", C.c_void_p),
                ("out_dim", C.c_int32), ("in_dim", C.c_int32)]


class Encoder(C.Structure):
    _fields_ = [("n_features", C.c_int32), ("n_layers", C.c_int32), ("activation", C.c_int32),
                ("kind", C.c_int32), ("layer", Linear * MAX_LAYERS)]


class Decoder(C.Structure):
    _fields_ = [("w", C.c_void_p), ("b", C.c_void_p), ("gw", C.c_void_p), ("gb", C.c_void_p),
                ("n_hidden", C.c_int32), ("hidden_activation", C.c_int32), ("hidden", Linear * MAX_DEC_HIDDEN)]


class Model(C.Structure):
    _fields_ = [("state_size", C.c_int32), ("n_encoders", C.c_int32), ("n_decoders", C.c_int32),
                ("flags", C.c_int32), ("init_state", C.c_void_p), ("g_init_state", C.c_void_p),
                ("enc", Encoder * MAX_ENCODERS), ("dec", Decoder * MAX_DECODERS)]


class Batch(C.Structure):
    _fields_ = [("x", C.c_void_p * MAX_ENCODERS), ("ldx", C.c_int32 * MAX_ENCODERS),
                ("y", C.c_void_p), ("nan_flags", C.c_void_p),
                ("batch", C.c_int32), ("batch_global", C.c_int32), ("n_seq", C.c_int32),
                ("flags_ready", C.c_int32),
                ("seq_data", C.c_int32 * MAX_ENCODERS), ("seq_enc", C.c_int32 * MAX_ENCODERS),
                ("tile_rows", C.c_void_p), ("tile_seq", C.c_void_p),
                ("drop_mask", C.c_void_p * MAX_ENCODERS)]


class AdamDesc(C.Structure):
    _fields_ = [("params", C.c_void_p), ("grads", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("steps", C.c_void_p), ("seg_start", C.c_void_p), ("seg_skip", C.c_void_p),
                ("n", C.c_int64),
                ("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double),
                ("weight_decay", C.c_double), ("n_seg", C.c_int32), ("maximize", C.c_int32)]


class StepOpts(C.Structure):
    _fields_ = [("adam", C.POINTER(AdamDesc)), ("next", C.POINTER(Batch)), ("accumulate_epoch", C.c_int32),
                ("reserved", C.c_int32), ("next_drop_p", C.POINTER(C.c_float)), ("next_drop_buf", C.c_void_p),
                ("next_drop_seed", C.c_uint64), ("next_drop_floats", C.c_uint64)]


#: advanced by every parameter update that goes through raw pointers OUTSIDE an engine's own step (multimodn_amd.optim.Adam.step
#: on its own): engines compare it, next to torch's version counters, before trusting the kernels' weight copies across calls
PARAM_WRITES = [0]


class MmnError(RuntimeError):
    pass


_lib: Optional[C.CDLL] = None


def load(path: Optional[str] = None) -> C.CDLL:
    """dlopen the HIP library and bind argument types.  Raises if it is absent or incomplete."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("MMN_LIB_PATH") or LIB_PATH        # (MMN_LIB_PATH: diagnostics - A/B runs of kernel variants)
    if not os.path.exists(p):
        raise MmnError(
            f"{p} not found: the MI355X HIP library is not built. Run `python -c 'import "
            f"__graft_entry__ as g; g.build()'` (needs hipcc). There is no CPU fallback for the "
            f"training hot path.")
    lib = C.CDLL(p)
    missing = [s for s in ABI_SYMBOLS if not hasattr(lib, s)]
    if missing:
        raise MmnError(f"{p} does not export {missing}")
    vp, i32, f32 = C.c_void_p, C.c_int, C.c_float
    lib.mmn_version.restype = i32
    lib.mmn_error_string.restype = C.c_char_p
    lib.mmn_error_string.argtypes = [i32]
    lib.mmn_last_hip_error.restype = i32
    lib.mmn_stats_floats.restype = C.c_size_t
    lib.mmn_stats_floats.argtypes = [C.POINTER(Model)]
    lib.mmn_epoch_doubles.restype = C.c_size_t
    lib.mmn_epoch_doubles.argtypes = [C.POINTER(Model)]
    lib.mmn_workspace_bytes.restype = C.c_size_t
    lib.mmn_workspace_bytes.argtypes = [C.POINTER(Model), i32]
    lib.mmn_plan_create.restype = i32
    lib.mmn_plan_create.argtypes = [C.POINTER(Model), i32, vp, C.c_size_t, vp, C.POINTER(vp)]
    lib.mmn_plan_destroy.restype = None
    lib.mmn_plan_destroy.argtypes = [vp]
    lib.mmn_nan_flags.restype = vp
    lib.mmn_nan_flags.argtypes = [vp]
    lib.mmn_nan_flags_set.restype = vp
    lib.mmn_nan_flags_set.argtypes = [vp, i32]
    lib.mmn_pack_invalidate.restype = None
    lib.mmn_pack_invalidate.argtypes = [vp]
    lib.mmn_pack_refresh.restype = i32
    lib.mmn_pack_refresh.argtypes = [vp, vp]
    lib.mmn_train_step_ex.restype = i32
    lib.mmn_train_step_ex.argtypes = [vp, C.POINTER(Batch), f32, f32, C.POINTER(StepOpts), vp]
    lib.mmn_epoch_small_rows.restype = i32
    lib.mmn_epoch_small_rows.argtypes = [vp]
    lib.mmn_train_epoch_small.restype = i32
    lib.mmn_train_epoch_small.argtypes = [vp, C.POINTER(Batch), vp, i32, f32, f32, C.POINTER(AdamDesc), vp]
    lib.mmn_wgrad_reduce.restype = i32
    lib.mmn_wgrad_reduce.argtypes = [vp, C.POINTER(Batch), f32, f32, C.POINTER(StepOpts), vp]
    lib.mmn_adam_fusable.restype = i32
    lib.mmn_adam_fusable.argtypes = [vp, C.POINTER(AdamDesc)]
    lib.mmn_epoch_write.restype = i32
    lib.mmn_epoch_write.argtypes = [vp, C.POINTER(C.c_double), vp]
    lib.mmn_prepare.restype = i32
    lib.mmn_prepare.argtypes = [vp, C.POINTER(Batch), i32, vp]
    lib.mmn_nan_scan.restype = i32
    lib.mmn_nan_scan.argtypes = [vp, C.POINTER(Batch), vp]
    lib.mmn_chain_kernel_name.restype = C.c_char_p
    lib.mmn_chain_kernel_name.argtypes = [vp, C.POINTER(Batch), i32]
    lib.mmn_chain_fwd.restype = i32
    lib.mmn_chain_fwd.argtypes = [vp, C.POINTER(Batch), f32, f32, i32, vp]
    lib.mmn_chain_bwd.restype = i32
    lib.mmn_chain_bwd.argtypes = [vp, C.POINTER(Batch), f32, vp]
    lib.mmn_chain_fwd_bwd.restype = i32
    lib.mmn_chain_fwd_bwd.argtypes = [vp, C.POINTER(Batch), f32, f32, vp]
    lib.mmn_wgrad.restype = i32
    lib.mmn_wgrad.argtypes = [vp, C.POINTER(Batch), vp]
    lib.mmn_reduce.restype = i32
    lib.mmn_reduce.argtypes = [vp, C.POINTER(Batch), vp]
    lib.mmn_eval_step_ex.restype = i32
    lib.mmn_eval_step_ex.argtypes = [vp, C.POINTER(Batch), i32, i32, vp, vp, vp]
    lib.mmn_dp_xbuf_bytes.restype = C.c_size_t
    lib.mmn_dp_xbuf_bytes.argtypes = [vp]
    lib.mmn_dp_xbuf_alloc.restype = i32
    lib.mmn_dp_xbuf_alloc.argtypes = [C.c_size_t, C.POINTER(vp), C.c_char_p]
    lib.mmn_dp_xbuf_open.restype = i32
    lib.mmn_dp_xbuf_open.argtypes = [C.c_char_p, C.POINTER(vp)]
    lib.mmn_dp_xbuf_close.restype = i32
    lib.mmn_dp_xbuf_close.argtypes = [vp, i32]
    lib.mmn_dp_oneshot_attach.restype = i32
    lib.mmn_dp_oneshot_attach.argtypes = [vp, i32, i32, C.POINTER(vp), i32]
    lib.mmn_dp_oneshot_error.restype = i32
    lib.mmn_dp_oneshot_error.argtypes = [vp]
    lib.mmn_dp_oneshot_detach.restype = i32
    lib.mmn_dp_oneshot_detach.argtypes = [vp]
    lib.mmn_dp_oneshot_diag.restype = i32
    lib.mmn_dp_oneshot_diag.argtypes = [vp, C.POINTER(C.c_uint32)]
    lib.mmn_per_sample_supported.restype = i32
    lib.mmn_per_sample_supported.argtypes = [vp]
    lib.mmn_adam_step_accumulate_oneshot.restype = i32
    lib.mmn_adam_step_accumulate_oneshot.argtypes = [vp, C.POINTER(AdamDesc), f32, f32, vp]
    lib.mmn_dp_rescale.restype = i32
    lib.mmn_dp_rescale.argtypes = [vp, vp, C.c_int64, i32, vp]
    lib.mmn_epoch_accumulate.restype = i32
    lib.mmn_epoch_accumulate.argtypes = [vp, f32, f32, vp]
    lib.mmn_train_step.restype = i32
    lib.mmn_train_step.argtypes = [vp, C.POINTER(Batch), f32, f32, i32, vp]
    lib.mmn_eval_step.restype = i32
    lib.mmn_eval_step.argtypes = [vp, C.POINTER(Batch), i32, vp]
    lib.mmn_adam_blocks.restype = i32
    lib.mmn_adam_blocks.argtypes = [C.c_int64]
    lib.mmn_adam_step.restype = i32
    lib.mmn_adam_step.argtypes = [C.POINTER(AdamDesc), vp]
    lib.mmn_adam_step_accumulate.restype = i32
    lib.mmn_adam_step_accumulate.argtypes = [vp, C.POINTER(AdamDesc), f32, f32, vp]
    lib.mmn_train_step_adam.restype = i32
    lib.mmn_train_step_adam.argtypes = [vp, C.POINTER(Batch), f32, f32, i32, C.POINTER(AdamDesc), vp]
    lib.mmn_reduce_adam.restype = i32
    lib.mmn_reduce_adam.argtypes = [vp, C.POINTER(Batch), C.POINTER(AdamDesc), vp]
    lib.mmn_regroup_rows.restype = i32
    lib.mmn_regroup_rows.argtypes = [i32, i32]
    lib.mmn_regroup.restype = i32
    lib.mmn_regroup.argtypes = [vp, C.POINTER(Batch), vp, C.POINTER(Batch), vp]
    lib.mmn_regroup_ex.restype = i32
    lib.mmn_regroup_ex.argtypes = [vp, C.POINTER(Batch), vp, C.POINTER(Batch), vp, vp]
    lib.mmn_regroup_multi.restype = i32
    lib.mmn_regroup_multi.argtypes = [vp, i32, C.POINTER(C.POINTER(Batch)), C.POINTER(vp), C.POINTER(C.POINTER(Batch)), C.POINTER(vp), vp]
    lib.mmn_epoch_reset.restype = i32
    lib.mmn_epoch_reset.argtypes = [vp, vp]
    lib.mmn_epoch_read.restype = i32
    lib.mmn_epoch_read.argtypes = [vp, C.POINTER(C.c_double), vp]
    lib.mmn_debug_buffer.restype = vp
    lib.mmn_debug_buffer.argtypes = [vp, i32, i32]
    lib.mmn_dropout_floats.restype = C.c_size_t
    lib.mmn_dropout_floats.argtypes = [vp, i32]
    lib.mmn_draw_dropout.restype = i32
    lib.mmn_draw_dropout.argtypes = [vp, C.POINTER(Batch), C.POINTER(C.c_float), C.c_uint64, vp, C.c_size_t, vp]
    lib.mmn_dropout_adopt.restype = i32
    lib.mmn_dropout_adopt.argtypes = [vp, C.POINTER(Batch), C.POINTER(C.c_float), vp, C.c_size_t, vp]
    lib.mmn_dropout_reset.restype = i32
    lib.mmn_dropout_reset.argtypes = [vp, vp]
    if lib.mmn_version() != VERSION:
        raise MmnError(f"{p}: ABI version {lib.mmn_version()} != expected {VERSION}; rebuild")
    lib.mmn_source_hash.restype = C.c_char_p
    if path is None and not os.environ.get("MMN_LIB_PATH"):    # (A/B builds of kernel variants are somebody's deliberate choice)
        from . import build as _build
        want, have = _build.source_hash(), lib.mmn_source_hash().decode()
        if have != want:
            raise MmnError(f"{p} was built from other sources (library {have}, tree {want}): run "
                           f"`python -m multimodn_amd.build` - a stale library must not run silently")
    if path is None:
        _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        lib = load()
        msg = lib.mmn_error_string(rc).decode()
        extra = f" (hipError {lib.mmn_last_hip_error()})" if rc == -4 else ""
        raise MmnError(f"{what} failed: {msg}{extra}")
