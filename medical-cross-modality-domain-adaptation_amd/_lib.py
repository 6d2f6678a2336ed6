"""ctypes binding of libpnp_hip.so (C-ABI declared in include/pnp_hip.h).

This is the ONLY route from the Python host code to the GPU arithmetic.  There is no CPU or
PyTorch fallback: if the shared library is missing or a call fails, we raise.
"""
import ctypes
import os

# torch FIRST: its wheel carries its own libamdhip64 / libhsa-runtime64.  Loaded after torch, libpnp_hip.so's DT_NEEDED libamdhip64.so.7
# resolves to that already-mapped runtime (one HIP runtime per process, shared streams and allocations).  Loaded BEFORE torch, the
# system copy under /opt/rocm comes in as a second HIP + HSA runtime and every launch from this library fails with
# "no ROCm-capable device is detected" (measured on the GPU box; tests/test_abi.py guards the order).
import torch  # noqa: F401  (import order is the point)

from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_uint32, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PNP_LIB") or os.path.join(_HERE, "libpnp_hip.so")   # PNP_LIB: timing-experiment builds (tools/)

PAD_ZERO = 0
PAD_SYMMETRIC = 1
OPT_CHUNK = 1024
DTYPE_F32, DTYPE_BF16, DTYPE_F64 = 0, 1, 2
COMM_ID_BYTES = 128
ABI_VERSION = 4


class ConvGeom(ctypes.Structure):
    """mirror of `pnp_conv_geom` (include/pnp_hip.h)"""
    _fields_ = [(n, c_int32) for n in
                ("N", "H", "W", "C", "K", "R", "S", "OH", "OW", "stride", "dil", "pad_t", "pad_l", "pad_mode", "dtype")]

    def key(self):
        return tuple(getattr(self, f) for f, _ in self._fields_)


class ProfRow(ctypes.Structure):
    """mirror of `pnp_prof_row` (include/pnp_hip.h)"""
    _fields_ = [("name", ctypes.c_char * 128), ("launches", c_int64), ("ms", ctypes.c_double), ("flops", ctypes.c_double),
                ("bytes", ctypes.c_double)]


PROF_CONV_FWD, PROF_CONV_DGRAD, PROF_CONV_WGRAD, PROF_CONV_DIRECT = 1, 2, 4, 8


class PnpError(RuntimeError):
    pass


_F = c_void_p  # device float* (passed as integer address)
_G = POINTER(ConvGeom)

# name -> (restype, argtypes); every symbol of include/pnp_hip.h must appear here (tests check it)
PROTOTYPES = {
    "pnp_abi_version": (c_int, []),
    "pnp_last_error": (c_char_p, []),
    "pnp_device_info": (c_int, [c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int), c_char_p, c_int]),
    "pnp_prof_enable": (c_int, [c_int32]),
    "pnp_prof_summary": (c_int, [POINTER(ProfRow), c_int32]),
    "pnp_conv2d_fwd": (c_int, [_F, _F, _F, _G, c_float, c_uint64, c_uint32, c_void_p]),
    "pnp_conv2d_dgrad_workspace_bytes": (c_size_t, [_G]),
    "pnp_conv2d_dgrad": (c_int, [_F, _F, _F, _G, c_void_p, c_size_t, c_void_p]),
    "pnp_conv2d_dgrad_add": (c_int, [_F, _F, _F, _F, _G, c_void_p, c_size_t, c_void_p]),
    "pnp_conv2d_wgrad_workspace_bytes": (c_size_t, [_G]),
    "pnp_conv2d_wgrad": (c_int, [_F, _F, _F, _G, c_void_p, c_size_t, c_void_p]),
    "pnp_conv2d_wgrad_acc": (c_int, [_F, _F, _F, _G, c_void_p, c_size_t, c_void_p]),
    "pnp_conv2d_fwd_workspace_bytes": (c_size_t, [_G]),
    "pnp_conv2d_fwd_ws": (c_int, [_F, _F, _F, _G, c_float, c_uint64, c_uint32, c_void_p, c_size_t, c_void_p]),
    "pnp_conv2d_fwd_stats_parts": (c_int32, [_G]),
    "pnp_conv2d_fwd_stats": (c_int, [_F, _F, _F, _G, c_float, c_uint64, c_uint32, _F, _F, c_size_t, c_void_p]),
    "pnp_bn_stats_finish": (c_int, [_F, c_int32, _F, _F, _F, _F, _F, c_int64, c_int32, c_float, c_void_p]),
    "pnp_bn_fold": (c_int, [_F, _F, _F, _F, _F, _F, c_int32, c_float, c_void_p]),
    "pnp_conv2d_fwd_bn": (c_int, [_F, _F, _F, _G, c_float, c_uint64, c_uint32, _F, _F, _F, c_int32, c_float, c_void_p]),
    "pnp_conv2d_fwd_bn_ws": (c_int, [_F, _F, _F, _G, c_float, c_uint64, c_uint32, _F, _F, _F, c_int32, c_float, c_void_p, c_size_t, c_void_p]),
    "pnp_conv2d_fwd_stats_ws_parts": (c_int32, [_G]),
    "pnp_conv2d_fwd_stats_ws": (c_int, [_F, _F, _F, _G, c_float, c_uint64, c_uint32, _F, _F, c_size_t, c_void_p, c_size_t, c_void_p]),
    "pnp_conv2d_wino_chosen": (c_int32, [_G, c_int32]),
    "pnp_conv2d_wino_mode": (c_int32, [c_int32]),
    "pnp_conv2d_wino_wgrad_mode": (c_int32, [c_int32]),
    "pnp_conv2d_wino_tile": (c_int32, [c_int32]),
    "pnp_conv2d_wino_x3": (c_int32, [c_int32]),
    "pnp_conv2d_x3_direct": (c_int32, [c_int32]),
    "pnp_conv2d_wino_filter_bytes": (c_size_t, [c_int32, c_int32]),
    "pnp_conv2d_wino_filter_bind": (c_int, [c_void_p, c_int32, c_void_p, c_size_t]),
    "pnp_weights_changed": (None, [c_void_p, c_void_p]),
    "pnp_conv2d_wino_filter_stats": (None, [POINTER(c_int64), POINTER(c_int64), c_int32]),
    "pnp_conv2d_fwd_naive": (c_int, [_F, _F, _F, _G, c_void_p]),
    "pnp_dropout": (c_int, [_F, _F, c_size_t, c_float, c_uint64, c_uint32, c_void_p]),
    "pnp_bn_workspace_bytes": (c_size_t, [c_int64, c_int32]),
    "pnp_bn_stats": (c_int, [_F, _F, _F, c_int64, c_int32, c_void_p, c_size_t, c_void_p]),
    "pnp_bn_stats_update": (c_int, [_F, _F, _F, _F, _F, c_int64, c_int32, c_float, c_void_p, c_size_t, c_void_p]),
    "pnp_bn_update_moving": (c_int, [_F, _F, _F, _F, c_int64, c_int32, c_float, c_void_p]),
    "pnp_bn_apply": (c_int, [_F, _F, _F, _F, _F, _F, c_int32, _F, c_int64, c_int32, c_float, c_float, c_void_p]),
    "pnp_bn_bwd": (c_int, [_F, _F, _F, _F, _F, _F, _F, _F, _F, _F, _F, c_int32, c_int64, c_int32, c_float, c_float, c_int32,
                           c_float, c_uint64, c_uint32, c_void_p, c_size_t, c_void_p]),
    "pnp_bn_bwd_acc": (c_int, [_F, _F, _F, _F, _F, _F, _F, _F, _F, _F, _F, _F, _F, c_int32, c_int64, c_int32, c_float, c_float, c_int32,
                               c_float, c_uint64, c_uint32, c_void_p, c_size_t, c_void_p]),
    "pnp_bn_bwd_reduce": (c_int, [_F, _F, _F, _F, _F, _F, _F, _F, _F, c_int64, c_int32, c_float, c_float, c_void_p, c_size_t, c_void_p]),
    "pnp_bn_bwd_apply": (c_int, [_F, _F, _F, _F, _F, _F, _F, _F, _F, _F, _F, c_int32, c_int64, c_int64, c_int32, c_float, c_float, c_int32,
                                 c_float, c_uint64, c_uint32, c_void_p]),
    "pnp_maxpool2_fwd": (c_int, [_F, _F, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "pnp_maxpool2_bwd": (c_int, [_F, _F, _F, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "pnp_ps_fwd": (c_int, [_F, _F, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "pnp_ps_bwd": (c_int, [_F, _F, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "pnp_sympad_fwd": (c_int, [_F, _F, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "pnp_sympad_bwd": (c_int, [_F, _F, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "pnp_seg_loss_workspace_bytes": (c_size_t, [c_int64, c_int32]),
    "pnp_seg_loss_fwd": (c_int, [_F, _F, _F, c_int64, c_int32, c_float, c_float, c_void_p, c_size_t, c_void_p]),
    "pnp_seg_loss_bwd": (c_int, [_F, _F, _F, c_int64, c_int32, c_float, c_float, c_float, c_void_p, c_size_t, c_void_p]),
    "pnp_seg_loss_bwd_norm": (c_int, [_F, _F, _F, c_int64, c_int64, c_int32, c_float, c_float, c_float, c_void_p, c_size_t, c_void_p]),
    "pnp_softmax_argmax": (c_int, [_F, _F, c_void_p, c_int64, c_int32, c_void_p]),
    "pnp_dice_eval": (c_int, [c_void_p, _F, _F, c_int64, c_int32, c_void_p, c_size_t, c_void_p]),
    "pnp_adam_step": (c_int, [_F, _F, _F, _F, c_size_t, _F, c_void_p, c_float, c_float, c_float, c_float, c_int32, c_void_p]),
    "pnp_rmsprop_step": (c_int, [_F, _F, _F, c_size_t, _F, c_void_p, c_float, c_float, c_float, c_void_p]),
    "pnp_momentum_step": (c_int, [_F, _F, _F, c_size_t, _F, c_void_p, c_float, c_float, c_void_p]),
    "pnp_clip": (c_int, [_F, c_size_t, c_void_p, c_float, c_float, c_void_p]),
    "pnp_l2_loss": (c_int, [_F, c_size_t, _F, _F, c_void_p, c_size_t, c_void_p]),
    "pnp_reduce_workspace_bytes": (c_size_t, [c_size_t]),
    "pnp_critic_input_fwd": (c_int, [_F, c_int32, c_int32, _F, c_int32, _F, c_int32, _F, c_int32, _F, c_int32, _F, c_int64, c_void_p]),
    "pnp_critic_input_bwd": (c_int, [_F, _F, c_int32, c_int32, _F, c_int32, _F, c_int32, _F, c_int32, _F, c_int32, c_int64, c_void_p]),
    "pnp_axpby": (c_int, [_F, _F, c_size_t, c_float, c_float, c_void_p]),
    "pnp_add": (c_int, [_F, _F, _F, c_size_t, c_void_p]),
    "pnp_wgan_loss": (c_int, [_F, _F, _F, _F, c_int32, c_float, c_float, c_float, c_float, _F, c_void_p]),
    "pnp_fill": (c_int, [_F, c_size_t, c_float, c_void_p]),
    "pnp_label_decomp": (c_int, [_F, _F, c_int64, c_int32, c_void_p]),
    "pnp_confusion_matrix": (c_int, [_F, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p]),
    "pnp_bn_moments": (c_int, [_F, _F, c_void_p, c_int32, c_void_p]),
    "pnp_bn_from_moments": (c_int, [c_void_p, c_int32, _F, _F, c_int32, c_void_p]),
    "pnp_cast_bf16": (c_int, [_F, c_void_p, c_size_t, c_void_p]),
    "pnp_filter_bf16": (c_int, [_F, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "pnp_conv2d_bf16r_served": (c_int32, [_G, c_int32]),
    "pnp_conv2d_fwd_bf16r_stats_parts": (c_int32, [_G]),
    "pnp_conv2d_fwd_bf16r": (c_int, [c_void_p, c_void_p, _F, c_void_p, _G, c_float, c_uint64, c_uint32, _F, _F, c_size_t, _F, _F, _F, c_int32,
                                     c_float, c_void_p]),
    "pnp_conv2d_dgrad_bf16r": (c_int, [c_void_p, c_void_p, _F, _F, c_void_p, _G, c_void_p]),
    "pnp_bn_apply_h": (c_int, [_F, _F, _F, _F, _F, _F, c_int32, _F, c_void_p, c_int64, c_int32, c_float, c_float, c_void_p]),
    "pnp_bn_bwd_acc_h": (c_int, [_F, _F, _F, _F, _F, _F, _F, _F, c_void_p, _F, _F, _F, _F, _F, c_int32, c_int64, c_int32, c_float, c_float,
                                 c_int32, c_float, c_uint64, c_uint32, c_void_p, c_size_t, c_void_p]),
    "pnp_bn_bwd_apply_h": (c_int, [_F, _F, _F, _F, _F, _F, _F, _F, _F, _F, c_void_p, _F, c_int32, c_int64, c_int64, c_int32, c_float, c_float,
                                   c_int32, c_float, c_uint64, c_uint32, c_void_p]),
    "pnp_dropout_h": (c_int, [_F, _F, c_void_p, c_size_t, c_float, c_uint64, c_uint32, c_void_p]),
    "pnp_conv2d_wgrad_bf16r_workspace_bytes": (c_size_t, [_G]),
    "pnp_conv2d_wgrad_bf16r": (c_int, [c_void_p, c_void_p, _F, c_int32, _G, c_void_p, c_size_t, c_void_p]),
    "pnp_step_params_bind": (c_int, [c_void_p]),
    "pnp_step_params_set": (c_int, [c_void_p, c_uint64, c_float, c_void_p]),
    "pnp_comm_load": (c_int, [c_char_p]),
    "pnp_comm_version": (c_int, [POINTER(c_int)]),
    "pnp_comm_unique_id": (c_int, [c_void_p]),
    "pnp_comm_init": (c_int, [c_int32, c_int32, c_void_p, POINTER(c_void_p)]),
    "pnp_comm_allreduce": (c_int, [c_void_p, c_void_p, c_size_t, c_int32, c_void_p]),
    "pnp_comm_destroy": (c_int, [c_void_p]),
}

_lib = None


def load():
    """Load libpnp_hip.so (once).  Raises PnpError if it has not been built — never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PnpError(
            "libpnp_hip.so not found at %s — build it with `python __graft_entry__.py` "
            "(or `make -C %s/csrc`). There is no CPU fallback." % (LIB_PATH, _HERE))
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)   # AttributeError here == ABI drift; let it propagate loudly
        fn.restype = res
        fn.argtypes = args
    if lib.pnp_abi_version() != ABI_VERSION:
        raise PnpError("libpnp_hip.so ABI version %d != %d (rebuild: python __graft_entry__.py)" % (lib.pnp_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(code, what):
    if code != 0:
        msg = load().pnp_last_error()
        raise PnpError("%s failed (%d): %s" % (what, code, msg.decode() if msg else "?"))


def prof_enable(mask):
    check(load().pnp_prof_enable(int(mask)), "pnp_prof_enable")


def prof_summary(max_rows=256):
    """[{name, launches, ms, flops, bytes}] per kernel symbol since the last call (waits for the recorded events)"""
    rows = (ProfRow * max_rows)()
    n = load().pnp_prof_summary(rows, max_rows)
    return [{"name": rows[i].name.decode(), "launches": int(rows[i].launches), "ms": rows[i].ms, "flops": rows[i].flops,
             "bytes": rows[i].bytes} for i in range(min(n, max_rows))]


def device_info(device=0):
    lib = load()
    cu, clk, lds = c_int(), c_int(), c_int()
    arch = ctypes.create_string_buffer(64)
    check(lib.pnp_device_info(device, ctypes.byref(cu), ctypes.byref(clk), ctypes.byref(lds), arch, 64), "pnp_device_info")
    return {"cu_count": cu.value, "clock_khz": clk.value, "lds_bytes_per_cu": lds.value, "arch": arch.value.decode()}
