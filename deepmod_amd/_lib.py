"""ctypes binding of libdeepmod_hip.so (include/deepmod_hip.h).  No fallbacks: if the HIP
library is missing or no gfx950 device is usable, every entry point raises."""
from __future__ import annotations

import ctypes
import os
from typing import Optional

# Multi-process GPU work on this stack (RCCL communicators between the ranks of a node) needs the ROCr runtime in its dmabuf IPC mode: the
# host driver does not support the legacy IPC handles, and with them hipIpcGetMemHandle - hence ncclCommInitRank's buffer exchange -
# fails with "invalid argument".  The runtime reads the variable when it initialises (the first HIP call of the process), so it is set
# here, before the library is loaded, unless the caller has decided otherwise.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "csrc", "libdeepmod_hip.so")

DM_OPT_PROFILE = 1
DM_OPT_PRECISION = 2
DM_OPT_ASYNC = 3
DM_OPT_RESERVED_CUS = 4
DM_OPT_F16X3_SHAPE = 5   # 16 (the product: 16x16x32 MFMAs) | 32 (experiment builds only: the 32x32x16 kernels of rounds 2-3)
DM_PREC_F32 = 0
DM_PREC_F16X3 = 1
DM_PREC_F16I8 = 3      # opt-in, reduced precision: int8 cross terms (2 issued matrix units per product instead of 3; worst window of 1e6 1.1e-4 instead of 9e-6 at weight scale 4)
DM_PREC_F16X3_ROLES = 2   # round-4 experiment (matrix / cell wave pairs, bit-identical to F16X3): only in a library built with -DDM_WITH_F16X3_ROLES (DM_INFO_HAS_F16X3_ROLES)
DM_INFO_PRECISION, DM_INFO_F16_REPRESENTABLE, DM_INFO_F16_LENGTH_SHIFT, DM_INFO_DEVICE, DM_INFO_HAS_F16X3_ROLES = 1, 2, 3, 4, 5
DM_INFO_HAS_F16S = 6         # 1: experiment build with the 32x32x16 kernels of rounds 2-3 (DM_WITH_F16S=1)
DM_OK, DM_EINVAL, DM_EDEVICE, DM_ENOMEM, DM_ESTATE, DM_ERCCL, DM_ERANGE = 0, -1, -2, -3, -4, -5, -6
(DM_MAP_STATUS, DM_MAP_N_ROWS, DM_MAP_LEFTCLIP, DM_MAP_RIGHTCLIP, DM_MAP_EV_LO, DM_MAP_EV_HI, DM_MAP_FIRST_MATCH_POS,
 DM_MAP_LAST_MATCH_POS, DM_MAP_NUM_INSERT, DM_MAP_NUM_DELETE, DM_MAP_NUM_MISMATCH, DM_MAP_STRAND, DM_MAP_POS_AFTER_CLIP,
 DM_MAP_EVENTS_AFTER_CLIP) = range(14)
DM_MAP_INFO_LEN = 16
DM_MAP_OK, DM_MAP_NO_MATCH, DM_MAP_NEED_ROWS = 0, 1, 2
DM_WEIGHT_FLOATS = 408402

_c = ctypes
_vp = ctypes.c_void_p
_i64 = ctypes.c_int64

# (name, restype, argtypes) — mirrors include/deepmod_hip.h one to one
SIGNATURES = [
    ("dm_last_error", _c.c_char_p, []),
    ("dm_version", _c.c_char_p, []),
    ("dm_build_flags", _c.c_char_p, []),
    ("dm_device_count", _c.c_int, []),
    ("dm_device_pci_bus_id", _c.c_int, [_c.c_int, _c.c_char_p, _c.c_int]),
    ("dm_model_create", _vp, [_c.c_int, _vp, _c.c_size_t, _c.c_int, _c.c_int, _c.c_int, _c.c_int]),
    ("dm_model_destroy", None, [_vp]),
    ("dm_model_set_option", _c.c_int, [_vp, _c.c_int, _i64]),
    ("dm_model_get_info", _c.c_int, [_vp, _c.c_int, _c.POINTER(_i64)]),
    ("dm_model_calibrate_i8", _c.c_int, [_vp, _i64, _c.c_double, _c.POINTER(_c.c_double), _c.POINTER(_c.c_int)]),
    ("dm_predict_windows", _c.c_int, [_vp, _vp, _i64, _vp, _vp]),
    ("dm_predict_read", _c.c_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp]),
    ("dm_predict_read_at", _c.c_int, [_vp, _vp, _i64, _vp, _i64, _vp, _vp]),
    ("dm_model_sync", _c.c_int, [_vp]),
    ("dm_profile_reset", _c.c_int, [_vp]),
    ("dm_profile_get", _c.c_int, [_vp, _c.POINTER(_c.c_double), _c.POINTER(_i64), _c.POINTER(_i64)]),
    ("dm_device_alloc", _vp, [_c.c_int, _c.c_size_t]),
    ("dm_device_free", _c.c_int, [_c.c_int, _vp]),
    ("dm_memcpy_h2d", _c.c_int, [_c.c_int, _vp, _vp, _c.c_size_t]),
    ("dm_memcpy_d2h", _c.c_int, [_c.c_int, _vp, _vp, _c.c_size_t]),
    ("dm_model_h2d_async", _c.c_int, [_vp, _vp, _vp, _c.c_size_t]),
    ("dm_model_h2d_ahead", _c.c_int, [_vp, _vp, _vp, _c.c_size_t]),
    ("dm_host_alloc", _vp, [_c.c_int, _c.c_size_t]),
    ("dm_host_free", _c.c_int, [_c.c_int, _vp]),
    ("dm_model_mark", _c.c_int, [_vp, _c.c_int]),
    ("dm_model_wait_mark", _c.c_int, [_vp, _c.c_int]),
    ("dm_summary_create", _vp, [_c.c_int, _i64]),
    ("dm_summary_destroy", None, [_vp]),
    ("dm_summary_length", _i64, [_vp]),
    ("dm_summary_add", _c.c_int, [_vp, _vp, _vp, _i64]),
    ("dm_summary_add_classified", _c.c_int, [_vp, _vp, _vp, _vp, _i64]),
    ("dm_summary_sync", _c.c_int, [_vp]),
    ("dm_summary_grow", _c.c_int, [_vp, _i64]),
    ("dm_rccl_unique_id", _c.c_int, [_vp]),
    ("dm_rccl_info", _c.c_int, [_c.c_char_p, _c.c_int, _c.POINTER(_c.c_int)]),
    ("dm_comm_create", _vp, [_c.c_int, _vp, _c.c_int, _c.c_int]),
    ("dm_comm_destroy", None, [_vp]),
    ("dm_comm_rank", _c.c_int, [_vp]),
    ("dm_comm_size", _c.c_int, [_vp]),
    ("dm_comm_barrier", _c.c_int, [_vp]),
    ("dm_comm_max_f64", _c.c_int, [_vp, _c.POINTER(_c.c_double)]),
    ("dm_comm_stats", _c.c_int, [_vp, _c.POINTER(_i64), _c.POINTER(_i64)]),
    ("dm_summary_reduce", _c.c_int, [_vp, _vp, _c.c_int]),
    ("dm_summary_reduce_scatter", _c.c_int, [_vp, _vp, _c.POINTER(_i64), _c.POINTER(_i64)]),
    ("dm_summary_fetch_slice", _c.c_int, [_vp, _vp, _vp, _vp]),
    ("dm_summary_fetch", _c.c_int, [_vp, _vp, _vp, _vp]),
    ("dm_summary_device_ptr", _vp, [_vp]),
    ("dm_summary_follow", _c.c_int, [_vp, _vp]),
    ("dm_bed_format", _i64, [_c.c_char_p, _c.c_char, _c.c_char, _vp, _vp, _vp, _i64, _vp, _i64]),
    ("dm_bed_format_at", _i64, [_c.c_char_p, _c.c_char, _c.c_char, _i64, _vp, _vp, _vp, _i64, _vp, _i64]),
    ("dm_cluster_create", _vp, [_c.c_int, _vp, _c.c_size_t]),
    ("dm_cluster_destroy", None, [_vp]),
    ("dm_cluster_predict", _c.c_int, [_vp, _vp, _i64, _vp]),
    ("dm_signal_create", _vp, [_c.c_int]),
    ("dm_signal_destroy", None, [_vp]),
    ("dm_map_read", _c.c_int, [_c.c_int, _i64, _c.c_char_p, _vp, _i64, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _i64, _vp]),
    ("dm_signal_event_stats", _c.c_int, [_vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _c.POINTER(_i64), _vp]),
    ("dm_signal_event_stats_batch", _c.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("dm_signal_plan_batch", _c.c_int, [_i64, _vp, _vp, _vp, _vp, _vp]),
    ("dm_signal_event_stats_device", _c.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("dm_events_merge", _i64, [_i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _c.c_int32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("dm_rows_create", _vp, [_c.c_char]),
    ("dm_rows_destroy", None, [_vp]),
    ("dm_rows_add_packed", _c.c_int, [_vp, _i64, _i64, _i64, _i64, _c.c_int32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("dm_rows_add_raw", _c.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _c.c_int32, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                   _vp, _c.c_int32, _vp, _vp, _vp]),
    ("dm_rows_add_mapped", _c.c_int, [_vp, _i64, _i64, _i64, _c.c_int32] + [_vp] * 16),
    ("dm_rows_info", _i64, [_vp, _c.POINTER(_i64), _c.POINTER(_i64), _c.POINTER(_i64), _vp, _vp, _i64, _c.POINTER(_i64)]),
    ("dm_rows_emit", _i64, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _c.POINTER(_c.c_int32)]),
    ("dm_rows_device_info", _c.c_int, [_vp, _c.POINTER(_i64), _c.POINTER(_i64)]),
    ("dm_rows_emit_device", _i64, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _c.POINTER(_c.c_int32)]),
    ("dm_rows_emit_resident", _i64, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _c.POINTER(_c.c_int32)]),
    ("dm_rows_assemble", _c.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64]),
]


class DeepModHipError(RuntimeError):
    code = None


class DeepModRangeError(DeepModHipError):
    """DM_ERANGE: the split-f16 kernel met an input it cannot represent; repeat the call with precision 'f32'."""
    code = DM_ERANGE


_LIB: Optional[ctypes.CDLL] = None


def load() -> ctypes.CDLL:
    """Load the in-tree HIP library (built by __graft_entry__.build()).  Raises if absent."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise DeepModHipError(
                "%s not found - build it first (python -c 'import __graft_entry__ as g; g.build()'); "
                "there is no CPU fallback" % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        for name, restype, argtypes in SIGNATURES:
            fn = getattr(lib, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _LIB = lib
    return _LIB


def check(rc: int) -> None:
    if rc != 0:
        msg = "deepmod_hip error %d: %s" % (rc, load().dm_last_error().decode("utf-8", "replace"))
        if rc == DM_ERANGE:
            raise DeepModRangeError(msg)
        err = DeepModHipError(msg)
        err.code = rc
        raise err


def pci_bus_id(device: int) -> str:
    buf = ctypes.create_string_buffer(64)
    check(load().dm_device_pci_bus_id(device, buf, 64))
    return buf.value.decode()


def last_error() -> str:
    return load().dm_last_error().decode("utf-8", "replace")
