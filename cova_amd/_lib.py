"""ctypes binding of libcovahip.so (the C-ABI declared in include/covahip.h).

The product path has no CPU fallback: if the shared library (built by
`__graft_entry__.build()` / `make -C cova_amd/csrc`) is missing this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcovahip.so")

OK = 0
MEM_HOST, MEM_DEVICE = 0, 1
AU_DELTA_UNIT, AU_DISCONT, AU_DROPPABLE = 1, 2, 4


class CovahipError(RuntimeError):
    def __init__(self, status: int, where: str, detail: str = ""):
        self.status = status
        try:
            text = lib().covahip_strerror(status).decode()
        except AttributeError:            # the sanitizer build of the host units (COVAHIP_HOST_SAN_LIB) has no covahip_strerror
            text = "error"
        msg = f"{where}: {text} (status {status})"
        if detail:
            msg += f" [{detail}]"
        super().__init__(msg)


class Box(C.Structure):
    _fields_ = [("left", C.c_int32), ("top", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
                ("area_px", C.c_int32)]


BOX_DTYPE = np.dtype([("left", "<i4"), ("top", "<i4"), ("width", "<i4"), ("height", "<i4"), ("area_px", "<i4")])

BBOX_DTYPE = np.dtype([
    ("left", "<f4"), ("top", "<f4"), ("width", "<f4"), ("height", "<f4"), ("area", "<f4"),
    ("_pad0", "<u4"),
    ("track_id", "<u8"), ("timestamp", "<u8"), ("class_id", "<u4"), ("confidence", "<f4"),
    ("has_track_id", "u1"), ("has_timestamp", "u1"), ("has_class_id", "u1"), ("has_confidence", "u1"),
    ("_pad1", "<u4"),
])
assert BBOX_DTYPE.itemsize == 56

AU_OUT_DTYPE = np.dtype([("id", "<u8"), ("pts", "<u8"), ("flags", "<u4"), ("list", "<u4")])

H264_INFO_DTYPE = np.dtype([(n, "<i4") for n in ("width_mbs", "height_mbs", "n_samples", "profile_idc", "level_idc", "entropy_cabac",
                                                "transform_8x8", "num_ref_frames", "frame_mbs_only", "weighted_pred",
                                                "weighted_bipred", "poc_type", "max_num_reorder_frames", "max_dec_frame_buffering")])
H264_SLICE_DTYPE = np.dtype([("nal_offset", "<u8"), ("nal_bytes", "<u4"), ("data_bit_offset", "<u4")] +
                            [(n, "<i4") for n in ("nal_type", "slice_type", "first_mb", "frame_num", "idr", "poc_lsb", "qp",
                                                  "cabac_init_idc", "num_ref_l0", "num_ref_l1", "direct_spatial", "nal_ref_idc", "has_mmco5")] + [("_pad", "<u4")])

KERNEL_TIME_DTYPE = np.dtype([("name", "S48"), ("total_ms", "<f8"), ("launches", "<i8")])


class GopFilterCfg(C.Structure):
    _fields_ = [("sort_iou", C.c_float), ("sort_maxage", C.c_uint32), ("sort_minhits", C.c_uint32),
                ("alpha", C.c_uint32), ("beta", C.c_uint32), ("infer_i", C.c_uint8)]


class AssocCfg(C.Structure):
    _fields_ = [("moving_iou", C.c_float), ("stationary_iou", C.c_float), ("stationary_maxage_s", C.c_uint64),
                ("scale_factor", C.c_float)]


# name -> (restype, argtypes); every symbol include/covahip.h declares
_P = C.c_void_p
_SZ = C.c_size_t
PROTOTYPES = {
    "covahip_strerror": (C.c_char_p, [C.c_int]),
    "covahip_version": (C.c_char_p, []),
    "covahip_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "covahip_device_pci_bus_id": (C.c_int, [C.c_int, C.c_char_p, C.c_int]),
    "covahip_ctx_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "covahip_ctx_destroy": (None, [_P]),
    "covahip_ctx_sync": (C.c_int, [_P]),
    "covahip_ctx_set_lanes": (C.c_int, [_P, C.c_int]),
    "covahip_ctx_get_lanes": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "covahip_last_hip_error": (C.c_char_p, [_P]),
    "covahip_device_info": (C.c_int, [_P, C.c_char_p, _SZ, C.POINTER(C.c_int), C.POINTER(_SZ)]),
    "covahip_malloc": (C.c_int, [_P, _SZ, C.POINTER(_P)]),
    "covahip_free": (C.c_int, [_P, _P]),
    "covahip_memcpy_h2d": (C.c_int, [_P, _P, _P, _SZ]),
    "covahip_memcpy_d2h": (C.c_int, [_P, _P, _P, _SZ]),
    "covahip_memset": (C.c_int, [_P, _P, C.c_int, _SZ]),
    "covahip_timer_start": (C.c_int, [_P, C.c_int]),
    "covahip_timer_stop": (C.c_int, [_P, C.c_int]),
    "covahip_timer_elapsed_ms": (C.c_int, [_P, C.c_int, C.POINTER(C.c_float)]),
    "covahip_profile_enable": (C.c_int, [_P, C.c_int]),
    "covahip_profile_filter": (C.c_int, [_P, C.c_char_p]),
    "covahip_profile_reset": (C.c_int, [_P]),
    "covahip_profile_read": (C.c_int, [_P, _P, C.c_int, C.POINTER(C.c_int)]),
    "covahip_blobnet_load": (C.c_int, [_P, _P, _SZ, C.c_int, C.c_int, C.c_int, C.c_int]),
    "covahip_blobnet_forward": (C.c_int, [_P, _P, C.c_int, _P, _P, C.c_int]),
    "covahip_blobnet_macs_per_frame": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "covahip_bboxcc": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, C.c_int]),
    "covahip_filter_forward": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P, C.c_int, _P, _P, C.c_int]),
    "covahip_filter_forward_frames": (C.c_int, [_P, _P, C.c_int, _P, C.c_int, C.c_int, _P, _P, C.c_int, _P, _P, C.c_int]),
    "covahip_pipe_create": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "covahip_pipe_destroy": (None, [_P]),
    "covahip_pipe_set_packed": (C.c_int, [_P, C.c_int]),
    "covahip_pipe_set_blocking_wait": (C.c_int, [_P, C.c_int]),
    "covahip_pipe_acquire": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(_P), C.POINTER(_P)]),
    "covahip_pipe_submit": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int]),
    "covahip_pipe_wait": (C.c_int, [_P, C.c_int]),
    "covahip_pipe_abort": (C.c_int, [_P, C.c_int]),
    "covahip_pipe_collect": (C.c_int, [_P, C.c_int, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(_P)]),
    "covahip_pipe_release": (C.c_int, [_P, C.c_int]),
    "covahip_boxes_to_bbox": (None, [_P, C.c_int, _P]),
    "covahip_bbox_serialize_vec": (_SZ, [_P, _SZ, _P, _SZ, C.POINTER(C.c_int)]),
    "covahip_bbox_deserialize_vec": (C.c_int, [_P, _SZ, _P, _SZ, C.POINTER(_SZ)]),
    "covahip_frame_serialize": (_SZ, [C.c_uint64, C.c_uint64, _P, _SZ, _P, _SZ, C.POINTER(C.c_int)]),
    "covahip_bbox_iou": (C.c_float, [_P, _P]),
    "covahip_tfrecord_example": (_SZ, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _SZ, C.POINTER(C.c_int)]),
    "covahip_bbox_csv": (_SZ, [_P, _SZ, C.c_int, _P, _SZ, C.POINTER(C.c_int)]),
    "covahip_tracks_export": (_SZ, [C.c_uint64, C.c_uint64, _P, _P, _SZ, _P, _SZ, C.POINTER(C.c_int)]),
    "covahip_assoc_default_cfg": (None, [C.POINTER(AssocCfg)]),
    "covahip_assoc_new": (C.c_int, [C.POINTER(AssocCfg), _P, _SZ, C.POINTER(_P)]),
    "covahip_assoc_free": (None, [_P]),
    "covahip_assoc_push_track": (C.c_int, [_P, C.c_uint64, C.c_uint64, _P, _SZ]),
    "covahip_assoc_push_track_frame": (C.c_int, [_P, _P, _SZ]),
    "covahip_assoc_push_dnn": (C.c_int, [_P, _P, _SZ]),
    "covahip_assoc_push_dnn_text": (C.c_int, [_P, C.c_char_p, _SZ]),
    "covahip_assoc_terminate": (C.c_int, [_P]),
    "covahip_assoc_csv": (_SZ, [_P, C.c_int, _P, _SZ, C.POINTER(C.c_int)]),
    "covahip_h264_open_mp4": (C.c_int, [_P, _SZ, C.POINTER(_P)]),
    "covahip_h264_close": (None, [_P]),
    "covahip_h264_get_info": (C.c_int, [_P, _P]),
    "covahip_h264_sample": (C.c_int, [_P, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_int)]),
    "covahip_h264_sample_slices": (C.c_int, [_P, C.c_int, _P, C.c_int, C.POINTER(C.c_int)]),
    "covahip_h264_decode_records": (C.c_int, [_P, C.c_int, _P, _SZ]),
    "covahip_h264_display_order": (C.c_int, [_P, C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "covahip_h264_colocated": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "covahip_h264_open_avcc": (C.c_int, [_P, _SZ, C.POINTER(_P)]),
    "covahip_h264_decode_au": (C.c_int, [_P, _P, _SZ, _P, _SZ, _P, C.POINTER(C.c_int64)]),
    "covahip_carrier_write_records": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P, _SZ]),
    "covahip_carrier_pack": (None, [_P, _SZ, _P]),
    "covahip_filter_forward_frames_packed": (C.c_int, [_P, _P, C.c_int, _P, C.c_int, C.c_int, _P, _P, C.c_int, _P, _P]),
    "covahip_stack_new": (C.c_int, [_SZ, C.c_uint, C.c_uint, C.POINTER(_P)]),
    "covahip_stack_free": (None, [_P]),
    "covahip_stack_push": (C.c_int, [_P, _P, _SZ, _P, _SZ, C.POINTER(C.c_int)]),
    "covahip_stack_out_dims": (None, [C.c_int, C.c_int, C.c_uint, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "covahip_sort_new": (C.c_int, [C.c_uint64, C.c_uint64, C.c_float, C.POINTER(_P)]),
    "covahip_sort_free": (None, [_P]),
    "covahip_sort_update": (C.c_int, [_P, _P, _SZ, C.c_uint64, _P, _SZ, C.POINTER(_SZ), _P, _SZ, C.POINTER(_SZ)]),
    "covahip_sort_finalize": (C.c_int, [_P, _P, _SZ, C.POINTER(_SZ), _P, _SZ, C.POINTER(_SZ)]),
    "covahip_sort_mark_seen": (C.c_int, [_P, C.c_uint64]),
    "covahip_sort_num_trackers": (C.c_int, [_P, C.POINTER(_SZ)]),
    "covahip_sort_tracker_info": (C.c_int, [_P, _SZ, C.POINTER(C.c_uint64), C.POINTER(C.c_int),
                                            C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), _P]),
    "covahip_sort_tracker_predict": (C.c_int, [_P, _SZ, C.c_uint64, _P]),
    "covahip_sort_tracker_update": (C.c_int, [_P, _SZ, _P]),
    "covahip_linear_assignment": (_SZ, [_P, _SZ, _SZ, _P, _SZ]),
    "covahip_gopfilter_default_cfg": (None, [C.POINTER(GopFilterCfg)]),
    "covahip_gopfilter_new": (C.c_int, [C.POINTER(GopFilterCfg), C.POINTER(_P)]),
    "covahip_gopfilter_free": (None, [_P]),
    "covahip_gopfilter_push_enc": (C.c_int, [_P, C.c_uint64, C.c_uint64, C.c_uint32]),
    "covahip_gopfilter_push_boxes": (C.c_int, [_P, _P, _SZ, C.c_uint64, _P, _SZ, C.POINTER(_SZ)]),
    "covahip_gopfilter_eos": (C.c_int, [_P, _P, _SZ, C.POINTER(_SZ)]),
    "covahip_gopfilter_take_dropped": (C.c_int, [_P, _P, _SZ, C.POINTER(_SZ)]),
    "covahip_gopfilter_take_track_export": (_SZ, [_P, _P, _SZ, C.POINTER(C.c_int)]),
    "covahip_gopfilter_counters": (C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                             C.POINTER(C.c_uint64)]),
}

# developer switches (include/covahip_dev.h): bound for tools/ and tests/, not part of the drop-in boundary
DEV_PROTOTYPES = {
    "covahip_blobnet_set_impl": (C.c_int, [_P, C.c_int]),
    "covahip_bboxcc_set_wave_cap": (C.c_int, [_P, C.c_int]),
    "covahip_blobnet_set_enc_plan": (C.c_int, [_P, C.c_int, C.c_int, C.c_int]),
    "covahip_dev_clock_mhz": (C.c_int, [_P, C.c_int, C.POINTER(C.c_float)]),
    "covahip_dev_graph_probe": (C.c_int, [_P, _P, C.c_int, _P, C.c_int, C.c_int, _P, _P, C.c_int, _P, C.c_int, C.POINTER(C.c_float),
                                          C.POINTER(C.c_float)]),
    "covahip_dev_bboxcc_overflow": (C.c_int, [_P, C.POINTER(C.c_int32)]),
    "covahip_dev_pipe_queue_plan": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
}

_lib = None


def lib() -> C.CDLL:
    """Loads libcovahip.so; raises (no fallback) when it has not been built."""
    global _lib
    if _lib is None:
        san = os.environ.get("COVAHIP_HOST_SAN_LIB")
        if san:
            # TEST HOOK (tests/test_sanitize_host.py): the CPU-only ASan / UBSan build of the HIP-free translation units.  It has
            # no GPU entry point at all -- those names stay unbound and any use of one raises AttributeError.
            L = C.CDLL(san)
            for name, (res, args) in list(PROTOTYPES.items()) + list(DEV_PROTOTYPES.items()):
                fn = getattr(L, name, None)
                if fn is not None:
                    fn.restype = res
                    fn.argtypes = args
            _lib = L
            return _lib
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C cova_amd/csrc` -- cova_amd has no CPU fallback")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in list(PROTOTYPES.items()) + list(DEV_PROTOTYPES.items()):
            fn = getattr(L, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(status: int, where: str, ctx=None) -> None:
    if status != OK:
        detail = ""
        if ctx is not None and status == 3:
            detail = lib().covahip_last_hip_error(ctx).decode()
        raise CovahipError(status, where, detail)
