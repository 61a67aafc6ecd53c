"""ctypes front-end of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product (cova_amd/, libcovahip.so) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libcovaoracle.so")


class RefBox(C.Structure):
    _fields_ = [("left", C.c_int32), ("top", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
                ("area", C.c_int32)]


BOX_DTYPE = np.dtype([("left", "<i4"), ("top", "<i4"), ("width", "<i4"), ("height", "<i4"), ("area", "<i4")])


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("blobnet_ref.c", "ccl_ref.c")]
    if force or not os.path.exists(_SO) or any(
            os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.run(["make", "-C", _HERE, "-B", "libcovaoracle.so"], check=True, capture_output=True)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.cova_ref_blobnet_num_params.restype = C.c_size_t
        L.cova_ref_blobnet_forward.restype = C.c_int
        L.cova_ref_blobnet_forward.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                               C.c_void_p]
        L.cova_ref_blobnet_encoder_level.restype = C.c_int
        L.cova_ref_blobnet_encoder_level.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                                     C.c_void_p]
        L.cova_ref_regionprops.restype = C.c_int
        L.cova_ref_regionprops.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                           C.c_void_p, C.POINTER(C.c_int)]
        L.cova_ref_regionprops_batch.restype = None
        L.cova_ref_regionprops_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                                 C.c_void_p, C.c_int]
        L.cova_ref_metapreprocess.restype = C.c_int
        L.cova_ref_metapreprocess.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.c_int, C.c_int,
                                              C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def blobnet_forward(weights_flat: np.ndarray, stack: np.ndarray, h: int, w: int):
    """stack u8 [B][4*h][w][4] -> (logits f32 [B][h][w], mask u8 [B][h][w])."""
    wts = np.ascontiguousarray(weights_flat, dtype=np.float32)
    assert wts.size == lib().cova_ref_blobnet_num_params()
    stack = np.ascontiguousarray(stack, dtype=np.uint8)
    b = stack.shape[0]
    assert stack.shape == (b, 4 * h, w, 4)
    logits = np.empty((b, h, w), dtype=np.float32)
    mask = np.empty((b, h, w), dtype=np.uint8)
    rc = lib().cova_ref_blobnet_forward(wts.ctypes.data, h, w, stack.ctypes.data, b, logits.ctypes.data,
                                        mask.ctypes.data)
    assert rc == 0
    return logits, mask


def blobnet_encoder_level(weights_flat: np.ndarray, stack1: np.ndarray, h: int, w: int, lvl: int):
    """Encoder level output of ONE frame as f32 [Cout][4][Ho][Wo]."""
    wts = np.ascontiguousarray(weights_flat, dtype=np.float32)
    stack1 = np.ascontiguousarray(stack1, dtype=np.uint8)
    hh, ww = h, w
    for _ in range(lvl + 1):
        hh, ww = (hh + 1) // 2, (ww + 1) // 2
    co = (16, 32, 64, 128)[lvl]
    out = np.empty((co, 4, hh, ww), dtype=np.float32)
    rc = lib().cova_ref_blobnet_encoder_level(wts.ctypes.data, h, w, stack1.ctypes.data, lvl, out.ctypes.data)
    assert rc == 0
    return out


def regionprops(mask: np.ndarray, area_thresh: int, max_boxes: int | None = None, want_labels: bool = False):
    """mask u8 [H][W] -> structured array of boxes (label order, filtered)."""
    mask = np.ascontiguousarray(mask, dtype=np.uint8)
    h, w = mask.shape
    cap = max_boxes if max_boxes is not None else h * w
    boxes = np.zeros(max(cap, 1), dtype=BOX_DTYPE)
    labels = np.zeros((h, w), dtype=np.int32) if want_labels else None
    nl = C.c_int(0)
    n = lib().cova_ref_regionprops(mask.ctypes.data, h, w, area_thresh, boxes.ctypes.data, cap,
                                   labels.ctypes.data if want_labels else None, C.byref(nl))
    res = boxes[:min(n, cap)].copy()
    if want_labels:
        return res, n, labels, nl.value
    return res, n


def regionprops_batch(masks: np.ndarray, area_thresh: int, max_boxes: int):
    masks = np.ascontiguousarray(masks, dtype=np.uint8)
    b, h, w = masks.shape
    boxes = np.zeros((b, max_boxes), dtype=BOX_DTYPE)
    counts = np.zeros(b, dtype=np.int32)
    lib().cova_ref_regionprops_batch(masks.ctypes.data, b, h, w, area_thresh, boxes.ctypes.data,
                                     counts.ctypes.data, max_boxes)
    return boxes, counts


def metapreprocess(frames: np.ndarray, size_per_buf: int, t: int, gamma: int = 1):
    """frames u8 [N][stride] -> (out u8 [n_out][t*size_per_buf], src_index i32 [n_out])."""
    frames = np.ascontiguousarray(frames, dtype=np.uint8)
    n, stride = frames.shape
    out = np.zeros((max(n, 1), t * size_per_buf), dtype=np.uint8)
    idx = np.zeros(max(n, 1), dtype=np.int32)
    k = lib().cova_ref_metapreprocess(frames.ctypes.data, n, stride, size_per_buf, t, gamma, out.ctypes.data,
                                      idx.ctypes.data)
    return out[:k].copy(), idx[:k].copy()
