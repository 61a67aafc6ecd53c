"""Generates tests/golden/demo_records_excerpt.npz: the per-macroblock records of 64 consecutive OUTPUT pictures of the reference's
demo/1m.mp4 (pictures 300 .. 363 in output order: a key frame at 500 is not in it, a vehicle crosses the scene), as this build's
entropy-decode front end produces them (covahip_h264_decode_records, cova_amd/csrc/h264_cabac.cpp).  Data derived from the
reference's demo video, so that the GPU box -- which has no /root/reference -- can run the hot path on real compressed-domain input.
Run here (CPU): python tests/golden/gen_demo_records.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cova_amd import _lib as L  # noqa: E402

lib = L.lib()
data = np.fromfile("/root/reference/demo/1m.mp4", dtype=np.uint8)
h = C.c_void_p()
assert lib.covahip_h264_open_mp4(data.ctypes.data, data.size, C.byref(h)) == 0
order = np.zeros(1802, np.int32)
n = C.c_int()
assert lib.covahip_h264_display_order(h, order.ctypes.data, 1802, C.byref(n)) == 0
first, count = 300, 64
rec = np.zeros((count, 45, 80, 4), np.uint8)
for k in range(count):
    assert lib.covahip_h264_decode_records(h, int(order[first + k]), rec[k].ctypes.data, rec[k].nbytes) == 0
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "demo_records_excerpt.npz"), records=rec,
                    samples=order[first:first + count])
print(rec.shape, os.path.getsize(os.path.join(ROOT, "tests", "golden", "demo_records_excerpt.npz")), "bytes")
