"""Developer helper (CPU; needs the reference's demo/1m.mp4): access units per second of the entropy-decode front end on ONE thread
-- the stream form (covahip_h264_open_avcc / decode_au is what the h264entropydec element runs; the file form decodes by sample
index).  usage: python tools/h264_rate.py [/root/reference/demo/1m.mp4]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cova_amd import _lib as L  # noqa: E402

path = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/demo/1m.mp4"
data = np.fromfile(path, dtype=np.uint8)
lib = L.lib()
h = C.c_void_p()
assert lib.covahip_h264_open_mp4(data.ctypes.data, data.nbytes, C.byref(h)) == 0
n = 1802
rec = np.zeros((45, 80, 4), np.uint8)
best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    for s in range(n):
        assert lib.covahip_h264_decode_records(h, s, rec.ctypes.data, rec.nbytes) == 0
    best = min(best, time.perf_counter() - t0)
print(f"{n} access units (1280x720, High@3.1, CABAC; 8 I / 564 P / 1,230 B) in {best:.3f} s on one thread: {n / best:.0f} frames/s per thread "
      f"({data.nbytes / best / 1e6:.1f} MB/s of container bytes)")
