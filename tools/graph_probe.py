"""Developer helper, ON THE GPU BOX: one carrier-frame step as six stream launches vs as launches of one captured HIP graph
(covahip_dev_graph_probe): what the gaps between the launches of a step cost."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cova_amd import synth, weights as W  # noqa: E402
from cova_amd.elements import BlobNetInfer, Context  # noqa: E402
from cova_amd import _lib as L  # noqa: E402

B, H, Wd = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 68, 120
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
ctx = Context(0)
net = BlobNetInfer(ctx, W.random_init(1234), H, Wd, max_batch=B)
frames, index = synth.carrier_batch(B, H, Wd, seed=1, streams=8)
d_frames = ctx.malloc(frames.nbytes)
ctx.h2d(d_frames, frames)
d_boxes, d_counts, d_mask = ctx.malloc(B * 256 * 20), ctx.malloc(B * 4), ctx.malloc(B * H * Wd)
lib = L.lib()
idx = np.ascontiguousarray(index, dtype=np.int32)
for rep in range(3):
    a, b = C.c_float(), C.c_float()
    rc = lib.covahip_dev_graph_probe(ctx.handle, d_frames, frames.shape[0], idx.ctypes.data, B, 1, d_boxes, d_counts, 256, d_mask, iters,
                                     C.byref(a), C.byref(b))
    print(f"rc={rc}: stream launches {a.value / iters * 1e3:.1f} us per step, graph launches {b.value / iters * 1e3:.1f} us per step")
