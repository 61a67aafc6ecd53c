"""Developer helper: where the host-buffer (PCIe-inclusive) call of the hot path spends its time."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cova_amd import synth, weights as W
from cova_amd.elements import BlobNetInfer, Context
B, H, Wd, MB = 256, 68, 120, 2048
ctx = Context(0)
net = BlobNetInfer(ctx, W.random_init(1234), H, Wd, max_batch=B)
stack = synth.stacked_batch(64, H, Wd, seed=1, streams=8)
stack = np.ascontiguousarray(np.concatenate([stack] * 4))
d_stack = ctx.malloc(stack.nbytes)
boxes = np.zeros((B, MB), dtype=np.dtype([("a", "i4", 5)]))
d_boxes = ctx.malloc(boxes.nbytes); d_counts = ctx.malloc(B * 4)
def t(fn, n=5):
    fn(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    ctx.sync()
    return (time.perf_counter() - t0) / n * 1e3
print("H2D stack %.1f MB: %.3f ms" % (stack.nbytes / 1e6, t(lambda: ctx.h2d(d_stack, stack))))
print("compute (device pointers): %.3f ms" % t(lambda: net.filter_device(d_stack, B, 1, d_boxes, d_counts, MB)))
print("D2H boxes %.1f MB: %.3f ms" % (boxes.nbytes / 1e6, t(lambda: ctx.d2h(boxes, d_boxes))))
print("host-buffer call: %.3f ms" % t(lambda: net.filter(stack, 1, max_boxes=MB)))
