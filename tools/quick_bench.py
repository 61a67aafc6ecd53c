"""Developer helper: per-kernel times of the hot path at a given batch (device-resident inputs)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cova_amd import synth, weights as W  # noqa: E402
from cova_amd.elements import BlobNetInfer, Context  # noqa: E402
from cova_amd import _lib as L  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
H, Wd = 68, 120
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ctx = Context(0)
ctx.set_lanes(int(os.environ.get("QB_LANES", "1")))      # per-kernel times want one step after the other
flat = W.random_init(1234)
if os.environ.get("QB_NEG_GAMMA"):          # mixed-sign BN gammas: the non-ALLPOS kernel variants
    _p = W.unflatten(flat.copy())
    for _i in range(4):
        _p[f"enc{_i}.bn.gamma"][::2] *= -1.0
    flat = W.flatten(_p)
net = BlobNetInfer(ctx, flat, H, Wd, max_batch=B)
if os.environ.get("QB_IMPL"):
    net.set_impl(os.environ["QB_IMPL"])
for spec in filter(None, os.environ.get("QB_PLAN", "").split(",")):      # level:nbands:nbuf (covahip_blobnet_set_enc_plan)
    net.set_enc_plan(*[int(x) for x in spec.split(":")])
FRAMES = os.environ.get("QB_INPUT", "stack") == "frames"     # carrier-frame entry point instead of stacks
if FRAMES:
    frames, index = synth.carrier_batch(B, H, Wd, seed=1, streams=8)
    d_frames = ctx.malloc(frames.nbytes)
    ctx.h2d(d_frames, frames)
else:
    stack = synth.stacked_batch(min(B, 64), H, Wd, seed=1, streams=8)
    stack = np.concatenate([stack] * (B // stack.shape[0] + 1))[:B]
    d_stack = ctx.malloc(stack.nbytes)
    ctx.h2d(d_stack, stack)
max_boxes = 256
d_boxes = ctx.malloc(B * max_boxes * 20)
d_counts = ctx.malloc(B * 4)
d_mask = ctx.malloc(B * H * Wd)


def step():
    if FRAMES:
        net.filter_frames_device(d_frames, frames.shape[0], index, B, 1, d_boxes, d_counts, max_boxes, d_mask)
    else:
        net.filter_device(d_stack, B, 1, d_boxes, d_counts, max_boxes, d_mask)


t_w = time.perf_counter()
n_w = 0
while n_w < 3 or time.perf_counter() - t_w < float(os.environ.get("QB_WARM_S", "0.3")):   # an idle chip's clocks take ~0.2 s to come up
    step()
    n_w += 1
    if n_w % 32 == 0:
        ctx.sync()
ctx.sync()
ctx.timer_start(0)
for _ in range(steps):
    step()
ctx.timer_stop(0)
ms = ctx.timer_ms(0) / steps
macs = net.macs_per_frame
print(f"B={B}: {ms*1e3:.1f} us/batch  {B/ms*1e3:.0f} frames/s  MFMA util (alg) {2*macs*B/(ms*1e-3)/2.5e15*100:.1f}%")
if len(sys.argv) > 3 and sys.argv[3] == "noprofile":
    sys.exit(0)
ctx.profile(True)
for _ in range(steps):
    step()
ctx.sync()
prof = ctx.profile_read()
tot = 0
for k, (t, n) in sorted(prof.items()):
    print(f"  {k:16s} {t/n*1e3:9.1f} us  x{n}")
    tot += t / n
print(f"  sum {tot*1e3:.1f} us")
