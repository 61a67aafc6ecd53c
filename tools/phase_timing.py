"""Developer helper, ON THE GPU BOX with a -DPHASE_TIMING build copied over cova_amd/libcovahip.so (see tools/README.md):
wall-clock share of each phase of enc_mfma's item loop (wave 0 of every workgroup, summed), levels 1..3."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cova_amd import synth, weights as W  # noqa: E402
from cova_amd.elements import BlobNetInfer, Context  # noqa: E402
from cova_amd import _lib as L  # noqa: E402

B, H, Wd = 256, 68, 120
ctx = Context(0)
net = BlobNetInfer(ctx, W.random_init(1234), H, Wd, max_batch=B)
if os.environ.get('QB_IMPL'):
    net.set_impl(os.environ['QB_IMPL'])
for spec in filter(None, os.environ.get("QB_PLAN", "").split(",")):      # level:nbands:nbuf
    net.set_enc_plan(*[int(x) for x in spec.split(":")])
NWG = {0: int(os.environ.get("NWG1", 512)), 16: 512, 32: 256, 48: 256, 64: 256}
TAIL_NAMES = ["barrier (previous frame)", "requesting band 0 + weights", "bands landing", "requesting the next band", "tiles", "barrier", "mask out", "bboxcc", "-"]
DEC_NAMES = ["barrier (previous frame)", "requesting the three tiles + block 0 weights", "tiles landing", "block 0: tiles",
             "block 1: weights + barrier", "block 1: tiles", "block 2: weights + barrier", "block 2: tiles + stores", "-"]
frames, index = synth.carrier_batch(B, H, Wd, seed=1, streams=8)
d_frames = ctx.malloc(frames.nbytes)
ctx.h2d(d_frames, frames)
d_boxes, d_counts, d_mask = ctx.malloc(B * 256 * 20), ctx.malloc(B * 4), ctx.malloc(B * H * Wd)
lib = ctypes.CDLL(L.LIB_PATH)
out = (ctypes.c_ulonglong * 80)()
names = ["bookkeeping", "barrier (previous item)", "DMA issue", "band landing", "temporal MLP in place", "skip slice out", "tiles: matrix part", "tiles: epilogue", "weights into registers (kernel start)"]
E23_NAMES = ["frame start: barrier, first six rows", "level 2: barrier A (+ weights at step 0)", "level 2: products, pooling", "level 2: barrier B", "level 2: next rows into the ring", "level 2: temporal MLP, band writes", "between the levels: weights, barrier, T = 0 out", "level 3: products + pooling", "level 3: epilogue"]
E1V_NAMES = ["bookkeeping, addresses, loads issued", "barrier (previous item)", "loads landing", "temporal MLP, LDS writes, skip stores", "barrier (band complete)", "weights landing", "tiles: matrix part", "tiles: epilogue", "partial logits (tail skip half)"]
steps = 20
for _ in range(3):
    net.filter_frames_device(d_frames, frames.shape[0], index, B, 1, d_boxes, d_counts, 256, d_mask)
ctx.sync()
lib.covahip_dev_phase_read(out, 1)
cc = (ctypes.c_ulonglong * 8)()
lib.covahip_dev_ccphase_read(cc, 1)
for _ in range(steps):
    net.filter_frames_device(d_frames, frames.shape[0], index, B, 1, d_boxes, d_counts, 256, d_mask)
ctx.sync()
lib.covahip_dev_phase_read(out, 1)
v = np.array(list(out), dtype=np.float64)
FUSED23 = os.environ.get("QB_IMPL") not in ("enc23_separate", "enc_general_tiles")
NWG[16] = 256 if FUSED23 else 512
for base, label in ((0, "enc1t (PRE)"), (16, "enc23" if FUSED23 else "enc2"), (32, "enc3"), (48, "dec012"), (64, "dec3cc")):
    if v[base:base + 9].sum() == 0:
        continue
    tot = v[base:base + 9].sum()
    print(f"{label}: {tot / steps / 100:.0f} us of workgroup time per launch (all workgroups)")
    for i, n in enumerate(E23_NAMES if base == 16 and FUSED23 else DEC_NAMES if base == 48 else TAIL_NAMES if base == 64 else E1V_NAMES if base == 0 and not os.environ.get('QB_IMPL') else names):
        print(f"   {n:40s} {100 * v[base + i] / tot:5.1f} %   {v[base + i] / steps / 100 / NWG[base]:7.2f} us per workgroup ({NWG[base]} of them)")
lib.covahip_dev_ccphase_read(cc, 1)
ccv = np.array(list(cc), dtype=np.float64)
if ccv.sum() > 0:
    print("bboxcc inside the fused tail (frame_wg, thread 0 of every workgroup):")
    for i, nme in enumerate(["A: bit planes", "B: row masks, prefix sums", "C: runs, unions", "D: statistics to the roots", "E: compaction, box stores"]):
        print(f"   {nme:40s} {ccv[i] / steps / 100 / 256:7.2f} us per workgroup")
