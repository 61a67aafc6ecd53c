"""Error behaviour of the C-ABI on a GPU box: loud failures, no silent fallbacks."""
import ctypes as C

import numpy as np
import pytest

from cova_amd import _lib as L
from cova_amd import synth, weights as W
from cova_amd.elements import BboxCc, BlobNetInfer, Context

pytestmark = pytest.mark.gpu


def test_forward_before_load_and_bad_args(weights_flat):
    ctx = Context(0)
    lib = L.lib()
    stack = synth.stacked_batch(1, 45, 80, seed=1)
    mask = np.zeros((1, 45, 80), np.uint8)
    assert lib.covahip_blobnet_forward(ctx.handle, stack.ctypes.data, 1, None, mask.ctypes.data, L.MEM_HOST) == 4  # NOT_LOADED
    net = BlobNetInfer(ctx, weights_flat, 45, 80, max_batch=2)
    assert lib.covahip_blobnet_forward(ctx.handle, stack.ctypes.data, 3, None, mask.ctypes.data, L.MEM_HOST) == 1  # batch > max_batch
    assert lib.covahip_blobnet_forward(ctx.handle, None, 1, None, mask.ctypes.data, L.MEM_HOST) == 1
    assert lib.covahip_blobnet_forward(ctx.handle, stack.ctypes.data, 0, None, None, L.MEM_HOST) == 0         # empty batch is fine
    assert lib.covahip_blobnet_forward(ctx.handle, stack.ctypes.data, 1, None, mask.ctypes.data, 7) == 1      # bad mem_kind
    assert net.macs_per_frame == 76_893_440
    ctx.close()


def test_bad_weight_blobs_rejected(weights_flat):
    ctx = Context(0)
    lib = L.lib()
    blob = bytearray(W.to_bytes(weights_flat))
    assert lib.covahip_blobnet_load(ctx.handle, bytes(blob[:-4]), len(blob) - 4, 68, 120, 4, 1) == 6         # truncated
    bad = bytearray(blob); bad[0] ^= 0xFF
    assert lib.covahip_blobnet_load(ctx.handle, bytes(bad), len(bad), 68, 120, 4, 1) == 6                     # magic
    assert lib.covahip_blobnet_load(ctx.handle, bytes(blob), len(blob), 68, 120, 3, 1) == 5                   # timestep != 4
    assert lib.covahip_blobnet_load(ctx.handle, bytes(blob), len(blob), 8, 8, 4, 1) == 5                      # grid too small
    assert lib.covahip_blobnet_load(ctx.handle, bytes(blob), len(blob), 68, 120, 4, 1) == 0
    assert BlobNetInfer(ctx, weights_flat, 68, 120, max_batch=1).macs_per_frame == 170_411_520
    ctx.close()


def test_failed_load_leaves_no_model_and_limits_are_checked_at_load(weights_flat):
    """A load that fails half way (workspace OOM) must not leave a half-built model behind; geometry the kernels
    cannot run (width not a multiple of 4) is rejected by the load, not by the first forward."""
    ctx = Context(0)
    lib = L.lib()
    blob = W.to_bytes(weights_flat)
    assert lib.covahip_blobnet_load(ctx.handle, blob, len(blob), 45, 80, 4, 2) == 0
    assert lib.covahip_blobnet_load(ctx.handle, blob, len(blob), 68, 120, 4, 1 << 22) == 3       # ~1 TB of workspace
    stack = synth.stacked_batch(1, 68, 120, seed=1)
    mask = np.zeros((1, 68, 120), np.uint8)
    assert lib.covahip_blobnet_forward(ctx.handle, stack.ctypes.data, 1, None, mask.ctypes.data, L.MEM_HOST) == 4
    boxes = np.zeros((1, 16), dtype=L.BOX_DTYPE)
    counts = np.zeros(1, dtype=np.int32)
    assert lib.covahip_filter_forward(ctx.handle, stack.ctypes.data, 1, 1, boxes.ctypes.data, counts.ctypes.data, 16,
                                      None, None, L.MEM_HOST) == 4
    assert lib.covahip_blobnet_load(ctx.handle, blob, len(blob), 45, 53, 4, 1) == 5             # 854 / 16 = 53 macroblocks
    assert lib.covahip_blobnet_forward(ctx.handle, stack.ctypes.data, 1, None, mask.ctypes.data, L.MEM_HOST) == 4
    assert lib.covahip_blobnet_load(ctx.handle, blob, len(blob), 68, 120, 4, 1) == 0            # and the ctx is still usable
    assert lib.covahip_blobnet_forward(ctx.handle, stack.ctypes.data, 1, None, mask.ctypes.data, L.MEM_HOST) == 0
    ctx.close()


def test_bboxcc_geometry_limits_and_empty_batch(ctx):
    lib = L.lib()
    cc = BboxCc(ctx, cc_threshold=1, max_boxes=16)
    boxes = np.zeros((1, 16), dtype=L.BOX_DTYPE)
    counts = np.zeros(1, dtype=np.int32)
    big = np.zeros((1, 400, 400), np.uint8)                   # 40,000 blocks do not fit the LDS union-find
    rc = lib.covahip_bboxcc(ctx.handle, big.ctypes.data, 1, 400, 400, 1, boxes.ctypes.data, counts.ctypes.data, 16, L.MEM_HOST)
    assert rc == 5
    assert lib.covahip_bboxcc(ctx.handle, big.ctypes.data, 0, 400, 400, 1, boxes.ctypes.data, counts.ctypes.data, 16, L.MEM_HOST) == 0
    assert lib.covahip_bboxcc(ctx.handle, big.ctypes.data, 1, 0, 400, 1, boxes.ctypes.data, counts.ctypes.data, 16, L.MEM_HOST) == 1
    # a 1440p macroblock grid (90 x 160 -> 45 x 80 blocks) still fits the LDS union-find (limit ~6,700 blocks)
    m = synth.random_masks(2, 90, 160, 0.1, seed=4)
    from oracle import ref
    cc2 = BboxCc(ctx, cc_threshold=1, max_boxes=8192)
    b, c = cc2.regionprops(m)
    rb, rc_ = ref.regionprops_batch(m, 1, 8192)
    np.testing.assert_array_equal(c, rc_)
    for i in range(2):
        np.testing.assert_array_equal(b[i, :c[i]]["left"], rb[i, :c[i]]["left"])
        np.testing.assert_array_equal(b[i, :c[i]]["area_px"], rb[i, :c[i]]["area"])


def test_strerror_covers_every_status():
    lib = L.lib()
    msgs = {lib.covahip_strerror(i) for i in range(9)}
    assert len(msgs) == 9 and b"unknown status" not in msgs
