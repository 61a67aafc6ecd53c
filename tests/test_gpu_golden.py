"""The HIP path against the COMMITTED golden fixtures directly (tests/golden/: torch-f64 BlobNet logits, scipy +
block-raster-key CCL boxes), the carrier-frame entry against the oracle directly at 68x120, and the lanes of a ctx
(batches in flight, include/covahip.h) against one step after the other."""
import numpy as np
import pytest

from cova_amd import _lib as L
from cova_amd import synth
from cova_amd.elements import BboxCc, BlobNetInfer
from oracle import ref
from tests.golden_util import blobnet_golden, blobnet_tolerance, ccl_golden

pytestmark = pytest.mark.gpu

FIELDS = (("left", "left"), ("top", "top"), ("width", "width"), ("height", "height"), ("area_px", "area"))


def _rows(boxes):
    return np.stack([boxes[f] for f, _ in FIELDS], axis=1).astype(np.int64) if len(boxes) else np.zeros((0, 5), np.int64)


def _same_boxes(boxes, counts, rboxes, rcounts):
    np.testing.assert_array_equal(counts, rcounts)
    for i in range(len(counts)):
        n = int(counts[i])
        for f, g in FIELDS:
            np.testing.assert_array_equal(boxes[i, :n][f], rboxes[i, :n][g], err_msg=f"frame {i} {f}")


@pytest.mark.parametrize("hw", ["45x80", "67x120", "68x120"])
def test_hip_blobnet_against_committed_golden_logits(ctx, weights_flat, hw):
    """tests/golden/blobnet_golden.npz holds logits of an independent float64 torch.nn.functional composition
    (tests/golden/gen_blobnet_golden.py).  Every entry of the HIP path: BlobNet alone, the fused path on stacks, the
    carrier-frame entry (the stack cut back into its four carrier frames)."""
    stack, gold = blobnet_golden(weights_flat)[hw]
    h, w = map(int, hw.split("x"))
    b = stack.shape[0]
    net = BlobNetInfer(ctx, weights_flat, h, w, max_batch=b)
    atol, rtol = blobnet_tolerance(gold)
    tol = atol + rtol * np.abs(gold)
    logits, mask = net.infer(stack)
    assert (np.abs(logits - gold) <= tol).all(), float(np.abs(logits - gold).max())
    np.testing.assert_array_equal(mask, (logits > 0).astype(np.uint8))
    assert (np.abs(gold[mask != (gold > 0)]) <= atol).all()
    # the fused path on the same stacks: same mask, boxes of that mask
    boxes, counts, fmask = net.filter(stack, cc_threshold=1, max_boxes=2048, want_mask=True)
    np.testing.assert_array_equal(fmask, mask)
    _same_boxes(boxes, counts, *ref.regionprops_batch(mask, 1, 2048))
    # carrier-frame entry: stack i = frames 4i .. 4i+3 (row block k = T index k)
    frames = stack.reshape(b * 4, h, w, 4)
    index = np.arange(b * 4, dtype=np.int32).reshape(b, 4)
    cboxes, ccounts, cmask, clogits = net.filter_frames(frames, index, 1, max_boxes=2048, want_mask=True, want_logits=True)
    np.testing.assert_array_equal(clogits, logits)
    np.testing.assert_array_equal(cmask, mask)
    _same_boxes(cboxes, ccounts, *ref.regionprops_batch(mask, 1, 2048))


@pytest.mark.parametrize("cap", [0, 24, -1])
def test_hip_bboxcc_against_committed_golden_boxes(ctx, cap):
    """tests/golden/ccl_golden.json: masks with the boxes scipy.ndimage.label + the explicit block-raster ordering give
    (tests/golden/gen_ccl_golden.py).  Through the automatic kernel choice, the wave kernel with its overflow pass
    (capacity 24) and the workgroup kernel only."""
    for name, (m, exp) in sorted(ccl_golden().items()):
        for thr in (1, 2, 4, 30):
            cc = BboxCc(ctx, cc_threshold=thr, max_boxes=((m.shape[0] + 1) // 2) * ((m.shape[1] + 1) // 2))
            cc.set_wave_cap(cap)
            try:
                # a batch of copies, so that the wave kernel (one wave per frame, several frames per workgroup) is exercised too
                boxes, counts = cc.regionprops(np.repeat(m[None], 5, axis=0))
            finally:
                cc.set_wave_cap(0)
            want = exp[exp[:, 4] >= thr]
            for i in range(5):
                assert counts[i] == len(want), (name, thr, i)
                np.testing.assert_array_equal(_rows(boxes[i, :counts[i]]), want, err_msg=f"{name} thr {thr}")


def test_carrier_frame_entry_against_the_oracle_at_1080p(ctx, weights_flat):
    """covahip_filter_forward_frames -- the entry bench.py times -- at 68x120, 40 frames of 4 streams, default kernels,
    directly against oracle/ref.py (logits within the stated tolerance, boxes bit-exact on the HIP mask)."""
    h, w, b = 68, 120, 40
    frames, index = synth.carrier_batch(b, h, w, seed=4242, streams=4)
    stack = np.concatenate([frames[index[:, k]] for k in range(4)], axis=1)
    net = BlobNetInfer(ctx, weights_flat, h, w, max_batch=b)
    boxes, counts, mask, logits = net.filter_frames(frames, index, 1, max_boxes=2048, want_mask=True, want_logits=True)
    ref_logits, _ = ref.blobnet_forward(weights_flat, stack, h, w)
    atol, rtol = blobnet_tolerance(ref_logits)
    err = np.abs(logits - ref_logits)
    assert (err <= atol + rtol * np.abs(ref_logits)).all(), float(err.max())
    np.testing.assert_array_equal(mask, (logits > 0).astype(np.uint8))
    assert (np.abs(ref_logits[mask != (ref_logits > 0)]) <= atol).all()
    _same_boxes(boxes, counts, *ref.regionprops_batch(mask, 1, 2048))
    assert counts.max() > 0


# ------------------------------------------------------------------------------------------------ lanes
def _device_run(ctx, net, d_in, n_frames, index, b, d_out):
    d_boxes, d_counts, d_mask = d_out
    net.filter_frames_device(d_in, n_frames, index, b, 1, d_boxes, d_counts, 512, d_mask)


def _fetch(ctx, b, h, w, d_out):
    boxes = np.zeros((b, 512), dtype=L.BOX_DTYPE)
    counts = np.zeros(b, dtype=np.int32)
    mask = np.zeros((b, h, w), dtype=np.uint8)
    ctx.d2h(boxes, d_out[0]); ctx.d2h(counts, d_out[1]); ctx.d2h(mask, d_out[2])
    return boxes, counts, mask


@pytest.mark.parametrize("lanes", [2, 3])
def test_lanes_give_the_bits_of_one_step_after_the_other(ctx, weights_flat, lanes):
    """Six device-pointer calls with DIFFERENT inputs enqueued back to back on a ctx with 2 / 3 lanes (they overlap on the GPU,
    each lane with its own workspace, stack table and output buffers) against the same six calls with one lane."""
    h, w, b = 45, 80, 24
    assert ctx.lanes() == 1          # the default: no hidden concurrency
    net = BlobNetInfer(ctx, weights_flat, h, w, max_batch=b)
    inputs = [synth.carrier_batch(b, h, w, seed=100 + k, streams=1 + k % 3) for k in range(6)]
    d_in = []
    for fr, _ in inputs:
        p = ctx.malloc(fr.nbytes)
        ctx.h2d(p, fr)
        d_in.append(p)
    outs = [(ctx.malloc(b * 512 * 20), ctx.malloc(b * 4), ctx.malloc(b * h * w)) for _ in range(6)]
    try:
        ctx.set_lanes(1)
        for k in range(6):
            _device_run(ctx, net, d_in[k], inputs[k][0].shape[0], inputs[k][1], b, outs[k])
        want = [_fetch(ctx, b, h, w, outs[k]) for k in range(6)]
        for o in outs:
            for p, n in zip(o, (b * 512 * 20, b * 4, b * h * w)):
                ctx._lib.covahip_memset(ctx.handle, p, 0xEE, n)
        ctx.set_lanes(lanes)
        assert ctx.lanes() == lanes
        for rep in range(3):             # the second and third round find every lane's stack table changed again
            for k in range(6):
                _device_run(ctx, net, d_in[k], inputs[k][0].shape[0], inputs[k][1], b, outs[k])
        ctx.sync()
        for k in range(6):
            boxes, counts, mask = _fetch(ctx, b, h, w, outs[k])
            np.testing.assert_array_equal(mask, want[k][2])
            np.testing.assert_array_equal(counts, want[k][1])
            for i in range(b):
                n = min(int(counts[i]), 512)
                assert boxes[i, :n].tobytes() == want[k][0][i, :n].tobytes()
        # ... and against the oracle, so that "the same" is also "right"
        stack = np.concatenate([inputs[5][0][inputs[5][1][:, t]] for t in range(4)], axis=1)
        ref_logits, _ = ref.blobnet_forward(weights_flat, stack, h, w)
        atol, _ = blobnet_tolerance(ref_logits)
        assert (np.abs(ref_logits[want[5][2] != (ref_logits > 0)]) <= atol).all()
        _same_boxes(want[5][0], want[5][1], *ref.regionprops_batch(want[5][2], 1, 512))
    finally:
        ctx.set_lanes(1)
        for p in d_in:
            ctx.free(p)
        for o in outs:
            for p in o:
                ctx.free(p)


def test_lanes_order_behind_and_before_primary_stream_work(ctx, weights_flat):
    """A filter call sees a copy enqueued before it, and a copy / stand-alone call after it sees its results, with two lanes
    and no explicit sync in between (rules in include/covahip.h)."""
    h, w, b = 45, 80, 8
    net = BlobNetInfer(ctx, weights_flat, h, w, max_batch=b)
    fr_a, idx = synth.carrier_batch(b, h, w, seed=7, streams=2)
    fr_b, _ = synth.carrier_batch(b, h, w, seed=8, streams=2)
    d_in = ctx.malloc(fr_a.nbytes)
    out = (ctx.malloc(b * 512 * 20), ctx.malloc(b * 4), ctx.malloc(b * h * w))
    ctx.set_lanes(2)
    try:
        res = []
        for fr in (fr_a, fr_b, fr_a):
            ctx.h2d(d_in, fr)                                   # primary stream; the next call must see it
            _device_run(ctx, net, d_in, fr.shape[0], idx, b, out)
            res.append(_fetch(ctx, b, h, w, out))               # d2h right behind the call: must see its results
        assert res[0][2].tobytes() == res[2][2].tobytes() and res[0][1].tobytes() == res[2][1].tobytes()
        assert res[0][2].tobytes() != res[1][2].tobytes()
        _, _, hmask, _ = net.filter_frames(fr_b, idx, 1, max_boxes=512, want_mask=True)      # host-pointer call: primary stream
        np.testing.assert_array_equal(hmask, res[1][2])
        # stand-alone bboxcc on the mask a lane has just written, without a sync in between
        _device_run(ctx, net, d_in, fr_a.shape[0], idx, b, out)
        cc = BboxCc(ctx, 1, 512)
        d_b2, d_c2 = ctx.malloc(b * 512 * 20), ctx.malloc(b * 4)
        cc.regionprops_device(out[2], b, h, w, d_b2, d_c2)
        c2 = np.zeros(b, dtype=np.int32)
        ctx.d2h(c2, d_c2)
        np.testing.assert_array_equal(c2, res[0][1])
        ctx.free(d_b2); ctx.free(d_c2)
    finally:
        ctx.set_lanes(1)
        ctx.free(d_in)
        for p in out:
            ctx.free(p)


def test_default_ctx_runs_calls_in_order_on_shared_output_buffers(weights_flat):
    """The drop-in contract (include/covahip.h; boundary precedent cova-rs/nvdsbbox/nvdsbbox.h:7-14: no hidden concurrency): on a
    ctx nobody has touched, device-pointer filter calls enqueued back to back WITHOUT a sync and sharing ONE set of output
    buffers leave the last call's results there -- every call is ordered behind the one before it."""
    from cova_amd.elements import Context
    c = Context(0)
    try:
        assert c.lanes() == 1
        h, w, b = 45, 80, 24
        net = BlobNetInfer(c, weights_flat, h, w, max_batch=b)
        inputs = [synth.carrier_batch(b, h, w, seed=300 + k, streams=2) for k in range(4)]
        d_in = []
        for fr, _ in inputs:
            p = c.malloc(fr.nbytes)
            c.h2d(p, fr)
            d_in.append(p)
        out = (c.malloc(b * 512 * 20), c.malloc(b * 4), c.malloc(b * h * w))
        want = []
        for k in range(4):
            _device_run(c, net, d_in[k], inputs[k][0].shape[0], inputs[k][1], b, out)
            c.sync()
            want.append(_fetch(c, b, h, w, out))
        assert want[2][2].tobytes() != want[3][2].tobytes()
        for rep in range(5):
            for k in range(4):          # four calls in flight on the same outputs, no sync in between
                _device_run(c, net, d_in[k], inputs[k][0].shape[0], inputs[k][1], b, out)
            boxes, counts, mask = _fetch(c, b, h, w, out)
            np.testing.assert_array_equal(mask, want[3][2])
            np.testing.assert_array_equal(counts, want[3][1])
            for i in range(b):
                n = min(int(counts[i]), 512)
                assert boxes[i, :n].tobytes() == want[3][0][i, :n].tobytes()
    finally:
        c.close()


def test_real_video_records_through_the_hot_path(ctx):
    """Real compressed-domain input: the records of 64 consecutive output pictures of the reference's demo/1m.mp4 (entropy-decoded
    here by this build's front end, tests/golden/gen_demo_records.py; the video itself cannot travel to the GPU box) as ONE stream
    through the carrier-frame entry point with the blob-like weight set, against the oracle; the vehicle that crosses the scene
    comes out as boxes."""
    import os
    from cova_amd import weights as W
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "demo_records_excerpt.npz"))
    frames = z["records"]
    n, h, w = frames.shape[0], 45, 80
    flat = W.blob_like(7)
    net = BlobNetInfer(ctx, flat, h, w, max_batch=n)
    boxes, counts, mask, logits = net.filter_frames(frames, None, 1, max_boxes=512, want_mask=True, want_logits=True)
    stack = np.stack([np.concatenate([frames[i - k] for k in range(4)], axis=0) for i in range(3, n)])
    ref_logits, _ = ref.blobnet_forward(flat, stack, h, w)
    atol, rtol = blobnet_tolerance(ref_logits)
    assert (np.abs(logits - ref_logits) <= atol + rtol * np.abs(ref_logits)).all()
    assert (np.abs(ref_logits[mask != (ref_logits > 0)]) <= atol).all()
    _same_boxes(boxes, counts, *ref.regionprops_batch(mask, 1, 512))
    # P pictures carry the motion (B pictures of this stream are almost entirely skipped / direct): while the vehicle is in the scene
    # (the first 21 pictures) every picture with content of its own -- this 60 Hz stream repeats every frame once, the repeats are
    # B pictures without a vector -- has a blob of >= 20 macroblocks
    big = [int(boxes[i, :counts[i]]["area_px"].max()) if counts[i] else 0 for i in range(len(counts))]
    assert sum(a >= 12 for a in big) >= 11 and all(a >= 20 for a in big[0:21:2])
    # ... and where the blob is, the stream's own motion bytes are large (a third of the box's macroblocks move by a pixel or more;
    # 0.5 % of the picture's do)
    i = int(np.argmax(big))
    b = boxes[i, :counts[i]][int(np.argmax(boxes[i, :counts[i]]["area_px"]))]
    cur = frames[i + 3]
    inside = cur[b["top"]:b["top"] + b["height"], b["left"]:b["left"] + b["width"], 1:3].max(axis=-1)
    assert (inside >= 4).mean() >= 0.3
