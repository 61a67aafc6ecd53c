"""covahip_pipe_* (pinned, three-stream pipelined host path with on-device box compaction) against the synchronous
carrier-frame entry point on the same batches."""
import numpy as np
import pytest

from cova_amd import _lib as L
from cova_amd import synth
from cova_amd.elements import BlobNetInfer, FilterPipe

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("blocking_wait", [False, True])
def test_pipelined_batches_equal_synchronous_calls(ctx, weights_flat, blocking_wait):
    h, w, b, streams, n_batches = 45, 80, 48, 4, 7
    net = BlobNetInfer(ctx, weights_flat, h, w, max_batch=64)
    data = [synth.carrier_batch(b, h, w, seed=300 + 10 * k, streams=streams) for k in range(n_batches)]
    ref = []
    for frames, index in data:
        boxes, counts, mask, _ = net.filter_frames(frames, index, 4, max_boxes=512, want_mask=True)
        ref.append((boxes, counts, mask))
    # blocking_wait (round 5): the waiting thread sleeps on the completion interrupt instead of spinning -- same results
    pipe = FilterPipe(net, max_batch=64, max_frames=64 + 3 * streams, max_boxes=512, n_slots=3, want_mask=True, blocking_wait=blocking_wait)
    got, in_flight = {}, []
    for k, (frames, index) in enumerate(data):
        acq = pipe.acquire()
        while acq is None:                       # every slot busy: drain the oldest
            s0, k0 = in_flight.pop(0)
            c, o, bx, m = pipe.collect(s0)
            got[k0] = (c.copy(), o.copy(), bx.copy(), m.copy())
            acq = pipe.acquire()
        slot, pf, pi = acq
        pf[:frames.shape[0]] = frames
        pi[:b] = index
        pipe.submit(slot, frames.shape[0], b, 4)
        in_flight.append((slot, k))
    assert len(in_flight) == 3                   # three batches were in flight at once
    for s0, k0 in in_flight:
        c, o, bx, m = pipe.collect(s0)
        got[k0] = (c.copy(), o.copy(), bx.copy(), m.copy())
    for k in range(n_batches):
        boxes, counts, mask = ref[k]
        c, o, bx, m = got[k]
        np.testing.assert_array_equal(c, counts)
        np.testing.assert_array_equal(m, mask)
        np.testing.assert_array_equal(o, np.concatenate([[0], np.cumsum(np.minimum(counts, 512))]))
        for i in range(b):
            np.testing.assert_array_equal(bx[o[i]:o[i + 1]], boxes[i, :min(counts[i], 512)])
    assert sum(int(g[1][-1]) for g in got.values()) > 0
    pipe.close()


def test_more_boxes_than_the_speculative_copy_and_truncation(ctx, weights_flat):
    """Noise-like masks give hundreds of boxes per frame: more than the part of the packed array that is copied back
    unconditionally, and with a small max_boxes every frame is truncated -- offsets follow min(count, max_boxes)."""
    h, w, b = 68, 120, 16
    net = BlobNetInfer(ctx, weights_flat, h, w, max_batch=16)
    frames, index = synth.carrier_batch(b, h, w, seed=77, streams=2)
    for max_boxes in (2048, 8):
        boxes, counts, _, _ = net.filter_frames(frames, index, 1, max_boxes=max_boxes)
        pipe = FilterPipe(net, max_batch=16, max_frames=32, max_boxes=max_boxes, n_slots=2)
        slot, pf, pi = pipe.acquire()
        pf[:frames.shape[0]] = frames
        pi[:b] = index
        pipe.submit(slot, frames.shape[0], b, 1)
        c, o, bx, m = pipe.collect(slot)
        assert m is None
        np.testing.assert_array_equal(c, counts)
        assert counts.min() > 8 and int(o[b]) == int(np.minimum(counts, max_boxes).sum())
        for i in range(b):
            np.testing.assert_array_equal(bx[o[i]:o[i + 1]], boxes[i, :min(counts[i], max_boxes)])
        pipe.close()


def test_batches_past_the_self_scan_limit_take_the_scan_kernel(ctx, weights_flat):
    """The pack kernel's workgroups sum the counts in front of them themselves up to 1,024 frames; larger batches (the pipe admits
    65,536) get their offsets from a one-pass scan kernel instead of batch^2 / 2 reads.  Same packed result."""
    h, w, b, streams = 35, 60, 1100, 4
    net = BlobNetInfer(ctx, weights_flat, h, w, max_batch=b)
    frames, index = synth.carrier_batch(b, h, w, seed=911, streams=streams)
    boxes, counts, _, _ = net.filter_frames(frames, index, 2, max_boxes=64)
    pipe = FilterPipe(net, max_batch=b, max_frames=frames.shape[0], max_boxes=64, n_slots=2)
    for _ in range(2):
        slot, pf, pi = pipe.acquire()
        pf[:frames.shape[0]] = frames
        pi[:b] = index
        pipe.submit(slot, frames.shape[0], b, 2)
        c, o, bx, _m = pipe.collect(slot)
        np.testing.assert_array_equal(c, counts)
        np.testing.assert_array_equal(o, np.concatenate([[0], np.cumsum(np.minimum(counts, 64))]))
        assert int(o[b]) > b                          # more than a box per frame: the offsets are not trivial
        for i in range(0, b, 37):
            np.testing.assert_array_equal(bx[o[i]:o[i + 1]], boxes[i, :min(counts[i], 64)])
        np.testing.assert_array_equal(bx[o[b - 1]:o[b]], boxes[b - 1, :min(counts[b - 1], 64)])
    pipe.acquire()
    pipe.close()


@pytest.mark.parametrize("lanes", [1, 2, 3, 4])
def test_copy_streams_keep_off_the_lanes_hardware_queues(ctx, weights_flat, lanes):
    """Round 6: HIP multiplexes its streams onto a few hardware queues; a copy stream that shares one with a lane holds that lane's
    kernels up behind its markers (113 - 182 us per batch depending on where the streams land -- creation order alone).  The pipe
    MEASURES where candidate streams land (pipe.hip, streams_share_queue) and takes an upload stream that shares its queue with no
    active lane; the results of a batch then leave on the lane that ran it (up to three lanes).  Results unchanged, whatever the
    lane count."""
    h, w, b = 45, 80, 24
    ctx.set_lanes(lanes)
    pipe = None
    try:
        net = BlobNetInfer(ctx, weights_flat, h, w, max_batch=b)
        frames, index = synth.carrier_batch(b, h, w, seed=5, streams=2)
        boxes, counts, _, _ = net.filter_frames(frames, index, 2, max_boxes=128)
        pipe = FilterPipe(net, max_batch=b, max_frames=frames.shape[0], max_boxes=128, n_slots=4, want_mask=(lanes == 2))
        up, down = pipe.queue_plan()
        assert up == 0 if lanes <= 3 else up <= 1, (lanes, up, down)
        assert 0 <= down <= 1
        for rnd in range(3):
            slots = []
            for _ in range(4):
                slot, pf, pi = pipe.acquire()
                pf[:frames.shape[0]] = frames
                pi[:b] = index
                pipe.submit(slot, frames.shape[0], b, 2)
                slots.append(slot)
            for slot in slots:
                c, o, bx, _m = pipe.collect(slot)
                np.testing.assert_array_equal(c, counts)
                for i in range(b):
                    np.testing.assert_array_equal(bx[o[i]:o[i + 1]], boxes[i, :min(counts[i], 128)])
        pipe.acquire()
    finally:
        if pipe is not None:
            pipe.close()
        ctx.set_lanes(1)


def test_pipe_argument_checks(ctx, weights_flat):
    lib = L.lib()
    net = BlobNetInfer(ctx, weights_flat, 45, 80, max_batch=8)
    import ctypes as C
    h = C.c_void_p()
    assert lib.covahip_pipe_create(ctx.handle, 16, 32, 64, 2, 0, C.byref(h)) == 1      # max_batch above the model's
    pipe = FilterPipe(net, max_batch=8, max_frames=16, max_boxes=64, n_slots=1)
    slot, pf, pi = pipe.acquire()
    assert pipe.acquire() is None                                                      # the only slot is taken
    assert lib.covahip_pipe_submit(pipe._h, slot, 2, 1, 1) == 1                        # fewer than four frames
    pi[0] = (3, 2, 1, 9)
    assert lib.covahip_pipe_submit(pipe._h, slot, 6, 1, 1) == 1                        # index outside the frames
    assert lib.covahip_pipe_collect(pipe._h, slot, None, None, None, None) == 1        # nothing submitted
    assert lib.covahip_pipe_release(pipe._h, slot) == 1                                # nothing collected
    pipe.close()


def test_abort_returns_an_acquired_slot_and_bad_submits_leave_the_pipe_usable(ctx, weights_flat):
    """covahip_pipe_abort: an acquired slot that is not submitted (a failed submit, a shutdown with a partly filled batch) goes back
    to the pool; a submit with impossible sizes is refused before anything is enqueued and the slot stays acquired."""
    h, w, b = 45, 80, 8
    net = BlobNetInfer(ctx, weights_flat, h, w, max_batch=8)
    frames, index = synth.carrier_batch(b, h, w, seed=9, streams=2)
    pipe = FilterPipe(net, max_batch=8, max_frames=16, max_boxes=64, n_slots=2)
    a = pipe.acquire()
    c = pipe.acquire()
    assert a is not None and c is not None and pipe.acquire() is None           # both slots taken
    assert pipe._lib.covahip_pipe_submit(pipe._h, a[0], 3, b, 1) == 1            # fewer than four frames: refused
    assert pipe._lib.covahip_pipe_submit(pipe._h, a[0], frames.shape[0], 99, 1) == 1
    assert pipe._lib.covahip_pipe_abort(pipe._h, 7) == 1                          # no such slot
    pipe.abort(a[0])
    assert pipe._lib.covahip_pipe_abort(pipe._h, a[0]) == 1                       # not acquired any more
    again = pipe.acquire()
    assert again is not None and again[0] == a[0]
    slot, pf, pi = again
    pf[:frames.shape[0]] = frames
    pi[:b] = index
    pipe.submit(slot, frames.shape[0], b, 2)
    counts, offsets, boxes, _ = pipe.collect(slot)
    rboxes, rcounts, _, _ = net.filter_frames(frames, index, 2, max_boxes=64)
    np.testing.assert_array_equal(counts, rcounts)
    pipe.abort(c[0])
    pipe.close()


def test_packed_records_give_identical_results(ctx, weights_flat):
    """covahip_pipe_set_packed: two-byte records (min(type, 6) | min(mv_x, 6) << 3 | min(mv_y, 6) << 6, covahip_carrier_pack) instead
    of the decoder's four bytes per macroblock -- half the PCIe bytes, the same boxes bit for bit (the network clips at 6 anyway,
    utils/model/preprocessing.py:6-7).  The pack itself against its definition, values above 6 and the unused byte included."""
    from cova_amd.elements import pack_frames
    h, w, b, streams = 68, 120, 48, 3
    net = BlobNetInfer(ctx, weights_flat, h, w, max_batch=b)
    frames, index = synth.carrier_batch(b, h, w, seed=23, streams=streams)
    frames = frames.copy()
    frames[::3, ::5, ::7, 1] = 200          # a motion-vector magnitude far above the clip
    frames[..., 3] = 0xA5                   # the byte the network ignores
    rec = pack_frames(frames)
    m = np.minimum(frames[..., :3].astype(np.uint16), 6)
    np.testing.assert_array_equal(rec, m[..., 0] | (m[..., 1] << 3) | (m[..., 2] << 6))
    res = []
    for packed in (False, True):
        pipe = FilterPipe(net, max_batch=b, max_frames=frames.shape[0], max_boxes=512, n_slots=2, want_mask=True, packed=packed)
        slot, pf, pi = pipe.acquire()
        pf[:frames.shape[0]] = rec if packed else frames
        pi[:b] = index
        pipe.submit(slot, frames.shape[0], b, 1)
        counts, offsets, boxes, mask = pipe.collect(slot)
        res.append((counts.copy(), offsets.copy(), boxes.copy(), mask.copy()))
        pipe.close()
    for a, c in zip(res[0], res[1]):
        np.testing.assert_array_equal(a, c)
    assert res[0][0].sum() > 0


def test_blocking_wait_only_with_idle_slots_and_device_pci_address(ctx, weights_flat):
    """covahip_pipe_set_blocking_wait is refused while a slot is acquired or in flight (the events it replaces may be waited on);
    covahip_device_pci_bus_id names the device the way sysfs does (what multigpu.pin_to_gpu reads the NUMA node from)."""
    import ctypes as C
    import os
    import re
    lib = L.lib()
    net = BlobNetInfer(ctx, weights_flat, 45, 80, max_batch=8)
    pipe = FilterPipe(net, max_batch=8, max_frames=16, max_boxes=64, n_slots=2)
    slot, pf, pi = pipe.acquire()
    assert lib.covahip_pipe_set_blocking_wait(pipe._h, 1) != 0
    pipe.abort(slot)
    assert lib.covahip_pipe_set_blocking_wait(pipe._h, 1) == 0
    assert lib.covahip_pipe_set_blocking_wait(pipe._h, 0) == 0
    assert lib.covahip_pipe_set_blocking_wait(None, 1) != 0
    pipe.close()
    buf = C.create_string_buffer(32)
    assert lib.covahip_device_pci_bus_id(0, buf, len(buf)) == 0
    addr = buf.value.decode().lower()
    assert re.fullmatch(r"[0-9a-f]{4}:[0-9a-f]{2}:[0-9a-f]{2}\.[0-9a-f]", addr), addr
    assert lib.covahip_device_pci_bus_id(0, buf, 4) != 0 and lib.covahip_device_pci_bus_id(9999, buf, len(buf)) != 0
    if os.path.isdir("/sys/bus/pci/devices"):
        assert os.path.exists(os.path.join("/sys/bus/pci/devices", addr))
