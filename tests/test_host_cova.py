"""cova GoP frame filter (C++ behind the C-ABI) vs the numpy restatement on scripted timelines."""
import numpy as np
import pytest

from cova_amd import _lib as L
from cova_amd import elements as E
from oracle import sort_ref as R

CLK = 1_000_000_000 // 30


def _bb(rows):
    out = np.zeros(len(rows), dtype=L.BBOX_DTYPE)
    for i, r in enumerate(rows):
        out[i] = E.make_bbox(*r)[0]
    return out


def _timeline(n_frames, gop_len, objects):
    """objects: list of (first_frame, last_frame, x0, y0, vx, vy, w, h)."""
    det = []
    for i in range(n_frames):
        d = []
        for (a, b, x0, y0, vx, vy, w, h) in objects:
            if a <= i <= b:
                d.append((x0 + vx * (i - a), y0 + vy * (i - a), w, h))
        det.append(d)
    return det


@pytest.mark.parametrize("cfg", [dict(), dict(infer_i=True), dict(alpha=8, beta=3), dict(sort_maxage=5, sort_minhits=3)])
def test_counters_and_forwarded_aus_match_restatement(cfg):
    n, gop = 900, 250
    objects = [(10, 120, 5, 5, 0.4, 0.2, 6, 6), (200, 420, 60, 30, -0.3, 0.0, 8, 5), (300, 330, 20, 20, 0, 0, 4, 4),
               (500, 800, 10, 40, 0.2, -0.1, 7, 7)]
    dets = _timeline(n, gop, objects)
    kw = dict(sort_maxage=10, sort_minhits=5, sort_iou=0.1)
    kw.update(cfg)
    c = E.Cova(**kw)
    r = R.GopFilter(**kw)
    forwarded = []
    # encoded AUs run ahead of the mask branch (the enc queue is unbounded in the reference pipeline)
    lead = 300
    for i in range(n + lead):
        if i < n:
            c.sink_enc_chain(i, i * CLK, delta_unit=(i % gop != 0))
            r.push_enc(i, i * CLK, 0 if i % gop == 0 else R.DELTA_UNIT)
        j = i - lead
        if 0 <= j < n:
            out = c.sink_mask_chain(E.serialize_vec(_bb(dets[j])), j * CLK)
            forwarded.extend(out)
            r.push_boxes([R.Bbox(*d) for d in dets[j]], j * CLK)
    assert c.eos("sink_enc") is None
    out = c.eos("sink_mask")
    forwarded.extend(out)
    r.eos()
    assert (c.dropped, c.decoded_dependency, c.decoded_inference) == (r.dropped, r.decoded_dependency,
                                                                      r.decoded_inference)
    exp = [b for lst in r.pushed for b in lst]
    assert [(int(a["id"]), int(a["pts"]), int(a["flags"])) for a in forwarded] == [tuple(b) for b in exp]
    # grouping into BufferLists is preserved
    got_lists = {}
    for a in forwarded:
        got_lists.setdefault(int(a["list"]), []).append(int(a["id"]))
    assert list(got_lists.values()) == [[b[0] for b in lst] for lst in r.pushed]
    assert c.decoded_inference >= 1
    # every AU is accounted for except those the reference silently discards
    assert c.dropped + c.decoded_dependency + c.decoded_inference <= n


@pytest.mark.parametrize("reorder", ["b_frames", "overlapping_gops", "far_ahead"])
def test_gop_walks_stop_early_only_while_the_gops_are_in_order(reorder):
    """Round 5: the port stops its walks over the buffered GoPs at the first one wholly outside the range it looks for -- exact
    only while the GoPs' pts ranges are in order, which it tracks.  Against the restatement (which walks everything, as the
    reference does, imp.rs:135-315): B-frame reordering inside closed GoPs (in order), GoPs whose leading pictures reach back
    below the previous GoP's last pts (NOT in order: the port must fall back to the full walks), and an encoded branch that
    is a thousand frames ahead (many buffered GoPs, the case the early exit is for)."""
    n, gop = 1500, 50
    objects = [(10, 120, 5, 5, 0.4, 0.2, 6, 6), (200, 420, 60, 30, -0.3, 0.0, 8, 5), (300, 330, 20, 20, 0, 0, 4, 4),
               (500, 800, 10, 40, 0.2, -0.1, 7, 7), (900, 1400, 30, 10, 0.05, 0.05, 5, 5)]
    dets = _timeline(n, gop, objects)
    kw = dict(sort_maxage=10, sort_minhits=5, sort_iou=0.1, alpha=4, beta=2)
    c = E.Cova(**kw)
    r = R.GopFilter(**kw)

    def enc_pts(i):                      # decode order -> presentation time
        k = i % gop
        if reorder == "overlapping_gops" and k in (1, 2) and i >= gop:
            return (i - 6) * CLK         # leading pictures of an open GoP: presented before the previous GoP's last pictures
        if k and k % 3 == 0 and k + 1 < gop:
            return (i + 1) * CLK         # a P picture decoded before the B picture that precedes it in output order
        if k and k % 3 == 1 and k > 1:
            return (i - 1) * CLK
        return i * CLK
    lead = 1000 if reorder == "far_ahead" else 120
    forwarded = []
    for i in range(n + lead):
        if i < n:
            c.sink_enc_chain(i, enc_pts(i), delta_unit=(i % gop != 0))
            r.push_enc(i, enc_pts(i), 0 if i % gop == 0 else R.DELTA_UNIT)
        j = i - lead
        if 0 <= j < n:
            forwarded.extend(c.sink_mask_chain(E.serialize_vec(_bb(dets[j])), j * CLK))
            r.push_boxes([R.Bbox(*d) for d in dets[j]], j * CLK)
    assert c.eos("sink_enc") is None
    forwarded.extend(c.eos("sink_mask"))
    r.eos()
    assert (c.dropped, c.decoded_dependency, c.decoded_inference) == (r.dropped, r.decoded_dependency, r.decoded_inference)
    exp = [b for lst in r.pushed for b in lst]
    assert [(int(a["id"]), int(a["pts"]), int(a["flags"])) for a in forwarded] == [tuple(b) for b in exp]
    assert len(exp) > 15


def test_key_frame_gets_discont_and_dependencies_droppable():
    c = E.Cova(sort_maxage=10, sort_minhits=5)
    for i in range(600):
        c.sink_enc_chain(i, i * CLK, delta_unit=(i % 250 != 0))
    out_all = []
    for j in range(300):
        dets = [(10, 10, 5, 5)] if 20 <= j < 60 else []
        out_all.extend(c.sink_mask_chain(E.serialize_vec(_bb(dets)), j * CLK))
    c.eos("sink_enc")
    out_all.extend(c.eos("sink_mask"))
    ids = [int(a["id"]) for a in out_all]
    assert ids and ids[0] == 0
    assert int(out_all[0]["flags"]) & L.AU_DISCONT and int(out_all[0]["flags"]) & L.AU_DROPPABLE
    last = out_all[-1]
    assert not int(last["flags"]) & L.AU_DROPPABLE          # the frame kept for inference
    assert all(int(a["flags"]) & L.AU_DROPPABLE for a in out_all[:-1])
    assert c.decoded_inference == 1 and c.decoded_dependency == len(out_all) - 1 == 20
    assert ids == list(range(21))                           # frames 0..19 for dependency, frame 20 inferred


def test_delta_unit_before_any_key_frame_is_an_error():
    c = E.Cova()
    with pytest.raises(L.CovahipError):
        c.sink_enc_chain(0, 0, delta_unit=True)


def test_every_access_unit_is_forwarded_or_reported_dropped_and_tracks_are_exported():
    """The element keeps one buffer per access unit it has handed to the filter: the filter must give every id
    back, either in an output list or through take_dropped (the reference frees a dropped GoP's buffers,
    cova/imp.rs:268-305), and nothing may stay referenced for longer than the GoP window."""
    import struct
    n, gop = 1500, 250
    objects = [(10, 120, 5, 5, 0.4, 0.2, 6, 6), (400, 620, 60, 30, -0.3, 0.0, 8, 5), (900, 1300, 10, 40, 0.2, -0.1, 7, 7)]
    dets = _timeline(n, gop, objects)
    c = E.Cova(sort_maxage=10, sort_minhits=5, sort_iou=0.1)
    forwarded, dropped, held_max = [], [], 0
    lead = 30
    wire = b""
    for i in range(n + lead):
        if i < n:
            c.sink_enc_chain(i, i * CLK, delta_unit=(i % gop != 0))
        j = i - lead
        if 0 <= j < n:
            forwarded.extend(int(a["id"]) for a in c.sink_mask_chain(E.serialize_vec(_bb(dets[j])), j * CLK))
            dropped.extend(c.take_dropped())
            wire += c.take_track_export()
        held_max = max(held_max, min(i, n - 1) + 1 - len(forwarded) - len(dropped))
    c.eos("sink_enc")
    forwarded.extend(int(a["id"]) for a in c.eos("sink_mask"))
    dropped.extend(c.take_dropped())
    wire += c.take_track_export()
    assert sorted(forwarded + dropped) == list(range(n))          # every id exactly once
    assert len(dropped) == c.dropped + (n - c.dropped - c.decoded_dependency - c.decoded_inference)
    assert held_max <= 2 * gop + lead + 60                        # bounded by the GoP window, not by the stream length
    # the three tracks left the tracker as length-delimited bincode Frames with the tracker's range_start
    frames, off = [], 0
    while off < len(wire):
        (fl,) = struct.unpack_from(">I", wire, off)
        frames.append(wire[off + 4:off + 4 + fl])
        off += 4 + fl
    assert off == len(wire) and len(frames) == 3
    for f in frames:
        range_start, oldest, nb = struct.unpack_from("<QQQ", f, 0)
        assert range_start == 0 and nb >= 5
    a = E.Associator([0])
    for f in frames:
        a.push_track_frame(f)                                     # the aggregator side parses them (track.rs:47-66)


def test_tracks_flushed_at_eos_carry_the_oldest_start_taken_before_the_flush():
    """cova/tracker.rs:91-118: flush() reads get_oldest_timestamp() BEFORE Sort::finalize() removes the active trackers, over
    all live ones; the aggregator retires pending detections by Frame.oldest (analysis-aggregator assoc.rs finalize_dnn).
    Two objects are still tracked when the stream ends: both EOS frames carry the start of the older one."""
    import struct
    n, gop = 500, 250
    objects = [(10, 80, 5, 5, 0.4, 0.2, 6, 6), (100, n - 1, 60, 30, -0.05, 0.0, 8, 5), (300, n - 1, 10, 10, 0.1, 0.05, 7, 7)]
    dets = _timeline(n, gop, objects)
    c = E.Cova(sort_maxage=10, sort_minhits=5, sort_iou=0.1)
    wire = b""
    for i in range(n):
        c.sink_enc_chain(i, i * CLK, delta_unit=(i % gop != 0))
    for i in range(n):
        c.sink_mask_chain(E.serialize_vec(_bb(dets[i])), i * CLK)
        wire += c.take_track_export()
    during = len(wire)
    c.eos("sink_enc")
    c.eos("sink_mask")
    wire += c.take_track_export()
    frames, off = [], 0
    while off < len(wire):
        (fl,) = struct.unpack_from(">I", wire, off)
        frames.append((off >= during, wire[off + 4:off + 4 + fl]))
        off += 4 + fl
    at_eos = [f for late, f in frames if late]
    assert len(frames) == 3 and len(at_eos) == 2
    for f in at_eos:
        range_start, oldest, nb = struct.unpack_from("<QQQ", f, 0)
        assert range_start == 0 and oldest == 100 * CLK and nb > 5     # range_start = PTS of the first update (tracker.rs:45)
    # the track that died during the stream (frame 91, before the second object appears): no tracker is left after that
    # update, and the reference folds from u64::MAX (tracker.rs:85-90)
    range_start, oldest, nb = struct.unpack_from("<QQQ", [f for late, f in frames if not late][0], 0)
    assert oldest == 2 ** 64 - 1
