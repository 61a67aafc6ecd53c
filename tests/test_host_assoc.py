"""analysis-aggregator association rules behind the C-ABI (covahip_assoc_*) vs oracle/assoc_ref.py
(reference: cova-rs/analysis-aggregator/src/server/assoc.rs, track.rs, dnn.rs)."""
import struct

import numpy as np
import pytest

from cova_amd import _lib as L
from cova_amd import elements as E
from oracle import assoc_ref as A
from oracle.sort_ref import Bbox

TS = 33_333_333


def _np(boxes):
    out = np.zeros(len(boxes), dtype=L.BBOX_DTYPE)
    for i, b in enumerate(boxes):
        out[i] = E.make_bbox(b.left, b.top, b.width, b.height)[0]
        out[i]["area"] = b.area
        for k in ("track_id", "timestamp", "class_id"):
            v = getattr(b, k)
            if v is not None:
                out[i][k], out[i]["has_" + k] = v, 1
    return out


def _track(tid, f0, n, x0, y0, vx, w=40.0, h=30.0):
    return [Bbox(x0 + vx * i, y0, w, h, track_id=tid, timestamp=(f0 + i) * TS) for i in range(n)]


def _det(frame, x, y, w, h, cls):
    return Bbox(x, y, w, h, timestamp=frame * TS, class_id=cls)


def _scenario(seed, n_trackers=2):
    """A message sequence as the aggregator's channel could deliver it: interleaved detections (every third
    frame) and finished tracks, per tracker range."""
    rng = np.random.default_rng(seed)
    range_starts = [r * 3000 * TS for r in range(n_trackers)]
    msgs = []
    for r, rs in enumerate(range_starts):
        f_base = r * 3000
        tracks = []
        for k in range(int(rng.integers(3, 7))):
            f0 = f_base + int(rng.integers(0, 300)) * 3
            tracks.append(_track(rs + k + 1, f0, int(rng.integers(10, 90)), float(rng.uniform(0, 900)), float(rng.uniform(0, 500)),
                                 float(rng.uniform(-6, 6)), float(rng.uniform(30, 120)), float(rng.uniform(30, 120))))
        stat = [(float(rng.uniform(0, 1000)), float(rng.uniform(0, 600)), 50.0, 40.0, int(rng.integers(0, 4)))
                for _ in range(int(rng.integers(1, 3)))]
        events = []   # (frame, order, kind, payload)
        for trk in tracks:
            end_f = trk[-1].timestamp // TS
            events.append((end_f + int(rng.integers(5, 40)), 1, "track", trk))
        for f in range(f_base, f_base + 1300, 3):
            dets = []
            for trk in tracks:
                f0, f1 = trk[0].timestamp // TS, trk[-1].timestamp // TS
                if f0 <= f <= f1 and rng.random() < 0.8:
                    b = trk[f - f0]
                    dets.append(_det(f, b.left + rng.uniform(-8, 8), b.top + rng.uniform(-8, 8), b.width * rng.uniform(0.8, 1.3),
                                     b.height * rng.uniform(0.8, 1.3), int(rng.integers(0, 3))))
            for (x, y, w, h, c) in stat:
                if f < f_base + 600 and rng.random() < 0.9:
                    dets.append(_det(f, x + rng.uniform(-2, 2), y + rng.uniform(-2, 2), w, h, c))
            if rng.random() < 0.1:
                dets.append(_det(f, float(rng.uniform(0, 1000)), float(rng.uniform(0, 600)), 20.0, 20.0, 5))
            if dets:
                events.append((f, 0, "dnn", dets))
        events.sort(key=lambda e: (e[0], e[1]))
        oldest = 0
        for f, _, kind, payload in events:
            if kind == "track":
                oldest = max(oldest, (f - 45) * TS)
                msgs.append(("track", rs, oldest, payload))
            else:
                msgs.append(("dnn", payload))
    return range_starts, msgs


def _run_both(range_starts, msgs, **kw):
    ref = A.Associator(range_starts, **kw)
    dut = E.Associator(range_starts, **kw)
    for m in msgs:
        if m[0] == "track":
            _, rs, oldest, trk = m
            ref.update_track(rs, oldest, [Bbox(b.left, b.top, b.width, b.height, b.area, b.track_id, b.timestamp) for b in trk])
            dut.push_track(rs, oldest, _np(trk))
        else:
            ref.update_dnn([Bbox(b.left, b.top, b.width, b.height, b.area, None, b.timestamp, b.class_id) for b in m[1]])
            dut.push_dnn(_np(m[1]))
    ref.terminate()
    dut.terminate()
    return ref, dut


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_association_matches_restatement(seed):
    range_starts, msgs = _scenario(seed)
    ref, dut = _run_both(range_starts, msgs, stationary_maxage=5)
    for name in ("track", "dnn", "assoc", "stationary"):
        assert dut.csv(name) == A.csv_text(ref.rows[name]), name
    assert len(ref.rows["assoc"]) > 0 and len(ref.rows["stationary"]) > 0   # the scenario exercises both paths


def test_class_vote_rules_and_tie_order():
    """assoc.rs:161-203: the most frequent class first; then every class seen at least twice, or -- when the maximum
    is one -- every class seen."""
    rs = [0]
    trk = _track(7, 0, 10, 100.0, 100.0, 0.0)
    for votes, expect in (([2, 2, 1, 1, 3], [2, 1]), ([4, 1, 3], [4, 1, 3]), ([5, 5, 5, 0], [5])):
        ref = A.Associator(rs)
        dut = E.Associator(rs)
        for a in (ref, ):
            a.update_track(0, 0, [Bbox(b.left, b.top, b.width, b.height, b.area, b.track_id, b.timestamp) for b in trk])
        dut.push_track(0, 0, _np(trk))
        for i, c in enumerate(votes):
            d = [_det(i, 100.0, 100.0, 40.0, 30.0, c)]
            ref.update_dnn([Bbox(x.left, x.top, x.width, x.height, x.area, None, x.timestamp, x.class_id) for x in d])
            dut.push_dnn(_np(d))
        late = [_det(50, 900.0, 900.0, 10.0, 10.0, 9)]           # a detection after the track's end finalises it
        ref.update_dnn([Bbox(x.left, x.top, x.width, x.height, x.area, None, x.timestamp, x.class_id) for x in late])
        dut.push_dnn(_np(late))
        text = dut.csv("assoc")
        assert text == A.csv_text(ref.rows["assoc"])
        classes = [int(r.split(",")[7]) for r in text.splitlines()[1:]]
        assert [classes[i] for i in range(0, len(classes), len(trk))] == expect


def test_strict_vs_non_strict_iou_quirk():
    """A detection arriving BEFORE its track matches with iou > moving_iou (assoc.rs:409), one arriving after it
    with iou >= moving_iou (assoc.rs:343)."""
    trk = _track(1, 0, 5, 0.0, 0.0, 0.0, 10.0, 10.0)
    tb = Bbox(0.0, 0.0, 10.0, 10.0)
    det = _det(2, 5.0, 0.0, 10.0, 10.0, 1)
    thr = float(tb.iou(det))                                      # scale_factor 1: compare the boxes as they are
    for first in ("dnn", "track"):
        dut = E.Associator([0], moving_iou=thr, scale_factor=1.0)
        if first == "dnn":
            dut.push_dnn(_np([det]))
            dut.push_track(0, 0, _np(trk))
        else:
            dut.push_track(0, 0, _np(trk))
            dut.push_dnn(_np([det]))
        dut.push_dnn(_np([_det(20, 500.0, 500.0, 5.0, 5.0, 3)]))  # finalises the track
        assert (len(dut.csv("assoc")) > 0) == (first == "track")


def test_tracks_pending_at_terminate_are_not_written():
    dut = E.Associator([0])
    dut.push_track(0, 0, _np(_track(1, 0, 5, 0.0, 0.0, 1.0)))
    dut.push_dnn(_np([_det(2, 2.0, 0.0, 40.0, 30.0, 1)]))
    dut.terminate()
    assert dut.csv("assoc") == "" and dut.csv("track").count("\n") == 6 and dut.csv("dnn").count("\n") == 2


def test_wire_ingest_scales_to_pixels_and_rebases_ids():
    """track.rs:59-66 on one payload of covahip_tracks_export; dnn.rs:57-86 on a split text stream."""
    rs = 5_000_000_000
    mb = _np([Bbox(1.0, 2.0, 3.0, 4.0, None, 3, rs + i * TS) for i in range(4)])
    frames = E.tracks_export(rs, rs, mb, np.array([4], np.uint32))
    (fl,) = struct.unpack_from(">I", frames, 0)
    dut = E.Associator([rs])
    dut.push_track_frame(frames[4:4 + fl])
    rows = dut.csv("track").splitlines()
    assert rows[1] == f"16.0,32.0,48.0,64.0,3072.0,{rs + 3},{rs},,"
    text = f"{rs + TS},10,20,30,40,2\n{rs + TS},1.5,2.5,3,4,0\n{rs + 2 * TS},7,7,7,7,1\n".encode()
    dut.push_dnn_text(text[:25])
    dut.push_dnn_text(text[25:])
    d = dut.csv("dnn").splitlines()
    assert d[1] == f"10.0,20.0,30.0,40.0,1200.0,,{rs + TS},2," and d[2].startswith("1.5,2.5,3.0,4.0,12.0,,") and len(d) == 4
    with pytest.raises(L.CovahipError):
        dut.push_dnn_text(b"1,2,3\n")


def test_errors():
    dut = E.Associator([0])
    with pytest.raises(L.CovahipError):
        dut.push_track(123, 0, _np(_track(1, 0, 3, 0.0, 0.0, 0.0)))     # unknown range_start (the reference unwraps)
    dut.push_track(0, 0, _np(_track(1, 0, 3, 0.0, 0.0, 0.0)))
    gap = _det(1, 0.0, 0.0, 40.0, 30.0, 1)
    gap.timestamp += 5                                                   # no track box at that timestamp
    with pytest.raises(L.CovahipError):
        dut.push_dnn(_np([gap]))
