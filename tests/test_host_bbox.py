"""Bbox / bincode wire format through the C-ABI vs the reference's KATs and hand-derived bytes."""
import json
import os
import struct

import numpy as np
import pytest

from cova_amd import _lib as L
from cova_amd import elements as E

KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))


def test_iou_reference_kats():
    for c in KATS["iou"]["cases"]:
        got = E.iou(E.make_bbox(*c["a"]), E.make_bbox(*c["b"]))
        exp = np.float32(c["expected_num"]) / np.float32(c["expected_den"])   # f32 division as in the Rust test
        assert np.float32(got) == exp


def test_bbox_new_area_is_box_area():
    boxes = np.zeros(1, dtype=L.BOX_DTYPE)
    boxes[0] = (3, 4, 5, 6, 11)       # pixel count 11 must NOT become Bbox.area (bbox.rs:23)
    bb = E.boxes_to_bbox(boxes)
    assert tuple(float(bb[0][k]) for k in ("left", "top", "width", "height", "area")) == (3, 4, 5, 6, 30)
    assert not bb[0]["has_track_id"] and not bb[0]["has_timestamp"]


def test_bincode_vec_bytes_exact():
    # bincode 1.3 default: u64 LE len; f32 LE fields; Option = 1-byte tag (+ payload)
    assert E.serialize_vec(np.zeros(0, dtype=L.BBOX_DTYPE)) == struct.pack("<Q", 0)
    a = E.make_bbox(0, 0, 2, 2)
    assert E.serialize_vec(a) == struct.pack("<Q5f4B", 1, 0, 0, 2, 2, 4, 0, 0, 0, 0)
    b = E.make_bbox(1.5, 2.5, 3, 4)
    b["has_track_id"], b["track_id"] = 1, 7
    b["has_timestamp"], b["timestamp"] = 1, 33_333_333
    b["has_class_id"], b["class_id"] = 1, 2
    b["has_confidence"], b["confidence"] = 1, 0.5
    exp = struct.pack("<Q5f", 1, 1.5, 2.5, 3, 4, 12) + struct.pack("<BQ", 1, 7) + struct.pack("<BQ", 1, 33_333_333) \
        + struct.pack("<BI", 1, 2) + struct.pack("<Bf", 1, 0.5)
    got = E.serialize_vec(b)
    assert got == exp and len(got) == 8 + 48
    # tracker output boxes: track_id + timestamp set -> 40 bytes per box
    c = E.make_bbox(1, 1, 1, 1)
    c["has_track_id"], c["has_timestamp"] = 1, 1
    assert len(E.serialize_vec(c)) == 8 + 40


def test_bincode_roundtrip_vec():           # bbox.rs:125-130 test_serde_vec
    rng = np.random.default_rng(0)
    v = np.zeros(17, dtype=L.BBOX_DTYPE)
    for k in ("left", "top", "width", "height", "area", "confidence"):
        v[k] = rng.random(17).astype(np.float32)
    v["track_id"] = rng.integers(0, 2**62, 17)
    v["timestamp"] = rng.integers(0, 2**62, 17)
    v["class_id"] = rng.integers(0, 2**31, 17)
    for k in ("has_track_id", "has_timestamp", "has_class_id", "has_confidence"):
        v[k] = rng.integers(0, 2, 17)
    back = E.deserialize_vec(E.serialize_vec(v))
    for i in range(17):
        for k in ("left", "top", "width", "height", "area"):
            assert back[i][k] == v[i][k]
        for has, k in (("has_track_id", "track_id"), ("has_timestamp", "timestamp"), ("has_class_id", "class_id"),
                       ("has_confidence", "confidence")):
            assert back[i][has] == v[i][has]
            if v[i][has]:
                assert back[i][k] == v[i][k]


def test_frame_bytes():                      # bbox/src/lib.rs:8-22
    a = E.make_bbox(0, 0, 2, 2)
    got = E.serialize_frame(5, 9, a)
    assert got == struct.pack("<QQ", 5, 9) + E.serialize_vec(a)


@pytest.mark.parametrize("bad", [b"", b"\x01", struct.pack("<Q", 3) + b"\x00" * 24,
                                 struct.pack("<Q5f4B", 1, 0, 0, 2, 2, 4, 2, 0, 0, 0),       # Option tag 2
                                 struct.pack("<Q5f4B", 1, 0, 0, 2, 2, 4, 0, 0, 0, 0) + b"x"])  # trailing byte
def test_bincode_malformed_rejected(bad):
    with pytest.raises(L.CovahipError):
        E.deserialize_vec(bad)
