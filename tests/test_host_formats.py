"""tfrecordsink / bboxsink / track-export byte formats (C-ABI) against independent Python parsers.
Reference: cova-rs/gst-plugins/src/tfrecordsink/imp.rs:69-198, bboxsink/imp.rs:252-270,
cova/tracker.rs:59-83."""
import ctypes as C
import csv
import io
import struct

import numpy as np

from cova_amd import _lib as L
from cova_amd import elements as E


def _crc32c(data: bytes) -> int:            # bitwise Castagnoli CRC (independent of the table version in C++)
    crc = 0xFFFFFFFF
    for b in data:
        crc ^= b
        for _ in range(8):
            crc = (crc >> 1) ^ (0x82F63B78 if crc & 1 else 0)
    return crc ^ 0xFFFFFFFF


def _masked(data: bytes) -> int:
    c = _crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(buf, off):
    v = shift = 0
    while True:
        b = buf[off]
        off += 1
        v |= (b & 0x7F) << shift
        shift += 7
        if not b & 0x80:
            return v, off


def _fields(buf):
    off, out = 0, []
    while off < len(buf):
        key, off = _varint(buf, off)
        assert key & 7 == 2                                 # every field here is length-delimited
        n, off = _varint(buf, off)
        out.append((key >> 3, buf[off:off + n]))
        off += n
    return out


def _parse_example(data: bytes):
    (f1, features), = _fields(data)
    assert f1 == 1
    out = {}
    for fno, entry in _fields(features):
        assert fno == 1
        kv = dict(_fields(entry))
        name = kv[1].decode()
        (one, blist), = _fields(kv[2])
        assert one == 1                                     # Feature.bytes_list
        out[name] = [v for f, v in _fields(blist) if f == 1]
    return out


def _example(rgba, gt, pad, w, h):
    lib = L.lib()
    n = rgba.shape[0]
    need = lib.covahip_tfrecord_example(rgba.ctypes.data, gt.ctypes.data if gt is not None else None, n, pad, w, h, None, 0, None)
    out = np.zeros(need, np.uint8)
    st = C.c_int()
    lib.covahip_tfrecord_example(rgba.ctypes.data, gt.ctypes.data if gt is not None else None, n, pad, w, h, out.ctypes.data,
                                 need, C.byref(st))
    assert st.value == 0
    return out.tobytes()


def test_tfrecord_framing_and_example_content():
    rng = np.random.default_rng(0)
    w, h = 20, 15
    for n, pad in ((1, 0), (3, 5), (2, 2)):
        rgba = rng.integers(0, 256, (n, h, w, 4), dtype=np.uint8)
        gt = rng.integers(0, 2, (n, h * w), dtype=np.uint8)
        rec = _example(rgba, gt, pad, w, h)
        (length,) = struct.unpack_from("<Q", rec, 0)
        assert len(rec) == 8 + 4 + length + 4
        assert struct.unpack_from("<I", rec, 8)[0] == _masked(rec[:8])
        data = rec[12:12 + length]
        assert struct.unpack_from("<I", rec, 12 + length)[0] == _masked(data)
        ex = _parse_example(data)
        assert sorted(ex) == ["gt", "mb_type", "mv_x", "mv_y"]
        total = max(n, pad)
        for ch, name in enumerate(("mb_type", "mv_x", "mv_y")):       # bytes 0/1/2 of every RGBA pixel
            assert len(ex[name]) == total
            for i in range(n):
                assert ex[name][i] == rgba[i, :, :, ch].tobytes()
            for i in range(n, total):
                assert ex[name][i] == bytes(w * h)                      # zero fill up to `gop`
        for i in range(n):
            assert ex["gt"][i] == gt[i].tobytes()


def test_crc32c_known_answer():
    # RFC 3720 test vector: 32 bytes of zeros -> 0x8A9136AA
    assert _crc32c(bytes(32)) == 0x8A9136AA


def test_bbox_csv_matches_serde_csv_conventions():
    lib = L.lib()
    b = np.zeros(3, dtype=L.BBOX_DTYPE)
    b[0] = E.make_bbox(0, 0, 2, 2)[0]
    b[1] = E.make_bbox(1.5, 2.25, 3, 4)[0]
    b[1]["has_track_id"], b[1]["track_id"] = 1, 7
    b[1]["has_timestamp"], b[1]["timestamp"] = 1, 33333333
    b[2] = E.make_bbox(0.1, 1e-7, 1e21, 3)[0]
    b[2]["has_class_id"], b[2]["class_id"] = 1, 2
    b[2]["has_confidence"], b[2]["confidence"] = 1, 0.5
    need = lib.covahip_bbox_csv(b.ctypes.data, 3, 1, None, 0, None)
    out = C.create_string_buffer(need)
    st = C.c_int()
    lib.covahip_bbox_csv(b.ctypes.data, 3, 1, out, need, C.byref(st))
    text = out.raw.decode()
    rows = list(csv.reader(io.StringIO(text)))
    assert rows[0] == ["left", "top", "width", "height", "area", "track_id", "timestamp", "class_id", "confidence"]
    assert rows[1] == ["0.0", "0.0", "2.0", "2.0", "4.0", "", "", "", ""]           # ryu: integral floats keep ".0"
    assert rows[2] == ["1.5", "2.25", "3.0", "4.0", "12.0", "7", "33333333", "", ""]
    assert rows[3][0] == "0.1" and rows[3][1] == "1e-7" and rows[3][2] == "1e21"     # shortest round-trip, e-notation
    assert rows[3][7:] == ["2", "0.5"]
    for r, src in zip(rows[1:], b):                                                  # values round-trip exactly as f32
        for k, name in enumerate(("left", "top", "width", "height", "area")):
            assert np.float32(float(r[k])) == src[name]


def test_track_export_frames():
    lib = L.lib()
    s = E._SortHandle(2, 2, 0.1)
    dead_all, lens_all = [], []
    for i in range(12):
        dets = np.zeros(1, dtype=L.BBOX_DTYPE)
        dets[0] = E.make_bbox(10 + i, 10, 5, 5)[0]
        dead, lens = s.update(dets if i < 7 else dets[:0], i * 100)
        if len(lens):
            dead_all.append(dead)
            lens_all.append(lens)
    assert dead_all
    boxes = np.concatenate(dead_all)
    lens = np.concatenate(lens_all).astype(np.uint32)
    need = lib.covahip_tracks_export(5, 9, boxes.ctypes.data, lens.ctypes.data, len(lens), None, 0, None)
    out = np.zeros(need, np.uint8)
    st = C.c_int()
    lib.covahip_tracks_export(5, 9, boxes.ctypes.data, lens.ctypes.data, len(lens), out.ctypes.data, need, C.byref(st))
    assert st.value == 0
    raw, off, k0 = out.tobytes(), 0, 0
    for n in lens:                                           # LengthDelimitedCodec: u32 big-endian length prefix
        (fl,) = struct.unpack_from(">I", raw, off)
        frame = raw[off + 4:off + 4 + fl]
        assert frame == E.serialize_frame(5, 9, boxes[k0:k0 + n])
        off += 4 + fl
        k0 += n
    assert off == len(raw)


def test_python_wrappers_agree_with_raw_calls():
    rng = np.random.default_rng(3)
    rgba = rng.integers(0, 256, (2, 6, 8, 4), dtype=np.uint8)
    gt = rng.integers(0, 2, (2, 6, 8), dtype=np.uint8)
    assert E.tfrecord_example(rgba, gt, gop=4) == _example(rgba, gt.reshape(2, -1), 4, 8, 6)
    b = E.make_bbox(1, 2, 3, 4)
    assert E.bbox_csv(b, with_header=False) == "1.0,2.0,3.0,4.0,12.0,,,,\n"
    assert E.tracks_export(0, 0, b, np.array([1], np.uint32)) == \
        struct.pack(">I", len(E.serialize_frame(0, 0, b))) + E.serialize_frame(0, 0, b)


def test_carrier_pack_records():
    """covahip_carrier_pack: min(type, 6) | min(mv_x, 6) << 3 | min(mv_y, 6) << 6 per macroblock, the fourth byte ignored --
    vector body and scalar tail (a length that is not a multiple of 16)."""
    from cova_amd.elements import pack_frames
    rng = np.random.default_rng(1)
    for shape in ((3, 68, 120, 4), (1, 5, 7, 4), (2, 45, 80, 4)):
        f = rng.integers(0, 256, size=shape, dtype=np.uint8)
        f[0] = rng.integers(0, 9, size=shape[1:])
        m = np.minimum(f[..., :3].astype(np.uint16), 6)
        np.testing.assert_array_equal(pack_frames(f), m[..., 0] | (m[..., 1] << 3) | (m[..., 2] << 6))
