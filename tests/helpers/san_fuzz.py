"""Seeded corruption of everything the host C-ABI parses from untrusted bytes -- bincode bbox vectors and frames, track-export
payloads, aggregator text, MP4 boxes, avcC, H.264 access units -- run by tests/test_sanitize_host.py in a python that has
the ASan runtime preloaded and COVAHIP_HOST_SAN_LIB pointing at the ASan + UBSan build of the HIP-free translation units
(make -C cova_amd/csrc host-san).  Every call must come back with a status; a crash, a sanitizer report (abort) or a hang
fails the test.  Prints one JSON line with the case counts."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cova_amd import _lib as L  # noqa: E402
from cova_amd import elements as E  # noqa: E402

DEMO = "/root/reference/demo/1m.mp4"
lib = L.lib()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
SCALE = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
counts = {}
EXTREMES = [b"\xff\xff\xff\xff", b"\x00\x00\x00\x00", b"\x7f\xff\xff\xff", b"\x80\x00\x00\x00", b"\x00\x00\x00\x01",
            b"\xff\xff\xff\x7f", b"\x00\x00\x00\x08", b"\x00\x01\x00\x00"]


def extreme(n=4):
    return np.frombuffer(EXTREMES[int(rng.integers(0, len(EXTREMES)))], dtype=np.uint8)[:n]


def corrupt(buf: np.ndarray, lo: int, hi: int):
    """One of: flip random bytes, overwrite a 32-bit field with an extreme value, truncate.  Returns (bytes, undo)."""
    kind = rng.integers(0, 4)
    b = buf.copy()
    if kind == 0:
        for _ in range(int(rng.integers(1, 9))):
            b[rng.integers(lo, hi)] = rng.integers(0, 256)
    elif kind == 1:
        at = int(rng.integers(lo, max(lo + 1, hi - 4)))
        b[at:at + 4] = extreme(len(b[at:at + 4]))
    elif kind == 2:
        b = b[:int(rng.integers(0, len(b) + 1))].copy()
    else:
        at = int(rng.integers(lo, hi))
        n = int(rng.integers(1, 64))
        b[at:at + n] = rng.integers(0, 256, len(b[at:at + n]), dtype=np.uint8)
    return b


def fuzz_bincode(n_cases):
    boxes = np.zeros(7, dtype=L.BBOX_DTYPE)
    for i in range(7):
        boxes[i] = E.make_bbox(3.5 * i, 2 + i, 4 + i, 5)[0]
        boxes[i]["has_track_id"], boxes[i]["track_id"] = i & 1, 40 + i
        boxes[i]["has_timestamp"], boxes[i]["timestamp"] = (i >> 1) & 1, 33_333_333 * i
        boxes[i]["has_class_id"], boxes[i]["class_id"] = (i >> 2) & 1, 2
        boxes[i]["has_confidence"], boxes[i]["confidence"] = i % 3 == 0, 0.5
    st = C.c_int()
    out = np.zeros(4096, dtype=np.uint8)
    n = lib.covahip_bbox_serialize_vec(boxes.ctypes.data, len(boxes), out.ctypes.data, out.size, C.byref(st))
    assert st.value == 0 and n > 0
    vec = out[:n].copy()
    fo = np.zeros(8192, dtype=np.uint8)
    nf = lib.covahip_frame_serialize(1000, 2000, boxes.ctypes.data, len(boxes), fo.ctypes.data, fo.size, C.byref(st))
    assert st.value == 0 and nf > 0
    frame = fo[:nf].copy()
    lens = np.array([3, 4], dtype=np.uint32)
    nt = lib.covahip_tracks_export(1000, 2000, boxes.ctypes.data, lens.ctypes.data, 2, fo.ctypes.data, fo.size, C.byref(st))
    assert st.value == 0 and nt > 0
    tracks = fo[:nt].copy()
    got = np.zeros(64, dtype=L.BBOX_DTYPE)
    ngot = C.c_size_t()
    cfg = L.AssocCfg(0.3, 0.5, 10, 1.0)
    starts = np.array([1000, 5000], dtype=np.uint64)
    for k in range(n_cases):
        b = corrupt(vec, 0, len(vec))
        rc = lib.covahip_bbox_deserialize_vec(b.ctypes.data if b.size else None, b.size, got.ctypes.data, int(rng.integers(0, 65)), C.byref(ngot))
        assert isinstance(rc, int)
        a = C.c_void_p()
        assert lib.covahip_assoc_new(C.byref(cfg), starts.ctypes.data, 2, C.byref(a)) == 0
        src = tracks if k & 1 else frame
        t = corrupt(src, 0, len(src))
        lib.covahip_assoc_push_track_frame(a, t.ctypes.data if t.size else None, t.size)
        if k % 4 == 0:
            txt = bytes(corrupt(np.frombuffer(b"0,10.5,20.0,30.0,40.0,2,0.9\n1,1,2,3,4,0,0.5\n", dtype=np.uint8), 0, 40))
            lib.covahip_assoc_push_dnn_text(a, txt, len(txt))
            lib.covahip_assoc_terminate(a)
            csv = C.create_string_buffer(1 << 14)
            lib.covahip_assoc_csv(a, int(rng.integers(0, 3)), csv, len(csv), C.byref(st))
        lib.covahip_assoc_free(a)
    counts["bincode_vec_frame_tracks_text"] = n_cases


def moov_range(data):
    raw = data.tobytes()
    at = raw.find(b"moov")
    assert at >= 4
    size = int.from_bytes(raw[at - 4:at], "big")
    return at - 4, min(len(raw), at - 4 + size)


def fuzz_mp4(n_cases):
    data = np.fromfile(DEMO, dtype=np.uint8)
    lo, hi = moov_range(data)
    # the file with everything but the container boxes and the first 64 access units' bytes cut off would change offsets: keep
    # the whole file, corrupt the moov box (sample tables, avcC inside stsd) in place and restore
    info = np.zeros(1, dtype=L.H264_INFO_DTYPE)
    sl = np.zeros(4, dtype=L.H264_SLICE_DTYPE)
    rec = np.zeros(45 * 80 * 4, dtype=np.uint8)
    n = C.c_int()
    opened = 0
    for k in range(n_cases):
        kind = rng.integers(0, 3)
        saved = []
        if kind == 0:       # a few bytes anywhere in moov
            for _ in range(int(rng.integers(1, 6))):
                at = int(rng.integers(lo, hi))
                saved.append((at, data[at:at + 1].copy()))
                data[at] = rng.integers(0, 256)
        elif kind == 1:     # a 32-bit field (box size, entry count, offset) set to an extreme
            at = int(rng.integers(lo, hi - 4)) & ~3
            saved.append((at, data[at:at + 4].copy()))
            data[at:at + 4] = extreme()
        else:               # a run of random bytes
            at = int(rng.integers(lo, hi - 64))
            m = int(rng.integers(2, 64))
            saved.append((at, data[at:at + m].copy()))
            data[at:at + m] = rng.integers(0, 256, m, dtype=np.uint8)
        length = data.size if k % 5 else int(rng.integers(0, data.size))       # every fifth case: a truncated file
        h = C.c_void_p()
        rc = lib.covahip_h264_open_mp4(data.ctypes.data, length, C.byref(h))
        if rc == 0:
            opened += 1
            lib.covahip_h264_get_info(h, info.ctypes.data)
            ns = int(info[0]["n_samples"])
            for s in (0, 1, ns // 2, ns - 1, ns, -1):
                off, size, sync = C.c_uint64(), C.c_uint32(), C.c_int()
                lib.covahip_h264_sample(h, s, C.byref(off), C.byref(size), C.byref(sync))
                lib.covahip_h264_sample_slices(h, s, sl.ctypes.data, 4, C.byref(n))
            if 0 < int(info[0]["width_mbs"]) * int(info[0]["height_mbs"]) * 4 <= rec.size:
                for s in (0, 1, 2):
                    lib.covahip_h264_decode_records(h, s, rec.ctypes.data, rec.size)
            order = np.zeros(64, dtype=np.int32)
            lib.covahip_h264_display_order(h, order.ctypes.data, 64, C.byref(n))
            lib.covahip_h264_close(h)
        for at, old in saved:
            data[at:at + len(old)] = old
    counts["mp4_boxes"] = n_cases
    counts["mp4_opened_despite_corruption"] = opened
    return data


def fuzz_avcc_and_access_units(data, n_cases):
    raw = data.tobytes()
    at = raw.find(b"avcC")
    size = int.from_bytes(raw[at - 4:at], "big")
    avcc = np.frombuffer(raw[at + 4:at - 4 + size], dtype=np.uint8).copy()
    h = C.c_void_p()
    assert lib.covahip_h264_open_mp4(data.ctypes.data, data.size, C.byref(h)) == 0
    aus = []
    for s in range(12):                      # one IDR, P and B pictures of the first GoP, in decode order
        off, sz, sync = C.c_uint64(), C.c_uint32(), C.c_int()
        assert lib.covahip_h264_sample(h, s, C.byref(off), C.byref(sz), C.byref(sync)) == 0
        aus.append(data[off.value:off.value + sz.value].copy())
    lib.covahip_h264_close(h)
    rec = np.zeros(45 * 80 * 4, dtype=np.uint8)
    hdr = np.zeros(1, dtype=L.H264_SLICE_DTYPE)
    key = C.c_int64()
    ok_parsed = 0
    for k in range(n_cases):
        a = corrupt(avcc, 0, len(avcc)) if k % 3 == 0 else avcc
        hs = C.c_void_p()
        rc = lib.covahip_h264_open_avcc(a.ctypes.data if a.size else None, a.size, C.byref(hs))
        if rc != 0:
            continue
        info = np.zeros(1, dtype=L.H264_INFO_DTYPE)
        lib.covahip_h264_get_info(hs, info.ctypes.data)
        cap = rec.size if 0 < int(info[0]["width_mbs"]) * int(info[0]["height_mbs"]) * 4 <= rec.size else 0
        for s, au in enumerate(aus[:int(rng.integers(1, len(aus) + 1))]):
            b = corrupt(au, 0, len(au)) if rng.integers(0, 3) else au         # two of three access units corrupted
            r2 = lib.covahip_h264_decode_au(hs, b.ctypes.data if b.size else None, b.size, rec.ctypes.data, cap, hdr.ctypes.data, C.byref(key))
            ok_parsed += r2 == 0
        lib.covahip_h264_close(hs)
    counts["avcc_and_access_units"] = n_cases
    counts["access_units_accepted"] = int(ok_parsed)


fuzz_bincode(int(8000 * SCALE))
if os.path.exists(DEMO):
    d = fuzz_mp4(int(1200 * SCALE))                      # (~20 ms per case under ASan: a 4 MB file's tables per open)
    fuzz_avcc_and_access_units(d, int(1200 * SCALE))
counts["total_cases"] = sum(v for k, v in counts.items() if k in ("bincode_vec_frame_tracks_text", "mp4_boxes", "avcc_and_access_units"))
print(json.dumps(counts))
