"""Entropy-decode front end, container + header layer on the reference's own demo video.
No reference output exists for this stage (the patched FFmpeg is an un-vendored submodule), so the checks are
structural: what the container and the H.264 headers of demo/1m.mp4 must say (README.md:94-114: 1280x720, 1,802
frames, GoP 250, High@3.1)."""
import ctypes as C
import os

import numpy as np
import pytest

from cova_amd import _lib as L

DEMO = "/root/reference/demo/1m.mp4"
pytestmark = pytest.mark.skipif(not os.path.exists(DEMO), reason="reference demo video not present (GPU box)")


@pytest.fixture(scope="module")
def demo():
    lib = L.lib()
    data = np.fromfile(DEMO, dtype=np.uint8)
    h = C.c_void_p()
    assert lib.covahip_h264_open_mp4(data.ctypes.data, data.size, C.byref(h)) == 0
    yield lib, h, data
    lib.covahip_h264_close(h)


def test_stream_parameters(demo):
    lib, h, _ = demo
    info = np.zeros(1, dtype=L.H264_INFO_DTYPE)
    assert lib.covahip_h264_get_info(h, info.ctypes.data) == 0
    i = info[0]
    assert (i["width_mbs"], i["height_mbs"], i["n_samples"]) == (80, 45, 1802)      # 1280x720, 1,802 access units
    assert (i["profile_idc"], i["level_idc"]) == (100, 31)                           # High@3.1
    assert i["entropy_cabac"] == 1 and i["transform_8x8"] == 1 and i["frame_mbs_only"] == 1
    assert i["num_ref_frames"] == 5 and i["poc_type"] == 0


def test_every_access_unit_has_one_slice_and_key_frames_every_250(demo):
    lib, h, data = demo
    sl = np.zeros(4, dtype=L.H264_SLICE_DTYPE)
    n = C.c_int()
    types, sync = [], []
    prev_end = None
    for s in range(1802):
        off, size, is_sync = C.c_uint64(), C.c_uint32(), C.c_int()
        assert lib.covahip_h264_sample(h, s, C.byref(off), C.byref(size), C.byref(is_sync)) == 0
        assert off.value + size.value <= data.size
        assert lib.covahip_h264_sample_slices(h, s, sl.ctypes.data, 4, C.byref(n)) == 0
        assert n.value == 1                                                          # one slice per picture
        x = sl[0]
        assert x["first_mb"] == 0 and x["nal_bytes"] <= size.value
        assert off.value <= x["nal_offset"] < off.value + size.value
        assert data[x["nal_offset"]] & 31 == x["nal_type"]
        assert x["data_bit_offset"] % 8 == 0                                         # cabac_alignment_one_bit: slice data starts on a byte
        assert 0 <= x["qp"] <= 51
        types.append(int(x["slice_type"]))
        if is_sync.value:
            sync.append(s)
            assert x["idr"] == 1 and x["nal_type"] == 5 and x["slice_type"] == 2 and x["frame_num"] == 0
            assert x["cabac_init_idc"] == -1
        else:
            assert x["idr"] == 0 and x["slice_type"] in (0, 1)                       # P and B pictures only between key frames
            assert x["cabac_init_idc"] in (0, 1, 2)
            assert 1 <= x["num_ref_l0"] <= 16 and 1 <= x["num_ref_l1"] <= 16    # x264 weightp duplicates references: more entries than frames
    assert sync == list(range(0, 1802, 250))                                          # GoP 250 (cova hard-codes it: cova/imp.rs:255-262)
    assert types.count(2) == 8 and types.count(0) + types.count(1) == 1794 and types.count(1) > types.count(0)


def test_macroblock_layer_is_refused_not_faked(demo):
    lib, h, _ = demo
    rec = np.zeros(80 * 45 * 4, np.uint8)
    assert lib.covahip_h264_decode_records(h, 0, rec.ctypes.data, rec.size) == 5    # COVAHIP_ERR_UNSUPPORTED
    assert not rec.any()


def test_truncated_and_foreign_files_are_rejected(demo):
    lib, _, data = demo
    h = C.c_void_p()
    assert lib.covahip_h264_open_mp4(data.ctypes.data, 1000, C.byref(h)) != 0       # moov sits behind mdat: not in the first KB
    junk = np.arange(4096, dtype=np.uint8)
    assert lib.covahip_h264_open_mp4(junk.ctypes.data, junk.size, C.byref(h)) != 0
    assert lib.covahip_h264_open_mp4(None, 0, C.byref(h)) == 1


def test_carrier_record_layout():
    """[mb_type, mv_x, mv_y, 0] per macroblock in raster order: what metapreprocess copies (imp.rs:311-312) and tfrecordsink
    splits into its three features (tfrecordsink/imp.rs:105-112)."""
    lib = L.lib()
    w, h = 5, 3
    rng = np.random.default_rng(0)
    mt, mx, my = (rng.integers(0, 8, w * h).astype(np.uint8) for _ in range(3))
    frame = np.full(w * h * 4 + 7, 0xAA, np.uint8)
    assert lib.covahip_carrier_write_records(mt.ctypes.data, mx.ctypes.data, my.ctypes.data, w, h, frame.ctypes.data, frame.size) == 0
    rec = frame[:w * h * 4].reshape(h, w, 4)
    np.testing.assert_array_equal(rec[..., 0].reshape(-1), mt)
    np.testing.assert_array_equal(rec[..., 1].reshape(-1), mx)
    np.testing.assert_array_equal(rec[..., 2].reshape(-1), my)
    assert not rec[..., 3].any() and (frame[w * h * 4:] == 0xAA).all()
    assert lib.covahip_carrier_write_records(mt.ctypes.data, mx.ctypes.data, my.ctypes.data, w, h, frame.ctypes.data, 10) == 7


def test_corrupted_headers_never_crash(demo):
    """The parser reads untrusted bytes: random corruption of the moov box / parameter sets / slice headers must end in
    an error status or in parsed values, never in a fault (the loop runs in this process: a fault would kill the run)."""
    lib, _, data = demo
    rng = np.random.default_rng(1)
    moov = int(np.flatnonzero((data[:-4] == ord("m")) & (data[1:-3] == ord("o")) & (data[2:-2] == ord("o")) & (data[3:-1] == ord("v")))[-1])
    sl = np.zeros(4, dtype=L.H264_SLICE_DTYPE)
    n = C.c_int()
    opened = 0
    for trial in range(300):
        d = data.copy()
        for _ in range(int(rng.integers(1, 12))):
            d[moov + int(rng.integers(0, data.size - moov))] = rng.integers(0, 256)      # inside moov (sample tables, avcC)
        if trial % 3 == 0:
            for _ in range(64):
                d[int(rng.integers(48, 200000))] = rng.integers(0, 256)                  # inside the first access units
        h = C.c_void_p()
        if lib.covahip_h264_open_mp4(d.ctypes.data, d.size, C.byref(h)) == 0:
            opened += 1
            for s in (0, 1, 2, 250, 1801):
                lib.covahip_h264_sample_slices(h, s, sl.ctypes.data, 4, C.byref(n))
            lib.covahip_h264_close(h)
    assert opened > 0
