"""Entropy-decode front end on the reference's own demo video: container, headers, picture
order and the CABAC macroblock layer.  No reference output exists for this stage (the patched FFmpeg is an un-vendored
submodule), so the checks are structural -- and strong: what the container and the H.264 headers of demo/1m.mp4 must say
(README.md:94-114: 1280x720, 1,802 frames, GoP 250, High@3.1); every one of its 1,802 slices must entropy-decode to exactly
3,600 macroblocks with end_of_slice_flag at the last one (a wrong CABAC table value or context rule cannot survive that);
the output order computed from picture order counts must be the order of the container's composition times."""
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
    # VUI bitstream_restriction: what x264 writes for 3 B frames with a B pyramid and 5 reference frames
    assert (i["max_num_reorder_frames"], i["max_dec_frame_buffering"]) == (2, 5)


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


def _decode_all(demo):
    lib, h, _ = demo
    sl = np.zeros(4, dtype=L.H264_SLICE_DTYPE)
    n = C.c_int()
    recs = np.zeros((1802, 45, 80, 4), np.uint8)
    types = np.zeros(1802, np.int32)
    for s in range(1802):
        assert lib.covahip_h264_sample_slices(h, s, sl.ctypes.data, 4, C.byref(n)) == 0
        types[s] = sl[0]["slice_type"]
        assert lib.covahip_h264_decode_records(h, s, recs[s].ctypes.data, recs[s].nbytes) == 0, f"access unit {s} (slice type {types[s]})"
    return recs, types


@pytest.fixture(scope="module")
def decoded(demo):
    return _decode_all(demo)


def test_every_slice_entropy_decodes_to_its_last_macroblock(decoded):
    """covahip_h264_decode_records returns OK only when the slice decoded exactly 3,600 macroblocks, end_of_slice_flag came with
    the last one and only trailing bits followed: 8 I + 564 P + 1,230 B slices, 6.49 M macroblocks, 4 MB of CABAC data."""
    recs, types = decoded
    assert (np.bincount(types, minlength=3) == [564, 1230, 8]).all()
    cls = recs[..., 0]
    assert cls.max() <= 7 and not recs[..., 3].any()
    # I pictures: intra classes only, and both kinds occur
    icls = cls[types == 2]
    assert set(np.unique(icls)) <= {5, 6, 7} and (icls == 5).any() and (icls == 6).any()
    # P pictures: mostly skipped or predicted, some intra; B pictures: overwhelmingly skipped / direct (307 bytes per picture on average)
    pcls, bcls = cls[types == 0], cls[types == 1]
    assert (pcls == 0).mean() > 0.5 and 0.01 < np.isin(pcls, (1, 2, 3)).mean() < 0.5 and np.isin(pcls, (5, 6)).mean() < 0.1
    assert not (pcls == 4).any()                                    # B_Direct_16x16 cannot occur in a P slice
    assert np.isin(bcls, (0, 4)).mean() > 0.9 and np.isin(bcls, (1, 2, 3)).any()
    # motion bytes: |mean motion vector| of the macroblock in quarter pixels (prediction + coded difference).  Intra macroblocks have
    # none; a few per cent of the macroblocks of this traffic scene move by a pixel or more, and they form blobs, not salt and pepper:
    # most moving macroblocks have a 4-neighbour that moves alike (a wrong median / skip / direct prediction would scatter them)
    mv = recs[..., 1:3]
    assert not mv[cls >= 5].any()
    for t, lo in ((0, 0.4), (1, 0.6)):
        mag = mv[types == t].astype(int).max(axis=-1)
        mov = mag >= 4
        assert 0.002 < mov.mean() < 0.2      # 2.6 % of the P pictures' macroblocks, 0.6 % of the B pictures' (one to three frames between references)
        agree = np.zeros_like(mov)
        for dy, dx in ((0, 1), (0, -1), (1, 0), (-1, 0)):
            agree |= np.roll(mov, (dy, dx), axis=(1, 2)) & (np.abs(np.roll(mag, (dy, dx), axis=(1, 2)) - mag) <= 2)
        assert (agree & mov).sum() / mov.sum() > lo
    pskip = (cls == 0) & (types == 0)[:, None, None]
    assert 0 < (mv.max(axis=-1)[pskip] > 0).mean() < 0.05          # P_Skip inherits its neighbours' motion, rarely a moving one


def test_output_order_from_picture_order_counts_is_the_containers_composition_order(demo):
    """covahip_h264_display_order (POC type 0, 8.2.1.1, from the slice headers) against an independent source in the same
    file: decode time (stts) + composition offset (ctts) of every sample."""
    import struct
    lib, h, data = demo
    order = np.zeros(1802, np.int32)
    n = C.c_int()
    assert lib.covahip_h264_display_order(h, order.ctypes.data, 1802, C.byref(n)) == 0 and n.value == 1802
    assert sorted(order.tolist()) == list(range(1802))
    raw = data.tobytes()

    def table(tag, entry):
        at = raw.rfind(tag)
        cnt = struct.unpack(">I", raw[at + 8:at + 12])[0]
        return [struct.unpack(entry, raw[at + 12 + struct.calcsize(entry) * i:at + 12 + struct.calcsize(entry) * (i + 1)]) for i in range(cnt)]
    dts, t = [], 0
    for cnt, delta in table(b"stts", ">II"):
        for _ in range(cnt):
            dts.append(t)
            t += delta
    off = [o for cnt, o in table(b"ctts", ">Ii") for _ in range(cnt)]
    assert len(dts) == len(off) == 1802
    cts = np.array(dts) + np.array(off)
    assert len(set(cts.tolist())) == 1802
    np.testing.assert_array_equal(order, np.argsort(cts, kind="stable"))
    assert (order[:1] == [0]).all() and (np.abs(order - np.arange(1802)) <= 8).all()      # reordering stays within a few frames
    assert not (order == np.arange(1802)).all()                                          # and there is some (B pictures)


def test_no_picture_is_overtaken_by_more_than_max_num_reorder_frames(demo):
    """E.2.1: max_num_reorder_frames bounds how many pictures can precede a picture in decoding order and follow it in output
    order -- the depth of h264entropydec's queue.  The order computed from the picture order counts must respect what the
    stream's own VUI promises (2), and reach it."""
    lib, h, _ = demo
    order = np.zeros(1802, dtype=np.int32)
    n = C.c_int()
    assert lib.covahip_h264_display_order(h, order.ctypes.data, order.size, C.byref(n)) == 0 and n.value == 1802
    pos = np.empty(1802, dtype=np.int64)
    pos[order] = np.arange(1802)                      # output position of the access unit decoded i-th
    worst = 0
    for i in range(1802):
        worst = max(worst, int((pos[max(0, i - 16):i] > pos[i]).sum()))
    assert worst == 2


def test_colocated_picture_of_every_b_picture_is_the_nearest_following_reference(demo):
    """RefPicList1[0] (the picture the colZeroFlag test of direct prediction looks at) from the marking process + list
    initialisation + modification commands of the slice headers: in this stream (x264, b-pyramid) it must be the reference
    picture that follows the B picture most closely in output order among those decoded before it -- a P picture for most, a
    reference B picture for the rest -- and always a short-term one."""
    lib, h, _ = demo
    order = np.zeros(1802, np.int32)
    n = C.c_int()
    assert lib.covahip_h264_display_order(h, order.ctypes.data, 1802, C.byref(n)) == 0
    rank = np.zeros(1802, int)
    rank[order] = np.arange(1802)
    sl = np.zeros(4, dtype=L.H264_SLICE_DTYPE)
    types, refs = [], []
    for s in range(1802):
        assert lib.covahip_h264_sample_slices(h, s, sl.ctypes.data, 4, C.byref(n)) == 0
        types.append(int(sl[0]["slice_type"]))
        refs.append(int(sl[0]["nal_ref_idc"]) != 0)
    col, short = C.c_int(), C.c_int()
    kinds = {0: 0, 1: 0}
    for s in range(1802):
        assert lib.covahip_h264_colocated(h, s, C.byref(col), C.byref(short)) == 0
        if types[s] != 1:
            assert col.value == -1
            continue
        later = [r for r in range(max(0, s - 40), s) if refs[r] and rank[r] > rank[s]]
        assert later and col.value == min(later, key=lambda r: rank[r]) and short.value == 1, f"access unit {s}"
        kinds[types[col.value]] += 1
    assert kinds[0] > kinds[1] > 100
    assert lib.covahip_h264_colocated(h, 1802, C.byref(col), C.byref(short)) != 0


def test_temporal_direct_pictures_move_half_as_far_as_the_p_picture_behind_them(demo, decoded):
    """The first GoP of the stream codes its reference B pictures with temporal direct prediction (direct_spatial 0): their
    vectors are the co-located P picture's, scaled by the ratio of the picture order count distances (8.4.1.2.3).  Such a B
    picture sits halfway between two P pictures, so its moving macroblocks are about as many as the following P picture's and
    move half as far -- which only comes out if the co-located picture, the list-0 mapping and DistScaleFactor are right."""
    lib, h, _ = demo
    recs, types = decoded
    order = np.zeros(1802, np.int32)
    n = C.c_int()
    assert lib.covahip_h264_display_order(h, order.ctypes.data, 1802, C.byref(n)) == 0
    sl = np.zeros(4, dtype=L.H264_SLICE_DTYPE)
    checked = 0
    for k in range(1, 24):
        s = int(order[k])
        assert lib.covahip_h264_sample_slices(h, s, sl.ctypes.data, 4, C.byref(n)) == 0
        if not (sl[0]["slice_type"] == 1 and sl[0]["direct_spatial"] == 0 and sl[0]["nal_ref_idc"] != 0):
            continue
        nxt = next(int(order[j]) for j in range(k + 1, k + 6) if types[int(order[j])] == 0)      # the P picture behind it
        mv_b, mv_p = recs[s][..., 1:3].astype(int), recs[nxt][..., 1:3].astype(int)
        mov_b, mov_p = mv_b.max(axis=-1) >= 4, mv_p.max(axis=-1) >= 4
        if mov_p.sum() < 50 or mov_b.sum() < 30:     # (the stream repeats frames: the B picture right behind the key frame shows the key frame again)
            continue
        assert 0.6 < mov_b.sum() / mov_p.sum() < 1.3
        ratio = mv_b[mov_b].mean(axis=0) / mv_p[mov_p].mean(axis=0)
        assert (0.35 < ratio).all() and (ratio < 0.75).all(), (k, ratio)
        checked += 1
    assert checked >= 4


def test_stream_form_in_decode_order_gives_the_records_of_the_file_form(demo, decoded):
    """covahip_h264_decode_au (parameter sets from the avcC box, access units in decode order: what the h264entropydec element
    calls) keeps its own reference marking and co-located motion; covahip_h264_decode_records (random access) finds them
    through the table made when the file was opened.  Same records, access unit by access unit, across two key frames."""
    lib, h, data = demo
    recs, _ = decoded
    raw = data.tobytes()
    at = raw.find(b"avcC")
    size = int.from_bytes(raw[at - 4:at], "big")
    avcc = np.frombuffer(raw[at + 4:at - 4 + size], dtype=np.uint8).copy()
    hs = C.c_void_p()
    assert lib.covahip_h264_open_avcc(avcc.ctypes.data, avcc.size, C.byref(hs)) == 0
    try:
        off, sz, sync = C.c_uint64(), C.c_uint32(), C.c_int()
        rec = np.zeros((45, 80, 4), np.uint8)
        key = C.c_int64()
        for s in range(520):
            assert lib.covahip_h264_sample(h, s, C.byref(off), C.byref(sz), C.byref(sync)) == 0
            au = data[off.value:off.value + sz.value]
            assert lib.covahip_h264_decode_au(hs, au.ctypes.data, au.size, rec.ctypes.data, rec.nbytes, None, C.byref(key)) == 0
            assert rec.tobytes() == recs[s].tobytes(), f"access unit {s}"
    finally:
        lib.covahip_h264_close(hs)


def test_config_1_demo_video_through_metapreprocess_into_tfrecords(demo, decoded):
    """BASELINE config 1: demo/1m.mp4 -> entropy decode -> metapreprocess -> tfrecordsink, here through the C-ABI objects behind
    those elements (cova_amd.elements): carrier frames in OUTPUT order, timestep 4, one tf.train.Example per emitted frame whose
    three features are bytes 0 / 1 / 2 of the CURRENT frame's records (tfrecordsink/imp.rs:105-112 splits the first w*h pixels)."""
    from cova_amd import elements as E
    from tests.test_host_formats import _parse_example
    lib, h, _ = demo
    recs, _ = decoded
    order = np.zeros(1802, np.int32)
    n = C.c_int()
    assert lib.covahip_h264_display_order(h, order.ctypes.data, 1802, C.byref(n)) == 0
    w_mb, h_mb, t = 80, 45, 4
    mp = E.MetaPreprocess(timestep=t, gamma=1)
    assert mp.set_caps(1280, 720) == (w_mb, h_mb * t)
    frame = np.zeros(1280 * 720 * 3 // 2, np.uint8)             # the I420 buffer avdec_h264 would push: records in its first bytes
    emitted = 0
    for k in range(40):
        frame[:w_mb * h_mb * 4] = recs[order[k]].reshape(-1)
        flow, out = mp.transform(frame)
        if k < t - 1:
            assert flow != E.FLOW_OK
            continue
        assert flow == E.FLOW_OK
        stack = out.reshape(t * h_mb, w_mb, 4)
        for j in range(t):                                     # row block j = the frame j steps back in OUTPUT order
            np.testing.assert_array_equal(stack[j * h_mb:(j + 1) * h_mb], recs[order[k - j]])
        ex = _parse_example(E.tfrecord_example(stack[:h_mb].reshape(1, h_mb, w_mb, 4))[12:-4])
        for ch, name in enumerate(("mb_type", "mv_x", "mv_y")):
            assert ex[name] == [recs[order[k]][..., ch].tobytes()]
        emitted += 1
    assert emitted == 40 - (t - 1)


def test_corrupted_slice_data_never_crashes(demo):
    """The macroblock layer reads untrusted bytes: flipping bytes inside an access unit must end in an error status (or in
    records, when the damage happens to decode), never in a fault or a hang."""
    lib, _, data = demo
    rng = np.random.default_rng(7)
    rec = np.zeros(45 * 80 * 4, np.uint8)
    off, size, sync = C.c_uint64(), C.c_uint32(), C.c_int()
    outcomes = {0: 0, "err": 0}
    for trial in range(120):
        d = data.copy()
        hh = C.c_void_p()
        assert lib.covahip_h264_open_mp4(d.ctypes.data, d.size, C.byref(hh)) == 0
        s = int(rng.choice([0, 1, 2, 3, 5, 250, 251, 260, 1801]))
        lib.covahip_h264_sample(hh, s, C.byref(off), C.byref(size), C.byref(sync))
        for _ in range(int(rng.integers(1, 6))):
            d[off.value + 8 + int(rng.integers(0, max(1, size.value - 8)))] = rng.integers(0, 256)
        rc = lib.covahip_h264_decode_records(hh, s, rec.ctypes.data, rec.size)
        outcomes[0 if rc == 0 else "err"] += 1
        # a buffer that is too small is refused before anything is written
        assert lib.covahip_h264_decode_records(hh, s, rec.ctypes.data, 100) == 7
        lib.covahip_h264_close(hh)
    assert outcomes["err"] > 60      # almost every hit desynchronises the arithmetic decoder, and that is noticed


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


# A second, unrelated stream that ships with this image (imageio's test clip: 320x240, High@4.0, picture order count type 2, one
# reference frame, I and P pictures, another encoder configuration than the demo video's): nothing in the front end is fitted to one file.
SECOND = "/opt/conda/lib/python3.9/site-packages/imageio/resources/images/realshort.mp4"


@pytest.mark.skipif(not os.path.exists(SECOND), reason="imageio's test clip is not in this image")
def test_a_second_stream_decodes_to_the_last_macroblock_of_every_slice():
    lib = L.lib()
    data = np.fromfile(SECOND, dtype=np.uint8)
    h = C.c_void_p()
    assert lib.covahip_h264_open_mp4(data.ctypes.data, data.size, C.byref(h)) == 0
    try:
        info = np.zeros(1, dtype=L.H264_INFO_DTYPE)
        assert lib.covahip_h264_get_info(h, info.ctypes.data) == 0
        n_s, wmb, hmb = int(info[0]["n_samples"]), int(info[0]["width_mbs"]), int(info[0]["height_mbs"])
        assert (wmb, hmb) == (20, 15) and n_s == 36 and info[0]["entropy_cabac"] == 1 and info[0]["poc_type"] == 2
        sl = np.zeros(4, dtype=L.H264_SLICE_DTYPE)
        n = C.c_int()
        rec = np.zeros((hmb, wmb, 4), np.uint8)
        types = []
        moving = 0
        for s in range(n_s):
            assert lib.covahip_h264_sample_slices(h, s, sl.ctypes.data, 4, C.byref(n)) == 0 and n.value == 1
            types.append(int(sl[0]["slice_type"]))
            assert lib.covahip_h264_decode_records(h, s, rec.ctypes.data, rec.nbytes) == 0, f"access unit {s}"
            assert rec[..., 0].max() <= 7 and not rec[..., 3].any()
            if types[-1] == 2:
                assert set(np.unique(rec[..., 0])) <= {5, 6, 7} and not rec[..., 1:3].any()
            else:
                moving += int((rec[..., 1:3].max(axis=-1) > 0).sum())
        assert types.count(2) == 2 and types.count(0) == 34 and moving > 0
        # picture order count type 2: output order = decode order
        order = np.zeros(n_s, np.int32)
        assert lib.covahip_h264_display_order(h, order.ctypes.data, n_s, C.byref(n)) == 0 and (order == np.arange(n_s)).all()
    finally:
        lib.covahip_h264_close(h)

