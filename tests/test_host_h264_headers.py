"""Picture order counts from hand-built headers: the cases no stream in this image exercises.  The access units below carry
a parameter-set pair and slice HEADERS only (the macroblock layer is a few stuffing bytes, so covahip_h264_decode_au reports
bad slice data -- after it has parsed the header and stepped the picture order count, which is what is checked)."""
import ctypes as C

import numpy as np

from cova_amd import _lib as L


class Bits:
    def __init__(self):
        self.b = []

    def u(self, n, v):
        self.b += [(v >> (n - 1 - i)) & 1 for i in range(n)]
        return self

    def ue(self, v):
        n = (v + 1).bit_length()
        return self.u(n - 1, 0).u(n, v + 1)

    def se(self, v):
        return self.ue(2 * v - 1 if v > 0 else -2 * v)

    def rbsp(self):
        bits = self.b + [1]
        bits += [0] * (-len(bits) % 8)
        return bytes(int("".join(map(str, bits[i:i + 8])), 2) for i in range(0, len(bits), 8))


def nal(header, payload):
    out = bytearray([header])
    zeros = 0
    for x in payload:                       # emulation prevention (7.4.1)
        if zeros >= 2 and x <= 3:
            out.append(3)
            zeros = 0
        out.append(x)
        zeros = zeros + 1 if x == 0 else 0
    return bytes(out)


def parameter_sets(poc_type):
    sps = Bits().u(8, 77).u(8, 0).u(8, 31).ue(0).ue(0).ue(poc_type)       # Main@3.1, log2_max_frame_num 4
    if poc_type == 0:
        sps.ue(0)                                                       # log2_max_pic_order_cnt_lsb 4
    sps.ue(4).u(1, 0).ue(7).ue(5).u(1, 1).u(1, 1).u(1, 0).u(1, 0)        # 4 reference frames, 8x6 macroblocks, frames only
    pps = Bits().ue(0).ue(0).u(1, 1).u(1, 0).ue(0).ue(0).ue(0).u(1, 0).u(2, 0).se(0).se(0).se(0).u(1, 0).u(1, 0).u(1, 0)
    s, p = nal(0x67, sps.rbsp()), nal(0x68, pps.rbsp())
    return bytes([1, 77, 0, 31, 0xFF, 0xE1]) + len(s).to_bytes(2, "big") + s + bytes([1]) + len(p).to_bytes(2, "big") + p


def access_unit(poc_type, idr, ref, frame_num, poc_lsb=0, mmco5=False):
    h = Bits().ue(0).ue(7 if idr else 5).ue(0).u(4, frame_num)         # first_mb 0; slice type I / P ("all slices" forms)
    if idr:
        h.ue(0)
    if poc_type == 0:
        h.u(4, poc_lsb)
    if not idr:
        h.u(1, 0).u(1, 0)                                               # no reference count override, no list modification
    if ref:
        if idr:
            h.u(1, 0).u(1, 0)
        elif mmco5:
            h.u(1, 1).ue(5).ue(0)                                       # adaptive marking: operation 5, end
        else:
            h.u(1, 0)
    if not idr:
        h.ue(0)                                                         # cabac_init_idc
    h.se(0)                                                             # slice_qp_delta
    body = nal((0x60 if ref else 0) | (5 if idr else 1), h.rbsp() + b"\x55" * 8)
    return len(body).to_bytes(4, "big") + body


def order_keys(poc_type, units):
    lib = L.lib()
    avcc = np.frombuffer(parameter_sets(poc_type), dtype=np.uint8).copy()
    h = C.c_void_p()
    assert lib.covahip_h264_open_avcc(avcc.ctypes.data, avcc.size, C.byref(h)) == 0
    keys, hdrs = [], []
    try:
        for u in units:
            au = np.frombuffer(access_unit(poc_type, **u), dtype=np.uint8).copy()
            key = C.c_int64(-1)
            hdr = np.zeros(1, dtype=L.H264_SLICE_DTYPE)
            rc = lib.covahip_h264_decode_au(h, au.ctypes.data, au.size, None, 0, hdr.ctypes.data, C.byref(key))
            assert rc in (0, 8), rc        # OK or COVAHIP_ERR_BAD_DATA (the stuffing bytes), never "unsupported" / "invalid"
            keys.append(key.value)
            hdrs.append(hdr[0].copy())
    finally:
        lib.covahip_h264_close(h)
    return keys, hdrs


def test_headers_parse():
    keys, hdrs = order_keys(0, [dict(idr=True, ref=True, frame_num=0, poc_lsb=0),
                                dict(idr=False, ref=True, frame_num=1, poc_lsb=6, mmco5=True)])
    assert [int(h["idr"]) for h in hdrs] == [1, 0]
    assert [int(h["frame_num"]) for h in hdrs] == [0, 1]
    assert [int(h["poc_lsb"]) for h in hdrs] == [0, 6]
    assert [int(h["has_mmco5"]) for h in hdrs] == [0, 1]


def test_memory_management_operation_5_restarts_the_count_type_0():
    """8.2.1: after a reference picture with memory_management_control_operation 5 the next picture's prevPicOrderCntMsb is 0
    and prevPicOrderCntLsb the picture's count after tempPicOrderCnt was taken off (0 for a frame); C.4.4: everything decoded
    before that picture leaves first.  So the picture itself opens a new output period at count 0, and lsb 2 behind it means
    count 2 -- not "8 -> 2 without a wrap", which would put it before pictures that were already output."""
    keys, _ = order_keys(0, [dict(idr=True, ref=True, frame_num=0, poc_lsb=0),
                             dict(idr=False, ref=True, frame_num=1, poc_lsb=4),
                             dict(idr=False, ref=True, frame_num=2, poc_lsb=8, mmco5=True),
                             dict(idr=False, ref=True, frame_num=1, poc_lsb=2),
                             dict(idr=False, ref=False, frame_num=2, poc_lsb=14),      # 14 is nearer to 2 going down: count -2
                             dict(idr=False, ref=True, frame_num=2, poc_lsb=4),
                             dict(idr=True, ref=True, frame_num=0, poc_lsb=0)])
    base = 1 << 31
    period = [k >> 32 for k in keys]
    count = [(k & 0xFFFFFFFF) - base for k in keys]
    assert period == [0, 0, 1, 1, 1, 1, 2]
    assert count == [0, 4, 0, 2, -2, 4, 0]
    assert sorted(range(7), key=lambda i: keys[i]) == [0, 1, 4, 2, 3, 5, 6]


def test_memory_management_operation_5_restarts_the_count_type_2():
    """8.2.1.3: prevFrameNumOffset is 0 behind such a picture and its frame_num counts as 0 (7.4.3), so frame_num 1 behind it
    is the second picture of a new period, not a wrap of the old one."""
    keys, _ = order_keys(2, [dict(idr=True, ref=True, frame_num=0),
                             dict(idr=False, ref=True, frame_num=1),
                             dict(idr=False, ref=True, frame_num=2, mmco5=True),
                             dict(idr=False, ref=True, frame_num=1),
                             dict(idr=False, ref=False, frame_num=2)])
    base = 1 << 31
    assert [k >> 32 for k in keys] == [0, 0, 1, 1, 1]
    assert [(k & 0xFFFFFFFF) - base for k in keys] == [0, 2, 0, 2, 3]
    assert keys == sorted(keys)
