// Entropy-decode front end: container, headers, picture order; the macroblock layer is
// h264_cabac.cpp.
//
// The reference feeds its filter from a patched FFmpeg `avdec_h264` that stops after entropy decoding and writes one
// 4-byte record [mb_type, mv_x, mv_y, -] per macroblock into the first bytes of its I420 output frame (README.md:94-114,
// pipeline/cova/pipeline.py:84-99; consumers: metapreprocess/imp.rs:233,311-312, tfrecordsink/imp.rs:105-112).  That
// decoder is an un-vendored submodule, so what its record bytes mean exactly is not known here; the
// layout is.  Built, and verified against the reference's own demo/1m.mp4 (tests/test_host_h264.py):
//   * ISO-BMFF demux of the video track (avcC, stsz / stsc / stco / co64 / stss): access units with their key-frame flag,
//   * NAL unit split (length prefixed), emulation-prevention removal,
//   * SPS / PPS parse, slice header parse up to the first bit of slice_data() (ITU-T H.264 7.3.2.1, 7.3.2.2, 7.3.3),
//   * picture order counts (8.2.1, types 0 and 2) and the output (display) order of the access units -- avdec_h264 hands
//     frames downstream in that order, and metapreprocess stacks consecutive OUTPUT frames,
//   * covahip_h264_decode_records: CABAC macroblock layer of a whole-picture slice -> records (h264_cabac.h says what the
//     three bytes are: macroblock class and the |mean motion vector| as the standard reconstructs it),
//   * what direct prediction needs of the reference pictures: reference marking (8.2.5), RefPicList1[0] (8.2.4.2.3, 8.2.4.3)
//     and the co-located picture's "does not move" bits for colZeroFlag (8.4.1.2.2) -- no pixels,
//   * the carrier layout writer (covahip_carrier_write_records).
// Refused loudly (COVAHIP_ERR_UNSUPPORTED), never faked: CAVLC, field / MBAFF coding, several slices per picture, FMO,
// scaling matrices, chroma formats other than 4:2:0, cabac_init_idc 1 / 2.
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <new>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

#include "covahip.h"
#include "h264_cabac.h"

namespace {

struct BitReader {
    const uint8_t *p;
    size_t n, pos = 0;   // bit position
    bool bad = false;
    BitReader(const uint8_t *d, size_t len) : p(d), n(len * 8) {}
    uint32_t u(int bits) {
        uint32_t v = 0;
        for (int i = 0; i < bits; i++) {
            if (pos >= n) { bad = true; return 0; }
            v = (v << 1) | ((p[pos >> 3] >> (7 - (pos & 7))) & 1);
            pos++;
        }
        return v;
    }
    uint32_t ue() {
        int z = 0;
        while (!bad && u(1) == 0 && z < 32) z++;
        if (z >= 32) { bad = true; return 0; }
        return z ? ((1u << z) - 1 + u(z)) : 0;
    }
    int32_t se() {
        const uint32_t k = ue();
        return (k & 1) ? (int32_t)((k + 1) / 2) : -(int32_t)(k / 2);
    }
};

std::vector<uint8_t> unescape(const uint8_t *d, size_t n) {   // 7.4.1: drop emulation_prevention_three_byte
    std::vector<uint8_t> o;
    o.reserve(n);
    for (size_t i = 0; i < n; i++) {
        if (i + 2 < n && d[i] == 0 && d[i + 1] == 0 && d[i + 2] == 3) {
            o.push_back(0); o.push_back(0);
            i += 2;
        } else {
            o.push_back(d[i]);
        }
    }
    return o;
}

struct Sps {
    int id = 0;
    int profile = 0, level = 0, chroma_format = 1, log2_max_frame_num = 4, poc_type = 0, log2_max_poc_lsb = 4;
    int delta_pic_order_always_zero = 0, num_ref_frames = 0, width_mbs = 0, height_map_units = 0, frame_mbs_only = 1, mbaff = 0;
    int direct_8x8 = 0, bit_depth = 8;
    int max_num_reorder = -1, max_dec_frame_buffering = -1;   // VUI bitstream_restriction (E.1.1); -1: the stream does not say
    bool ok = false;
};
struct Pps {
    int id = 0, sps_id = 0;
    int entropy_cabac = 0, bottom_field_pic_order = 0, num_slice_groups = 1, num_ref_l0 = 1, num_ref_l1 = 1, weighted_pred = 0,
        weighted_bipred = 0, pic_init_qp = 26, deblocking_control = 0, redundant_pic_cnt = 0, transform_8x8 = 0;
    bool ok = false;
};

bool parse_sps(const std::vector<uint8_t> &rbsp, Sps &s) {
    if (rbsp.size() < 5) return false;
    BitReader r(rbsp.data() + 1, rbsp.size() - 1);
    s.profile = (int)r.u(8); r.u(8); s.level = (int)r.u(8);
    s.id = (int)r.ue();
    if (s.profile == 100 || s.profile == 110 || s.profile == 122 || s.profile == 244 || s.profile == 44 || s.profile == 83 ||
        s.profile == 86 || s.profile == 118 || s.profile == 128) {
        s.chroma_format = (int)r.ue();
        if (s.chroma_format == 3) r.u(1);
        s.bit_depth = 8 + (int)std::max(r.ue(), r.ue());   // luma, chroma: samples of more than 8 bits are refused at open (I_PCM sizes)
        r.u(1);
        if (r.u(1)) return false;   // seq_scaling_matrix_present_flag: not handled
    }
    s.log2_max_frame_num = (int)r.ue() + 4;
    s.poc_type = (int)r.ue();
    if (s.poc_type == 0) s.log2_max_poc_lsb = (int)r.ue() + 4;
    else if (s.poc_type == 1) {
        s.delta_pic_order_always_zero = (int)r.u(1);
        r.se(); r.se();
        const uint32_t n = r.ue();
        for (uint32_t i = 0; i < n && !r.bad; i++) r.se();
    }
    s.num_ref_frames = (int)r.ue();
    r.u(1);
    s.width_mbs = (int)r.ue() + 1;
    s.height_map_units = (int)r.ue() + 1;
    s.frame_mbs_only = (int)r.u(1);
    if (!s.frame_mbs_only) s.mbaff = (int)r.u(1);
    s.direct_8x8 = (int)r.u(1);
    s.ok = !r.bad && s.log2_max_frame_num <= 16 && s.log2_max_poc_lsb <= 16 && s.width_mbs <= 1024 && s.height_map_units <= 1024 &&
           s.chroma_format <= 3 && s.num_ref_frames <= 16;
    if (!s.ok) return false;
    // frame cropping, then the VUI as far as bitstream_restriction (E.1.1): max_num_reorder_frames is how many pictures a decoder
    // has to hold back before the first one may leave in output order (h264entropydec's queue).  A VUI that does not parse
    // leaves the two fields unknown; the parameters above stand.
    if (r.u(1)) { r.ue(); r.ue(); r.ue(); r.ue(); }
    if (r.u(1) && !r.bad) {
        auto hrd = [&]() {
            const uint32_t cnt = r.ue() + 1;
            r.u(8);
            for (uint32_t i = 0; i < cnt && i < 32 && !r.bad; i++) { r.ue(); r.ue(); r.u(1); }
            r.u(20);
        };
        if (r.u(1)) { if (r.u(8) == 255) { r.u(16); r.u(16); } }   // aspect ratio
        if (r.u(1)) r.u(1);                                          // overscan
        if (r.u(1)) { r.u(4); if (r.u(1)) r.u(24); }                 // video signal type (+ colour description)
        if (r.u(1)) { r.ue(); r.ue(); }                              // chroma sample location
        if (r.u(1)) { r.u(32); r.u(32); r.u(1); }                    // timing
        const uint32_t nal_hrd = r.u(1);
        if (nal_hrd) hrd();
        const uint32_t vcl_hrd = r.u(1);
        if (vcl_hrd) hrd();
        if (nal_hrd || vcl_hrd) r.u(1);
        r.u(1);                                                      // pic_struct_present_flag
        if (r.u(1)) {
            r.u(1); r.ue(); r.ue(); r.ue(); r.ue();
            const uint32_t reorder = r.ue(), buffering = r.ue();
            if (!r.bad && reorder <= 16 && buffering <= 16 && reorder <= buffering) {
                s.max_num_reorder = (int)reorder;
                s.max_dec_frame_buffering = (int)buffering;
            }
        }
    }
    return true;
}

bool parse_pps(const std::vector<uint8_t> &rbsp, Pps &p) {
    if (rbsp.size() < 3) return false;
    BitReader r(rbsp.data() + 1, rbsp.size() - 1);
    p.id = (int)r.ue();
    p.sps_id = (int)r.ue();
    p.entropy_cabac = (int)r.u(1);
    p.bottom_field_pic_order = (int)r.u(1);
    p.num_slice_groups = (int)r.ue() + 1;
    if (p.num_slice_groups > 1) return false;   // FMO: not handled
    p.num_ref_l0 = (int)r.ue() + 1;
    p.num_ref_l1 = (int)r.ue() + 1;
    p.weighted_pred = (int)r.u(1);
    p.weighted_bipred = (int)r.u(2);
    p.pic_init_qp = (int)r.se() + 26;
    r.se(); r.se();
    p.deblocking_control = (int)r.u(1);
    r.u(1);
    p.redundant_pic_cnt = (int)r.u(1);
    // more_rbsp_data(): something besides the trailing bits is left
    size_t last = rbsp.size() - 1;
    while (last > 1 && rbsp[last] == 0) last--;
    int tz = 0;
    while (tz < 7 && !((rbsp[last] >> tz) & 1)) tz++;
    const size_t end_bits = (last - 1) * 8 + (7 - tz);   // bits of payload before rbsp_stop_one_bit, relative to byte 1
    if (r.pos < end_bits) p.transform_8x8 = (int)r.u(1);
    p.ok = !r.bad && p.num_ref_l0 <= 32 && p.num_ref_l1 <= 32;
    return p.ok;
}

struct Sample { uint64_t off; uint32_t size; bool sync; };

// What the reference picture lists and the marking process need from a slice header beyond covahip_h264_slice (kept out of the
// public struct): the list-1 modification commands (7.3.3.1) and dec_ref_pic_marking (7.3.3.3)
struct RplOp { int idc; uint32_t val; };
struct Mmco { int op; uint32_t a, b; };
struct SliceExt {
    std::vector<RplOp> l0, l1;
    std::vector<Mmco> mmco;
    bool adaptive = false, long_term_reference = false;
};

// The reference pictures (frames only) as far as RefPicList1[0] needs them: decoded reference picture marking (8.2.5) and the
// initialisation + modification of list 1 (8.2.4.2.3, 8.2.4.3).  `id` names a picture for the caller (sample index / running count).
struct RefPic { int id; int frame_num; int64_t key; bool lt; int lt_idx; };
struct Dpb {
    std::vector<RefPic> refs;
    // RefPicList0 / RefPicList1 of the P or B picture described by (sl, ext, key), before it is decoded: indices into `refs`,
    // -1 where the list has no picture (8.2.4.2.1 / 8.2.4.2.3 initialisation, 8.2.4.3 modification; frames only)
    void build_lists(const Sps &sp, const covahip_h264_slice &sl, const SliceExt &ext, int64_t key, std::vector<int> &l0,
                     std::vector<int> &l1) const {
        const int max_fn = 1 << sp.log2_max_frame_num;
        auto pic_num = [&](const RefPic &r) { return r.frame_num > sl.frame_num ? r.frame_num - max_fn : r.frame_num; };
        std::vector<int> lt;
        for (int i = 0; i < (int)refs.size(); i++)
            if (refs[i].lt) lt.push_back(i);
        std::sort(lt.begin(), lt.end(), [&](int a, int b) { return refs[a].lt_idx < refs[b].lt_idx; });
        l0.clear();
        l1.clear();
        if (sl.slice_type == 1) {
            std::vector<int> before, after;
            for (int i = 0; i < (int)refs.size(); i++)
                if (!refs[i].lt) (refs[i].key < key ? before : after).push_back(i);
            std::sort(before.begin(), before.end(), [&](int a, int b) { return refs[a].key > refs[b].key; });
            std::sort(after.begin(), after.end(), [&](int a, int b) { return refs[a].key < refs[b].key; });
            l0 = before;
            l1 = after;
            l0.insert(l0.end(), after.begin(), after.end());
            l1.insert(l1.end(), before.begin(), before.end());
            l0.insert(l0.end(), lt.begin(), lt.end());
            l1.insert(l1.end(), lt.begin(), lt.end());
            if (l1.size() > 1 && l1 == l0) std::swap(l1[0], l1[1]);
        } else {
            for (int i = 0; i < (int)refs.size(); i++)
                if (!refs[i].lt) l0.push_back(i);
            std::sort(l0.begin(), l0.end(), [&](int a, int b) { return pic_num(refs[a]) > pic_num(refs[b]); });
            l0.insert(l0.end(), lt.begin(), lt.end());
        }
        auto modify = [&](std::vector<int> &l, const std::vector<RplOp> &ops, int n_active) {
            const size_t n = (size_t)std::max(1, n_active);
            l.resize(n, -1);
            size_t at = 0;
            int pred = sl.frame_num;   // CurrPicNum (frames)
            for (const RplOp &op : ops) {
                int pick = -1;
                if (op.idc == 0 || op.idc == 1) {
                    const int64_t d = (int64_t)op.val + 1;   // (a damaged header may carry any 32-bit value)
                    int64_t nw = op.idc == 0 ? (int64_t)pred - d : (int64_t)pred + d;
                    nw = ((nw % max_fn) + max_fn) % max_fn;
                    const int nowrap = (int)nw;
                    pred = nowrap;
                    const int pn = nowrap > sl.frame_num ? nowrap - max_fn : nowrap;
                    for (int i = 0; i < (int)refs.size(); i++)
                        if (!refs[i].lt && pic_num(refs[i]) == pn) pick = i;
                } else if (op.idc == 2) {
                    for (int i = 0; i < (int)refs.size(); i++)
                        if (refs[i].lt && refs[i].lt_idx == (int)op.val) pick = i;
                }
                if (pick < 0 || at >= n) continue;   // a command that names no reference picture: ignored
                l.insert(l.begin() + (long)at, pick);
                at++;
                for (size_t k = at; k < l.size(); k++)
                    if (l[k] == pick) { l.erase(l.begin() + (long)k); break; }
                l.resize(n, -1);
            }
        };
        modify(l0, ext.l0, sl.num_ref_l0);
        if (sl.slice_type == 1) modify(l1, ext.l1, sl.num_ref_l1);
    }
    // RefPicList1[0] of a B picture.  false: no such picture.
    bool list1_first(const Sps &sp, const covahip_h264_slice &sl, const SliceExt &ext, int64_t key, int &id, bool &short_term) const {
        std::vector<int> l0, l1;
        build_lists(sp, sl, ext, key, l0, l1);
        if (l1.empty() || l1[0] < 0) return false;
        id = refs[(size_t)l1[0]].id;
        short_term = !refs[(size_t)l1[0]].lt;
        return true;
    }
    // after the picture (sl, ext) with identifier id has been decoded
    void mark(const Sps &sp, const covahip_h264_slice &sl, const SliceExt &ext, int id, int64_t key) {
        if (sl.nal_ref_idc == 0) return;
        const int max_fn = 1 << sp.log2_max_frame_num;
        auto pic_num = [&](const RefPic &r) { return r.frame_num > sl.frame_num ? r.frame_num - max_fn : r.frame_num; };
        auto drop_lt_idx = [&](int idx) {
            refs.erase(std::remove_if(refs.begin(), refs.end(), [&](const RefPic &r) { return r.lt && r.lt_idx == idx; }), refs.end());
        };
        RefPic cur{id, sl.frame_num, key, false, 0};
        if (sl.idr) {
            refs.clear();
            if (ext.long_term_reference) { cur.lt = true; cur.lt_idx = 0; }
        } else if (ext.adaptive) {
            for (const Mmco &m : ext.mmco) {
                if (m.op == 1 || m.op == 3) {
                    const int64_t pn = (int64_t)sl.frame_num - ((int64_t)m.a + 1);
                    for (size_t i = 0; i < refs.size(); i++)
                        if (!refs[i].lt && pic_num(refs[i]) == pn) {
                            if (m.op == 1) { refs.erase(refs.begin() + (long)i); }
                            else {
                                RefPic moved = refs[i];
                                refs.erase(refs.begin() + (long)i);
                                drop_lt_idx((int)m.b);
                                moved.lt = true; moved.lt_idx = (int)m.b;
                                refs.push_back(moved);
                            }
                            break;
                        }
                } else if (m.op == 2) {
                    drop_lt_idx((int)m.a);
                } else if (m.op == 4) {
                    const int keep = (int)m.a - 1;   // MaxLongTermFrameIdx; -1: none
                    refs.erase(std::remove_if(refs.begin(), refs.end(), [&](const RefPic &r) { return r.lt && r.lt_idx > keep; }), refs.end());
                } else if (m.op == 5) {
                    refs.clear();
                    cur.frame_num = 0;
                } else if (m.op == 6) {
                    drop_lt_idx((int)m.a);
                    cur.lt = true; cur.lt_idx = (int)m.a;
                }
            }
        } else {
            // sliding window (8.2.5.3): the short-term picture with the smallest FrameNumWrap goes when the buffer is full
            const size_t cap = (size_t)std::max(1, sp.num_ref_frames);
            while (refs.size() >= cap) {
                int victim = -1;
                for (int i = 0; i < (int)refs.size(); i++)
                    if (!refs[i].lt && (victim < 0 || pic_num(refs[i]) < pic_num(refs[(size_t)victim]))) victim = i;
                if (victim < 0) break;
                refs.erase(refs.begin() + victim);
            }
        }
        refs.push_back(cur);
    }
};

// One entry of a reference picture list as later pictures need it: which picture (caller's id, -1: none), its picture order
// count key, long-term or not
struct RefEntry { int32_t id; int64_t key; bool lt; };
void list_entries(const Dpb &dpb, const std::vector<int> &l, std::vector<RefEntry> &out) {
    out.clear();
    for (int i : l) out.push_back(i < 0 ? RefEntry{-1, 0, false} : RefEntry{dpb.refs[(size_t)i].id, dpb.refs[(size_t)i].key, dpb.refs[(size_t)i].lt});
}
// Temporal direct (8.4.1.2.3): for every entry of the co-located picture's lists the lowest index in the current list 0 that
// names the same picture, and DistScaleFactor of every current list-0 index against RefPicList1[0]
void temporal_tables(const std::vector<RefEntry> &l0, const std::vector<RefEntry> &l1, int64_t key, const std::vector<RefEntry> &col_l0,
                     const std::vector<RefEntry> &col_l1, int8_t col_to_l0[64], int16_t dist_scale[32]) {
    auto clip3 = [](int64_t lo, int64_t hi, int64_t v) { return v < lo ? lo : (v > hi ? hi : v); };
    for (int list = 0; list < 2; list++) {
        const std::vector<RefEntry> &cl = list ? col_l1 : col_l0;
        for (int i = 0; i < 32; i++) {
            int8_t m = -1;
            if (i < (int)cl.size() && cl[(size_t)i].id >= 0)
                for (int j = 0; j < (int)l0.size() && j < 32; j++)
                    if (l0[(size_t)j].id == cl[(size_t)i].id) { m = (int8_t)j; break; }
            col_to_l0[list * 32 + i] = m;
        }
    }
    for (int i = 0; i < 32; i++) {
        dist_scale[i] = h264::DIST_SCALE_NONE;
        if (i >= (int)l0.size() || l0[(size_t)i].id < 0 || l0[(size_t)i].lt || l1.empty() || l1[0].id < 0) continue;
        const int64_t tb = clip3(-128, 127, key - l0[(size_t)i].key), td = clip3(-128, 127, l1[0].key - l0[(size_t)i].key);
        if (td == 0) continue;
        const int64_t tx = (16384 + (td < 0 ? -td : td) / 2) / td;
        dist_scale[i] = (int16_t)clip3(-1024, 1023, (tb * tx + 32) >> 6);
    }
}

uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
uint64_t be64(const uint8_t *p) { return ((uint64_t)be32(p) << 32) | be32(p + 4); }

}  // namespace

struct covahip_h264 {
    const uint8_t *data = nullptr;
    size_t len = 0;
    int nal_len_size = 4;
    Sps sps;
    Pps pps;
    std::vector<Sample> samples;
    std::vector<int64_t> order_key;     // per sample: (IDR period << 32) + picture order count + 2^31; empty when not computable
    std::vector<int32_t> display;       // sample indices in output order
    struct PocState { int64_t prev_msb = 0, prev_lsb = 0, period = -1, fn_off = 0, prev_fn = 0; } poc;   // stream form (covahip_h264_decode_au)
    // What direct prediction needs of a reference picture (h264_cabac.h): the "does not move" bits of its macroblocks (colZeroFlag,
    // 8.4.1.2.2), the motion of their corner blocks and the pictures its own lists named (temporal direct, 8.4.1.2.3)
    struct PicMotion {
        std::vector<uint16_t> still;
        std::vector<h264::ColMb> motion;
        std::vector<RefEntry> l0, l1;
    };
    typedef std::shared_ptr<PicMotion> Still;
    // file form: per sample the sample that is its RefPicList1[0] (-1: none / not B) and whether that is a short-term picture
    // (from a pass over the slice headers at open); bits of decoded reference pictures are kept for the B pictures that follow
    std::vector<int32_t> col_sample;
    std::vector<uint8_t> col_short;
    std::vector<std::vector<RefEntry>> lists0, lists1;   // per sample: its reference picture lists (empty for I pictures)
    mutable std::mutex still_mu;
    mutable std::map<int, Still> still_cache;
    mutable std::vector<int> still_order;   // insertion order, oldest first
    // stream form: the reference pictures so far and their bits, by running picture count
    Dpb dpb;
    int au_count = 0;
    std::map<int, Still> still_live;
};

namespace {

// Finds box `type` directly inside [off, end); returns payload range.
bool find_box(const uint8_t *d, size_t off, size_t end, const char *type, size_t &pb, size_t &pe) {
    while (off + 8 <= end) {
        uint64_t sz = be32(d + off);
        size_t hdr = 8;
        if (sz == 1) { if (off + 16 > end) return false; sz = be64(d + off + 8); hdr = 16; }
        if (sz == 0) sz = end - off;
        if (sz < hdr || sz > end - off) return false;
        if (std::memcmp(d + off + 4, type, 4) == 0) { pb = off + hdr; pe = off + sz; return true; }
        off += sz;
    }
    return false;
}

int parse_slice_header(const covahip_h264 *h, const uint8_t *nal, size_t n, covahip_h264_slice *s, SliceExt *ext = nullptr) {
    if (n < 2) return COVAHIP_ERR_BAD_DATA;
    const int nal_ref_idc = (nal[0] >> 5) & 3, nal_type = nal[0] & 31;
    // the header never needs more than a few dozen bytes; unescape a bounded prefix
    const std::vector<uint8_t> rb = unescape(nal + 1, n - 1 < 96 ? n - 1 : 96);
    BitReader r(rb.data(), rb.size());
    const Sps &sp = h->sps;
    const Pps &pp = h->pps;
    s->nal_type = nal_type;
    s->nal_ref_idc = nal_ref_idc;
    s->has_mmco5 = 0;
    s->first_mb = (int)r.ue();
    const int st = (int)r.ue();
    s->slice_type = st % 5;   // 0 P, 1 B, 2 I, 3 SP, 4 SI
    if (r.ue() != (uint32_t)pp.id) return COVAHIP_ERR_UNSUPPORTED;   // pic_parameter_set_id: only the first PPS of avcC is kept
    s->frame_num = (int)r.u(sp.log2_max_frame_num);
    if (!sp.frame_mbs_only) return COVAHIP_ERR_UNSUPPORTED;   // field / MBAFF coding
    s->idr = nal_type == 5;
    if (s->idr) r.ue();       // idr_pic_id
    s->poc_lsb = 0;
    if (sp.poc_type == 0) {
        s->poc_lsb = (int)r.u(sp.log2_max_poc_lsb);
        if (pp.bottom_field_pic_order) r.se();
    } else if (sp.poc_type == 1 && !sp.delta_pic_order_always_zero) {
        r.se();
        if (pp.bottom_field_pic_order) r.se();
    }
    if (pp.redundant_pic_cnt) r.ue();
    s->direct_spatial = 0;
    if (s->slice_type == 1) s->direct_spatial = (int)r.u(1);
    s->num_ref_l0 = pp.num_ref_l0;
    s->num_ref_l1 = pp.num_ref_l1;
    if (s->slice_type == 0 || s->slice_type == 1 || s->slice_type == 3) {
        if (r.u(1)) {   // num_ref_idx_active_override_flag
            s->num_ref_l0 = (int)r.ue() + 1;
            if (s->slice_type == 1) s->num_ref_l1 = (int)r.ue() + 1;
        }
        if (s->num_ref_l0 < 1 || s->num_ref_l0 > 32 || s->num_ref_l1 < 1 || s->num_ref_l1 > 32) return COVAHIP_ERR_BAD_DATA;
    }
    // ref_pic_list_modification (7.3.3.1)
    if (s->slice_type != 2 && s->slice_type != 4) {
        for (int list = 0; list < (s->slice_type == 1 ? 2 : 1); list++)
            if (r.u(1)) {
                uint32_t op;
                do {
                    op = r.ue();
                    if (op == 0 || op == 1 || op == 2) {
                        const uint32_t v = r.ue();
                        if (ext) (list == 0 ? ext->l0 : ext->l1).push_back(RplOp{(int)op, v});
                    }
                } while (op != 3 && !r.bad);
            }
    }
    // pred_weight_table (7.3.3.2)
    if ((pp.weighted_pred && (s->slice_type == 0 || s->slice_type == 3)) || (pp.weighted_bipred == 1 && s->slice_type == 1)) {
        r.ue();
        if (sp.chroma_format != 0) r.ue();
        for (int list = 0; list < (s->slice_type == 1 ? 2 : 1); list++) {
            const int cnt = list == 0 ? s->num_ref_l0 : s->num_ref_l1;
            for (int i = 0; i < cnt && !r.bad; i++) {
                if (r.u(1)) { r.se(); r.se(); }
                if (sp.chroma_format != 0 && r.u(1)) { r.se(); r.se(); r.se(); r.se(); }
            }
        }
    }
    // dec_ref_pic_marking (7.3.3.3)
    if (nal_ref_idc != 0) {
        if (s->idr) {
            r.u(1);                                   // no_output_of_prior_pics_flag
            const uint32_t ltr = r.u(1);              // long_term_reference_flag
            if (ext) ext->long_term_reference = ltr != 0;
        } else if (r.u(1)) {
            if (ext) ext->adaptive = true;
            uint32_t op;
            do {
                op = r.ue();
                Mmco m{(int)op, 0, 0};
                if (op == 1 || op == 3) m.a = r.ue();   // difference_of_pic_nums_minus1
                if (op == 2) m.a = r.ue();              // long_term_pic_num
                if (op == 3) m.b = r.ue();              // long_term_frame_idx
                if (op == 6) m.a = r.ue();              // long_term_frame_idx
                if (op == 4) m.a = r.ue();              // max_long_term_frame_idx_plus1
                if (op == 5) s->has_mmco5 = 1;          // resets the picture order count (8.2.1)
                if (ext && op != 0) ext->mmco.push_back(m);
            } while (op != 0 && !r.bad);
        }
    }
    s->cabac_init_idc = -1;
    if (pp.entropy_cabac && s->slice_type != 2 && s->slice_type != 4) s->cabac_init_idc = (int)r.ue();
    s->qp = pp.pic_init_qp + r.se();
    if (s->slice_type == 3 || s->slice_type == 4) {
        if (s->slice_type == 3) r.u(1);
        r.se();
    }
    if (pp.deblocking_control) {
        const uint32_t idc = r.ue();
        if (idc != 1) { r.se(); r.se(); }
    }
    if (r.bad) return COVAHIP_ERR_BAD_DATA;
    size_t pos = r.pos;
    if (pp.entropy_cabac) pos = (pos + 7) & ~(size_t)7;   // cabac_alignment_one_bit
    s->data_bit_offset = (uint32_t)pos;                   // in the unescaped RBSP, after the NAL header byte
    s->nal_bytes = (uint32_t)n;
    return COVAHIP_OK;
}

// Picture order count of the next access unit in decode order (8.2.1.1 type 0, 8.2.1.3 type 2; frames only) as a key whose
// ascending order is the output order: (IDR period << 32) + POC + 2^31.  false for POC type 1.
bool poc_step(const Sps &sp, covahip_h264::PocState &st, const covahip_h264_slice &sl, int64_t &key, int64_t *out_key = nullptr) {
    if (sp.poc_type == 1) return false;
    const int64_t max_lsb = 1ll << sp.log2_max_poc_lsb, max_fn = 1ll << sp.log2_max_frame_num;
    const bool idr = sl.idr != 0, ref = sl.nal_ref_idc != 0;
    int64_t poc;
    if (idr) { st.period++; st.prev_msb = st.prev_lsb = 0; st.fn_off = 0; }
    if (st.period < 0) st.period = 0;
    if (sp.poc_type == 0) {
        const int64_t lsb = sl.poc_lsb;
        int64_t msb = st.prev_msb;
        if (lsb < st.prev_lsb && st.prev_lsb - lsb >= max_lsb / 2) msb = st.prev_msb + max_lsb;
        else if (lsb > st.prev_lsb && lsb - st.prev_lsb > max_lsb / 2) msb = st.prev_msb - max_lsb;
        poc = msb + lsb;
        if (ref) {
            st.prev_msb = msb; st.prev_lsb = lsb;
        }
    } else {
        if (!idr && sl.frame_num < st.prev_fn) st.fn_off += max_fn;
        poc = idr ? 0 : 2 * (st.fn_off + sl.frame_num) - (ref ? 0 : 1);
    }
    st.prev_fn = sl.frame_num;
    // A reference picture that carries memory_management_control_operation 5 starts an output period the way an IDR picture
    // does: everything decoded before it leaves first (C.4.4), its own count becomes 0 once tempPicOrderCnt is subtracted
    // (8.2.1, frames) and the pictures behind it count from there (prevPicOrderCntMsb/Lsb 0, prevFrameNumOffset 0, frame_num 0)
    // `key` is the count the picture predicts with (its lists, temporal direct); `out_key` the one it leaves the decoder by
    key = (st.period << 32) + poc + (1ll << 31);
    if (ref && sl.has_mmco5) {
        st.period++;
        st.prev_msb = st.prev_lsb = 0;
        st.fn_off = 0;
        st.prev_fn = 0;
        if (out_key) *out_key = (st.period << 32) + (1ll << 31);
    } else if (out_key) *out_key = key;
    return true;
}

int au_slices(const covahip_h264 *h, const uint8_t *au, size_t len, uint64_t base, covahip_h264_slice *out, int cap, int *n,
              SliceExt *first_ext);

// Output order of a file's access units: within an IDR period pictures leave the decoder by ascending POC.  Leaves `display`
// empty when a header does not parse or for POC type 1.
void compute_display_order(covahip_h264 *h) {
    const size_t n = h->samples.size();
    h->order_key.assign(n, 0);
    h->col_sample.assign(n, -1);
    h->col_short.assign(n, 0);
    h->lists0.assign(n, {});
    h->lists1.assign(n, {});
    covahip_h264::PocState st;
    std::vector<int64_t> out_key(n, 0);
    Dpb dpb;
    for (size_t i = 0; i < n; i++) {
        covahip_h264_slice sl[1];
        SliceExt ext;
        int cnt = 0;
        const Sample &sm = h->samples[i];
        const int rc = au_slices(h, h->data + sm.off, sm.size, sm.off, sl, 1, &cnt, &ext);
        if ((rc != COVAHIP_OK && rc != COVAHIP_ERR_OVERFLOW) || cnt < 1 || !poc_step(h->sps, st, sl[0], h->order_key[i], &out_key[i])) {
            h->order_key.clear();
            h->col_sample.assign(n, -1);
            h->lists0.assign(n, {});
            h->lists1.assign(n, {});
            return;
        }
        // RefPicList1[0] of a B picture (the co-located picture of its direct prediction), then the marking of this picture
        int64_t key = h->order_key[i];
        if (sl[0].slice_type == 0 || sl[0].slice_type == 1) {
            std::vector<int> l0, l1;
            dpb.build_lists(h->sps, sl[0], ext, key, l0, l1);
            list_entries(dpb, l0, h->lists0[i]);
            list_entries(dpb, l1, h->lists1[i]);
            if (sl[0].slice_type == 1 && !h->lists1[i].empty() && h->lists1[i][0].id >= 0) {
                h->col_sample[i] = h->lists1[i][0].id;
                h->col_short[i] = h->lists1[i][0].lt ? 0 : 1;
            }
        }
        if (sl[0].has_mmco5) key = (st.period << 32) + (1ll << 31);   // tempPicOrderCnt subtracted: the picture's own count becomes 0
        dpb.mark(h->sps, sl[0], ext, (int)i, key);
    }
    h->display.resize(n);
    for (size_t i = 0; i < n; i++) h->display[i] = (int32_t)i;
    std::stable_sort(h->display.begin(), h->display.end(), [&](int32_t a, int32_t b) { return out_key[a] < out_key[b]; });
}

// Slice NAL units of one access unit (length-prefixed NAL units at au[0, len)); nal_offset = base + offset inside au.
int au_slices(const covahip_h264 *h, const uint8_t *au, size_t len, uint64_t base, covahip_h264_slice *out, int cap, int *n,
              SliceExt *first_ext) {
    size_t p = 0;
    int cnt = 0;
    while (p + h->nal_len_size <= len) {
        size_t l = 0;
        for (int k = 0; k < h->nal_len_size; k++) l = (l << 8) | au[p + k];
        p += h->nal_len_size;
        if (l == 0 || l > len - p) return COVAHIP_ERR_BAD_DATA;
        const int t = au[p] & 31;
        if (t == 1 || t == 5) {
            if (cnt < cap) {
                covahip_h264_slice sl;
                std::memset(&sl, 0, sizeof sl);
                int rc = parse_slice_header(h, au + p, l, &sl, cnt == 0 ? first_ext : nullptr);
                if (rc) return rc;
                sl.nal_offset = base + (uint64_t)p;
                out[cnt] = sl;
            }
            cnt++;
        }
        p += l;
    }
    if (p != len) return COVAHIP_ERR_BAD_DATA;
    *n = cnt;
    return cnt > cap ? COVAHIP_ERR_OVERFLOW : COVAHIP_OK;
}

// SPS / PPS of an AVCDecoderConfigurationRecord (the avcC box payload = the codec_data of video/x-h264,stream-format=avc caps)
int parse_avcc(covahip_h264 *h, const uint8_t *a, size_t n) {
    if (n < 7 || a[0] != 1) return COVAHIP_ERR_BAD_DATA;
    h->nal_len_size = (a[4] & 3) + 1;
    size_t p = 5;
    const int nsps = a[p++] & 31;
    for (int k = 0; k < nsps && p + 2 <= n; k++) {
        const size_t l = ((size_t)a[p] << 8) | a[p + 1];
        if (p + 2 + l > n) break;
        if (k == 0 && !parse_sps(unescape(a + p + 2, l), h->sps)) return COVAHIP_ERR_UNSUPPORTED;
        p += 2 + l;
    }
    const int npps = p < n ? a[p++] : 0;
    for (int k = 0; k < npps && p + 2 <= n; k++) {
        const size_t l = ((size_t)a[p] << 8) | a[p + 1];
        if (p + 2 + l > n) break;
        if (k == 0 && !parse_pps(unescape(a + p + 2, l), h->pps)) return COVAHIP_ERR_UNSUPPORTED;
        p += 2 + l;
    }
    if (!h->sps.ok || !h->pps.ok) return COVAHIP_ERR_BAD_DATA;
    if (h->pps.sps_id != h->sps.id) return COVAHIP_ERR_UNSUPPORTED;   // only the first SPS / PPS are kept
    if (h->sps.bit_depth != 8) return COVAHIP_ERR_UNSUPPORTED;        // High 10 and up
    return COVAHIP_OK;
}

// Entropy-decodes the single whole-picture slice `sl` whose NAL unit starts at nal.
// col: the motion of RefPicList1[0] (NULL: not known / long-term / not a B slice); l0, l1, key: this picture's lists and count
// (temporal direct); mine (may be NULL) receives this picture's motion for the pictures after it.
int decode_slice_records(const covahip_h264 *h, const uint8_t *nal, const covahip_h264_slice &sl, uint8_t *records, size_t cap,
                         const covahip_h264::PicMotion *col, const std::vector<RefEntry> *l0, const std::vector<RefEntry> *l1, int64_t key,
                         covahip_h264::PicMotion *mine) {
    const int wmb = h->sps.width_mbs, hmb = h->sps.height_map_units;
    if (records && cap < (size_t)wmb * hmb * 4) return COVAHIP_ERR_OVERFLOW;
    // CAVLC streams, field / MBAFF coding: not built -- refused, never faked
    if (!h->pps.entropy_cabac || !h->sps.frame_mbs_only) return COVAHIP_ERR_UNSUPPORTED;
    if (sl.slice_type != 2 && sl.cabac_init_idc != 0) return COVAHIP_ERR_UNSUPPORTED;
    const std::vector<uint8_t> rbsp = unescape(nal + 1, sl.nal_bytes - 1);
    h264::SliceParams sp;
    sp.slice_type = sl.slice_type;
    sp.first_mb = sl.first_mb;
    sp.qp = sl.qp;
    sp.cabac_init_idc = sl.cabac_init_idc < 0 ? 0 : sl.cabac_init_idc;
    sp.num_ref_l0 = sl.num_ref_l0;
    sp.num_ref_l1 = sl.num_ref_l1;
    sp.width_mbs = wmb;
    sp.height_mbs = hmb;
    sp.transform_8x8 = h->pps.transform_8x8;
    sp.direct_8x8_inference = h->sps.direct_8x8;
    sp.chroma_format = h->sps.chroma_format;
    sp.direct_spatial = sl.direct_spatial;
    int8_t col_to_l0[64];
    int16_t dist_scale[32];
    if (sl.slice_type == 1 && col) {
        sp.col_still = col->still.data();
        if (!sl.direct_spatial && l0 && l1) {
            temporal_tables(*l0, *l1, key, col->l0, col->l1, col_to_l0, dist_scale);
            sp.col_motion = col->motion.data();
            sp.col_to_l0 = col_to_l0;
            sp.dist_scale = dist_scale;
        }
    }
    if (mine) {
        mine->still.assign((size_t)wmb * hmb, 0);
        mine->motion.assign((size_t)wmb * hmb, h264::ColMb());
        sp.still_out = mine->still.data();
        sp.motion_out = mine->motion.data();
    }
    std::string why;
    const int rc = h264::parse_slice_cabac(rbsp.data(), rbsp.size(), sl.data_bit_offset, sp, records, &why);
    if (rc && getenv("COVAHIP_H264_DEBUG")) fprintf(stderr, "covahip h264: %s\n", why.c_str());
    return rc;
}

// File form: the "does not move" bits of sample `sample` (a reference picture), decoding it -- and, for a B reference picture,
// first the picture its own direct prediction looks at -- when no earlier call has left them in the cache.
covahip_h264::Still still_of(const covahip_h264 *h, int sample, int depth);
int decode_sample(const covahip_h264 *h, int sample, uint8_t *records, size_t cap, int depth) {
    covahip_h264_slice sl[2];
    int n = 0;
    const int rc = covahip_h264_sample_slices(h, sample, sl, 2, &n);
    if (rc == COVAHIP_ERR_OVERFLOW || (rc == COVAHIP_OK && n != 1)) return COVAHIP_ERR_UNSUPPORTED;   // several slices per picture
    if (rc) return rc;
    const bool tables = (size_t)sample < h->col_sample.size() && h->lists0.size() == h->col_sample.size();
    covahip_h264::Still col;
    if (sl[0].slice_type == 1 && tables && h->col_sample[(size_t)sample] >= 0 && h->col_short[(size_t)sample] && depth < 8)
        col = still_of(h, h->col_sample[(size_t)sample], depth + 1);
    covahip_h264::Still mine;
    if (sl[0].nal_ref_idc != 0 && tables) {
        mine = std::make_shared<covahip_h264::PicMotion>();
        mine->l0 = h->lists0[(size_t)sample];
        mine->l1 = h->lists1[(size_t)sample];
    }
    const int rc2 = decode_slice_records(h, h->data + sl[0].nal_offset, sl[0], records, cap, col.get(),
                                         tables ? &h->lists0[(size_t)sample] : nullptr, tables ? &h->lists1[(size_t)sample] : nullptr,
                                         tables && !h->order_key.empty() ? h->order_key[(size_t)sample] : 0, mine.get());
    if (rc2 == COVAHIP_OK && mine) {
        std::lock_guard<std::mutex> lock(h->still_mu);
        if (!h->still_cache.count(sample)) {
            h->still_cache[sample] = mine;
            h->still_order.push_back(sample);
            if (h->still_order.size() > 24) {   // a reference picture serves the B pictures up to the next one: a handful is plenty
                h->still_cache.erase(h->still_order.front());
                h->still_order.erase(h->still_order.begin());
            }
        }
    }
    return rc2;
}
covahip_h264::Still still_of(const covahip_h264 *h, int sample, int depth) {
    {
        std::lock_guard<std::mutex> lock(h->still_mu);
        auto it = h->still_cache.find(sample);
        if (it != h->still_cache.end()) return it->second;
    }
    if (decode_sample(h, sample, nullptr, 0, depth) != COVAHIP_OK) return nullptr;
    std::lock_guard<std::mutex> lock(h->still_mu);
    auto it = h->still_cache.find(sample);
    return it != h->still_cache.end() ? it->second : nullptr;
}

}  // namespace

extern "C" {

int covahip_h264_display_order(const covahip_h264 *h, int32_t *samples, int cap, int *n) {
    if (!h || !n || (!samples && cap)) return COVAHIP_ERR_INVALID_ARG;
    if (h->display.empty()) return COVAHIP_ERR_UNSUPPORTED;
    *n = (int)h->display.size();
    for (int i = 0; i < cap && i < *n; i++) samples[i] = h->display[i];
    return cap < *n ? COVAHIP_ERR_OVERFLOW : COVAHIP_OK;
}

int covahip_h264_open_mp4(const uint8_t *file, size_t len, covahip_h264 **out) {
    if (!file || !out || len < 16) return COVAHIP_ERR_INVALID_ARG;
    *out = nullptr;
    size_t mb, me, tb, te, b, e;
    if (!find_box(file, 0, len, "moov", mb, me)) return COVAHIP_ERR_BAD_DATA;
    covahip_h264 *h = new (std::nothrow) covahip_h264();
    if (!h) return COVAHIP_ERR_INVALID_ARG;
    h->data = file;
    h->len = len;
    // the first track that carries an avcC
    size_t toff = mb;
    bool found = false;
    while (!found && find_box(file, toff, me, "trak", tb, te)) {
        toff = te;
        size_t sb, se;
        if (!(find_box(file, tb, te, "mdia", b, e) && find_box(file, b, e, "minf", b, e) && find_box(file, b, e, "stbl", sb, se))) continue;
        size_t db, de;
        if (!find_box(file, sb, se, "stsd", db, de)) continue;
        // avcC sits inside the avc1 sample entry: search the stsd payload for its tag
        size_t a = 0;
        for (size_t i = db; i + 8 < de; i++)
            if (std::memcmp(file + i, "avcC", 4) == 0) { a = i + 4; break; }
        if (!a || a + 8 > de) continue;
        {
            const int rc = parse_avcc(h, file + a, de - a);
            if (rc) { delete h; return rc; }
        }
        // sample sizes, chunk offsets, samples per chunk, sync samples
        size_t zb, ze, cb, ce, scb, sce;
        if (!find_box(file, sb, se, "stsz", zb, ze) || !find_box(file, sb, se, "stsc", scb, sce)) { delete h; return COVAHIP_ERR_BAD_DATA; }
        const bool co64 = !find_box(file, sb, se, "stco", cb, ce);
        if (co64 && !find_box(file, sb, se, "co64", cb, ce)) { delete h; return COVAHIP_ERR_BAD_DATA; }
        if (zb + 12 > ze || cb + 8 > ce || scb + 8 > sce) { delete h; return COVAHIP_ERR_BAD_DATA; }
        const uint32_t fixed = be32(file + zb + 4), ns = be32(file + zb + 8);
        const uint32_t nchunks = be32(file + cb + 4), nsc = be32(file + scb + 4);
        if (ns > (1u << 26) || nchunks > (1u << 26) || nsc > (1u << 20)) { delete h; return COVAHIP_ERR_BAD_DATA; }
        if ((!fixed && zb + 12 + 4ull * ns > ze) || cb + 8 + (co64 ? 8ull : 4ull) * nchunks > ce || scb + 8 + 12ull * nsc > sce) { delete h; return COVAHIP_ERR_BAD_DATA; }
        h->samples.resize(ns);
        uint32_t si = 0, sck = 0, per = 0;
        for (uint32_t c = 0; c < nchunks && si < ns; c++) {
            // samples per chunk: the last stsc entry whose first_chunk <= c + 1 (entries are in chunk order: one running index)
            while (sck < nsc && be32(file + scb + 8 + 12 * sck) <= c + 1) per = be32(file + scb + 8 + 12 * sck++ + 4);
            uint64_t off = co64 ? be64(file + cb + 8 + 8ull * c) : be32(file + cb + 8 + 4ull * c);
            for (uint32_t k = 0; k < per && si < ns; k++, si++) {
                const uint32_t sz = fixed ? fixed : be32(file + zb + 12 + 4ull * si);
                if (off > len || sz > len - off) { delete h; return COVAHIP_ERR_BAD_DATA; }
                h->samples[si] = Sample{off, sz, false};
                off += sz;
            }
        }
        if (si != ns) { delete h; return COVAHIP_ERR_BAD_DATA; }
        size_t yb, ye;
        if (find_box(file, sb, se, "stss", yb, ye) && yb + 8 <= ye) {
            const uint32_t n = be32(file + yb + 4);
            for (uint32_t k = 0; k < n && yb + 8 + 4ull * k + 4 <= ye; k++) {
                const uint32_t s1 = be32(file + yb + 8 + 4ull * k);
                if (s1 >= 1 && s1 <= ns) h->samples[s1 - 1].sync = true;
            }
        } else {
            for (auto &s : h->samples) s.sync = true;
        }
        found = true;
    }
    if (!found) { delete h; return COVAHIP_ERR_BAD_DATA; }
    compute_display_order(h);
    *out = h;
    return COVAHIP_OK;
}

void covahip_h264_close(covahip_h264 *h) { delete h; }

int covahip_h264_get_info(const covahip_h264 *h, covahip_h264_info *info) {
    if (!h || !info) return COVAHIP_ERR_INVALID_ARG;
    info->width_mbs = h->sps.width_mbs;
    info->height_mbs = h->sps.height_map_units * (2 - h->sps.frame_mbs_only);
    info->n_samples = (int)h->samples.size();
    info->profile_idc = h->sps.profile;
    info->level_idc = h->sps.level;
    info->entropy_cabac = h->pps.entropy_cabac;
    info->transform_8x8 = h->pps.transform_8x8;
    info->num_ref_frames = h->sps.num_ref_frames;
    info->frame_mbs_only = h->sps.frame_mbs_only;
    info->weighted_pred = h->pps.weighted_pred;
    info->weighted_bipred = h->pps.weighted_bipred;
    info->poc_type = h->sps.poc_type;
    info->max_num_reorder_frames = h->sps.max_num_reorder;
    info->max_dec_frame_buffering = h->sps.max_dec_frame_buffering;
    return COVAHIP_OK;
}

int covahip_h264_sample(const covahip_h264 *h, int sample, uint64_t *offset, uint32_t *size, int *is_sync) {
    if (!h || sample < 0 || sample >= (int)h->samples.size()) return COVAHIP_ERR_INVALID_ARG;
    if (offset) *offset = h->samples[sample].off;
    if (size) *size = h->samples[sample].size;
    if (is_sync) *is_sync = h->samples[sample].sync ? 1 : 0;
    return COVAHIP_OK;
}

int covahip_h264_sample_slices(const covahip_h264 *h, int sample, covahip_h264_slice *out, int cap, int *n) {
    if (!h || !n || sample < 0 || sample >= (int)h->samples.size() || (!out && cap)) return COVAHIP_ERR_INVALID_ARG;
    const Sample &s = h->samples[sample];
    return au_slices(h, h->data + s.off, s.size, s.off, out, cap, n, nullptr);
}

int covahip_h264_decode_records(const covahip_h264 *h, int sample, uint8_t *records, size_t cap) {
    if (!h || sample < 0 || sample >= (int)h->samples.size()) return COVAHIP_ERR_INVALID_ARG;
    if (records && cap < (size_t)h->sps.width_mbs * h->sps.height_map_units * 4) return COVAHIP_ERR_OVERFLOW;
    return decode_sample(h, sample, records, cap, 0);
}

int covahip_h264_colocated(const covahip_h264 *h, int sample, int *col_sample, int *short_term) {
    if (!h || sample < 0 || sample >= (int)h->samples.size() || !col_sample || !short_term) return COVAHIP_ERR_INVALID_ARG;
    if (h->col_sample.size() != h->samples.size()) return COVAHIP_ERR_UNSUPPORTED;
    *col_sample = h->col_sample[(size_t)sample];
    *short_term = h->col_short[(size_t)sample];
    return COVAHIP_OK;
}

int covahip_h264_open_avcc(const uint8_t *avcc, size_t len, covahip_h264 **out) {
    if (!avcc || !out) return COVAHIP_ERR_INVALID_ARG;
    *out = nullptr;
    covahip_h264 *h = new (std::nothrow) covahip_h264();
    if (!h) return COVAHIP_ERR_INVALID_ARG;
    const int rc = parse_avcc(h, avcc, len);
    if (rc) { delete h; return rc; }
    *out = h;
    return COVAHIP_OK;
}

int covahip_h264_decode_au(covahip_h264 *h, const uint8_t *au, size_t len, uint8_t *records, size_t cap, covahip_h264_slice *hdr,
                           int64_t *order_key) {
    if (!h || !au) return COVAHIP_ERR_INVALID_ARG;
    if (records && cap < (size_t)h->sps.width_mbs * h->sps.height_map_units * 4) return COVAHIP_ERR_OVERFLOW;
    covahip_h264_slice sl[2];
    SliceExt ext;
    int n = 0;
    const int rc = au_slices(h, au, len, 0, sl, 2, &n, &ext);
    if (rc == COVAHIP_ERR_OVERFLOW || (rc == COVAHIP_OK && n != 1)) return COVAHIP_ERR_UNSUPPORTED;
    if (rc) return rc;
    int64_t key = 0, okey = 0;
    const bool have_key = poc_step(h->sps, h->poc, sl[0], key, &okey);
    if (order_key) {
        if (!have_key) return COVAHIP_ERR_UNSUPPORTED;
        *order_key = okey;
    }
    if (hdr) *hdr = sl[0];
    // access units arrive in decoding order: the reference pictures so far give RefPicList1[0] of a B picture, whose "does
    // not move" bits (kept while the picture is a reference) feed the colZeroFlag test of its direct prediction
    covahip_h264::Still col, mine;
    std::vector<RefEntry> l0e, l1e;
    if (have_key && (sl[0].slice_type == 0 || sl[0].slice_type == 1)) {
        std::vector<int> l0, l1;
        h->dpb.build_lists(h->sps, sl[0], ext, key, l0, l1);
        list_entries(h->dpb, l0, l0e);
        list_entries(h->dpb, l1, l1e);
        if (sl[0].slice_type == 1 && !l1e.empty() && l1e[0].id >= 0 && !l1e[0].lt) {
            auto it = h->still_live.find(l1e[0].id);
            if (it != h->still_live.end()) col = it->second;
        }
    }
    if (have_key && sl[0].nal_ref_idc != 0) {
        mine = std::make_shared<covahip_h264::PicMotion>();
        mine->l0 = l0e;
        mine->l1 = l1e;
    }
    const int rc2 = decode_slice_records(h, au + sl[0].nal_offset, sl[0], records, cap, col.get(), &l0e, &l1e, key, mine.get());
    if (have_key) {
        const int id = h->au_count++;
        if (sl[0].has_mmco5) key = (h->poc.period << 32) + (1ll << 31);
        h->dpb.mark(h->sps, sl[0], ext, id, key);
        if (mine && rc2 == COVAHIP_OK) h->still_live[id] = mine;
        for (auto it = h->still_live.begin(); it != h->still_live.end();) {   // pictures that left the reference set
            bool live = false;
            for (const RefPic &r : h->dpb.refs) live = live || r.id == it->first;
            it = live ? std::next(it) : h->still_live.erase(it);
        }
    }
    return rc2;
}

int covahip_carrier_write_records(const uint8_t *mb_type, const uint8_t *mv_x, const uint8_t *mv_y, int width_mbs, int height_mbs,
                                  uint8_t *frame, size_t frame_bytes) {
    if (!mb_type || !mv_x || !mv_y || !frame || width_mbs <= 0 || height_mbs <= 0) return COVAHIP_ERR_INVALID_ARG;
    const size_t n = (size_t)width_mbs * height_mbs;
    if (frame_bytes < n * 4) return COVAHIP_ERR_OVERFLOW;
    for (size_t i = 0; i < n; i++) {   // byte 0 / 1 / 2 = mb_type / mv_x / mv_y (tfrecordsink/imp.rs:105-112), byte 3 unused
        frame[4 * i] = mb_type[i];
        frame[4 * i + 1] = mv_x[i];
        frame[4 * i + 2] = mv_y[i];
        frame[4 * i + 3] = 0;
    }
    return COVAHIP_OK;
}

}  // extern "C"
