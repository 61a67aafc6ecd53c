"""TEST INFRASTRUCTURE ONLY -- CPU restatement of CoVA's analysis-aggregator association rules.

Follows cova-rs/analysis-aggregator/src/server/assoc.rs (Associator, :63-441; message loop :443-507),
track.rs:59-66 (scale_dim(16), track ids re-based by range_start) and dnn.rs:57-86 (text rows
"timestamp,left,top,width,height,class_id").  Only tests/ may import this module.

Parity status: UNPINNED -- the reference has no test, fixture or golden file for the aggregator, its
sources cannot be built here (Rust toolchain absent) and three of its outputs depend on HashMap
iteration order.  This restatement fixes those orders (documented below); the product (hostlib.cpp,
covahip_assoc_*) must reproduce this file row for row.

Deterministic choices where the reference iterates a HashMap:
  * finalize_trk: classes are visited in ascending class id; `max_by_key` keeps the LAST maximum, so
    among equally frequent classes the largest id is "the most frequent" one; the remaining classes
    follow in ascending id.
  * terminate: ranges are visited in ascending range_start.

Reference quirks kept (tests name them): a pending detection matches a NEW track with `iou > moving_iou`
but an existing track with `iou >= moving_iou`; `finalize_stationary`'s "at least two detections"
filter compares range_start with range_end and therefore never drops anything; `terminate` calls
finalize_trk(range_end), which cannot match any remaining track, so tracks still pending at the end
never reach assoc.csv.
"""
from __future__ import annotations

import copy

import numpy as np

from .sort_ref import Bbox

f32 = np.float32
U64_MAX = (1 << 64) - 1
TIMESTEP = 33_333_333
TIMESTEP_3 = 100_000_000


def scale_dim(b: Bbox, s) -> None:          # bbox.rs:58-67
    s = f32(s)
    if s == 1:
        return
    b.left, b.top, b.width, b.height = f32(b.left * s), f32(b.top * s), f32(b.width * s), f32(b.height * s)
    b.area = f32(b.area * f32(s * s))


def scale(b: Bbox, s) -> None:              # bbox.rs:69-82 (centroid stays)
    s = f32(s)
    if s == 1:
        return
    x = f32(b.left + f32(b.width / f32(2)))
    y = f32(b.top + f32(b.height / f32(2)))
    b.width, b.height = f32(b.width * s), f32(b.height * s)
    b.left = f32(x - f32(b.width / f32(2)))
    b.top = f32(y - f32(b.height / f32(2)))
    b.area = f32(b.area * f32(s * s))


class Stationary:                            # assoc.rs:11-58
    def __init__(self, range_start, range_end, bbox: Bbox):
        self.range_start, self.range_end = range_start, range_end
        self.start = self.end = bbox.timestamp
        self.class_id = bbox.class_id
        self.bbox = bbox
        self.track_id = None

    def to_vec(self):
        out = []
        for ts in range(self.start, self.end, TIMESTEP_3):
            for i in range(2):
                b = copy.copy(self.bbox)
                b.timestamp = ts + i * TIMESTEP
                b.track_id = self.track_id
                out.append(b)
        return out


class Associator:
    def __init__(self, range_starts, moving_iou=0.15, stationary_iou=0.3, stationary_maxage=120, scale_factor=1.3):
        rs = sorted(range_starts) + [U64_MAX]                               # assoc.rs:476-489
        self.tracker_range = {rs[i]: rs[i + 1] for i in range(len(range_starts))}
        self.rows = {"track": [], "dnn": [], "assoc": [], "stationary": []}
        self.tracks, self.dnns, self.stationary, self.finalized_stationary = [], [], [], []
        self.track2class = {}
        self.moving_iou, self.stationary_iou = f32(moving_iou), f32(stationary_iou)
        self.stationary_maxage = int(stationary_maxage) * 1_000_000_000
        self.scale_factor = f32(scale_factor)
        self.max_track_id = 0

    # ---- assoc.rs:127-215
    def finalize_trk(self, ts):
        keep = []
        for rs, re, trk in self.tracks:
            if rs <= ts < re and trk[-1].timestamp < ts:
                classes = self.track2class.pop(trk[0].track_id, None)
                class_ids = []
                if classes:
                    count = {}
                    for c in classes:
                        count[c] = count.get(c, 0) + 1
                    best, freq = None, -1
                    for c in sorted(count):
                        if count[c] >= freq:
                            best, freq = c, count[c]
                    del count[best]
                    class_ids.append(best)
                    for c in sorted(count):
                        if freq != 1:
                            if count[c] >= 2:
                                class_ids.append(c)
                        else:
                            class_ids.append(c)
                for c in class_ids:
                    for b in trk:
                        b.class_id = c
                        self.rows["assoc"].append(copy.copy(b))
            else:
                keep.append((rs, re, trk))
        self.tracks = keep

    # ---- assoc.rs:220-267
    def finalize_dnn(self, range_start, range_end, ts):
        keep = []
        for matched, b in self.dnns:
            if range_start <= b.timestamp < range_end and b.timestamp < ts:
                if not matched:
                    best, best_iou = None, None
                    for s in self.stationary:
                        if s.range_start != range_start or s.class_id != b.class_id:
                            continue
                        iou = s.bbox.iou(b)
                        if iou >= self.stationary_iou and (best is None or iou >= best_iou):
                            best, best_iou = s, iou
                    if best is not None:
                        best.end = b.timestamp
                    else:
                        self.stationary.append(Stationary(range_start, range_end, b))
            else:
                keep.append((matched, b))
        self.dnns = keep

    # ---- assoc.rs:271-287
    def finalize_stationary(self, ts):
        keep = []
        for s in self.stationary:
            if s.range_start <= ts < s.range_end and self.stationary_maxage + s.end < ts:
                if s.range_start != s.range_end:      # the reference's (always true) "two detections" filter
                    self.finalized_stationary.append(s)
            else:
                keep.append(s)
        self.stationary = keep

    def _match(self, trk, ts):
        for b in trk:
            if b.timestamp == ts:
                s = copy.copy(b)
                scale(s, self.scale_factor)
                return s
        raise ValueError("track has no box at the detection's timestamp (the reference unwraps here)")

    # ---- assoc.rs:296-367
    def update_dnn(self, boxes):
        seen = []
        for b in boxes:
            if b.timestamp not in seen:
                seen.append(b.timestamp)
        for ts in seen:
            self.finalize_stationary(ts)
            self.finalize_trk(ts)
        for b in boxes:
            ts = b.timestamp
            self.rows["dnn"].append(copy.copy(b))
            matched = False
            for rs, re, trk in self.tracks:
                if rs <= ts < re and trk[0].timestamp <= ts:
                    s = self._match(trk, ts)
                    if s.iou(b) >= self.moving_iou:
                        self.track2class.setdefault(s.track_id, []).append(b.class_id)
                        matched = True
            self.dnns.append((matched, b))

    # ---- assoc.rs:370-431
    def update_track(self, range_start, oldest, trk):
        range_end = self.tracker_range[range_start]
        for b in trk:
            self.rows["track"].append(copy.copy(b))
        self.max_track_id = max(self.max_track_id, trk[0].track_id)
        t0, t1 = trk[0].timestamp, trk[-1].timestamp
        for i, (matched, b) in enumerate(self.dnns):
            if t0 <= b.timestamp <= t1:
                s = self._match(trk, b.timestamp)
                if s.iou(b) > self.moving_iou:
                    self.track2class.setdefault(s.track_id, []).append(b.class_id)
                    self.dnns[i] = (True, b)
        self.tracks.append((range_start, range_end, trk))
        self.finalize_dnn(range_start, range_end, oldest)

    # ---- track.rs:47-66: what a tracker connection does to a received Frame before the associator sees it
    def ingest_frame(self, range_start, oldest, boxes):
        out = []
        for b in boxes:
            b = copy.copy(b)
            scale_dim(b, 16.0)
            b.track_id += range_start
            out.append(b)
        self.update_track(range_start, oldest, out)

    # ---- assoc.rs:434-467
    def terminate(self):
        for rs in sorted(self.tracker_range):
            re = self.tracker_range[rs]
            self.finalize_trk(re)
            self.finalize_dnn(rs, re, re)
            self.finalize_stationary(re)
        tid = self.max_track_id + 1
        for s in self.finalized_stationary:
            s.track_id = tid
            tid += 1
            self.rows["stationary"].extend(s.to_vec())


def _fmt_f32(v) -> str:
    """Text of an f32 as serde/csv writes it (crate ryu, pretty::format32): shortest round-trip digits
    d1..dn x 10^k, kk = n + k; plain notation while 0 <= k, kk <= 13 or 0 < kk <= 13 or -6 < kk <= 0,
    else exponent notation."""
    v = np.float32(v)
    if v == 0:
        return "-0.0" if np.signbit(v) else "0.0"
    mant, exp = np.format_float_scientific(v, unique=True, trim="-").split("e")
    neg = "-" if mant.startswith("-") else ""
    digits = mant.replace("-", "").replace(".", "")
    n, kk = len(digits), int(exp) + 1
    k = kk - n
    if k >= 0 and kk <= 13:
        return neg + digits + "0" * k + ".0"
    if 0 < kk <= 13:
        return neg + digits[:kk] + "." + digits[kk:]
    if -6 < kk <= 0:
        return neg + "0." + "0" * (-kk) + digits
    return neg + digits[0] + ("." + digits[1:] if n > 1 else "") + f"e{kk - 1}"


def csv_text(rows) -> str:
    """serde field order of bbox.rs:12-27; header only when there is at least one record."""
    if not rows:
        return ""
    out = ["left,top,width,height,area,track_id,timestamp,class_id,confidence"]
    for b in rows:
        opt = lambda v, f=str: "" if v is None else f(v)
        out.append(",".join([_fmt_f32(b.left), _fmt_f32(b.top), _fmt_f32(b.width), _fmt_f32(b.height), _fmt_f32(b.area),
                             opt(b.track_id), opt(b.timestamp), opt(b.class_id), opt(b.confidence, _fmt_f32)]))
    return "\n".join(out) + "\n"
