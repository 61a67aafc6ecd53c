"""numpy restatement of CoVA's SORT tracker and cova GoP filter -- TEST INFRASTRUCTURE ONLY.

Only tests/ may import this.  It follows, line by line where it matters:
  cova-rs/bbox/src/bbox.rs:17-56                     Bbox::new / iou
  cova-rs/sort/src/state.rs:9-28                     into_z / from_x (top uses width: quirk)
  cova-rs/sort/src/tracker/mod.rs:33-151             KalmanBoxTracker
  cova-rs/sort/src/tracker/motion_model.rs:38-55     F, Q
  cova-rs/sort/src/tracker/linear_observation_model.rs:33-47   H, R
  cova-rs/sort/src/lib.rs:25-214                     linear_assignment / match_dets / update / finalize
  cova-rs/gst-plugins/src/cova/imp.rs:90-432, cova/tracker.rs:43-60   GoP filter
Third-party pieces (sources absent; PARITY UNPINNED beyond the reference's own KATs):
  adskalman 0.13.0 predict + Joseph-form update -> textbook formulas in float32;
  linear_assignment 0.0.2 -> scipy.optimize.linear_sum_assignment (any optimal perfect
  matching of the zero-padded square matrix; the reference only tests edge membership).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np
from scipy.optimize import linear_sum_assignment

f32 = np.float32


@dataclass
class Bbox:
    left: float
    top: float
    width: float
    height: float
    area: float = None
    track_id: int | None = None
    timestamp: int | None = None
    class_id: int | None = None
    confidence: float | None = None

    def __post_init__(self):
        self.left, self.top, self.width, self.height = map(f32, (self.left, self.top, self.width, self.height))
        if self.area is None:
            self.area = f32(self.width * self.height)

    def iou(self, t: "Bbox") -> np.float32:
        sx2, sy2 = f32(self.left + self.width), f32(self.top + self.height)
        tx2, ty2 = f32(t.left + t.width), f32(t.top + t.height)
        xl, yt = max(self.left, t.left), max(self.top, t.top)
        xr, yb = min(sx2, tx2), min(sy2, ty2)
        if xr <= xl or yb <= yt:
            return f32(0.0)
        inter = f32(f32(xr - xl) * f32(yb - yt))
        union = f32(f32(self.area + t.area) - inter)
        return f32(inter / union)


def into_z(b: Bbox):
    return np.array([b.left + b.width / f32(2), b.top + b.height / f32(2), b.area, b.width / b.height], dtype=f32)


def from_x(x) -> Bbox:
    with np.errstate(invalid="ignore", divide="ignore"):
        width = np.sqrt(f32(x[2] * x[3]))
        height = f32(x[2] / width)
        return Bbox(x[0] - width / f32(2), x[1] - width / f32(2), width, height)


F = np.eye(7, dtype=f32)
F[0, 4] = F[1, 5] = F[2, 6] = 1.0
Q = np.diag(np.array([1, 1, 1, 1, 0.01, 0.01, 0.0001], dtype=f32))
H = np.zeros((4, 7), dtype=f32)
H[:4, :4] = np.eye(4, dtype=f32)
R = np.diag(np.array([1, 1, 10, 10], dtype=f32))


class Tracker:
    def __init__(self, tid: int, bbox: Bbox, start: int):
        self.id, self.start, self.last_match = tid, start, start
        self.seen_ts: list[int] = []
        self.active = False
        self.history: list[Bbox] = []
        self.hits = self.time_since_update = self.hit_streaks = self.age = 0
        self.x = np.concatenate([into_z(bbox), np.zeros(3, dtype=f32)]).astype(f32)
        self.P = np.diag(np.array([10, 10, 10, 10, 1e4, 1e4, 1e4], dtype=f32))
        self.prior = None

    def predict(self, ts: int) -> Bbox:
        if self.x[6] + self.x[2] <= 0:
            self.x[6] = 0
        xp = (F @ self.x).astype(f32)
        Pp = ((F @ self.P).astype(f32) @ F.T + Q).astype(f32)
        self.prior = (xp, Pp)
        b = from_x(xp)
        b.track_id, b.timestamp = self.id, ts
        self.age += 1
        self.time_since_update += 1
        self.history.append(b)
        return b

    def update(self, det: Bbox | None):
        if det is None:
            self.hit_streaks = 0
            return
        self.hits += 1
        self.hit_streaks += 1
        if self.hit_streaks >= 5:
            self.time_since_update = 0
            self.last_match = det.timestamp
        xp, Pp = self.prior
        z = into_z(det)
        S = (H @ Pp @ H.T + R).astype(f32)
        K = (Pp @ H.T @ np.linalg.inv(S.astype(np.float64)).astype(f32)).astype(f32)
        self.x = (xp + K @ (z - H @ xp)).astype(f32)
        A = (np.eye(7, dtype=f32) - K @ H).astype(f32)
        self.P = (A @ Pp @ A.T + K @ R @ K.T).astype(f32)
        self.history[-1].class_id = det.class_id
        self.history[-1].confidence = det.confidence

    def is_seen(self) -> bool:
        return any(self.start <= ts <= self.last_match for ts in self.seen_ts)

    def trim_dead_history(self):
        drop_idx = len(self.history) - self.time_since_update
        self.history = self.history[:drop_idx]


def linear_assignment(cost: np.ndarray):
    """cost [n_trk][n_det] f32 -> set of (i, j) edges (lib.rs:25-56)."""
    nt, nd = cost.shape
    n = max(nt, nd)
    sq = np.zeros((n, n), dtype=np.float64)
    sq[:nt, :nd] = cost
    rows, cols = linear_sum_assignment(sq)
    return sorted((int(i), int(j)) for i, j in zip(rows, cols) if i < nt and j < nd and cost[i, j] != f32(2.0))


class Sort:
    def __init__(self, max_age: int, min_hits: int, iou_threshold: float):
        self.max_age, self.min_hits, self.iou_threshold = max_age, min_hits, f32(iou_threshold)
        self.trackers: list[Tracker] = []
        self.frame_count = self.id_counter = 0

    def match_dets(self, preds, dets):
        if not preds or not dets:
            return []
        cost = np.zeros((len(preds), len(dets)), dtype=f32)
        for i, p in enumerate(preds):
            wgt = f32(1.0) if self.trackers[i].active else f32(2.0)
            for j, d in enumerate(dets):
                cost[i, j] = f32(-d.iou(p) + wgt)
        out = []
        for i, j in linear_assignment(cost):
            thr = f32(1.0) - self.iou_threshold if self.trackers[i].active else f32(2.0) - self.iou_threshold
            if cost[i, j] <= f32(thr):
                out.append((i, j))
        return out

    def update(self, dets: list[Bbox], pts: int):
        self.frame_count += 1
        preds = [t.predict(pts) for t in self.trackers]
        matches = self.match_dets(preds, dets)
        matched_d = {j for _, j in matches}
        unmatched = [j for j in range(len(dets)) if j not in matched_d]
        m = dict(matches)
        for i, t in enumerate(self.trackers):
            if i in m:
                dets[m[i]].timestamp = pts
                t.update(dets[m[i]])
            else:
                t.update(None)
        for t in self.trackers:
            if not t.active and t.hit_streaks >= self.min_hits:
                t.active = True
        dead, keep = [], []
        for t in self.trackers:
            if t.time_since_update <= self.max_age:
                keep.append(t)
            elif t.active:
                t.trim_dead_history()
                dead.append(t)
        self.trackers = keep
        for j in unmatched:
            self.trackers.append(Tracker(self.id_counter, dets[j], pts))
            self.id_counter += 1
        return dead

    def mark_seen(self, ts: int):
        for t in self.trackers:
            t.seen_ts.append(ts)

    def finalize(self):
        out = [t for t in self.trackers if t.active and len(t.history) > self.min_hits]
        self.trackers = [t for t in self.trackers if not t.active]
        return out


# ----------------------------------------------------------------------------- cova GoP filter
SECOND = 1_000_000_000
DELTA_UNIT, DISCONT, DROPPABLE = 1, 2, 4


@dataclass
class Gop:
    min: int
    max: int
    inl: list = field(default_factory=list)
    out: list = field(default_factory=list)
    finalized: bool = False


class GopFilter:
    def __init__(self, sort_iou=0.1, sort_maxage=30, sort_minhits=30, alpha=0, beta=0, infer_i=False):
        self.cfg = dict(iou=sort_iou, maxage=sort_maxage, minhits=sort_minhits, alpha=alpha, beta=beta, infer_i=infer_i)
        self.bufs: list[Gop] = []
        self.sort = None
        self.dropped = self.decoded_dependency = self.decoded_inference = 0
        self.pushed: list[list] = []   # BufferLists pushed downstream: lists of (id, pts, flags)

    def push_enc(self, au_id, pts, flags):
        if not flags & DELTA_UNIT:
            if self.bufs:
                self.bufs[-1].finalized = True
            self.bufs.append(Gop(pts, pts, [[au_id, pts, flags | DISCONT]]))
        else:
            g = self.bufs[-1]
            if pts < g.min:
                g.min = pts
            elif pts > g.max:
                g.max = pts
            g.inl.append([au_id, pts, flags])

    def push_boxes(self, boxes: list[Bbox], pts: int):
        c = self.cfg
        if self.sort is None:
            self.sort = Sort(c["maxage"], c["minhits"], c["iou"])
        dead = self.sort.update(boxes, pts)
        min_required = None
        if dead:
            min_required = 0
            for t in dead:
                if not t.is_seen():
                    min_required = max(min_required, t.start)
        clk = SECOND // 30
        maxage_pts = clk * (c["maxage"] + 10)
        max_track = pts - maxage_pts if pts >= maxage_pts else 0
        if min_required is not None:
            mt = min_required
            inferenced = dd = di = 0
            for g in reversed(self.bufs):
                if not (mt <= g.max and g.min <= max_track):
                    continue
                hit = False
                for b in g.out:
                    if mt < b[1]:
                        inferenced += 1
                        hit = True
                        break
                if hit:
                    continue
                while g.inl:
                    b = g.inl.pop(0)
                    if inferenced > 0:
                        break  # popped AU is lost (reference behaviour)
                    if mt <= b[1]:
                        self.sort.mark_seen(b[1])
                        di += 1
                        g.out.append(b)
                        inferenced += 1
                        break
                    b[2] |= DROPPABLE
                    dd += 1
                    g.out.append(b)
            if inferenced < c["beta"]:
                for g in reversed(self.bufs):
                    if not (mt <= g.max and g.min <= max_track) or not g.out:
                        continue
                    extra_decode = min(len(g.inl), c["alpha"])
                    extra_infer = min(extra_decode, c["beta"] - inferenced)
                    if extra_decode == 0 or extra_infer == 0:
                        continue
                    step, rem = divmod(extra_decode, extra_infer)
                    for _ in range(rem):
                        b = g.inl.pop(0); b[2] |= DROPPABLE; dd += 1; g.out.append(b)
                    for _ in range(extra_infer):
                        for _ in range(max(step - 1, 0)):
                            b = g.inl.pop(0); b[2] |= DROPPABLE; dd += 1; g.out.append(b)
                        b = g.inl.pop(0)
                        self.sort.mark_seen(b[1]); di += 1; g.out.append(b); inferenced += 1
            assert inferenced > 0
            self.decoded_inference += di
            self.decoded_dependency += dd
        dropped = di = 0
        gop_pts = clk * 250
        droppable = pts - gop_pts if pts >= gop_pts else 0
        keep = []
        for g in self.bufs:
            if not (g.finalized and g.max <= droppable):
                keep.append(g)
                continue
            if c["infer_i"] and g.inl:
                b = g.inl.pop(0)
                if not b[2] & DELTA_UNIT:
                    di += 1
                    g.out.append(b)
                else:
                    dropped += 1
            if g.out:
                self.pushed.append([tuple(b) for b in g.out])
            dropped += len(g.inl)
        self.bufs = keep
        self.decoded_inference += di
        self.dropped += dropped

    def eos(self):
        for g in self.bufs:
            self.dropped += len(g.inl)
            if g.out:
                self.pushed.append([tuple(b) for b in g.out])
        self.bufs = []
