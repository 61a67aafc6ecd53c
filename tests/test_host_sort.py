"""SORT (C++ behind the C-ABI) vs the reference's KATs and the numpy restatement."""
import json
import os

import numpy as np
import pytest
from scipy.optimize import linear_sum_assignment

from cova_amd import _lib as L
from cova_amd import elements as E
from oracle import sort_ref as R

KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))


def _bb(rows):
    out = np.zeros(len(rows), dtype=L.BBOX_DTYPE)
    for i, r in enumerate(rows):
        out[i] = E.make_bbox(*r)[0]
    return out


def test_linear_assignment_reference_kats():
    for c in KATS["linear_assignment"]["cases"]:
        cm = np.array(c["colmajor"], dtype=np.float32).reshape(c["cols"], c["rows"]).T + np.float32(c["offset"])
        assert E.linear_assignment(cm) == sorted(map(tuple, c["expected"]))
        assert R.linear_assignment(cm) == sorted(map(tuple, c["expected"]))


def test_linear_assignment_cost_optimal_random():
    rng = np.random.default_rng(1)
    for _ in range(50):
        nr, nc = rng.integers(1, 12, 2)
        cost = (1.0 - rng.random((nr, nc))).astype(np.float32)
        edges = E.linear_assignment(cost)
        assert len(edges) == min(nr, nc)
        n = max(nr, nc)
        sq = np.zeros((n, n)); sq[:nr, :nc] = cost
        r, c = linear_sum_assignment(sq)
        assert abs(sum(cost[i, j] for i, j in edges) - sq[r, c].sum()) < 1e-5


def test_iou_matrix_kat():
    k = KATS["iou_matrix"]
    dets, preds = _bb(k["dets"]), _bb(k["preds"])
    got = [[-E.iou(dets[j:j + 1], preds[i:i + 1]) for j in range(len(dets))] for i in range(len(preds))]
    assert got == k["expected_rows_preds_cols_dets"]


def test_new_sort_kat():
    d = KATS["sort_default"]
    s = E._SortHandle(d["max_age"], d["min_hits"], d["iou_threshold"])
    dets = _bb(KATS["new_sort"]["dets"])
    dead, lens = s.update(dets, 0)
    assert len(dead) == 0 and s.num_trackers() == KATS["new_sort"]["expect_trackers"]
    for i in range(2):
        st = s.tracker_info(i)["state"]
        for k in ("left", "top", "width", "height", "area"):
            assert st[k] == dets[i][k]


def test_observation_model_kat():
    """cova-rs/sort/src/lib.rs:250-274: two fresh trackers, predict(0) on each: history.last() (ids and timestamps cleared)
    equals the detection exactly -- through the C-ABI (covahip_sort_tracker_predict)."""
    d, k = KATS["sort_default"], KATS["observation_model"]
    s = E._SortHandle(d["max_age"], d["min_hits"], d["iou_threshold"])
    dets = _bb(k["dets"])
    s.update(dets, 0)
    assert s.num_trackers() == k["expect_trackers"]
    for i in range(s.num_trackers()):
        last = s.tracker_predict(i, k["predict_ts"])
        assert last["has_track_id"] == 1 and last["has_timestamp"] == 1 and last["timestamp"] == k["predict_ts"]
        for f in ("left", "top", "width", "height", "area"):
            assert last[f] == dets[i][f]                       # exact equality, as the reference asserts
        assert last["has_class_id"] == 0 and last["has_confidence"] == 0
    assert s.num_trackers() == k["expect_trackers"]


def test_create_tracker_kat():
    """cova-rs/sort/src/tracker/mod.rs:154-165: new -> predict(0) -> update(Some(next)) (the reference's smoke test)."""
    k = KATS["create_tracker"]
    s = E._SortHandle(3, 3, 0.2)
    s.update(_bb([k["bbox"]]), 0)                              # KalmanBoxTracker::new(0, &bbox, 0) (Sort creates it)
    assert s.num_trackers() == 1 and s.tracker_info(0)["id"] == 0
    prior = s.tracker_predict(0, 0)
    for f, v in zip(("left", "top", "width", "height"), k["bbox"]):
        assert prior[f] == v
    s.tracker_update(0, _bb([k["next_bbox"]])[0])
    info = s.tracker_info(0)
    assert info["hit_streaks"] == 1 and np.isfinite([info["state"][f] for f in ("left", "top", "width", "height")]).all()
    s.tracker_update(0, None)                                  # update(None): the streak ends
    assert s.tracker_info(0)["hit_streaks"] == 0


def test_match_dets_kat():
    """lib.rs:384-407: after the second frame only tracker 1 is matched (to detection 0), so
    tracker 0 has hit_streaks 0, tracker 1 has 1, and two new trackers are born."""
    d = KATS["sort_default"]
    k = KATS["match_dets"]
    s = E._SortHandle(d["max_age"], d["min_hits"], d["iou_threshold"])
    s.update(_bb(k["first_dets"]), 0)
    s.update(_bb(k["second_dets"]), 1)
    assert k["expected_matches"] == [[1, 0]]
    assert s.num_trackers() == 2 + 2
    assert s.tracker_info(0)["hit_streaks"] == 0
    assert s.tracker_info(1)["hit_streaks"] == 1
    r = R.Sort(d["max_age"], d["min_hits"], d["iou_threshold"])
    r.update([R.Bbox(*b) for b in k["first_dets"]], 0)
    preds = [t.predict(0) for t in r.trackers]
    assert r.match_dets(preds, [R.Bbox(*b) for b in k["second_dets"]]) == [(1, 0)]


def _scripted_sequence(n_frames, seed):
    """Two objects crossing + a third appearing late + random clutter."""
    rng = np.random.default_rng(seed)
    frames = []
    for i in range(n_frames):
        dets = []
        dets.append((5 + 1.5 * i, 10 + 0.5 * i, 8 + 0.05 * i, 6))
        dets.append((70 - 1.2 * i, 12 + 0.4 * i, 7, 9))
        if 20 <= i < 70:
            dets.append((30 + 0.3 * (i - 20), 40 - 0.5 * (i - 20), 5, 5))
        if i % 7 == 3:
            dets.append((float(rng.uniform(0, 100)), float(rng.uniform(0, 60)), 3, 3))
        if 40 <= i < 46:                       # object 0 missed for a few frames
            dets.pop(0)
        jit = rng.normal(0, 0.15, (len(dets), 4))
        frames.append([tuple(np.float32(v + j) for v, j in zip(d, jj)) for d, jj in zip(dets, jit)])
    return frames


@pytest.mark.parametrize("params", [(3, 3, 0.2), (10, 5, 0.1), (60, 30, 0.1)])
def test_sequence_matches_numpy_restatement(params):
    max_age, min_hits, iou = params
    frames = _scripted_sequence(120, seed=max_age)
    s = E._SortHandle(max_age, min_hits, iou)
    r = R.Sort(max_age, min_hits, iou)
    n_dead_total = 0
    for i, dets in enumerate(frames):
        pts = i * 33_333_333
        dead, lens = s.update(_bb(dets), pts)
        rdead = r.update([R.Bbox(*d) for d in dets], pts)
        assert [len(t.history) for t in rdead] == list(lens)
        flat = [b for t in rdead for b in t.history]
        assert len(flat) == len(dead)
        for g, e in zip(dead, flat):
            assert g["track_id"] == e.track_id and g["timestamp"] == e.timestamp
            for k in ("left", "top", "width", "height"):
                assert abs(float(g[k]) - float(getattr(e, k))) <= 1e-3 * max(1.0, abs(float(getattr(e, k))))
        n_dead_total += len(rdead)
        assert s.num_trackers() == len(r.trackers)
        for j, t in enumerate(r.trackers):
            info = s.tracker_info(j)
            assert (info["id"], info["active"], info["hit_streaks"], info["time_since_update"]) == \
                (t.id, t.active, t.hit_streaks, t.time_since_update)
    fin, lens = s.finalize()
    rfin = r.finalize()
    assert [len(t.history) for t in rfin] == list(lens)
    assert n_dead_total + len(rfin) >= 1


def test_zero_iou_threshold_solves_the_full_problem():
    """at iou_threshold <= 0 an edge of IoU 0 passes the filter (cost 1 <= 1 - 0), so a detection that overlaps
    nothing is MATCHED to an idle tracker instead of starting a new one (lib.rs:108-131) -- the port's idle-tracker pruning must
    not run there.  One active idle tracker + one inactive idle one: the optimum is unique (cost 1 vs 2), so the numpy
    restatement (scipy) and the port must agree on which tracker takes the detection."""
    for iou in (0.0, -0.25):
        s = E._SortHandle(10, 2, iou)
        r = R.Sort(10, 2, iou)
        seq = [[(10, 10, 6, 6)], [(10.5, 10, 6, 6)], [(11, 10, 6, 6)],          # tracker 0 becomes active
               [(11.5, 10, 6, 6), (60, 40, 5, 5)],                              # tracker 1 is born (inactive)
               [(90, 5, 4, 4)],                                                 # overlaps neither: goes to the ACTIVE idle tracker
               [(90.5, 5, 4, 4), (30, 30, 3, 3), (31, 50, 3, 3)]]               # more detections than trackers
        for i, dets in enumerate(seq):
            pts = i * 33_333_333
            dead, lens = s.update(_bb(dets), pts)
            rdead = r.update([R.Bbox(*d) for d in dets], pts)
            assert [len(t.history) for t in rdead] == list(lens)
            assert s.num_trackers() == len(r.trackers), (iou, i)
            for j, t in enumerate(r.trackers):
                info = s.tracker_info(j)
                assert (info["id"], info["active"], info["hit_streaks"], info["time_since_update"]) == \
                    (t.id, t.active, t.hit_streaks, t.time_since_update), (iou, i, j)
            if i == 4:
                assert s.num_trackers() == 2 and s.tracker_info(0)["hit_streaks"] == 4   # matched at IoU 0: no tracker was born
        assert s.num_trackers() == 4        # (an edge of cost exactly 2.0 -- inactive tracker, IoU 0 -- is dropped by the `!= 2.0` filter)


def test_duplicate_detections_match_numpy_restatement():
    """Two identical detections on one tracker, and a duplicate pair that overlaps nothing: one of a pair matches / both start
    trackers, the same way in the port (pruned problem, iou_threshold > 0) and in the restatement."""
    s = E._SortHandle(10, 2, 0.1)
    r = R.Sort(10, 2, 0.1)
    seq = [[(10, 10, 6, 6)], [(10.2, 10, 6, 6)], [(10.4, 10, 6, 6), (10.4, 10, 6, 6)],
           [(10.6, 10, 6, 6), (50, 50, 4, 4), (50, 50, 4, 4)], [(10.8, 10, 6, 6), (50, 50, 4, 4)]]
    for i, dets in enumerate(seq):
        pts = i * 33_333_333
        dead, lens = s.update(_bb(dets), pts)
        rdead = r.update([R.Bbox(*d) for d in dets], pts)
        assert [len(t.history) for t in rdead] == list(lens)
        assert s.num_trackers() == len(r.trackers), i
        got = sorted((s.tracker_info(j)["hit_streaks"], s.tracker_info(j)["time_since_update"]) for j in range(s.num_trackers()))
        exp = sorted((t.hit_streaks, t.time_since_update) for t in r.trackers)
        assert got == exp, i


def test_updates_after_finalize_match_numpy_restatement():
    """finalize() (lib.rs:207-213) takes the active trackers out and keeps the others: the updates that follow run on the
    reordered set (the port keeps its prediction arrays in tracker order and histories partly unwritten -- both must survive it)."""
    max_age, min_hits, iou = 10, 5, 0.1
    frames = _scripted_sequence(120, seed=4)
    s = E._SortHandle(max_age, min_hits, iou)
    r = R.Sort(max_age, min_hits, iou)
    for i, dets in enumerate(frames):
        pts = i * 33_333_333
        dead, lens = s.update(_bb(dets), pts)
        rdead = r.update([R.Bbox(*d) for d in dets], pts)
        assert [len(t.history) for t in rdead] == list(lens)
        assert s.num_trackers() == len(r.trackers)
        if i in (50, 51, 90):
            fin, flens = s.finalize()
            rfin = r.finalize()
            assert [len(t.history) for t in rfin] == list(flens)
            flat = [b for t in rfin for b in t.history]
            assert [(g["track_id"], g["timestamp"]) for g in fin] == [(e.track_id, e.timestamp) for e in flat]
            assert s.num_trackers() == len(r.trackers)
        for j, t in enumerate(r.trackers):
            info = s.tracker_info(j)
            assert (info["id"], info["active"], info["hit_streaks"], info["time_since_update"]) == \
                (t.id, t.active, t.hit_streaks, t.time_since_update)


def test_young_track_ages_while_matched_quirk():
    """tracker/mod.rs:77-80: time_since_update only resets once hit_streaks >= 5."""
    s = E._SortHandle(30, 3, 0.1)
    for i in range(4):
        s.update(_bb([(10, 10, 5, 5)]), i)
    info = s.tracker_info(0)
    assert info["hit_streaks"] == 3 and info["time_since_update"] == 3 and info["active"]
    for i in range(4, 6):
        s.update(_bb([(10, 10, 5, 5)]), i)
    assert s.tracker_info(0)["time_since_update"] == 0


def test_from_x_top_uses_width_quirk():
    """state.rs:26: top = y - width/2 (not height/2)."""
    s = E._SortHandle(30, 3, 0.1)
    s.update(_bb([(10, 20, 8, 2)]), 0)
    st = s.tracker_info(0)["state"]
    assert float(st["left"]) == pytest.approx(10.0, abs=1e-5)
    assert float(st["top"]) == pytest.approx(21.0 - 4.0, abs=1e-5)   # cy = 21, width/2 = 4


def test_sorttracker_element_bincode_io():
    st = E.SortTracker(iou_threshold=0.1, maxage=10, minhits=3)
    st.set_caps(80, 45)
    outs = []
    seq = [[(10, 10, 5, 5)]] * 8 + [[]] * 14
    for i, dets in enumerate(seq):
        outs.append(E.deserialize_vec(st.transform(E.serialize_vec(_bb(dets)), i)))
    dead = [o for o in outs if len(o)]
    assert len(dead) == 1
    assert all(b["has_track_id"] and b["has_timestamp"] for b in dead[0])
    # predicted boxes of frames 1..7 survive trim_dead_history (the 11 unmatched predictions are cut)
    assert [int(b["timestamp"]) for b in dead[0]] == list(range(1, 8))
    assert len(E.deserialize_vec(st.sink_event_eos())) == 0
