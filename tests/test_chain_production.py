"""bboxcc -> sorttracker / cova at the EXPERIMENT's parameters: cc-threshold 1 (experiment/cova/config.yaml:59),
sort maxage 60 / minhits 30 / iou 0.1 (config.yaml:67, experiment/cova/launch.py:43-44), on blob-like masks -- what a
trained BlobNet emits -- whose objects move fast enough that the assignment problem has no exact ties that matter (an
object whose integer box repeats makes a young, inactive tracker predict its detection exactly: cost 2 - 1 = 1, the cost
of every non-overlapping active tracker; duplicate inactive trackers tie with each other; which optimum a Kuhn-Munkres
implementation then returns is not defined, and the crate the reference pins, linear_assignment @a992de6, is not in the
reference tree -- INTEGRATION.md, "Assignment ties").  Every frame is checked for such a tie before it is compared.
The C++ ports behind the C-ABI against oracle/sort_ref.py (numpy f32 Kalman + scipy's assignment) frame by frame."""
import numpy as np
import pytest

from cova_amd import _lib as L
from cova_amd import elements as E
from oracle import ref
from oracle import sort_ref as R

CLK = 1_000_000_000 // 30
H, W = 45, 80
GOP = 250
SORT = dict(sort_iou=0.1, sort_maxage=60, sort_minhits=30)
CC_THRESHOLD = 1


def moving_blob_masks(n, seed, h=H, w=W, n_objects=40):
    """u8 [n][h][w]: ellipses of 2.5 .. 6 macroblocks radius that enter at a border and cross the grid at 0.8 .. 1.6
    macroblocks per frame with a slowly changing size: the integer box of an object changes every frame (a slower object's
    box repeats for several frames, which is exactly the stationary tie of the module comment); about three are visible at
    a time, each for 40 .. 100 frames."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    objs = []
    for _ in range(n_objects):
        a = int(rng.integers(0, max(1, n - 40)))
        ang, speed = rng.uniform(0, 2 * np.pi), rng.uniform(0.8, 1.6)
        vy, vx = speed * np.sin(ang), speed * np.cos(ang)
        cy, cx = rng.uniform(5, h - 5), rng.uniform(5, w - 5)
        if abs(vx) > abs(vy):
            cx = 2.0 if vx > 0 else w - 3.0
        else:
            cy = 2.0 if vy > 0 else h - 3.0
        objs.append(dict(a=a, cy=cy, cx=cx, vy=vy, vx=vx, ry=rng.uniform(2.5, 6), rx=rng.uniform(2.5, 6), g=rng.uniform(-0.004, 0.004)))
    m = np.zeros((n, h, w), np.uint8)
    for i in range(n):
        for o in objs:
            k = i - o["a"]
            if k >= 0:
                ry, rx = o["ry"] * (1 + o["g"] * k), o["rx"] * (1 + o["g"] * k)
                m[i] |= ((((yy - o["cy"] - o["vy"] * k) / ry) ** 2 + ((xx - o["cx"] - o["vx"] * k) / rx) ** 2) <= 1).astype(np.uint8)
    return m


def effective_tie(r_sort, dets, pts):
    """True when the product's solver and scipy's pick different optima of this frame's assignment problem AND the difference
    survives SORT's thresholds (lib.rs:117-127), i.e. the frame's outcome depends on how ties are broken."""
    if not r_sort.trackers or not dets:
        return False
    f32 = np.float32
    preds = []
    for t in r_sort.trackers:          # Tracker.predict without its side effects
        x = t.x.copy()
        if x[6] + x[2] <= 0:
            x[6] = 0
        preds.append(R.from_x((R.F @ x).astype(f32)))
    rc = r_sort
    cost = np.zeros((len(preds), len(dets)), dtype=f32)
    for a, p in enumerate(preds):
        wgt = f32(1.0) if rc.trackers[a].active else f32(2.0)
        for j, d in enumerate(dets):
            cost[a, j] = f32(-d.iou(p) + wgt)

    def kept(edges):
        return sorted((a, j) for a, j in edges
                      if cost[a, j] <= (f32(1.0) if rc.trackers[a].active else f32(2.0)) - f32(SORT["sort_iou"]))
    return kept(E.linear_assignment(cost)) != kept(R.linear_assignment(cost))


def _to_box_dtype(rb):
    bx = np.zeros(len(rb), dtype=L.BOX_DTYPE)
    for f, g in (("left", "left"), ("top", "top"), ("width", "width"), ("height", "height"), ("area_px", "area")):
        bx[f] = rb[g]
    return bx


def run_stream(boxes_per_frame, n_frames):
    """One stream through sorttracker and cova (C++ behind the C-ABI) and through the oracle, compared frame by frame.
    boxes_per_frame: list of BOX_DTYPE arrays (bboxcc output).  Returns (tracks finished, frames forwarded for inference)."""
    st = E.SortTracker(iou_threshold=SORT["sort_iou"], maxage=SORT["sort_maxage"], minhits=SORT["sort_minhits"])
    st.set_caps(W, H)
    cv = E.Cova(**SORT)
    r_sort = R.Sort(SORT["sort_maxage"], SORT["sort_minhits"], SORT["sort_iou"])
    r_cova = R.GopFilter(**SORT)
    for i in range(n_frames):
        cv.sink_enc_chain(i, i * CLK, delta_unit=(i % GOP != 0))
        r_cova.push_enc(i, i * CLK, 0 if i % GOP == 0 else R.DELTA_UNIT)
    forwarded, n_tracks = [], 0
    for i in range(n_frames):
        bx = boxes_per_frame[i]
        wire = E.serialize_vec(E.boxes_to_bbox(bx))
        dets = [R.Bbox(float(b["left"]), float(b["top"]), float(b["width"]), float(b["height"])) for b in bx]
        if effective_tie(r_sort, dets, i * CLK):
            pytest.fail(f"frame {i}: this stream has an assignment tie that matters -- pick another seed (module comment)")
        dead = E.deserialize_vec(st.transform(wire, i * CLK))
        exp_tracks = r_sort.update(dets, i * CLK)
        exp = [b for t in exp_tracks for b in t.history]
        n_tracks += len(exp_tracks)
        assert len(dead) == len(exp), f"frame {i}: {len(dead)} boxes of dead tracks, oracle {len(exp)}"
        for g, e in zip(dead, exp):
            assert int(g["track_id"]) == e.track_id and int(g["timestamp"]) == e.timestamp, f"frame {i}"
            np.testing.assert_allclose([g["left"], g["top"], g["width"], g["height"]], [e.left, e.top, e.width, e.height],
                                       rtol=2e-3, atol=2e-3)
        assert st.sort.num_trackers() == len(r_sort.trackers), f"frame {i}: tracker count"
        forwarded.extend(cv.sink_mask_chain(wire, i * CLK))
        r_cova.push_boxes(dets, i * CLK)
    fin = E.deserialize_vec(st.sink_event_eos())
    fin_exp = r_sort.finalize()
    assert len(fin) == sum(len(t.history) for t in fin_exp)
    n_tracks += len(fin_exp)
    assert cv.eos("sink_enc") is None
    forwarded.extend(cv.eos("sink_mask"))
    r_cova.eos()
    assert (cv.dropped, cv.decoded_dependency, cv.decoded_inference) == (r_cova.dropped, r_cova.decoded_dependency, r_cova.decoded_inference)
    assert [(int(a["id"]), int(a["pts"]), int(a["flags"])) for a in forwarded] == [tuple(b) for lst in r_cova.pushed for b in lst]
    inferred = cv.decoded_inference
    cv.close()
    return n_tracks, inferred


# Such ties are the rule, not the exception, at these parameters: of the seeds 31 .. 89 only 37, 39, 49, 50, 52, 57, 64, 66 ..
# 69, 71, 75, 81, 85, 86 give 600 frames without one (a box that repeats while an object leaves the grid, twin trackers
# born from two fragments of one object).
@pytest.mark.parametrize("seed", [37, 39, 49])
def test_production_parameters_host_chain(seed):
    """CPU: regionprops = the oracle (bit-identical to the HIP kernel, tests/test_gpu_bboxcc.py)."""
    n = 600
    masks = moving_blob_masks(n, seed)
    boxes = []
    for i in range(n):
        rb, cnt = ref.regionprops(masks[i], CC_THRESHOLD, 256)
        boxes.append(_to_box_dtype(rb[:cnt]))
    assert 2 < np.mean([len(b) for b in boxes]) < 5
    tracks, inferred = run_stream(boxes, n)
    assert tracks >= 5 and inferred >= 2


@pytest.mark.gpu
def test_production_parameters_gpu_chain(ctx):
    """GPU: four streams multiplexed through the HIP bboxcc in batches of 64 (nvstreammux order), demultiplexed into
    per-stream sorttracker / cova."""
    n, n_streams = 600, 4
    masks = [moving_blob_masks(n, (50, 52, 57, 64)[s]) for s in range(n_streams)]
    mux = np.stack([masks[s][i] for i in range(n) for s in range(n_streams)])
    cc = E.BboxCc(ctx, CC_THRESHOLD, 256)
    boxes = np.zeros((len(mux), 256), dtype=L.BOX_DTYPE)
    counts = np.zeros(len(mux), dtype=np.int32)
    for b0 in range(0, len(mux), 64):
        boxes[b0:b0 + 64], counts[b0:b0 + 64] = cc.regionprops(mux[b0:b0 + 64])
    rb, rc = ref.regionprops_batch(mux, CC_THRESHOLD, 256)
    np.testing.assert_array_equal(counts, rc)
    total_tracks = 0
    for s in range(n_streams):
        per_frame = []
        for i in range(n):
            j = i * n_streams + s
            k = int(counts[j])
            assert boxes[j, :k].tobytes() == _to_box_dtype(rb[j, :k]).tobytes()
            per_frame.append(boxes[j, :k].copy())
        tracks, inferred = run_stream(per_frame, n)
        total_tracks += tracks
    assert total_tracks >= 20
