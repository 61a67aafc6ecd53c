"""The GStreamer elements of gst/libgstcova.so (reference names / pads / properties) driven buffer by
buffer through gst/gst_element_driver and compared with the oracle restatements."""
import json
import os
import struct
import subprocess

import numpy as np
import pytest

from cova_amd import _lib as L
from cova_amd import elements as E
from oracle import ref
from oracle import sort_ref as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GST = os.path.join(ROOT, "gst")
DRIVER = os.path.join(GST, "gst_element_driver")
CONDA = "/opt/conda"
CLK = 1_000_000_000 // 30

pytestmark = pytest.mark.skipif(
    not (os.path.exists(os.path.join(CONDA, "lib", "libgstreamer-1.0.so")) and os.path.exists(DRIVER)
         and os.path.exists(os.path.join(GST, "libgstcova.so"))),
    reason="GStreamer 1.x under /opt/conda or the built plugin is missing")


def _env(tmp):
    env = dict(os.environ)
    env.update({
        "GST_PLUGIN_PATH": GST, "GST_PLUGIN_SYSTEM_PATH": os.path.join(CONDA, "lib", "gstreamer-1.0"),
        "LD_LIBRARY_PATH": os.path.join(CONDA, "lib"), "GST_REGISTRY": str(tmp / "registry.bin"),
        # conda ships an older libstdc++ than the one libcovahip.so was built against
        "LD_PRELOAD": "/usr/lib/x86_64-linux-gnu/libstdc++.so.6", "GST_DEBUG": "1",
    })
    return env


def _write(path, recs):
    with open(path, "wb") as f:
        for kind, pts, flags, payload in recs:
            f.write(struct.pack("<BQII", ord(kind), pts, flags, len(payload)) + payload)


def _read(path):
    out = []
    data = open(path, "rb").read()
    off = 0
    while off < len(data):
        kind, pts, flags, n = struct.unpack_from("<BQII", data, off)
        off += 17
        out.append((chr(kind), pts, flags, data[off:off + n]))
        off += n
    return out


def _run(args, tmp):
    r = subprocess.run([DRIVER] + args, env=_env(tmp), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_inspect_lists_reference_elements_and_properties(tmp_path):
    insp = os.path.join(CONDA, "bin", "gst-inspect-1.0")
    want = {"metapreprocess": ["timestep", "gamma"], "bboxcc": ["cc-threshold"],
            "sorttracker": ["iou-threshold", "maxage", "minhits"],
            "cova": ["sort-iou", "sort-maxage", "sort-minhits", "port", "infer-i", "debug", "alpha", "beta", "dropped",
                     "decoded-dependency", "decoded-inference"],
            "blobnetinfer": ["model-weights-file", "gpu-id"], "bboxsink": ["location"],
            "blobnetfilter": ["model-weights-file", "gpu-id", "batch-size", "batched-push-timeout", "cc-threshold"],
            "maskcopy": ["unique-id", "gpu-id", "timestep"],
            "tfrecordsink": ["location", "gt", "gop"], "h264entropydec": ["max-threads", "records"]}
    for el, props in want.items():
        r = subprocess.run([insp, el], env=_env(tmp_path), capture_output=True, text=True, timeout=60)
        assert r.returncode == 0, r.stdout + r.stderr
        for p in props:
            assert f"  {p} " in r.stdout or f"  {p}:" in r.stdout, (el, p)
    r = subprocess.run([insp, "cova"], env=_env(tmp_path), capture_output=True, text=True)
    for pad in ("sink_mask", "sink_enc", "src"):
        assert f"'{pad}'" in r.stdout


@pytest.mark.parametrize("t,gamma", [(4, 1), (4, 2), (1, 1)])
def test_metapreprocess_element(tmp_path, t, gamma):
    w, h, n = 320, 240, 9                    # 20 x 15 macroblocks
    spb = (w // 16) * (h // 16) * 4
    rng = np.random.default_rng(t * 7 + gamma)
    frames = rng.integers(0, 256, (n, w * h * 3 // 2), dtype=np.uint8)
    _write(tmp_path / "in.rec", [("B", i * CLK, 0, frames[i].tobytes()) for i in range(n)])
    info = _run(["harness", f"metapreprocess timestep={t} gamma={gamma}",
                 f"video/x-raw,format=I420,width={w},height={h},framerate=30/1", str(tmp_path / "in.rec"),
                 str(tmp_path / "out.rec")], tmp_path)
    exp, idx = ref.metapreprocess(frames, spb, t, gamma)
    outs = _read(tmp_path / "out.rec")
    assert info["pulled"] == len(outs) == len(exp)
    assert f"width=(int){w // 16}" in info["out_caps"] and f"height=(int){h // 16 * t}" in info["out_caps"]
    assert "RGBA" in info["out_caps"]
    for k, (kind, pts, flags, payload) in enumerate(outs):
        assert pts == int(idx[k]) * CLK                      # output inherits the PTS of the current frame
        assert payload == exp[k].tobytes()


def _bb(rows):
    out = np.zeros(len(rows), dtype=L.BBOX_DTYPE)
    for i, r in enumerate(rows):
        out[i] = E.make_bbox(*r)[0]
    return out


def test_sorttracker_element(tmp_path):
    seq = [[(10 + 0.5 * i, 10, 5, 5), (40, 20 + 0.3 * i, 6, 4)] for i in range(30)] + [[]] * 15
    _write(tmp_path / "in.rec", [("B", i * CLK, 0, E.serialize_vec(_bb(d))) for i, d in enumerate(seq)])
    info = _run(["harness", "sorttracker maxage=10 minhits=5 iou-threshold=0.1", "bbox,width=80,height=45",
                 str(tmp_path / "in.rec"), str(tmp_path / "out.rec")], tmp_path)
    outs = _read(tmp_path / "out.rec")
    assert info["pulled"] == len(seq) + 1                    # one buffer per input + the EOS finalize() buffer
    r = R.Sort(10, 5, 0.1)
    for i, d in enumerate(seq):
        dead = r.update([R.Bbox(*b) for b in d], i * CLK)
        got = E.deserialize_vec(outs[i][3])
        flat = [b for t in dead for b in t.history]
        assert len(got) == len(flat)
        for g, e in zip(got, flat):
            assert g["track_id"] == e.track_id and g["timestamp"] == e.timestamp
            assert abs(float(g["left"]) - float(e.left)) < 1e-3 * max(1, abs(float(e.left)))
    fin = E.deserialize_vec(outs[-1][3])
    assert len(fin) == sum(len(t.history) for t in r.finalize())
    assert sum(len(E.deserialize_vec(o[3])) for o in outs) > 0


@pytest.mark.parametrize("props,kw", [("sort-maxage=10 sort-minhits=5", dict(sort_maxage=10, sort_minhits=5)),
                                      ("sort-maxage=10 sort-minhits=5 infer-i=true",
                                       dict(sort_maxage=10, sort_minhits=5, infer_i=True))])
def test_cova_element(tmp_path, props, kw):
    n, gop, lead = 700, 250, 280
    dets = [[(5 + 0.4 * (i - 10), 5 + 0.2 * (i - 10), 6, 6)] if 10 <= i <= 120 else
            ([(60 - 0.3 * (i - 200), 30, 8, 5)] if 200 <= i <= 420 else []) for i in range(n)]
    recs = []
    r = R.GopFilter(**kw)
    for i in range(n + lead):
        if i < n:
            recs.append(("E", i * CLK, 0 if i % gop == 0 else 1, struct.pack("<I", i)))
            r.push_enc(i, i * CLK, 0 if i % gop == 0 else R.DELTA_UNIT)
        j = i - lead
        if 0 <= j < n:
            recs.append(("M", j * CLK, 0, E.serialize_vec(_bb(dets[j]))))
            r.push_boxes([R.Bbox(*d) for d in dets[j]], j * CLK)
    recs += [("e", 0, 0, b""), ("m", 0, 0, b"")]
    r.eos()
    _write(tmp_path / "in.rec", recs)
    info = _run(["cova", props, str(tmp_path / "in.rec"), str(tmp_path / "out.rec")], tmp_path)
    assert (info["dropped"], info["decoded_dependency"], info["decoded_inference"]) == \
        (r.dropped, r.decoded_dependency, r.decoded_inference)
    assert info["eos"] == 1                                  # EOS forwarded once both sinks saw it
    assert info["held_end"] == 0
    outs = _read(tmp_path / "out.rec")
    lists, cur = [], None
    for kind, pts, flags, payload in outs:
        if kind == "L":
            cur = []
            lists.append(cur)
        else:
            au = struct.unpack("<I", payload)[0]
            cur.append((au, pts, bool(flags & 0x40), bool(flags & 0x1000)))  # GST_BUFFER_FLAG_DISCONT (64) / DROPPABLE (4096, cf. identity drop-buffer-flags=4096)
    lists = [l for l in lists if l]
    assert [[b[0] for b in l] for l in lists] == [[b[0] for b in l] for l in r.pushed]
    for got, exp in zip(lists, r.pushed):
        for (au, pts, discont, droppable), (eid, epts, eflags) in zip(got, exp):
            assert pts == epts and discont == bool(eflags & R.DISCONT) and droppable == bool(eflags & R.DROPPABLE)


def test_cova_element_releases_dropped_access_units(tmp_path):
    """Most access units are dropped by design; the element must release them when the filter discards them (the
    reference frees a dropped GoP's buffers, cova/imp.rs:268-305), not keep the whole bitstream until EOS."""
    n, gop, lead = 1500, 250, 30
    dets = [[(5 + 0.4 * (i - 10), 5 + 0.2 * (i - 10), 6, 6)] if 10 <= i <= 120 else
            ([(60 - 0.05 * (i - 900), 30, 8, 5)] if 900 <= i <= 1300 else []) for i in range(n)]
    recs = []
    for i in range(n + lead):
        if i < n:
            recs.append(("E", i * CLK, 0 if i % gop == 0 else 1, struct.pack("<I", i)))
        j = i - lead
        if 0 <= j < n:
            recs.append(("M", j * CLK, 0, E.serialize_vec(_bb(dets[j]))))
    recs += [("e", 0, 0, b""), ("m", 0, 0, b"")]
    _write(tmp_path / "in.rec", recs)
    info = _run(["cova", "sort-maxage=10 sort-minhits=5", str(tmp_path / "in.rec"), str(tmp_path / "out.rec")], tmp_path)
    assert info["eos"] == 1 and info["held_end"] == 0
    assert info["held_max"] <= 2 * gop + lead + 60 < n       # bounded by the GoP window, not by the stream length
    assert info["dropped"] > n // 2


@pytest.mark.gpu
def test_blobnet_bboxcc_pipeline(tmp_path, weights_flat):
    """blobnetinfer ! bboxcc on the GPU: bincode boxes identical to the C-ABI path on the same inputs."""
    from cova_amd import synth, weights as W
    from cova_amd.elements import BlobNetInfer, Context
    h, w, n = 45, 80, 3
    stack = synth.stacked_batch(n, h, w, seed=31)
    wpath = tmp_path / "weights.bin"
    wpath.write_bytes(W.to_bytes(weights_flat))
    _write(tmp_path / "in.rec", [("B", i * CLK, 0, stack[i].tobytes()) for i in range(n)])
    info = _run(["harness", f"blobnetinfer model-weights-file={wpath} ! bboxcc cc-threshold=1",
                 f"video/x-raw,format=RGBA,width={w},height={4 * h},framerate=30/1", str(tmp_path / "in.rec"),
                 str(tmp_path / "out.rec")], tmp_path)
    outs = _read(tmp_path / "out.rec")
    assert info["pulled"] == n and "bbox" in info["out_caps"]
    ctx = Context(0)
    net = BlobNetInfer(ctx, weights_flat, h, w, max_batch=n)
    boxes, counts, mask = net.filter(stack, cc_threshold=1, max_boxes=2048, want_mask=True)
    for i in range(n):
        exp = E.serialize_vec(E.boxes_to_bbox(boxes[i, :counts[i]]))
        assert outs[i][3] == exp and outs[i][1] == i * CLK
    ctx.close()


def test_bboxsink_element(tmp_path):
    """bboxsink (bboxsink/imp.rs:199-270): bincode Vec<Bbox> buffers -> one CSV file, header with the first record."""
    frames = [_bb([(1.5, 2.0, 3.0, 4.0), (10.0, 20.0, 5.0, 6.0)]), _bb([]), _bb([(0.25, 0.5, 100000.0, 2.0)])]
    frames[0]["track_id"], frames[0]["has_track_id"] = [7, 8], 1
    frames[0]["timestamp"], frames[0]["has_timestamp"] = 33333333, 1
    _write(tmp_path / "in.rec", [("B", i * CLK, 0, E.serialize_vec(f)) for i, f in enumerate(frames)])
    out = tmp_path / "boxes.csv"
    info = _run(["sink", f"bboxsink location={out}", "bbox,width=120,height=68", str(tmp_path / "in.rec")], tmp_path)
    assert info["pushed"] == 3
    exp = E.bbox_csv(frames[0], with_header=True) + E.bbox_csv(frames[2], with_header=False)
    assert out.read_text() == exp
    assert exp.splitlines()[1] == "1.5,2.0,3.0,4.0,12.0,7,33333333,,"


@pytest.mark.parametrize("gop", [0, 3])
def test_tfrecordsink_element(tmp_path, gop):
    """tfrecordsink (tfrecordsink/imp.rs:69-198,534-606): one Example per frame, or per GoP (closed by the next key
    frame, zero filled to `gop`; an open GoP at EOS is not written)."""
    from tests.test_host_formats import _parse_example, _masked
    w, h, n = 20, 15, 8
    rng = np.random.default_rng(gop)
    frames = rng.integers(0, 256, (n, h, w, 4), dtype=np.uint8)
    gt = rng.integers(0, 2, (n, h * w), dtype=np.uint8)
    (tmp_path / "gt.bin").write_bytes(gt.tobytes())
    key = [0, 2, 5]                                                     # key frames (not DELTA_UNIT)
    _write(tmp_path / "in.rec", [("B", i * CLK, 0 if i in key else 1, frames[i].tobytes()) for i in range(n)])
    out = tmp_path / "out.tfrecord"
    _run(["sink", f"tfrecordsink location={out} gt={tmp_path / 'gt.bin'} gop={gop}",
          f"video/x-raw,format=RGBA,width={w},height={h},framerate=30/1", str(tmp_path / "in.rec")], tmp_path)
    raw, off, recs = out.read_bytes(), 0, []
    while off < len(raw):
        (length,) = struct.unpack_from("<Q", raw, off)
        assert struct.unpack_from("<I", raw, off + 8)[0] == _masked(raw[off:off + 8])
        data = raw[off + 12:off + 12 + length]
        assert struct.unpack_from("<I", raw, off + 12 + length)[0] == _masked(data)
        recs.append(_parse_example(data))
        off += 16 + length
    groups = [[i] for i in range(n)] if gop == 0 else [[0, 1], [2, 3, 4]]   # frames 5..7 stay in the open GoP
    assert len(recs) == len(groups)
    for ex, g in zip(recs, groups):
        total = max(len(g), gop)
        for ch, name in enumerate(("mb_type", "mv_x", "mv_y")):
            assert len(ex[name]) == total
            for k, i in enumerate(g):
                assert ex[name][k] == frames[i, :, :, ch].tobytes()
            for k in range(len(g), total):
                assert ex[name][k] == bytes(w * h)
        for k, i in enumerate(g):
            assert ex["gt"][k] == gt[i].tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("batch_size,timeout_us,pause,records", [(64, 0, False, False), (32, 0, False, False), (64, 20000, True, False),
                                                                 (64, 0, False, True), (32, 0, False, True)])
def test_blobnetfilter_batching_element(tmp_path, weights_flat, batch_size, timeout_us, pause, records):
    """blobnetfilter (N request pads; stands for metapreprocess ! nvstreammux ! nvinfer ! nvstreamdemux ! maskcopy ! bboxcc):
    8 streams x 64 carrier frames -> per stream, per frame from the fourth on, the bincode boxes of the C-ABI path on the
    stacks metapreprocess would have built, with the frame's PTS; batches are formed across the streams.
    records (round 5): the sink caps are application/x-cova-records and every buffer is the frame's packed two-byte records
    (what `h264entropydec records=true` emits) -- byte for byte the same payloads as from the I420 carrier frames."""
    from cova_amd import synth, weights as W
    from cova_amd.elements import BlobNetInfer, Context
    h, w, n_streams, n = 45, 80, 8, 64
    carriers = [synth.carrier_frames(n, h, w, seed=700 + s, n_objects=5) for s in range(n_streams)]
    wpath = tmp_path / "weights.bin"
    wpath.write_bytes(W.to_bytes(weights_flat))
    recs = []
    for i in range(n):
        for s in range(n_streams):
            recs.append(("B", i * CLK, s << 8, E.pack_frames(carriers[s][i]).tobytes() if records else carriers[s][i].tobytes()))
        if pause and i == 12:
            recs.append(("s", 150, 0, b""))                  # 150 ms without input: the open batch (16 stacks) leaves by timeout
    recs += [("e", 0, s << 8, b"") for s in range(n_streams)]
    _write(tmp_path / "in.rec", recs)
    info = _run(["mux", f"blobnetfilter model-weights-file={wpath} batch-size={batch_size} batched-push-timeout={timeout_us} "
                 f"cc-threshold=4 max-boxes=512", str(n_streams),
                 f"application/x-cova-records,width-mbs={w},height-mbs={h},framerate=30/1" if records else
                 f"video/x-raw,format=I420,width={w * 16},height={h * 16},framerate=30/1",
                 str(tmp_path / "in.rec"), str(tmp_path / "out.rec")], tmp_path)
    n_out = n_streams * (n - 3)
    assert info["buffers"] == n_out and info["eos"] == n_streams
    full = -(-n_out // batch_size)
    assert info["batches"] == (full + 1 if pause else full)                  # the pause splits one batch in two
    outs = _read(tmp_path / "out.rec")
    per_stream = {s: [] for s in range(n_streams)}
    for kind, pts, pad, payload in outs:
        per_stream[pad].append((pts, payload))
    ctx = Context(0)
    net = BlobNetInfer(ctx, weights_flat, h, w, max_batch=n - 3)
    total = 0
    for s in range(n_streams):
        stack = np.stack([np.concatenate([carriers[s][i - k] for k in range(4)], axis=0) for i in range(3, n)])
        boxes, counts, mask = net.filter(stack, cc_threshold=4, max_boxes=512, want_mask=True)
        assert [p for p, _ in per_stream[s]] == [i * CLK for i in range(3, n)]        # in order, PTS of the current frame
        for j, (_, payload) in enumerate(per_stream[s]):
            assert payload == E.serialize_vec(E.boxes_to_bbox(boxes[j, :counts[j]]))
            total += int(counts[j])
        if s in (0, n_streams - 1):
            # ... and not only "what the C-ABI path gives": the oracle's regionprops on the HIP mask, serialised, is the payload
            rb, rc = ref.regionprops_batch(mask, 4, 512)
            for j, (_, payload) in enumerate(per_stream[s]):
                bx = np.zeros(int(rc[j]), dtype=L.BOX_DTYPE)
                for f, g in (("left", "left"), ("top", "top"), ("width", "width"), ("height", "height"), ("area_px", "area")):
                    bx[f] = rb[j, :rc[j]][g]
                assert payload == E.serialize_vec(E.boxes_to_bbox(bx))
    assert total > 0
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n_streams,batch_size,timeout_us", [(16, 48, 0), (24, 128, 300), (3, 16, 0)])
def test_blobnetfilter_concurrent_streams(tmp_path, weights_flat, n_streams, batch_size, timeout_us):
    """One streaming thread per sink pad, all pushing at once (the element's lock-free reservation, the flush hand-over
    and the ordered pusher threads under contention): every stream carries the same frames, so every src pad must
    deliver, in PTS order, byte for byte what a stream gets when it runs through the element alone."""
    from cova_amd import weights as W
    wpath = tmp_path / "weights.bin"
    wpath.write_bytes(W.to_bytes(weights_flat))
    n, warm = 900, 40

    def run(streams, bs, to):
        env = _env(tmp_path)
        env.update(MUXBENCH_SAME="1", MUXBENCH_WARM=str(warm))
        r = subprocess.run([DRIVER, "muxbench", f"blobnetfilter model-weights-file={wpath} batch-size={bs} "
                            f"batched-push-timeout={to} cc-threshold=2 max-boxes=1024", str(streams), "1280", "720", str(n)],
                           env=env, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr[-2000:]
        return json.loads(r.stdout.strip().splitlines()[-1])

    alone = run(1, 64, 0)
    assert alone["buffers_out"] == n + warm - 3 and alone["in_order"]
    many = run(n_streams, batch_size, timeout_us)
    assert many["buffers_out"] == n_streams * (n + warm - 3) and many["eos"] == n_streams
    assert many["in_order"] and many["pads_agree"]
    assert many["pad0_sum"] == alone["pad0_sum"]


DEMO = "/root/reference/demo/1m.mp4"


@pytest.mark.skipif(not os.path.exists(DEMO), reason="reference demo video not present (GPU box)")
def test_config_1_through_the_elements_entropy_decoder_metapreprocess_tfrecordsink(tmp_path):
    """BASELINE config 1 through the plugin: the access units of demo/1m.mp4 (decode order, as qtdemux ! h264parse would hand them
    over) -> h264entropydec -> metapreprocess timestep=4 -> tfrecordsink.  The decoder element reorders into output order; every
    Example's features are the record bytes of the current output frame (checked against the C-ABI front end on the file)."""
    import ctypes as C
    from tests.test_host_formats import _parse_example, _masked
    lib = L.lib()
    data = np.fromfile(DEMO, dtype=np.uint8)
    h = C.c_void_p()
    assert lib.covahip_h264_open_mp4(data.ctypes.data, data.size, C.byref(h)) == 0
    n_au = 300                                   # the first GoP and the start of the second: one IDR in the middle of the run
    raw = data.tobytes()
    at = raw.find(b"avcC")
    avcc = raw[at + 4:at - 4 + struct.unpack(">I", raw[at - 4:at])[0]]
    recs_in = []
    off, size, sync = C.c_uint64(), C.c_uint32(), C.c_int()
    for s in range(n_au):
        lib.covahip_h264_sample(h, s, C.byref(off), C.byref(size), C.byref(sync))
        recs_in.append(("B", s * CLK, 0 if sync.value else 1, raw[off.value:off.value + size.value]))
    _write(tmp_path / "in.rec", recs_in)
    caps = f"video/x-h264,stream-format=avc,alignment=au,framerate=30/1,codec_data=(buffer){avcc.hex()}"
    info = _run(["harness", "h264entropydec max-threads=1 ! metapreprocess timestep=4", caps, str(tmp_path / "in.rec"),
                 str(tmp_path / "out.rec")], tmp_path)
    assert "width=(int)80" in info["out_caps"] and "height=(int)180" in info["out_caps"] and "RGBA" in info["out_caps"]
    outs = _read(tmp_path / "out.rec")
    assert len(outs) == n_au - 3
    # what the C-ABI front end says: records per access unit and the output order
    order = np.zeros(1802, np.int32)
    n = C.c_int()
    assert lib.covahip_h264_display_order(h, order.ctypes.data, 1802, C.byref(n)) == 0
    order = [int(s) for s in order if s < n_au]
    assert sorted(order[:n_au]) == list(range(n_au))       # the first 300 access units are the first 300 output pictures (closed GoPs)
    rec = np.zeros((n_au, 45, 80, 4), np.uint8)
    for s in range(n_au):
        assert lib.covahip_h264_decode_records(h, s, rec[s].ctypes.data, rec[s].nbytes) == 0
    for k, (kind, pts, flags, payload) in enumerate(outs):
        cur = k + 3
        assert pts == order[cur] * CLK                      # the stacked frame carries the timestamp of its current picture
        stack = np.frombuffer(payload, np.uint8).reshape(4 * 45, 80, 4)
        for j in range(4):
            np.testing.assert_array_equal(stack[j * 45:(j + 1) * 45], rec[order[cur - j]])
    # round 5: records=true -> application/x-cova-records, a frame = the packed two-byte records of its picture (16 KB at 1080p
    # instead of a zero-filled 3 MB I420 frame), in the same output order with the same timestamps
    info = _run(["harness", "h264entropydec records=true", caps, str(tmp_path / "in.rec"), str(tmp_path / "rec.rec")], tmp_path)
    assert "application/x-cova-records" in info["out_caps"] and "width-mbs=(int)80" in info["out_caps"] and "height-mbs=(int)45" in info["out_caps"]
    routs = _read(tmp_path / "rec.rec")
    assert len(routs) == n_au
    for k, (kind, pts, flags, payload) in enumerate(routs):
        assert pts == order[k] * CLK and len(payload) == 45 * 80 * 2
        np.testing.assert_array_equal(np.frombuffer(payload, np.uint16).reshape(45, 80), E.pack_frames(rec[order[k]]))
    lib.covahip_h264_close(h)
    # the same chain into the sink: one Example per emitted frame
    out = tmp_path / "out.tfrecord"
    gt = np.random.default_rng(0).integers(0, 2, (n_au, 45 * 80), dtype=np.uint8)      # the ground-truth file the sink reads beside the frames
    (tmp_path / "gt.bin").write_bytes(gt.tobytes())
    _run(["sink", f"h264entropydec ! metapreprocess timestep=1 ! tfrecordsink location={out} gt={tmp_path / 'gt.bin'} async=false sync=false",
          caps, str(tmp_path / "in.rec")], tmp_path)
    raw_out, o, k = out.read_bytes(), 0, 0
    while o < len(raw_out):
        (length,) = struct.unpack_from("<Q", raw_out, o)
        assert struct.unpack_from("<I", raw_out, o + 8)[0] == _masked(raw_out[o:o + 8])
        ex = _parse_example(raw_out[o + 12:o + 12 + length])
        for ch, name in enumerate(("mb_type", "mv_x", "mv_y")):
            assert ex[name] == [rec[order[k]][..., ch].tobytes()]
        assert ex["gt"] == [gt[k].tobytes()]
        o += 16 + length
        k += 1
    assert k == n_au
