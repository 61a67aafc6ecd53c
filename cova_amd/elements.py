"""Host-side mirror of the CoVA GStreamer elements on the compressed-domain hot path.

Same element names, property names/defaults and buffer semantics as the reference's
Rust/C++ elements, but every piece of arithmetic or state goes through the C-ABI of
libcovahip.so (include/covahip.h) -- this module only marshals buffers, exactly what a
GStreamer element's transform()/chain() vfunc would do.

  metapreprocess  cova-rs/gst-plugins/src/metapreprocess/imp.rs   -> MetaPreprocess
  nvinfer(BlobNet) config/blobnet/*.txt, model/tasks.py            -> BlobNetInfer
  maskcopy        gst-plugins/gst-maskcopy/gstmaskcopy.cpp         -> folded into BlobNetInfer
                                                                      (mask is already GRAY8 {0,1})
  bboxcc          cova-rs/gst-plugins/src/bboxcc/{imp,process}.rs  -> BboxCc
  sorttracker     cova-rs/gst-plugins/src/sorttracker/imp.rs       -> SortTracker
  cova            cova-rs/gst-plugins/src/cova/{imp,tracker}.rs    -> Cova
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L
from . import weights as W

FLOW_OK = "ok"
FLOW_DROPPED = "dropped"  # gst_base::BASE_TRANSFORM_FLOW_DROPPED


def _ptr(a: np.ndarray) -> int:
    return a.ctypes.data


# ---------------------------------------------------------------------------------- ctx
class Context:
    """One GPU context (device + HIP stream + BlobNet workspace)."""

    def __init__(self, device_id: int = 0):
        self._lib = L.lib()
        h = C.c_void_p()
        L.check(self._lib.covahip_ctx_create(device_id, C.byref(h)), "covahip_ctx_create")
        self.handle = h
        self.device_id = device_id

    def close(self):
        if getattr(self, "handle", None):
            self._lib.covahip_ctx_destroy(self.handle)
            self.handle = None

    __del__ = close

    def sync(self):
        L.check(self._lib.covahip_ctx_sync(self.handle), "covahip_ctx_sync", self.handle)

    def set_lanes(self, n: int):
        """Batches in flight (covahip_ctx_set_lanes): 1 (default) = strictly in call order, 2 = consecutive device-pointer
        filter calls overlap."""
        L.check(self._lib.covahip_ctx_set_lanes(self.handle, n), "covahip_ctx_set_lanes", self.handle)

    def lanes(self) -> int:
        n = C.c_int()
        L.check(self._lib.covahip_ctx_get_lanes(self.handle, C.byref(n)), "covahip_ctx_get_lanes", self.handle)
        return n.value

    def clock_mhz(self, busy_us: int = 200) -> float:
        """Shader clock held right now, measured beside whatever the ctx has in flight (covahip_dev_clock_mhz)."""
        mhz = C.c_float()
        L.check(self._lib.covahip_dev_clock_mhz(self.handle, busy_us, C.byref(mhz)), "covahip_dev_clock_mhz", self.handle)
        return mhz.value

    def info(self):
        name = C.create_string_buffer(256)
        cu = C.c_int()
        mem = C.c_size_t()
        L.check(self._lib.covahip_device_info(self.handle, name, 256, C.byref(cu), C.byref(mem)), "device_info")
        return {"name": name.value.decode(), "num_cu": cu.value, "hbm_bytes": mem.value}

    # device memory -------------------------------------------------------------
    def malloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        L.check(self._lib.covahip_malloc(self.handle, nbytes, C.byref(p)), "covahip_malloc", self.handle)
        return p.value

    def free(self, dptr: int):
        L.check(self._lib.covahip_free(self.handle, dptr), "covahip_free", self.handle)

    def h2d(self, dptr: int, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        L.check(self._lib.covahip_memcpy_h2d(self.handle, dptr, _ptr(arr), arr.nbytes), "h2d", self.handle)

    def d2h(self, arr: np.ndarray, dptr: int):
        assert arr.flags["C_CONTIGUOUS"]
        L.check(self._lib.covahip_memcpy_d2h(self.handle, _ptr(arr), dptr, arr.nbytes), "d2h", self.handle)

    # timing --------------------------------------------------------------------
    def timer_start(self, slot=0):
        L.check(self._lib.covahip_timer_start(self.handle, slot), "timer_start", self.handle)

    def timer_stop(self, slot=0):
        L.check(self._lib.covahip_timer_stop(self.handle, slot), "timer_stop", self.handle)

    def timer_ms(self, slot=0) -> float:
        ms = C.c_float()
        L.check(self._lib.covahip_timer_elapsed_ms(self.handle, slot, C.byref(ms)), "timer_elapsed", self.handle)
        return ms.value

    def profile(self, on: bool, only: str | None = None):
        L.check(self._lib.covahip_profile_filter(self.handle, only.encode() if only else None), "profile_filter")
        L.check(self._lib.covahip_profile_enable(self.handle, int(on)), "profile_enable")
        L.check(self._lib.covahip_profile_reset(self.handle), "profile_reset")

    def profile_read(self):
        buf = np.zeros(64, dtype=L.KERNEL_TIME_DTYPE)
        n = C.c_int()
        L.check(self._lib.covahip_profile_read(self.handle, _ptr(buf), 64, C.byref(n)), "profile_read")
        return {r["name"].decode(): (float(r["total_ms"]), int(r["launches"])) for r in buf[:n.value]}


# ---------------------------------------------------------------------- metapreprocess
class MetaPreprocess:
    """`metapreprocess` (BaseTransform, NeverInPlace): properties `timestep`, `gamma`."""

    def __init__(self, timestep: int = 1, gamma: int = 1):  # DEFAULT_TIMESTEP / DEFAULT_GAMMA = 1
        self.timestep = timestep
        self.gamma = gamma
        self._h = None
        self._lib = L.lib()

    def transform_caps(self, width: int, height: int):
        """sink caps I420 w x h  ->  src caps RGBA (w/16) x (h/16*timestep) (imp.rs:247-286)."""
        ow, oh = C.c_int(), C.c_int()
        self._lib.covahip_stack_out_dims(width, height, self.timestep, C.byref(ow), C.byref(oh))
        return ow.value, oh.value

    def set_caps(self, width: int, height: int):
        ow, oh = self.transform_caps(width, height)
        out_size = ow * oh * 4                       # RGBA VideoInfo size
        self.size_per_buf = out_size // self.timestep  # imp.rs:233
        self.out_size = out_size
        self.stop()
        h = C.c_void_p()
        L.check(self._lib.covahip_stack_new(self.size_per_buf, self.timestep, self.gamma, C.byref(h)), "stack_new")
        self._h = h
        return ow, oh

    def transform(self, inbuf: bytes | np.ndarray):
        """Returns (FLOW_OK, out_bytes) or (FLOW_DROPPED, None)."""
        a = np.frombuffer(inbuf, dtype=np.uint8) if not isinstance(inbuf, np.ndarray) else inbuf.reshape(-1)
        a = np.ascontiguousarray(a, dtype=np.uint8)
        out = np.empty(self.out_size, dtype=np.uint8)
        em = C.c_int()
        L.check(self._lib.covahip_stack_push(self._h, _ptr(a), a.size, _ptr(out), out.size, C.byref(em)), "stack_push")
        return (FLOW_OK, out) if em.value else (FLOW_DROPPED, None)

    def stop(self):
        if self._h:
            self._lib.covahip_stack_free(self._h)
            self._h = None

    __del__ = stop


# ------------------------------------------------------------------- nvinfer(BlobNet)
class BlobNetInfer:
    """Stands where `nvinfer` (BlobNet TensorRT engine) + `maskcopy` stand in the reference
    pipeline: batched RGBA stacks in, GRAY8 {0,1} masks (and optionally logits) out."""

    def __init__(self, ctx: Context, weights_flat: np.ndarray, h_mb: int, w_mb: int, max_batch: int,
                 timestep: int = 4):
        self.ctx, self.h, self.w, self.t, self.max_batch = ctx, h_mb, w_mb, timestep, max_batch
        self._lib = L.lib()
        blob = W.to_bytes(weights_flat)
        L.check(self._lib.covahip_blobnet_load(ctx.handle, blob, len(blob), h_mb, w_mb, timestep, max_batch),
                "covahip_blobnet_load", ctx.handle)

    def set_enc_plan(self, level: int, nbands: int, nbuf: int = 1):
        """Developer switch (include/covahip_dev.h): band plan of encoder level 1..3; nbands = 0 -> automatic."""
        L.check(self._lib.covahip_blobnet_set_enc_plan(self.ctx.handle, level, nbands, nbuf), "covahip_blobnet_set_enc_plan")

    def set_impl(self, impl: str):
        """Developer switch (include/covahip_dev.h): decoder blocks 0..2 as one launch ("mfma", default) or three
        ("dec_separate"); "enc1_legacy": encoder level 1 on the 32x32x16 kernel instead of the sixteen-channel-wave one;
        "enc_general_tiles": encoder levels 2 and 3 on the general tile form instead of the row-aligned one;
        "enc23_separate" / "enc23_force": encoder levels 2 + 3 always as two launches / as one launch (enc23_mfma) whenever it
        fits (the default takes the single launch when the batch fills the chip and the rows fill their tile columns);
        "tail_skip_tensor": the last decoder block reads the level-0 skip tensor (rounds 1-4) instead of the partial logits the
        level-1 kernel computes from it (round 5 default; another summation order, not another value)."""
        L.check(self._lib.covahip_blobnet_set_impl(self.ctx.handle, {"mfma": 1, "dec_separate": 4, "enc1_legacy": 5, "enc_general_tiles": 6, "enc23_separate": 7, "enc23_force": 8, "tail_skip_tensor": 9, "tail_band_tiles": 10}[impl]), "set_impl")

    @property
    def macs_per_frame(self) -> int:
        v = C.c_int64()
        L.check(self._lib.covahip_blobnet_macs_per_frame(self.ctx.handle, C.byref(v)), "macs_per_frame")
        return v.value

    def infer(self, stack: np.ndarray, want_logits: bool = True):
        """stack u8 [B][t*h][w][4] (host) -> (logits f32 [B][h][w] | None, mask u8 [B][h][w])."""
        stack = np.ascontiguousarray(stack, dtype=np.uint8)
        b = stack.shape[0]
        assert stack.shape == (b, self.t * self.h, self.w, 4), stack.shape
        logits = np.empty((b, self.h, self.w), dtype=np.float32) if want_logits else None
        mask = np.empty((b, self.h, self.w), dtype=np.uint8)
        L.check(self._lib.covahip_blobnet_forward(self.ctx.handle, _ptr(stack), b, _ptr(logits) if want_logits else None,
                                                  _ptr(mask), L.MEM_HOST), "covahip_blobnet_forward", self.ctx.handle)
        return logits, mask

    def infer_device(self, d_stack: int, batch: int, d_logits: int | None, d_mask: int | None):
        L.check(self._lib.covahip_blobnet_forward(self.ctx.handle, d_stack, batch, d_logits, d_mask, L.MEM_DEVICE),
                "covahip_blobnet_forward", self.ctx.handle)

    def filter(self, stack: np.ndarray, cc_threshold: int, max_boxes: int = 256, want_mask: bool = False):
        """Fused BlobNet -> mask -> bboxcc on host buffers: returns (boxes [B][max_boxes], counts [B], mask|None)."""
        stack = np.ascontiguousarray(stack, dtype=np.uint8)
        b = stack.shape[0]
        boxes = np.zeros((b, max_boxes), dtype=L.BOX_DTYPE)
        counts = np.zeros(b, dtype=np.int32)
        mask = np.empty((b, self.h, self.w), dtype=np.uint8) if want_mask else None
        L.check(self._lib.covahip_filter_forward(self.ctx.handle, _ptr(stack), b, cc_threshold, _ptr(boxes), _ptr(counts),
                                                 max_boxes, None, _ptr(mask) if want_mask else None, L.MEM_HOST),
                "covahip_filter_forward", self.ctx.handle)
        return boxes, counts, mask

    def filter_frames(self, frames: np.ndarray, stack_index: np.ndarray | None, cc_threshold: int, max_boxes: int = 256,
                      want_mask: bool = False, want_logits: bool = False):
        """The hot path fed with carrier frames u8 [F][h][w][4] (host) and the stack -> frame index table i32 [B][4]
        (None: one stream in order): returns (boxes, counts, mask|None, logits|None) like filter()."""
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
        f = frames.shape[0]
        assert frames.shape == (f, self.h, self.w, 4), frames.shape
        idx = None if stack_index is None else np.ascontiguousarray(stack_index, dtype=np.int32).reshape(-1, 4)
        b = f - 3 if idx is None else idx.shape[0]
        boxes = np.zeros((b, max_boxes), dtype=L.BOX_DTYPE)
        counts = np.zeros(b, dtype=np.int32)
        mask = np.empty((b, self.h, self.w), dtype=np.uint8) if want_mask else None
        logits = np.empty((b, self.h, self.w), dtype=np.float32) if want_logits else None
        L.check(self._lib.covahip_filter_forward_frames(self.ctx.handle, _ptr(frames), f, None if idx is None else _ptr(idx), b,
                                                        cc_threshold, _ptr(boxes), _ptr(counts), max_boxes,
                                                        _ptr(logits) if want_logits else None,
                                                        _ptr(mask) if want_mask else None, L.MEM_HOST),
                "covahip_filter_forward_frames", self.ctx.handle)
        return boxes, counts, mask, logits

    def filter_frames_device(self, d_frames: int, n_frames: int, stack_index: np.ndarray | None, batch: int, cc_threshold: int,
                             d_boxes: int, d_counts: int, max_boxes: int, d_mask: int | None = None):
        idx = None if stack_index is None else np.ascontiguousarray(stack_index, dtype=np.int32).reshape(-1, 4)
        L.check(self._lib.covahip_filter_forward_frames(self.ctx.handle, d_frames, n_frames, None if idx is None else _ptr(idx),
                                                        batch, cc_threshold, d_boxes, d_counts, max_boxes, None, d_mask,
                                                        L.MEM_DEVICE),
                "covahip_filter_forward_frames", self.ctx.handle)

    def filter_device(self, d_stack: int, batch: int, cc_threshold: int, d_boxes: int, d_counts: int, max_boxes: int,
                      d_mask: int | None = None):
        L.check(self._lib.covahip_filter_forward(self.ctx.handle, d_stack, batch, cc_threshold, d_boxes, d_counts,
                                                 max_boxes, None, d_mask, L.MEM_DEVICE),
                "covahip_filter_forward", self.ctx.handle)


def pack_frames(frames: np.ndarray) -> np.ndarray:
    """covahip_carrier_pack: carrier frames u8 [..][h][w][4] -> two-byte records u16 [..][h][w]."""
    frames = np.ascontiguousarray(frames, dtype=np.uint8)
    out = np.empty(frames.shape[:-1], dtype=np.uint16)
    L.lib().covahip_carrier_pack(frames.ctypes.data_as(C.c_void_p), frames.size // 4, out.ctypes.data_as(C.c_void_p))
    return out


class FilterPipe:
    """covahip_pipe_*: batches of carrier frames in pinned host memory, H2D / kernels / D2H of consecutive batches
    overlapped on three HIP streams, boxes compacted on the device."""

    def __init__(self, net: "BlobNetInfer", max_batch: int, max_frames: int, max_boxes: int = 256, n_slots: int = 3,
                 want_mask: bool = False, packed: bool = False, blocking_wait: bool = False):
        self.net, self.max_batch, self.max_frames, self.max_boxes = net, max_batch, max_frames, max_boxes
        self.packed = packed
        self._lib = L.lib()
        h = C.c_void_p()
        L.check(self._lib.covahip_pipe_create(net.ctx.handle, max_batch, max_frames, max_boxes, n_slots, int(want_mask),
                                              C.byref(h)), "covahip_pipe_create", net.ctx.handle)
        self._h = h
        if packed:      # the slots take two-byte records (pack_frames) instead of the decoder's four bytes per macroblock
            L.check(self._lib.covahip_pipe_set_packed(h, 1), "covahip_pipe_set_packed")
        if blocking_wait:   # collect() sleeps until the results have landed instead of spinning (covahip_pipe_set_blocking_wait)
            L.check(self._lib.covahip_pipe_set_blocking_wait(h, 1), "covahip_pipe_set_blocking_wait", net.ctx.handle)
        self._batch = {}
        self._views = {}     # slot -> numpy views of its pinned input buffers (the addresses never change)
        self._rviews = {}    # slot -> full-size numpy views of its result buffers
        self._held = []      # collected slots whose result views are still handed out

    def close(self):
        if getattr(self, "_h", None):
            self._lib.covahip_pipe_destroy(self._h)
            self._h = None

    __del__ = close

    def queue_plan(self):
        """(active lanes on the upload stream's hardware queue, active lanes on the result stream's): what the probe at creation
        found (include/covahip_dev.h)."""
        a, b = C.c_int(-1), C.c_int(-1)
        L.check(self._lib.covahip_dev_pipe_queue_plan(self._h, C.byref(a), C.byref(b)), "covahip_dev_pipe_queue_plan")
        return a.value, b.value

    def acquire(self):
        """-> (slot, frames u8 [max_frames][h][w][4] -- u16 [max_frames][h][w] records when packed --, index i32 [max_batch][4]):
        numpy views of the slot's pinned buffers, or None when every slot is in flight.  Result views of earlier collect() calls
        become invalid."""
        for held in self._held:
            L.check(self._lib.covahip_pipe_release(self._h, held), "covahip_pipe_release")
        self._held = []
        slot, fp, ip = C.c_int(), C.c_void_p(), C.c_void_p()
        rc = self._lib.covahip_pipe_acquire(self._h, C.byref(slot), C.byref(fp), C.byref(ip))
        if rc == 7:
            return None
        L.check(rc, "covahip_pipe_acquire")
        if slot.value not in self._views:
            n = self.max_frames * self.net.h * self.net.w * 4
            if self.packed:
                frames = np.ctypeslib.as_array(C.cast(fp.value, C.POINTER(C.c_uint16)), shape=(n // 4,)).reshape(self.max_frames, self.net.h, self.net.w)
            else:
                frames = np.ctypeslib.as_array(C.cast(fp.value, C.POINTER(C.c_uint8)), shape=(n,)).reshape(self.max_frames, self.net.h, self.net.w, 4)
            index = np.ctypeslib.as_array(C.cast(ip.value, C.POINTER(C.c_int32)), shape=(self.max_batch * 4,)).reshape(self.max_batch, 4)
            self._views[slot.value] = (frames, index)
        return (slot.value,) + self._views[slot.value]

    def submit(self, slot: int, n_frames: int, batch: int, cc_threshold: int):
        L.check(self._lib.covahip_pipe_submit(self._h, slot, n_frames, batch, cc_threshold), "covahip_pipe_submit", self.net.ctx.handle)
        self._batch[slot] = batch

    def abort(self, slot: int):
        """Gives an acquired, unsubmitted slot back (covahip_pipe_abort)."""
        L.check(self._lib.covahip_pipe_abort(self._h, slot), "covahip_pipe_abort")

    def collect(self, slot: int):
        """-> (counts [B], offsets [B+1], packed boxes [offsets[B]], mask [B][h][w] | None): views, valid until the next
        acquire()."""
        cp, op, bp, mp = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        L.check(self._lib.covahip_pipe_collect(self._h, slot, C.byref(cp), C.byref(op), C.byref(bp), C.byref(mp)),
                "covahip_pipe_collect", self.net.ctx.handle)
        b = self._batch.pop(slot)
        self._held.append(slot)
        # the result buffers of a slot never move: wrap them ONCE at their full size and slice per batch (np.ctypeslib.as_array costs
        # 10 - 20 us a call; three or four of them per batch made a Python caller host-bound at ~130 us per batch on a slow box)
        key = (slot, cp.value, op.value, bp.value, mp.value)
        full = self._rviews.get(slot)
        if full is None or full[0] != key:
            counts_f = np.ctypeslib.as_array(C.cast(cp.value, C.POINTER(C.c_int32)), shape=(self.max_batch,))
            offsets_f = np.ctypeslib.as_array(C.cast(op.value, C.POINTER(C.c_int32)), shape=(self.max_batch + 1,))
            boxes_f = np.ctypeslib.as_array(C.cast(bp.value, C.POINTER(C.c_int32)), shape=(self.max_batch * self.max_boxes * 5,)).view(L.BOX_DTYPE)
            mask_f = None
            if mp.value:
                mask_f = np.ctypeslib.as_array(C.cast(mp.value, C.POINTER(C.c_uint8)),
                                               shape=(self.max_batch * self.net.h * self.net.w,)).reshape(self.max_batch, self.net.h, self.net.w)
            full = (key, counts_f, offsets_f, boxes_f, mask_f)
            self._rviews[slot] = full
        _, counts_f, offsets_f, boxes_f, mask_f = full
        offsets = offsets_f[:b + 1]
        return counts_f[:b], offsets, boxes_f[:int(offsets[b])], (mask_f[:b] if mask_f is not None else None)


# -------------------------------------------------------------------------------- bbox
def boxes_to_bbox(boxes: np.ndarray) -> np.ndarray:
    """covahip_box[] (CC stats) -> covahip_bbox[] via Bbox::new (process.rs:47)."""
    boxes = np.ascontiguousarray(boxes, dtype=L.BOX_DTYPE)
    out = np.zeros(boxes.shape[0], dtype=L.BBOX_DTYPE)
    L.lib().covahip_boxes_to_bbox(_ptr(boxes), boxes.shape[0], _ptr(out))
    return out


def make_bbox(left, top, width, height) -> np.ndarray:
    b = np.zeros(1, dtype=L.BBOX_DTYPE)
    b["left"], b["top"], b["width"], b["height"] = left, top, width, height
    b["area"] = np.float32(width) * np.float32(height)
    return b


def serialize_vec(bboxes: np.ndarray) -> bytes:
    bboxes = np.ascontiguousarray(bboxes, dtype=L.BBOX_DTYPE)
    n = bboxes.shape[0]
    need = L.lib().covahip_bbox_serialize_vec(_ptr(bboxes), n, None, 0, None)
    out = np.empty(need, dtype=np.uint8)
    st = C.c_int()
    L.lib().covahip_bbox_serialize_vec(_ptr(bboxes), n, _ptr(out), need, C.byref(st))
    L.check(st.value, "covahip_bbox_serialize_vec")
    return out.tobytes()


def deserialize_vec(data: bytes) -> np.ndarray:
    a = np.frombuffer(data, dtype=np.uint8)
    n = C.c_size_t()
    cap = max(len(data) // 24, 1)
    out = np.zeros(cap, dtype=L.BBOX_DTYPE)
    L.check(L.lib().covahip_bbox_deserialize_vec(_ptr(a) if a.size else None, a.size, _ptr(out), cap, C.byref(n)),
            "covahip_bbox_deserialize_vec")
    return out[:n.value].copy()


def serialize_frame(range_start: int, oldest: int, bboxes: np.ndarray) -> bytes:
    bboxes = np.ascontiguousarray(bboxes, dtype=L.BBOX_DTYPE)
    n = bboxes.shape[0]
    need = L.lib().covahip_frame_serialize(range_start, oldest, _ptr(bboxes), n, None, 0, None)
    out = np.empty(need, dtype=np.uint8)
    st = C.c_int()
    L.lib().covahip_frame_serialize(range_start, oldest, _ptr(bboxes), n, _ptr(out), need, C.byref(st))
    L.check(st.value, "covahip_frame_serialize")
    return out.tobytes()


def iou(a: np.ndarray, b: np.ndarray) -> float:
    a = np.ascontiguousarray(a, dtype=L.BBOX_DTYPE)
    b = np.ascontiguousarray(b, dtype=L.BBOX_DTYPE)
    return float(L.lib().covahip_bbox_iou(_ptr(a), _ptr(b)))


def _sized_call(fn, what: str) -> bytes:
    """Two-pass size query / fill convention of the byte-format entry points."""
    need = fn(None, 0, None)
    buf = np.zeros(max(need, 1), np.uint8)
    st = C.c_int()
    fn(_ptr(buf), need, C.byref(st))
    L.check(st.value, what)
    return buf[:need].tobytes()


def tfrecord_example(rgba: np.ndarray, gt=None, gop: int = 0) -> bytes:
    """One framed TFRecord record as `tfrecordsink` writes per GoP (tfrecordsink/imp.rs:69-198): features
    mb_type / mv_x / mv_y = bytes 0 / 1 / 2 of every RGBA pixel per frame, gt = label bytes, zero-filled to `gop` frames."""
    rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
    n, h, w, _ = rgba.shape
    g = None if gt is None else np.ascontiguousarray(gt, dtype=np.uint8).reshape(n, h * w)
    lib = L.lib()
    return _sized_call(lambda o, c, st: lib.covahip_tfrecord_example(_ptr(rgba), None if g is None else _ptr(g), n, gop, w, h,
                                                                     o, c, st), "tfrecord_example")


def bbox_csv(bboxes: np.ndarray, with_header: bool = True) -> str:
    """The CSV rows `bboxsink` writes (bboxsink/imp.rs:252-270; serde field order of bbox.rs:12-27)."""
    b = np.ascontiguousarray(bboxes, dtype=L.BBOX_DTYPE)
    lib = L.lib()
    return _sized_call(lambda o, c, st: lib.covahip_bbox_csv(_ptr(b), b.shape[0], int(with_header), o, c, st), "bbox_csv").decode()


def tracks_export(range_start: int, oldest: int, dead_boxes: np.ndarray, track_lens: np.ndarray) -> bytes:
    """Bytes the `cova` element's tracker task sends per finished GoP (cova/tracker.rs:59-83): per track a 4-byte
    big-endian length followed by the bincode `Frame`."""
    b = np.ascontiguousarray(dead_boxes, dtype=L.BBOX_DTYPE)
    t = np.ascontiguousarray(track_lens, dtype=np.uint32)
    lib = L.lib()
    return _sized_call(lambda o, c, st: lib.covahip_tracks_export(range_start, oldest, _ptr(b), _ptr(t), t.shape[0], o, c, st),
                       "tracks_export")


# ------------------------------------------------------------------------------ bboxcc
class BboxCc:
    """`bboxcc` (BaseTransform, AlwaysInPlace): property `cc-threshold` (default 30)."""

    def __init__(self, ctx: Context, cc_threshold: int = 30, max_boxes: int = 1024):
        self.ctx, self.cc_threshold, self.max_boxes = ctx, cc_threshold, max_boxes
        self._lib = L.lib()

    def set_wave_cap(self, cap: int):
        """Developer switch (include/covahip_dev.h): > 0 run capacity of the wave-per-frame kernel, < 0 workgroup kernel only, 0 auto."""
        L.check(self._lib.covahip_bboxcc_set_wave_cap(self.ctx.handle, cap), "covahip_bboxcc_set_wave_cap")

    def overflow_stats(self):
        """Developer read-out (include/covahip_dev.h): {batch, overflowed pass 1, overflowed pass 2, capacity of pass 1} of the last
        large-batch device-pointer call."""
        out = (C.c_int32 * 4)()
        L.check(self._lib.covahip_dev_bboxcc_overflow(self.ctx.handle, out), "covahip_dev_bboxcc_overflow", self.ctx.handle)
        return {"batch": out[0], "overflow_pass1": out[1], "overflow_pass2": out[2], "cap_pass1": out[3]}

    def regionprops(self, masks: np.ndarray):
        """masks u8 [B][H][W] (host) -> (boxes [B][max_boxes], counts [B])."""
        masks = np.ascontiguousarray(masks, dtype=np.uint8)
        if masks.ndim == 2:
            masks = masks[None]
        b, h, w = masks.shape
        boxes = np.zeros((b, self.max_boxes), dtype=L.BOX_DTYPE)
        counts = np.zeros(b, dtype=np.int32)
        L.check(self._lib.covahip_bboxcc(self.ctx.handle, _ptr(masks), b, h, w, self.cc_threshold, _ptr(boxes),
                                         _ptr(counts), self.max_boxes, L.MEM_HOST), "covahip_bboxcc", self.ctx.handle)
        return boxes, counts

    def regionprops_device(self, d_masks: int, b: int, h: int, w: int, d_boxes: int, d_counts: int):
        L.check(self._lib.covahip_bboxcc(self.ctx.handle, d_masks, b, h, w, self.cc_threshold, d_boxes, d_counts,
                                         self.max_boxes, L.MEM_DEVICE), "covahip_bboxcc", self.ctx.handle)

    def transform_ip(self, buf: bytes | np.ndarray, width: int, height: int) -> bytes:
        """GRAY8 mask buffer -> bincode Vec<Bbox> bytes (imp.rs:232-272)."""
        m = np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else buf
        m = m.reshape(height, -1)[:, :width]  # process.rs:14-15: reshape(1, height)
        boxes, counts = self.regionprops(m)
        n = int(counts[0])
        if n > self.max_boxes:
            raise L.CovahipError(7, "bboxcc.transform_ip", f"{n} boxes > max_boxes {self.max_boxes}")
        return serialize_vec(boxes_to_bbox(boxes[0, :n]))


# ------------------------------------------------------------------------- sorttracker
class _SortHandle:
    def __init__(self, max_age: int, min_hits: int, iou_threshold: float):
        self._lib = L.lib()
        h = C.c_void_p()
        L.check(self._lib.covahip_sort_new(max_age, min_hits, iou_threshold, C.byref(h)), "sort_new")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self._lib.covahip_sort_free(self.h)
            self.h = None

    __del__ = close

    def _collect(self, fn, *args):
        cap, capt = 4096, 256
        while True:
            boxes = np.zeros(cap, dtype=L.BBOX_DTYPE)
            lens = np.zeros(capt, dtype=np.uint32)
            nb, nt = C.c_size_t(), C.c_size_t()
            st = fn(*args, _ptr(boxes), cap, C.byref(nb), _ptr(lens), capt, C.byref(nt))
            if st == 7:
                raise L.CovahipError(st, "sort output capacity")  # state already advanced; sized generously
            L.check(st, "sort")
            return boxes[:nb.value].copy(), lens[:nt.value].copy()

    def update(self, dets: np.ndarray, pts: int):
        dets = np.ascontiguousarray(dets, dtype=L.BBOX_DTYPE)
        return self._collect(lambda *a: self._lib.covahip_sort_update(self.h, _ptr(dets), dets.shape[0], pts, *a))

    def finalize(self):
        return self._collect(lambda *a: self._lib.covahip_sort_finalize(self.h, *a))

    def mark_seen(self, ts: int):
        L.check(self._lib.covahip_sort_mark_seen(self.h, ts), "sort_mark_seen")

    def num_trackers(self) -> int:
        n = C.c_size_t()
        L.check(self._lib.covahip_sort_num_trackers(self.h, C.byref(n)), "sort_num_trackers")
        return n.value

    def tracker_info(self, i: int):
        tid, act, hs, tsu = C.c_uint64(), C.c_int(), C.c_uint64(), C.c_uint64()
        st = np.zeros(1, dtype=L.BBOX_DTYPE)
        L.check(self._lib.covahip_sort_tracker_info(self.h, i, C.byref(tid), C.byref(act), C.byref(hs), C.byref(tsu),
                                                    _ptr(st)), "sort_tracker_info")
        return {"id": tid.value, "active": bool(act.value), "hit_streaks": hs.value,
                "time_since_update": tsu.value, "state": st[0]}

    def tracker_predict(self, i: int, ts: int):
        """KalmanBoxTracker::predict(ts) on tracker i (tracker/mod.rs:104-121); returns history.last()."""
        st = np.zeros(1, dtype=L.BBOX_DTYPE)
        L.check(self._lib.covahip_sort_tracker_predict(self.h, i, ts, _ptr(st)), "sort_tracker_predict")
        return st[0]

    def tracker_update(self, i: int, det=None):
        """KalmanBoxTracker::update(Some(det) / None) on tracker i (tracker/mod.rs:71-102)."""
        if det is None:
            L.check(self._lib.covahip_sort_tracker_update(self.h, i, None), "sort_tracker_update")
        else:
            d = np.ascontiguousarray(det, dtype=L.BBOX_DTYPE).reshape(1)
            L.check(self._lib.covahip_sort_tracker_update(self.h, i, _ptr(d)), "sort_tracker_update")


class SortTracker:
    """`sorttracker` (BaseTransform, NeverInPlace): `iou-threshold` 0.1, `maxage` 30, `minhits` 30."""

    def __init__(self, iou_threshold: float = 0.1, maxage: int = 30, minhits: int = 30):
        self.iou_threshold, self.maxage, self.minhits = iou_threshold, maxage, minhits
        self.sort = None

    def set_caps(self, width: int = 0, height: int = 0):  # Sort created in set_caps (imp.rs:209-236)
        self.sort = _SortHandle(self.maxage, self.minhits, self.iou_threshold)

    def transform(self, inbuf: bytes, pts: int) -> bytes:
        if self.sort is None:
            self.set_caps()
        dead, _ = self.sort.update(deserialize_vec(inbuf), pts)
        return serialize_vec(dead)

    def sink_event_eos(self) -> bytes:
        fin, _ = self.sort.finalize()
        return serialize_vec(fin)


def linear_assignment(cost: np.ndarray):
    """cost [n_rows][n_cols] f32 -> sorted list of (row, col) (sort/src/lib.rs:25-56)."""
    cost = np.asarray(cost, dtype=np.float32)
    nr, nc = cost.shape
    colmajor = np.ascontiguousarray(cost.T).reshape(-1)
    pairs = np.zeros(2 * max(nr, nc, 1), dtype=np.uint32)
    n = L.lib().covahip_linear_assignment(_ptr(colmajor), nr, nc, _ptr(pairs), pairs.size // 2)
    return sorted((int(pairs[2 * k]), int(pairs[2 * k + 1])) for k in range(n))


# -------------------------------------------------------------------------------- cova
class Cova:
    """`cova` element: pads sink_mask (bbox), sink_enc (encoded AUs), src.  Properties as in
    cova/imp.rs:22-29,590-634; read-only counters dropped / decoded-dependency / decoded-inference."""

    def __init__(self, sort_iou: float = 0.1, sort_maxage: int = 30, sort_minhits: int = 30, port: int = 0,
                 infer_i: bool = False, debug: bool = False, alpha: int = 0, beta: int = 0):
        self.port = port           # the socket itself belongs to the GStreamer element (gst/gstcova.c); here the bytes
        self._lib = L.lib()        # it would carry are read with take_track_export()
        cfg = L.GopFilterCfg(sort_iou, sort_maxage, sort_minhits, alpha, beta, int(infer_i))
        h = C.c_void_p()
        L.check(self._lib.covahip_gopfilter_new(C.byref(cfg), C.byref(h)), "gopfilter_new")
        self._h = h
        self._eos = [False, False]

    def close(self):
        if getattr(self, "_h", None):
            self._lib.covahip_gopfilter_free(self._h)
            self._h = None

    __del__ = close

    def sink_enc_chain(self, au_id: int, pts: int, delta_unit: bool):
        L.check(self._lib.covahip_gopfilter_push_enc(self._h, au_id, pts, L.AU_DELTA_UNIT if delta_unit else 0),
                "gopfilter_push_enc")

    def _out(self, fn):
        cap = 1 << 16
        out = np.zeros(cap, dtype=L.AU_OUT_DTYPE)
        n = C.c_size_t()
        L.check(fn(_ptr(out), cap, C.byref(n)), "gopfilter")
        return out[:n.value].copy()

    def sink_mask_chain(self, bbox_bytes: bytes, pts: int) -> np.ndarray:
        boxes = deserialize_vec(bbox_bytes)
        return self._out(lambda o, c, n: self._lib.covahip_gopfilter_push_boxes(self._h, _ptr(boxes), boxes.shape[0],
                                                                                pts, o, c, n))

    def eos(self, pad: str):
        """EOS on `sink_enc` / `sink_mask`; flushes once both have seen it (imp.rs:361-432)."""
        self._eos[0 if pad == "sink_enc" else 1] = True
        if all(self._eos):
            return self._out(lambda o, c, n: self._lib.covahip_gopfilter_eos(self._h, o, c, n))
        return None

    def take_dropped(self) -> list:
        """Ids of the access units discarded for good since the last call (their buffers can be released)."""
        ids = []
        buf = np.zeros(4096, dtype=np.uint64)
        while True:
            n = C.c_size_t()
            L.check(self._lib.covahip_gopfilter_take_dropped(self._h, _ptr(buf), buf.size, C.byref(n)), "take_dropped")
            ids.extend(int(x) for x in buf[:n.value])
            if n.value < buf.size:
                return ids

    def take_track_export(self) -> bytes:
        """Length-delimited bincode Frames of the tracks finished since the last call (cova/tracker.rs:59-83)."""
        return _sized_call(lambda o, c, st: self._lib.covahip_gopfilter_take_track_export(self._h, o, c, st),
                           "take_track_export")

    def _counters(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        L.check(self._lib.covahip_gopfilter_counters(self._h, C.byref(a), C.byref(b), C.byref(c)), "counters")
        return a.value, b.value, c.value

    dropped = property(lambda s: s._counters()[0])
    decoded_dependency = property(lambda s: s._counters()[1])
    decoded_inference = property(lambda s: s._counters()[2])


# ---------------------------------------------------------------------------------------- analysis-aggregator join
class Associator:
    """The association rules of CoVA's analysis-aggregator (assoc.rs:63-507) without its sockets: tracker frames
    and DNN detections are pushed in arrival order, `csv(name)` returns the text of track / dnn / assoc /
    stationary .csv.  Defaults as main.rs:32-39."""

    FILES = {"track": 0, "dnn": 1, "assoc": 2, "stationary": 3}

    def __init__(self, range_starts, moving_iou: float = 0.15, stationary_iou: float = 0.3, stationary_maxage: int = 120,
                 scale_factor: float = 1.3):
        self._lib = L.lib()
        cfg = L.AssocCfg(moving_iou, stationary_iou, stationary_maxage, scale_factor)
        rs = np.ascontiguousarray(range_starts, dtype=np.uint64)
        h = C.c_void_p()
        L.check(self._lib.covahip_assoc_new(C.byref(cfg), _ptr(rs), rs.shape[0], C.byref(h)), "assoc_new")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.covahip_assoc_free(self._h)
            self._h = None

    __del__ = close

    def push_track(self, range_start: int, oldest: int, boxes: np.ndarray):
        b = np.ascontiguousarray(boxes, dtype=L.BBOX_DTYPE)
        L.check(self._lib.covahip_assoc_push_track(self._h, range_start, oldest, _ptr(b), b.shape[0]), "assoc_push_track")

    def push_track_frame(self, payload: bytes):
        buf = np.frombuffer(payload, dtype=np.uint8)
        L.check(self._lib.covahip_assoc_push_track_frame(self._h, _ptr(buf), buf.shape[0]), "assoc_push_track_frame")

    def push_dnn(self, boxes: np.ndarray):
        b = np.ascontiguousarray(boxes, dtype=L.BBOX_DTYPE)
        L.check(self._lib.covahip_assoc_push_dnn(self._h, _ptr(b), b.shape[0]), "assoc_push_dnn")

    def push_dnn_text(self, text: bytes):
        L.check(self._lib.covahip_assoc_push_dnn_text(self._h, text, len(text)), "assoc_push_dnn_text")

    def terminate(self):
        L.check(self._lib.covahip_assoc_terminate(self._h), "assoc_terminate")

    def csv(self, name: str) -> str:
        which = self.FILES[name]
        return _sized_call(lambda o, c, st: self._lib.covahip_assoc_csv(self._h, which, o, c, st), "assoc_csv").decode()
