/*
 * gstblobnetfilter.c -- the batching filter element `blobnetfilter` and the `maskcopy` element of plugin "cova".
 *
 * blobnetfilter stands where the reference assembles
 *     metapreprocess (per stream) -> nvstreammux -> nvinfer(BlobNet) -> nvstreamdemux -> maskcopy -> bboxcc (per stream)
 * (pipeline/cova/pipeline.py:104-261): N request sink pads take the carrier frames of N entropy-decoder branches
 * (what `metapreprocess` takes: I420 caps of the picture, the macroblock records in the first w/16 * h/16 * 4 bytes,
 * cova-rs/gst-plugins/src/metapreprocess/imp.rs:233,311-312), batches them like nvstreammux (`batch-size`,
 * `batched-push-timeout`, pipeline.py:146-164), runs ONE covahip_pipe submission per batch -- temporal stacking as an
 * index gather on the GPU, BlobNet, threshold, connected components -- and pushes, per stream, what `bboxcc` pushes:
 * a bincode Vec<Bbox> buffer with the PTS of the frame (cova-rs/gst-plugins/src/bboxcc/imp.rs:232-272).
 * Batches are filled in pinned host memory; copy-in, kernels and copy-out of consecutive batches overlap
 * (include/covahip.h, covahip_pipe_*).  The per-stream elements `blobnetinfer` / `bboxcc` stay for compatibility.
 *
 * maskcopy keeps the reference element's name and properties (gst-plugins/gst-maskcopy/gstmaskcopy.cpp:39-46,
 * 102-125: unique-id, gpu-id, timestep; GRAY8 out).  In the reference it turns nvinfer's segmentation metadata into a
 * GRAY8 {0,1} mask; here `blobnetinfer` already emits that mask, so maskcopy passes GRAY8 buffers through.
 */
#include <gst/base/gstbasetransform.h>
#include <gst/gst.h>
#include <gst/video/video.h>
#include <string.h>

#include "covahip.h"

GST_DEBUG_CATEGORY_EXTERN(cova_debug);
#define GST_CAT_DEFAULT cova_debug

#define BF_TIMESTEP 4
#define BF_SLOTS 6

/* ===================================================================== blobnetfilter */
typedef struct {
    GstPad *sink, *src;
    guint idx;
    GstBuffer *hist[BF_TIMESTEP - 1]; /* the last three carrier frames (references, no copies): [0] newest */
    gint hist_pos[BF_TIMESTEP - 1];  /* their position in the slot being filled, -1 = not copied into it yet */
    guint n_seen;
    gboolean eos, caps_sent;
} BfPad;

typedef struct { guint pad; GstClockTime pts, duration; } BfMeta;
typedef struct { int slot; int n_stacks; BfMeta *meta; } BfFlight;
#define BF_PUSHERS 8
typedef struct _GstBlobNetFilter GstBlobNetFilter;
typedef struct {   /* one share of a finished batch: the stacks of the src pads with pad % BF_PUSHERS == group */
    GstBlobNetFilter *s;
    BfFlight *fl;
    const int32_t *counts, *offsets;
    const covahip_box *boxes;
    guint group;
} BfPushTask;

struct _GstBlobNetFilter {
    GstElement parent;
    GMutex lock, push_lock;
    GCond cond, flush_cond;
    gboolean flushing;        /* a flush has dropped the lock to push results: the slot pointers are in flux */
    gint pending;             /* streaming threads still copying their frame into the slot being filled (lock not held) */
    gchar *weights;
    guint gpu_id, batch_size, cc_threshold, max_boxes;
    guint64 timeout_us;
    covahip_ctx *ctx;
    covahip_pipe *pipe;
    gint w_mb, h_mb;
    gsize frame_bytes;
    GPtrArray *pads;          /* BfPad* */
    /* the batch being filled */
    int slot;
    uint8_t *pf;
    int32_t *pi;
    int n_frames, n_stacks, max_frames;
    BfMeta *meta;
    gint64 first_us;
    GQueue flights;           /* BfFlight*, oldest first: submitted, not yet taken by the collector */
    guint in_flight;          /* submitted batches whose results have not all been pushed yet */
    GCond slot_cond;          /* a slot was released / a flight was queued / all flights are done */
    GCond push_cond;          /* with push_lock: the last share of a batch has been pushed */
    GThread *timer, *collector;
    GThreadPool *pushers;
    gint push_left;           /* shares of the current batch still being pushed */
    GstFlowReturn push_ret;
    gboolean stop, failed;
    guint64 batches, frames_out;
};
typedef struct { GstElementClass parent_class; } GstBlobNetFilterClass;
G_DEFINE_TYPE(GstBlobNetFilter, gst_blobnetfilter, GST_TYPE_ELEMENT)
enum { BF_PROP_0, BF_PROP_WEIGHTS, BF_PROP_GPU, BF_PROP_BATCH, BF_PROP_TIMEOUT, BF_PROP_CC, BF_PROP_MAXBOXES, BF_PROP_BATCHES };

static BfPad *bf_pad_of(GstBlobNetFilter *s, GstPad *sink) { return (BfPad *)gst_pad_get_element_private(sink); }

static gboolean bf_ensure_model(GstBlobNetFilter *s) {   /* lock held */
    gchar *blob = NULL;
    gsize len = 0;
    int rc;
    if (s->pipe) return TRUE;
    if (s->failed) return FALSE;
    s->failed = TRUE;
    if (!s->weights || !g_file_get_contents(s->weights, &blob, &len, NULL)) {
        GST_ELEMENT_ERROR(s, RESOURCE, OPEN_READ, ("cannot read weights file '%s'", s->weights ? s->weights : "(unset)"), (NULL));
        return FALSE;
    }
    if (covahip_ctx_create((int)s->gpu_id, &s->ctx) != COVAHIP_OK) {
        g_free(blob);
        GST_ELEMENT_ERROR(s, RESOURCE, OPEN_READ, ("no HIP device %u", s->gpu_id), (NULL));
        return FALSE;
    }
    rc = covahip_blobnet_load(s->ctx, blob, len, s->h_mb, s->w_mb, BF_TIMESTEP, (int)s->batch_size);
    g_free(blob);
    s->max_frames = BF_TIMESTEP * (int)s->batch_size;
    if (rc == COVAHIP_OK) rc = covahip_pipe_create(s->ctx, (int)s->batch_size, s->max_frames, (int)s->max_boxes, BF_SLOTS, 0, &s->pipe);
    if (rc == COVAHIP_OK) rc = covahip_pipe_acquire(s->pipe, &s->slot, &s->pf, &s->pi);
    if (rc != COVAHIP_OK) {
        GST_ELEMENT_ERROR(s, LIBRARY, INIT, ("covahip: %s (%s)", covahip_strerror(rc), covahip_last_hip_error(s->ctx)), (NULL));
        return FALSE;
    }
    s->meta = g_new0(BfMeta, s->batch_size);
    s->failed = FALSE;
    return TRUE;
}

/* One share of a finished batch: the boxes of every stack of this share's src pads as bincode Vec<Bbox> buffers with
 * the frame's PTS (bboxcc/imp.rs:232-272).  A src pad is served by exactly one pusher thread, batches are pushed one
 * after the other: buffers leave every pad in order. */
static void bf_push_share(gpointer data, gpointer user) {
    BfPushTask *t = data;
    GstBlobNetFilter *s = t->s;
    GstFlowReturn ret = GST_FLOW_OK;
    covahip_bbox *bb = g_new(covahip_bbox, s->max_boxes ? s->max_boxes : 1);
    for (int i = 0; i < t->fl->n_stacks; i++) {
        const BfMeta *mt = &t->fl->meta[i];
        if (mt->pad % BF_PUSHERS != t->group) continue;
        const int n = t->offsets[i + 1] - t->offsets[i];
        int st = 0;
        GstMapInfo m;
        BfPad *p = g_ptr_array_index(s->pads, mt->pad);
        if (t->counts[i] > n) GST_WARNING_OBJECT(s, "frame with %d boxes truncated to max-boxes = %d", t->counts[i], n);
        covahip_boxes_to_bbox(t->boxes + t->offsets[i], n, bb);                /* Bbox::new, process.rs:47 */
        const gsize len = covahip_bbox_serialize_vec(bb, (size_t)n, NULL, 0, NULL);
        GstBuffer *b = gst_buffer_new_allocate(NULL, len, NULL);
        gst_buffer_map(b, &m, GST_MAP_WRITE);
        covahip_bbox_serialize_vec(bb, (size_t)n, m.data, m.size, &st);
        gst_buffer_unmap(b, &m);
        GST_BUFFER_PTS(b) = mt->pts;
        GST_BUFFER_DURATION(b) = mt->duration;
        const GstFlowReturn r = gst_pad_push(p->src, b);
        if (r != GST_FLOW_OK && r != GST_FLOW_NOT_LINKED && ret == GST_FLOW_OK) ret = r;
    }
    g_free(bb);
    g_mutex_lock(&s->push_lock);
    if (ret != GST_FLOW_OK) s->push_ret = ret;
    if (--s->push_left == 0) g_cond_broadcast(&s->push_cond);
    g_mutex_unlock(&s->push_lock);
    g_free(t);
}

/* Collector thread: takes the submitted batches in order, waits for their results (the GPU works on the next batch
 * meanwhile, the streaming threads fill the one after), fans the pushes out to the pusher threads, releases the slot. */
static gpointer bf_collector(gpointer data) {
    GstBlobNetFilter *s = data;
    g_mutex_lock(&s->lock);
    while (TRUE) {
        BfFlight *fl;
        const int32_t *counts = NULL, *offsets = NULL;
        const covahip_box *boxes = NULL;
        int rc;
        while (!s->stop && g_queue_is_empty(&s->flights)) g_cond_wait(&s->slot_cond, &s->lock);
        if (g_queue_is_empty(&s->flights)) break;   /* stop */
        fl = g_queue_pop_head(&s->flights);
        g_mutex_unlock(&s->lock);
        rc = covahip_pipe_wait(s->pipe, fl->slot);   /* blocks on the D2H event only: no pipe state is touched */
        g_mutex_lock(&s->lock);
        if (rc == COVAHIP_OK) rc = covahip_pipe_collect(s->pipe, fl->slot, &counts, &offsets, &boxes, NULL);
        g_mutex_unlock(&s->lock);
        if (rc == COVAHIP_OK) {
            g_mutex_lock(&s->push_lock);
            s->push_left = BF_PUSHERS;
            g_mutex_unlock(&s->push_lock);
            for (guint g = 0; g < BF_PUSHERS; g++) {
                BfPushTask *t = g_new(BfPushTask, 1);
                t->s = s; t->fl = fl; t->counts = counts; t->offsets = offsets; t->boxes = boxes; t->group = g;
                g_thread_pool_push(s->pushers, t, NULL);
            }
            g_mutex_lock(&s->push_lock);
            while (s->push_left > 0) g_cond_wait(&s->push_cond, &s->push_lock);
            g_mutex_unlock(&s->push_lock);
        } else {
            GST_ELEMENT_ERROR(s, LIBRARY, FAILED, ("covahip_pipe_collect: %s", covahip_strerror(rc)), (NULL));
        }
        g_mutex_lock(&s->lock);
        if (rc == COVAHIP_OK) covahip_pipe_release(s->pipe, fl->slot);
        s->frames_out += (guint64)fl->n_stacks;
        s->in_flight--;
        g_cond_broadcast(&s->slot_cond);
        g_free(fl->meta);
        g_free(fl);
    }
    g_mutex_unlock(&s->lock);
    return NULL;
}

/* Waits until no other thread is in the middle of a flush.  The caller re-evaluates its reason to flush afterwards:
 * the batch it saw full has been submitted by then. */
static void bf_wait_idle(GstBlobNetFilter *s) {
    while (s->flushing) g_cond_wait(&s->flush_cond, &s->lock);
}
/* Submits the batch being filled (if any) and opens the next slot.  Lock held; it is dropped while results of
 * older batches are pushed downstream, `flushing` keeps every other thread away from the slot state meanwhile. */
static GstFlowReturn bf_flush_locked(GstBlobNetFilter *s);
static GstFlowReturn bf_flush(GstBlobNetFilter *s) {   /* caller: lock held, s->flushing false (bf_wait_idle) */
    GstFlowReturn ret;
    s->flushing = TRUE;
    while (g_atomic_int_get(&s->pending) > 0) g_thread_yield();   /* frames of this batch still being copied in (microseconds) */
    ret = bf_flush_locked(s);
    s->flushing = FALSE;
    g_cond_broadcast(&s->flush_cond);
    return ret;
}
static GstFlowReturn bf_flush_locked(GstBlobNetFilter *s) {
    int rc;
    if (s->n_stacks > 0) {
        BfFlight *fl = g_new0(BfFlight, 1);
        rc = covahip_pipe_submit(s->pipe, s->slot, s->n_frames, s->n_stacks, (int)s->cc_threshold);
        if (rc != COVAHIP_OK) {
            g_free(fl);
            GST_ELEMENT_ERROR(s, LIBRARY, FAILED, ("covahip_pipe_submit: %s (%s)", covahip_strerror(rc), covahip_last_hip_error(s->ctx)), (NULL));
            return GST_FLOW_ERROR;
        }
        fl->slot = s->slot;
        fl->n_stacks = s->n_stacks;
        fl->meta = g_memdup(s->meta, sizeof(BfMeta) * (guint)s->n_stacks);
        g_queue_push_tail(&s->flights, fl);
        s->in_flight++;
        s->batches++;
        s->slot = -1;
        g_cond_broadcast(&s->slot_cond);
    } else if (s->slot >= 0) {
        return GST_FLOW_OK;   /* frames without a complete stack yet (stream start): keep filling the same slot */
    }
    /* the next slot: when all of them are in flight, wait for the collector to release one (the lock is dropped in the
     * wait; `flushing` keeps the other streaming threads out meanwhile) */
    while (s->slot < 0) {
        rc = covahip_pipe_acquire(s->pipe, &s->slot, &s->pf, &s->pi);
        if (rc == COVAHIP_ERR_OVERFLOW) {
            s->slot = -1;
            if (s->stop) return GST_FLOW_FLUSHING;
            g_cond_wait(&s->slot_cond, &s->lock);
        } else if (rc != COVAHIP_OK) {
            return GST_FLOW_ERROR;
        }
    }
    s->n_frames = s->n_stacks = 0;
    for (guint i = 0; i < s->pads->len; i++) {
        BfPad *p = g_ptr_array_index(s->pads, i);
        for (int k = 0; k < BF_TIMESTEP - 1; k++) p->hist_pos[k] = -1;
    }
    return s->push_ret;
}

/* One carrier frame of one stream.  Positions in the slot are reserved under the lock; the 32 KB copies into the pinned
 * slot run outside it, so the streaming threads of the N decoder branches copy in parallel. */
static GstFlowReturn bf_chain(GstPad *pad, GstObject *parent, GstBuffer *buf) {
    GstBlobNetFilter *s = (GstBlobNetFilter *)parent;
    BfPad *p = bf_pad_of(s, pad);
    GstFlowReturn ret = GST_FLOW_OK;
    GstBuffer *need[BF_TIMESTEP - 1] = {NULL, NULL, NULL};   /* history frames this slot does not hold yet */
    int need_pos[BF_TIMESTEP - 1] = {0, 0, 0};
    uint8_t *pf;
    int pos;
    gboolean full = FALSE;
    if (gst_buffer_get_size(buf) < s->frame_bytes || !s->frame_bytes) {
        GST_ELEMENT_ERROR(s, STREAM, FORMAT, ("carrier frame of %" G_GSIZE_FORMAT " bytes, need %" G_GSIZE_FORMAT " (caps set?)",
                                              gst_buffer_get_size(buf), s->frame_bytes), (NULL));
        gst_buffer_unref(buf);
        return GST_FLOW_ERROR;
    }
    g_mutex_lock(&s->lock);
    bf_wait_idle(s);
    if (!bf_ensure_model(s)) ret = GST_FLOW_ERROR;
    /* room for this frame and, at worst, three history frames of its stream; a full batch whose last copier has not
     * come back yet is flushed by whoever arrives first */
    while (ret == GST_FLOW_OK && (s->n_frames + BF_TIMESTEP > s->max_frames || s->n_stacks >= (int)s->batch_size)) {
        if (s->n_stacks == 0) break;   /* nothing to submit: the streams' warm-up frames alone fill the slot */
        ret = bf_flush(s);
        bf_wait_idle(s);
    }
    if (ret == GST_FLOW_OK && s->n_frames + BF_TIMESTEP > s->max_frames) {
        GST_ELEMENT_ERROR(s, CORE, FAILED, ("batch-size %u is too small for %u streams", s->batch_size, s->pads->len), (NULL));
        ret = GST_FLOW_ERROR;
    }
    if (ret != GST_FLOW_OK) {
        g_mutex_unlock(&s->lock);
        gst_buffer_unref(buf);
        return ret;
    }
    pf = s->pf;
    pos = s->n_frames++;
    if (p->n_seen >= BF_TIMESTEP - 1) {
        /* a complete stack: T = 0 is this frame, T = k the frame k steps back (metapreprocess/imp.rs:307-320) */
        int32_t *row = s->pi + (gsize)s->n_stacks * BF_TIMESTEP;
        row[0] = pos;
        for (int k = 0; k < BF_TIMESTEP - 1; k++) {
            if (p->hist_pos[k] < 0) {   /* first use in this slot: the frame came with an earlier batch */
                p->hist_pos[k] = s->n_frames++;
                need[k] = gst_buffer_ref(p->hist[k]);
                need_pos[k] = p->hist_pos[k];
            }
            row[k + 1] = p->hist_pos[k];
        }
        s->meta[s->n_stacks].pad = p->idx;
        s->meta[s->n_stacks].pts = GST_BUFFER_PTS(buf);
        s->meta[s->n_stacks].duration = GST_BUFFER_DURATION(buf);
        if (s->n_stacks == 0) {
            s->first_us = g_get_monotonic_time();
            if (s->timeout_us) g_cond_signal(&s->cond);   /* the timeout thread starts its clock for this batch */
        }
        s->n_stacks++;
        full = s->n_stacks >= (int)s->batch_size;
    }
    /* history: newest first; the element keeps references, the frame bytes are copied once, into the slot */
    if (p->hist[BF_TIMESTEP - 2]) gst_buffer_unref(p->hist[BF_TIMESTEP - 2]);
    for (int k = BF_TIMESTEP - 2; k > 0; k--) { p->hist[k] = p->hist[k - 1]; p->hist_pos[k] = p->hist_pos[k - 1]; }
    p->hist[0] = buf;                 /* takes over the reference the chain function was given */
    p->hist_pos[0] = pos;
    p->n_seen++;
    g_atomic_int_inc(&s->pending);
    g_mutex_unlock(&s->lock);

    gst_buffer_extract(buf, 0, pf + (gsize)pos * s->frame_bytes, s->frame_bytes);   /* metapreprocess copies the same bytes (imp.rs:311-312) */
    for (int k = 0; k < BF_TIMESTEP - 1; k++)
        if (need[k]) {
            gst_buffer_extract(need[k], 0, pf + (gsize)need_pos[k] * s->frame_bytes, s->frame_bytes);
            gst_buffer_unref(need[k]);
        }

    g_atomic_int_add(&s->pending, -1);
    if (full) {   /* the frame that completed the batch submits it, unless a later arrival has done so already */
        g_mutex_lock(&s->lock);
        bf_wait_idle(s);
        if (s->n_stacks >= (int)s->batch_size) ret = bf_flush(s);
        g_mutex_unlock(&s->lock);
    }
    return ret;
}

/* nvstreammux's batched-push-timeout: a batch that does not fill up leaves after this long anyway */
static gpointer bf_timer(gpointer data) {
    GstBlobNetFilter *s = data;
    g_mutex_lock(&s->lock);
    while (!s->stop) {
        if (s->pipe && s->n_stacks > 0 && s->timeout_us > 0) {
            const gint64 due = s->first_us + (gint64)s->timeout_us;
            if (s->flushing) bf_wait_idle(s);
            else if (g_get_monotonic_time() >= due) bf_flush(s);
            else g_cond_wait_until(&s->cond, &s->lock, due);
        } else {
            g_cond_wait_until(&s->cond, &s->lock, g_get_monotonic_time() + 50 * G_TIME_SPAN_MILLISECOND);
        }
    }
    g_mutex_unlock(&s->lock);
    return NULL;
}

static gboolean bf_sink_event(GstPad *pad, GstObject *parent, GstEvent *ev) {
    GstBlobNetFilter *s = (GstBlobNetFilter *)parent;
    BfPad *p = bf_pad_of(s, pad);
    switch (GST_EVENT_TYPE(ev)) {
    case GST_EVENT_CAPS: {
        GstCaps *caps, *out;
        GstVideoInfo vi;
        gboolean ok;
        gst_event_parse_caps(ev, &caps);
        if (!gst_video_info_from_caps(&vi, caps)) { gst_event_unref(ev); return FALSE; }
        g_mutex_lock(&s->lock);
        {   /* metapreprocess' caps arithmetic (imp.rs:262-268): macroblock grid = picture / 16 */
            const gint w = GST_VIDEO_INFO_WIDTH(&vi) / 16, h = GST_VIDEO_INFO_HEIGHT(&vi) / 16;
            ok = (s->w_mb == 0 && s->h_mb == 0) || (s->w_mb == w && s->h_mb == h);
            if (ok) { s->w_mb = w; s->h_mb = h; s->frame_bytes = (gsize)w * h * 4; }
        }
        g_mutex_unlock(&s->lock);
        if (!ok) { GST_ELEMENT_ERROR(s, CORE, NEGOTIATION, ("all streams of a blobnetfilter must have one picture size"), (NULL)); gst_event_unref(ev); return FALSE; }
        out = gst_caps_new_simple("bbox", "width", G_TYPE_INT, s->w_mb, "height", G_TYPE_INT, s->h_mb, NULL);   /* bboxcc/imp.rs:199-211 */
        gst_event_unref(ev);
        ok = gst_pad_push_event(p->src, gst_event_new_caps(out));
        gst_caps_unref(out);
        return ok;
    }
    case GST_EVENT_EOS: {
        gboolean all = TRUE;
        g_mutex_lock(&s->lock);
        p->eos = TRUE;
        for (guint i = 0; i < s->pads->len; i++) all = all && ((BfPad *)g_ptr_array_index(s->pads, i))->eos;
        if (all && s->pipe) {   /* the last stream ended: the open batch and everything in flight leave, then EOS on every src pad */
            bf_wait_idle(s);
            bf_flush(s);
            while (s->in_flight > 0) g_cond_wait(&s->slot_cond, &s->lock);   /* every result has been pushed */
        }
        g_mutex_unlock(&s->lock);
        gst_event_unref(ev);
        if (all)
            for (guint i = 0; i < s->pads->len; i++) gst_pad_push_event(((BfPad *)g_ptr_array_index(s->pads, i))->src, gst_event_new_eos());
        return TRUE;
    }
    case GST_EVENT_STREAM_START:
    case GST_EVENT_SEGMENT:
        return gst_pad_push_event(p->src, ev);
    default:
        return gst_pad_event_default(pad, parent, ev);
    }
}

static GstPad *bf_request_new_pad(GstElement *e, GstPadTemplate *templ, const gchar *name, const GstCaps *caps) {
    GstBlobNetFilter *s = (GstBlobNetFilter *)e;
    BfPad *p = g_new0(BfPad, 1);
    gchar *sn, *rn;
    g_mutex_lock(&s->lock);
    p->idx = s->pads->len;
    sn = g_strdup_printf("sink_%u", p->idx);
    rn = g_strdup_printf("src_%u", p->idx);
    p->sink = gst_pad_new_from_template(templ, sn);
    p->src = gst_pad_new_from_template(gst_element_class_get_pad_template(GST_ELEMENT_GET_CLASS(e), "src_%u"), rn);
    for (int k = 0; k < BF_TIMESTEP - 1; k++) p->hist_pos[k] = -1;
    g_ptr_array_add(s->pads, p);
    g_mutex_unlock(&s->lock);
    g_free(sn);
    g_free(rn);
    gst_pad_set_element_private(p->sink, p);
    gst_pad_set_chain_function(p->sink, bf_chain);
    gst_pad_set_event_function(p->sink, bf_sink_event);
    gst_pad_set_active(p->src, TRUE);
    gst_pad_set_active(p->sink, TRUE);
    gst_element_add_pad(e, p->src);
    gst_element_add_pad(e, p->sink);
    return p->sink;
}

static GstStateChangeReturn bf_change_state(GstElement *e, GstStateChange t) {
    GstBlobNetFilter *s = (GstBlobNetFilter *)e;
    GstStateChangeReturn r;
    if (t == GST_STATE_CHANGE_READY_TO_PAUSED) {
        g_mutex_lock(&s->lock);
        s->stop = FALSE;
        s->push_ret = GST_FLOW_OK;
        if (!s->pushers) s->pushers = g_thread_pool_new(bf_push_share, s, BF_PUSHERS, FALSE, NULL);
        if (!s->timer) s->timer = g_thread_new("blobnetfilter-timeout", bf_timer, s);
        if (!s->collector) s->collector = g_thread_new("blobnetfilter-collect", bf_collector, s);
        g_mutex_unlock(&s->lock);
    }
    r = GST_ELEMENT_CLASS(gst_blobnetfilter_parent_class)->change_state(e, t);
    if (t == GST_STATE_CHANGE_PAUSED_TO_READY) {
        GThread *th, *tc;
        g_mutex_lock(&s->lock);
        s->stop = TRUE;
        g_cond_broadcast(&s->cond);
        g_cond_broadcast(&s->slot_cond);
        g_cond_broadcast(&s->flush_cond);
        th = s->timer;
        tc = s->collector;
        s->timer = s->collector = NULL;
        g_mutex_unlock(&s->lock);
        if (th) g_thread_join(th);
        if (tc) g_thread_join(tc);   /* drains the batches still queued */
        if (s->pushers) { g_thread_pool_free(s->pushers, FALSE, TRUE); s->pushers = NULL; }
    }
    return r;
}

static void bf_set_property(GObject *o, guint id, const GValue *v, GParamSpec *ps) {
    GstBlobNetFilter *s = (GstBlobNetFilter *)o;
    g_mutex_lock(&s->lock);
    switch (id) {
    case BF_PROP_WEIGHTS: g_free(s->weights); s->weights = g_value_dup_string(v); break;
    case BF_PROP_GPU: s->gpu_id = g_value_get_uint(v); break;
    case BF_PROP_BATCH: if (!s->pipe) s->batch_size = g_value_get_uint(v); break;
    case BF_PROP_TIMEOUT: s->timeout_us = g_value_get_uint64(v); break;
    case BF_PROP_CC: s->cc_threshold = g_value_get_uint(v); break;
    case BF_PROP_MAXBOXES: if (!s->pipe) s->max_boxes = g_value_get_uint(v); break;
    default: G_OBJECT_WARN_INVALID_PROPERTY_ID(o, id, ps);
    }
    g_mutex_unlock(&s->lock);
}
static void bf_get_property(GObject *o, guint id, GValue *v, GParamSpec *ps) {
    GstBlobNetFilter *s = (GstBlobNetFilter *)o;
    switch (id) {
    case BF_PROP_WEIGHTS: g_value_set_string(v, s->weights); break;
    case BF_PROP_GPU: g_value_set_uint(v, s->gpu_id); break;
    case BF_PROP_BATCH: g_value_set_uint(v, s->batch_size); break;
    case BF_PROP_TIMEOUT: g_value_set_uint64(v, s->timeout_us); break;
    case BF_PROP_CC: g_value_set_uint(v, s->cc_threshold); break;
    case BF_PROP_MAXBOXES: g_value_set_uint(v, s->max_boxes); break;
    case BF_PROP_BATCHES: g_value_set_uint64(v, s->batches); break;
    default: G_OBJECT_WARN_INVALID_PROPERTY_ID(o, id, ps);
    }
}
static void bf_finalize(GObject *o) {
    GstBlobNetFilter *s = (GstBlobNetFilter *)o;
    BfFlight *fl;
    while ((fl = g_queue_pop_head(&s->flights)) != NULL) { g_free(fl->meta); g_free(fl); }
    if (s->pipe) covahip_pipe_destroy(s->pipe);
    if (s->ctx) covahip_ctx_destroy(s->ctx);
    for (guint i = 0; i < s->pads->len; i++) {
        BfPad *p = g_ptr_array_index(s->pads, i);
        for (int k = 0; k < BF_TIMESTEP - 1; k++)
            if (p->hist[k]) gst_buffer_unref(p->hist[k]);
        g_free(p);
    }
    g_ptr_array_free(s->pads, TRUE);
    g_free(s->meta);
    g_free(s->weights);
    g_mutex_clear(&s->lock);
    g_mutex_clear(&s->push_lock);
    g_cond_clear(&s->cond);
    g_cond_clear(&s->flush_cond);
    g_cond_clear(&s->slot_cond);
    g_cond_clear(&s->push_cond);
    G_OBJECT_CLASS(gst_blobnetfilter_parent_class)->finalize(o);
}
static void gst_blobnetfilter_init(GstBlobNetFilter *s) {
    g_mutex_init(&s->lock);
    g_mutex_init(&s->push_lock);
    g_cond_init(&s->cond);
    g_cond_init(&s->flush_cond);
    g_cond_init(&s->slot_cond);
    g_cond_init(&s->push_cond);
    g_queue_init(&s->flights);
    s->pads = g_ptr_array_new();
    s->batch_size = 128;       /* experiment/cova/config.yaml:30-35 */
    s->timeout_us = 40000;     /* nvstreammux batched-push-timeout of the reference pipeline (pipeline.py:146-164) */
    s->cc_threshold = 30;      /* bboxcc default (imp.rs:16) */
    s->max_boxes = 256;
    s->slot = -1;
}
static void gst_blobnetfilter_class_init(GstBlobNetFilterClass *k) {
    GObjectClass *g = G_OBJECT_CLASS(k);
    GstElementClass *e = GST_ELEMENT_CLASS(k);
    g->set_property = bf_set_property;
    g->get_property = bf_get_property;
    g->finalize = bf_finalize;
    e->request_new_pad = bf_request_new_pad;
    e->change_state = bf_change_state;
    g_object_class_install_property(g, BF_PROP_WEIGHTS, g_param_spec_string("model-weights-file", "Weights",
        "BlobNet weight blob (cova_amd/weights.py format)", NULL, G_PARAM_READWRITE | GST_PARAM_MUTABLE_READY));
    g_object_class_install_property(g, BF_PROP_GPU, g_param_spec_uint("gpu-id", "GPU id", "HIP device to run on", 0, 15, 0,
        G_PARAM_READWRITE | GST_PARAM_MUTABLE_READY));
    g_object_class_install_property(g, BF_PROP_BATCH, g_param_spec_uint("batch-size", "Batch size",
        "Maximum number of frames (stacks) per GPU batch, over all streams (nvstreammux batch-size)", 1, 4096, 128,
        G_PARAM_READWRITE | GST_PARAM_MUTABLE_READY));
    g_object_class_install_property(g, BF_PROP_TIMEOUT, g_param_spec_uint64("batched-push-timeout", "Batched push timeout",
        "Microseconds after the first frame of a batch after which an incomplete batch is processed (0: wait for a full batch)",
        0, G_MAXUINT64, 40000, G_PARAM_READWRITE | GST_PARAM_MUTABLE_PLAYING));
    g_object_class_install_property(g, BF_PROP_CC, g_param_spec_uint("cc-threshold", "Threshold of Connected Components",
        "Connected component with area smaller than the threshold is ignored", 0, G_MAXUINT, 30, G_PARAM_READWRITE | GST_PARAM_MUTABLE_PLAYING));
    g_object_class_install_property(g, BF_PROP_MAXBOXES, g_param_spec_uint("max-boxes", "Max boxes",
        "Boxes kept per frame (more are dropped with a warning)", 1, 65536, 256, G_PARAM_READWRITE | GST_PARAM_MUTABLE_READY));
    g_object_class_install_property(g, BF_PROP_BATCHES, g_param_spec_uint64("batches", "Batches", "GPU batches submitted so far", 0,
        G_MAXUINT64, 0, G_PARAM_READABLE));
    gst_element_class_set_static_metadata(e, "BlobNet compressed-domain filter (batched)", "Filter/Video",
        "Batches the carrier frames of N streams and runs stacking + BlobNet + connected components on MI355X "
        "(replaces metapreprocess ! nvstreammux ! nvinfer ! nvstreamdemux ! maskcopy ! bboxcc)", "covahip");
    gst_element_class_add_pad_template(e, gst_pad_template_new("sink_%u", GST_PAD_SINK, GST_PAD_REQUEST,
        gst_caps_from_string("video/x-raw, format=(string)I420, width=(int)[16,2147483647], height=(int)[16,2147483647]")));
    gst_element_class_add_pad_template(e, gst_pad_template_new("src_%u", GST_PAD_SRC, GST_PAD_SOMETIMES,
        gst_caps_from_string("bbox, width=(int)[0,2147483647], height=(int)[0,2147483647]")));
}
GType gst_blobnetfilter_get_type_public(void) { return gst_blobnetfilter_get_type(); }

/* ===================================================================== maskcopy */
typedef struct {
    GstBaseTransform parent;
    guint unique_id, gpu_id, timestep;
} GstMaskCopy;
typedef struct { GstBaseTransformClass parent_class; } GstMaskCopyClass;
G_DEFINE_TYPE(GstMaskCopy, gst_maskcopy, GST_TYPE_BASE_TRANSFORM)
enum { MC_PROP_0, MC_PROP_UNIQUE_ID, MC_PROP_GPU_ID, MC_PROP_TIMESTEP };

static void mc_set_property(GObject *o, guint id, const GValue *v, GParamSpec *ps) {
    GstMaskCopy *s = (GstMaskCopy *)o;
    if (id == MC_PROP_UNIQUE_ID) s->unique_id = g_value_get_uint(v);
    else if (id == MC_PROP_GPU_ID) s->gpu_id = g_value_get_uint(v);
    else if (id == MC_PROP_TIMESTEP) s->timestep = g_value_get_uint(v);
    else G_OBJECT_WARN_INVALID_PROPERTY_ID(o, id, ps);
}
static void mc_get_property(GObject *o, guint id, GValue *v, GParamSpec *ps) {
    GstMaskCopy *s = (GstMaskCopy *)o;
    if (id == MC_PROP_UNIQUE_ID) g_value_set_uint(v, s->unique_id);
    else if (id == MC_PROP_GPU_ID) g_value_set_uint(v, s->gpu_id);
    else if (id == MC_PROP_TIMESTEP) g_value_set_uint(v, s->timestep);
    else G_OBJECT_WARN_INVALID_PROPERTY_ID(o, id, ps);
}
static GstFlowReturn mc_transform_ip(GstBaseTransform *bt, GstBuffer *buf) { return GST_FLOW_OK; }
static void gst_maskcopy_init(GstMaskCopy *s) {
    s->unique_id = 0; s->gpu_id = 0; s->timestep = 4;   /* gstmaskcopy.cpp:102-125 */
    gst_base_transform_set_passthrough(GST_BASE_TRANSFORM(s), TRUE);
}
static void gst_maskcopy_class_init(GstMaskCopyClass *k) {
    GObjectClass *g = G_OBJECT_CLASS(k);
    GstElementClass *e = GST_ELEMENT_CLASS(k);
    GstBaseTransformClass *b = GST_BASE_TRANSFORM_CLASS(k);
    g->set_property = mc_set_property;
    g->get_property = mc_get_property;
    g_object_class_install_property(g, MC_PROP_UNIQUE_ID, g_param_spec_uint("unique-id", "Unique ID",
        "Unique ID for the element (the nvinfer instance whose output is copied, in the reference)", 0, G_MAXUINT, 0, G_PARAM_READWRITE));
    g_object_class_install_property(g, MC_PROP_GPU_ID, g_param_spec_uint("gpu-id", "Set GPU Device ID", "Set GPU Device ID", 0, G_MAXUINT, 0,
        G_PARAM_READWRITE | GST_PARAM_MUTABLE_READY));
    g_object_class_install_property(g, MC_PROP_TIMESTEP, g_param_spec_uint("timestep", "Timestep",
        "Number of stacked frames per inference input (the mask is height / timestep rows)", 1, G_MAXUINT, 4, G_PARAM_READWRITE | GST_PARAM_MUTABLE_READY));
    gst_element_class_set_static_metadata(e, "Mask copy", "Filter/Video",
        "Hands the GRAY8 {0,1} mask of the BlobNet stage downstream (covahip: blobnetinfer already emits it)", "covahip");
    gst_element_class_add_pad_template(e, gst_pad_template_new("src", GST_PAD_SRC, GST_PAD_ALWAYS,
        gst_caps_from_string("video/x-raw, format=(string)GRAY8, width=(int)[1,2147483647], height=(int)[1,2147483647]")));
    gst_element_class_add_pad_template(e, gst_pad_template_new("sink", GST_PAD_SINK, GST_PAD_ALWAYS,
        gst_caps_from_string("video/x-raw, format=(string)GRAY8, width=(int)[1,2147483647], height=(int)[1,2147483647]")));
    b->transform_ip = mc_transform_ip;
}
GType gst_maskcopy_get_type_public(void) { return gst_maskcopy_get_type(); }
