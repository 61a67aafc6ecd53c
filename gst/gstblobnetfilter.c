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
 * Threads: the N streaming threads take their positions in the open slot with one compare-and-swap (no lock) and copy
 * their frame in; whoever completes a batch hands the slot to the SUBMITTER thread (the HIP calls); the COLLECTOR thread
 * waits for results in order and hands every finished batch to eight PUSHER threads (src pad i is always served by
 * pusher i % 8, batches in order, so every pad sees its frames in order; the last share of a batch releases the slot);
 * a TIMER thread implements batched-push-timeout.  The element mutex covers model load, batch hand-over and waiting for
 * a free slot only.
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
#define BF_LANES 3   /* batches in flight on the GPU (round 6: three; the pipe plans its copy streams for this count, pipe.hip) */

/* ===================================================================== blobnetfilter */
typedef struct {
    GstPad *sink, *src;
    guint idx;
    GstBuffer *hist[BF_TIMESTEP - 1]; /* the last three carrier frames (references, no copies): [0] newest */
    gint hist_pos[BF_TIMESTEP - 1];  /* their position in the slot being filled, -1 = not copied into it yet */
    guint n_seen;
    gboolean eos, caps_sent;
    gboolean records;                 /* sink caps application/x-cova-records: buffers are packed two-byte records already */
    gint src_ret;                     /* last GstFlowReturn of a push on src that was neither OK nor NOT_LINKED (atomic) */
} BfPad;

typedef struct { guint pad; GstClockTime pts, duration; } BfMeta;
typedef struct {   /* one batch on its way: filled slot -> submitter -> collector -> the pusher threads */
    int slot;
    int n_frames, n_stacks;
    guint cc_threshold;
    BfMeta *meta;
    const int32_t *counts, *offsets;   /* the slot's results (valid until the slot is released) */
    const covahip_box *boxes;
    volatile gint shares_left;         /* pusher threads that have not pushed their share yet: the last one releases the slot */
} BfFlight;
#define BF_PUSHERS 8
typedef struct _GstBlobNetFilter GstBlobNetFilter;
typedef struct { GstBlobNetFilter *s; guint group; GAsyncQueue *q; GThread *th; } BfPusher;

struct _GstBlobNetFilter {
    GstElement parent;
    GMutex lock;
    GCond cond, flush_cond;
    /* Reservation word of the slot being filled: frames taken (bits 0-20), stacks taken (21-41), a generation that
     * changes with every slot (42-62) and the "flush under way" bit (63).  A streaming thread takes its positions with one
     * compare-and-swap; a sleeping mutex there is a convoy (measured: 250 k frames/s with 4 to 64 threads alike) and a
     * spin lock still costs 5 us per frame with 8 threads on it. */
    guint64 state __attribute__((aligned(64)));
    guint64 done __attribute__((aligned(64)));   /* frames whose bytes (and table rows) are in the slot; a flush waits for done == frames taken */
    gboolean flushing __attribute__((aligned(64)));   /* lock held: same as the bit in `state`, for the threads that wait on flush_cond */
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
    int max_frames;
    BfMeta *meta;
    gint64 first_us;
    GMutex pipe_lock;         /* covahip_pipe_* calls (acquire / submit / collect / release) are one at a time */
    GQueue submit_q;          /* BfFlight*: filled batches waiting for the submitter thread */
    GThread *submitter;
    GQueue flights;           /* BfFlight*, oldest first: submitted, not yet taken by the collector */
    guint in_flight;          /* submitted batches whose results have not all been pushed yet */
    GCond slot_cond;          /* a slot was released / a flight was queued / all flights are done */
    GThread *timer, *collector;
    BfPusher pusher[BF_PUSHERS];   /* src pad i is served by pusher i % BF_PUSHERS, batches in order */
    volatile gint pushers_us;
    GstFlowReturn push_ret;
    gboolean stop, failed;
    guint64 batches, frames_out;
    gint64 slot_wait_us, gpu_wait_us, push_us;   /* where the time goes (read-only property "timing") */
    volatile gint chain_slow_us, chain_flush_us, chain_slow_n;   /* summed over the streaming threads (slow paths only) */
};
typedef struct { GstElementClass parent_class; } GstBlobNetFilterClass;
G_DEFINE_TYPE(GstBlobNetFilter, gst_blobnetfilter, GST_TYPE_ELEMENT)
enum { BF_PROP_0, BF_PROP_WEIGHTS, BF_PROP_GPU, BF_PROP_BATCH, BF_PROP_TIMEOUT, BF_PROP_CC, BF_PROP_MAXBOXES, BF_PROP_BATCHES, BF_PROP_TIMING };

#define BF_ST_BITS 21
#define BF_ST_MASK ((1ull << BF_ST_BITS) - 1)
#define BF_ST_FRAMES(x) ((int)((x) & BF_ST_MASK))
#define BF_ST_STACKS(x) ((int)(((x) >> BF_ST_BITS) & BF_ST_MASK))
#define BF_ST_ONE_STACK (1ull << BF_ST_BITS)
#define BF_ST_ONE_GEN (1ull << (2 * BF_ST_BITS))
#define BF_ST_FLUSH (1ull << 63)
static inline guint64 bf_state(GstBlobNetFilter *s) { return __atomic_load_n(&s->state, __ATOMIC_ACQUIRE); }

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
    /* two batches in flight: every pipe slot owns its output buffers (the ctx default is one lane, include/covahip.h) */
    rc = covahip_ctx_set_lanes(s->ctx, BF_LANES);
    if (rc == COVAHIP_OK) rc = covahip_blobnet_load(s->ctx, blob, len, s->h_mb, s->w_mb, BF_TIMESTEP, (int)s->batch_size);
    g_free(blob);
    if (rc == COVAHIP_OK) rc = covahip_pipe_create(s->ctx, (int)s->batch_size, BF_TIMESTEP * (int)s->batch_size, (int)s->max_boxes, BF_SLOTS, 0, &s->pipe);
    if (rc == COVAHIP_OK) rc = covahip_pipe_set_packed(s->pipe, 1);
    /* the collector sleeps while it waits for the GPU instead of spinning on the completion signal (BLOBNETFILTER_SPIN=1: the old way) */
    if (rc == COVAHIP_OK && !g_getenv("BLOBNETFILTER_SPIN")) rc = covahip_pipe_set_blocking_wait(s->pipe, 1);
    if (rc == COVAHIP_OK) rc = covahip_pipe_acquire(s->pipe, &s->slot, &s->pf, &s->pi);
    if (rc != COVAHIP_OK) {
        GST_ELEMENT_ERROR(s, LIBRARY, INIT, ("covahip: %s (%s)", covahip_strerror(rc), covahip_last_hip_error(s->ctx)), (NULL));
        return FALSE;
    }
    s->meta = g_new0(BfMeta, s->batch_size);
    s->failed = FALSE;
    __atomic_store_n(&s->done, 0, __ATOMIC_RELEASE);
    __atomic_store_n(&s->state, 0, __ATOMIC_RELEASE);
    __atomic_store_n(&s->max_frames, BF_TIMESTEP * (int)s->batch_size, __ATOMIC_RELEASE);   /* last: opens the lock-free path */
    return TRUE;
}

/* One share of a finished batch: the boxes of every stack of this pusher's src pads as bincode Vec<Bbox> buffers with
 * the frame's PTS (bboxcc/imp.rs:232-272).  A src pad is served by exactly one pusher thread and every pusher takes the
 * batches in order: buffers leave every pad in order.  Whoever pushes the last share of a batch releases its slot. */
static void bf_push_share(GstBlobNetFilter *s, BfFlight *fl, guint group, covahip_bbox **bbp, gsize *bb_cap) {
    GstFlowReturn ret = GST_FLOW_OK;
    const gint64 t0 = g_get_monotonic_time();
    for (int i = 0; i < fl->n_stacks; i++) {
        const BfMeta *mt = &fl->meta[i];
        if (mt->pad % BF_PUSHERS != group) continue;
        const int n = fl->offsets[i + 1] - fl->offsets[i];
        int st = 0;
        GstMapInfo m;
        BfPad *p = g_ptr_array_index(s->pads, mt->pad);
        covahip_bbox *bb;
        /* a src pad whose downstream has finished (EOS) takes no more buffers; the other streams go on */
        if (g_atomic_int_get(&p->src_ret) == GST_FLOW_EOS) continue;
        if ((gsize)n > *bb_cap) {   /* sized by what the batch holds, whatever max-boxes was when the thread started */
            g_free(*bbp);
            *bb_cap = (gsize)n + 64;
            *bbp = g_new(covahip_bbox, *bb_cap);
        }
        bb = *bbp;
        if (fl->counts[i] > n) GST_WARNING_OBJECT(s, "frame with %d boxes truncated to max-boxes = %d", fl->counts[i], n);
        covahip_boxes_to_bbox(fl->boxes + fl->offsets[i], n, bb);                /* Bbox::new, process.rs:47 */
        const gsize len = covahip_bbox_serialize_vec(bb, (size_t)n, NULL, 0, NULL);
        GstBuffer *b = gst_buffer_new_allocate(NULL, len, NULL);
        gst_buffer_map(b, &m, GST_MAP_WRITE);
        covahip_bbox_serialize_vec(bb, (size_t)n, m.data, m.size, &st);
        gst_buffer_unmap(b, &m);
        GST_BUFFER_PTS(b) = mt->pts;
        GST_BUFFER_DURATION(b) = mt->duration;
        const GstFlowReturn r = gst_pad_push(p->src, b);
        if (r != GST_FLOW_OK && r != GST_FLOW_NOT_LINKED) {
            /* EOS / FLUSHING concern this stream only (its own chain function returns them); errors stop the element.
             * A FLUSHING that comes from the element's own shutdown (pads already deactivated while the queued batches
             * drain, PAUSED -> READY) is not a state of the stream and is not remembered. */
            if (!(r == GST_FLOW_FLUSHING && s->stop)) g_atomic_int_set(&p->src_ret, (gint)r);
            if (r != GST_FLOW_EOS && r != GST_FLOW_FLUSHING && ret == GST_FLOW_OK) ret = r;
        } else if (g_atomic_int_get(&p->src_ret) == GST_FLOW_FLUSHING) {
            g_atomic_int_set(&p->src_ret, GST_FLOW_OK);   /* the flush is over */
        }
    }
    g_atomic_int_add(&s->pushers_us, (gint)(g_get_monotonic_time() - t0));
    const gboolean last = g_atomic_int_dec_and_test(&fl->shares_left);
    if (last) {
        g_mutex_lock(&s->pipe_lock);
        covahip_pipe_release(s->pipe, fl->slot);
        g_mutex_unlock(&s->pipe_lock);
    }
    if (last || ret != GST_FLOW_OK) {
        g_mutex_lock(&s->lock);
        if (ret != GST_FLOW_OK) s->push_ret = ret;
        if (last) {
            s->frames_out += (guint64)fl->n_stacks;
            s->in_flight--;
            g_cond_broadcast(&s->slot_cond);
        }
        g_mutex_unlock(&s->lock);
    }
    if (last) {
        g_free(fl->meta);
        g_free(fl);
    }
}
static gpointer bf_pusher(gpointer data) {
    BfPusher *pu = data;
    GstBlobNetFilter *s = pu->s;
    covahip_bbox *bb = NULL;
    gsize bb_cap = 0;
    gpointer item;
    while ((item = g_async_queue_pop(pu->q)) != (gpointer)pu)   /* the pusher's own address is the stop token */
        bf_push_share(s, item, pu->group, &bb, &bb_cap);
    g_free(bb);
    return NULL;
}

/* Submitter thread: enqueues the GPU work of the filled batches, in order. */
static gpointer bf_submitter(gpointer data) {
    GstBlobNetFilter *s = data;
    g_mutex_lock(&s->lock);
    while (TRUE) {
        BfFlight *fl;
        int rc;
        while (!s->stop && g_queue_is_empty(&s->submit_q)) g_cond_wait(&s->slot_cond, &s->lock);
        if (g_queue_is_empty(&s->submit_q)) break;   /* stop */
        fl = g_queue_pop_head(&s->submit_q);
        g_mutex_unlock(&s->lock);
        g_mutex_lock(&s->pipe_lock);
        rc = covahip_pipe_submit(s->pipe, fl->slot, fl->n_frames, fl->n_stacks, (int)fl->cc_threshold);
        if (rc != COVAHIP_OK) covahip_pipe_abort(s->pipe, fl->slot);   /* the slot goes back to the pool */
        g_mutex_unlock(&s->pipe_lock);
        if (rc != COVAHIP_OK)
            GST_ELEMENT_ERROR(s, LIBRARY, FAILED, ("covahip_pipe_submit: %s (%s)", covahip_strerror(rc), covahip_last_hip_error(s->ctx)), (NULL));
        g_mutex_lock(&s->lock);
        if (rc != COVAHIP_OK) {   /* nothing will come back for this batch */
            s->push_ret = GST_FLOW_ERROR;
            s->in_flight--;
            g_free(fl->meta);
            g_free(fl);
        } else {
            g_queue_push_tail(&s->flights, fl);
        }
        g_cond_broadcast(&s->slot_cond);
    }
    g_mutex_unlock(&s->lock);
    return NULL;
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
        while (g_queue_is_empty(&s->flights) && !(s->stop && s->in_flight == 0)) g_cond_wait(&s->slot_cond, &s->lock);
        if (g_queue_is_empty(&s->flights)) break;   /* stop, and nothing queued or being submitted any more */
        fl = g_queue_pop_head(&s->flights);
        g_mutex_unlock(&s->lock);
        const gint64 tw0 = g_get_monotonic_time();
        rc = covahip_pipe_wait(s->pipe, fl->slot);   /* blocks on the D2H event only: no pipe state is touched */
        const gint64 tw1 = g_get_monotonic_time();
        g_mutex_lock(&s->pipe_lock);
        if (rc == COVAHIP_OK) rc = covahip_pipe_collect(s->pipe, fl->slot, &counts, &offsets, &boxes, NULL);
        g_mutex_unlock(&s->pipe_lock);
        if (rc != COVAHIP_OK) GST_ELEMENT_ERROR(s, LIBRARY, FAILED, ("covahip_pipe_collect: %s", covahip_strerror(rc)), (NULL));
        g_mutex_lock(&s->lock);
        s->gpu_wait_us += tw1 - tw0;
        s->push_us += g_get_monotonic_time() - tw1;
        if (rc == COVAHIP_OK) {
            /* the pusher threads take it from here; this thread goes on to wait for the next batch */
            fl->counts = counts; fl->offsets = offsets; fl->boxes = boxes;
            g_atomic_int_set(&fl->shares_left, BF_PUSHERS);
            for (guint g = 0; g < BF_PUSHERS; g++) g_async_queue_push(s->pusher[g].q, fl);
        } else {
            s->push_ret = GST_FLOW_ERROR;
            s->in_flight--;
            g_cond_broadcast(&s->slot_cond);
            g_free(fl->meta);
            g_free(fl);
        }
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
static GstFlowReturn bf_flush(GstBlobNetFilter *s) {   /* caller: lock held, s->flushing false (bf_wait_idle) */
    GstFlowReturn ret = GST_FLOW_OK;
    int rc;
    /* from here on nobody reserves in this slot; the frames taken so far are the batch */
    guint64 st = __atomic_fetch_or(&s->state, BF_ST_FLUSH, __ATOMIC_ACQ_REL) & ~BF_ST_FLUSH;
    s->flushing = TRUE;
    while (__atomic_load_n(&s->done, __ATOMIC_ACQUIRE) != (guint64)BF_ST_FRAMES(st)) g_thread_yield();   /* copies still running (microseconds) */
    if (BF_ST_STACKS(st) > 0 && s->slot >= 0) {
        /* hand the filled slot to the submitter thread: the HIP calls of a submission (copies, a dozen launches) take
         * ~0.2 ms and must not hold up the streaming threads */
        BfFlight *fl = g_new0(BfFlight, 1);
        fl->slot = s->slot;
        fl->n_frames = BF_ST_FRAMES(st);
        fl->n_stacks = BF_ST_STACKS(st);
        fl->cc_threshold = s->cc_threshold;
        fl->meta = g_memdup(s->meta, sizeof(BfMeta) * (guint)fl->n_stacks);
        g_queue_push_tail(&s->submit_q, fl);
        s->in_flight++;
        s->batches++;
        s->slot = -1;
        g_cond_broadcast(&s->slot_cond);
    }
    /* the next slot: when all of them are in flight, wait for a pusher to release one (the lock is dropped in the wait;
     * `flushing` keeps the other streaming threads out meanwhile).  Frames without a complete stack yet (stream start)
     * stay where they are and the same slot goes on being filled. */
    while (s->slot < 0 && ret == GST_FLOW_OK) {
        if (s->push_ret != GST_FLOW_OK) { ret = s->push_ret; break; }   /* a failed element waits for nothing */
        g_mutex_lock(&s->pipe_lock);
        rc = covahip_pipe_acquire(s->pipe, &s->slot, &s->pf, &s->pi);
        g_mutex_unlock(&s->pipe_lock);
        if (rc == COVAHIP_OK) {
            for (guint i = 0; i < s->pads->len; i++) {
                BfPad *p = g_ptr_array_index(s->pads, i);
                for (int k = 0; k < BF_TIMESTEP - 1; k++) p->hist_pos[k] = -1;
            }
            __atomic_store_n(&s->done, 0, __ATOMIC_RELEASE);
            st = (st + BF_ST_ONE_GEN) & ~(BF_ST_ONE_GEN - 1) & ~BF_ST_FLUSH;   /* next generation, nothing taken */
        } else if (rc == COVAHIP_ERR_OVERFLOW) {
            s->slot = -1;
            if (s->stop) ret = GST_FLOW_FLUSHING;
            else {
                const gint64 t0 = g_get_monotonic_time();
                g_cond_wait(&s->slot_cond, &s->lock);
                s->slot_wait_us += g_get_monotonic_time() - t0;
            }
        } else {
            ret = GST_FLOW_ERROR;
        }
    }
    /* without a slot the flush bit stays: every frame takes the slow path and tries again from here */
    __atomic_store_n(&s->state, s->slot >= 0 ? st : (st | BF_ST_FLUSH), __ATOMIC_RELEASE);
    s->flushing = FALSE;
    g_cond_broadcast(&s->flush_cond);
    return ret != GST_FLOW_OK ? ret : s->push_ret;
}

/* Takes the slot positions of one frame (and of the history frames its stack needs that the slot does not hold yet) with
 * one compare-and-swap, then fills in what it owns: the stack's table row and meta entry, the stream's history.  FALSE:
 * no room, batch full, or a flush under way -- the caller takes the slow path.  `taken` frames are to be added to
 * s->done once their bytes are in the slot. */
static gboolean bf_try_reserve(GstBlobNetFilter *s, BfPad *p, GstBuffer *buf, uint8_t **pf, int *pos, GstBuffer **need,
                               int *need_pos, gboolean *full, int *taken) {
    const gboolean complete = p->n_seen >= BF_TIMESTEP - 1;
    guint64 st = bf_state(s);
    int extra, next;
    while (TRUE) {
        if ((st & BF_ST_FLUSH) || BF_ST_FRAMES(st) + BF_TIMESTEP > __atomic_load_n(&s->max_frames, __ATOMIC_ACQUIRE) || BF_ST_STACKS(st) >= (int)s->batch_size) return FALSE;
        /* the history positions are this stream's own; a flush resets them, but then the swap below fails (generation) */
        extra = 0;
        if (complete) for (int k = 0; k < BF_TIMESTEP - 1; k++) extra += p->hist_pos[k] < 0;
        if (__atomic_compare_exchange_n(&s->state, &st, st + (guint64)(1 + extra) + (complete ? BF_ST_ONE_STACK : 0), FALSE,
                                        __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE)) break;
    }
    *pf = s->pf;
    *pos = BF_ST_FRAMES(st);
    *taken = 1 + extra;
    next = *pos + 1;
    if (complete) {
        /* a complete stack: T = 0 is this frame, T = k the frame k steps back (metapreprocess/imp.rs:307-320) */
        const int ns = BF_ST_STACKS(st);
        int32_t *row = s->pi + (gsize)ns * BF_TIMESTEP;
        row[0] = *pos;
        for (int k = 0; k < BF_TIMESTEP - 1; k++) {
            if (p->hist_pos[k] < 0) {   /* first use in this slot: the frame came with an earlier batch */
                p->hist_pos[k] = next++;
                need[k] = gst_buffer_ref(p->hist[k]);
                need_pos[k] = p->hist_pos[k];
            }
            row[k + 1] = p->hist_pos[k];
        }
        s->meta[ns].pad = p->idx;
        s->meta[ns].pts = GST_BUFFER_PTS(buf);
        s->meta[ns].duration = GST_BUFFER_DURATION(buf);
        if (ns == 0) {
            __atomic_store_n(&s->first_us, g_get_monotonic_time(), __ATOMIC_RELEASE);
            if (s->timeout_us) g_cond_signal(&s->cond);   /* the timeout thread starts its clock for this batch */
        }
        *full = ns + 1 >= (int)s->batch_size;
    }
    /* history: newest first; the element keeps references, the frame bytes are copied once, into the slot */
    if (p->hist[BF_TIMESTEP - 2]) gst_buffer_unref(p->hist[BF_TIMESTEP - 2]);
    for (int k = BF_TIMESTEP - 2; k > 0; k--) { p->hist[k] = p->hist[k - 1]; p->hist_pos[k] = p->hist_pos[k - 1]; }
    p->hist[0] = buf;                 /* takes over the reference the chain function was given */
    p->hist_pos[0] = *pos;
    p->n_seen++;
    return TRUE;
}

/* One carrier frame of one stream.  The streaming threads of the N decoder branches take their slot positions without a
 * lock and copy their 32 KB into the pinned slot in parallel; the element's mutex is for the rest: loading the model,
 * submitting a batch, waiting for a free slot. */
/* FALSE: the buffer could not be mapped.  The slot position is already reserved (the batch's bookkeeping must complete), so it
 * is zeroed -- and the caller reports the failure: a warning on the bus and GST_FLOW_ERROR upstream, never a silent empty frame. */
static gboolean bf_pack_into(GstBuffer *buf, uint16_t *dst, gsize frame_bytes, gboolean records) {
    GstMapInfo mi;
    if (gst_buffer_map(buf, &mi, GST_MAP_READ)) {
        /* (the chain function checked the buffer it was given; a held history buffer is checked here, where it is read) */
        if (mi.size < (records ? frame_bytes / 2 : frame_bytes)) {
            gst_buffer_unmap(buf, &mi);
            memset(dst, 0, frame_bytes / 2);
            return FALSE;
        }
        /* application/x-cova-records (round 5): the producer -- h264entropydec records=true, or any front end that writes
         * covahip_carrier_pack's form -- hands over the two-byte records themselves: a 16 KB copy at 1080p instead of reading a
         * 32 KB carrier region and packing it (the feeders' 6.4 us per frame of the round-4 chain) */
        if (records) memcpy(dst, mi.data, frame_bytes / 2);
        else covahip_carrier_pack(mi.data, frame_bytes / 4, dst);
        gst_buffer_unmap(buf, &mi);
        return TRUE;
    }
    memset(dst, 0, frame_bytes / 2);
    return FALSE;
}

static GstFlowReturn bf_chain(GstPad *pad, GstObject *parent, GstBuffer *buf) {
    GstBlobNetFilter *s = (GstBlobNetFilter *)parent;
    BfPad *p = bf_pad_of(s, pad);
    GstFlowReturn ret = GST_FLOW_OK;
    GstBuffer *need[BF_TIMESTEP - 1] = {NULL, NULL, NULL};   /* history frames this slot does not hold yet */
    int need_pos[BF_TIMESTEP - 1] = {0, 0, 0};
    uint8_t *pf = NULL;
    int pos = 0, taken = 0;
    gboolean full = FALSE;
    {   /* this stream's own downstream has finished or is flushing: say so upstream, the other streams are not affected */
        const gint own = g_atomic_int_get(&p->src_ret);
        if (own == GST_FLOW_EOS || own == GST_FLOW_FLUSHING) { gst_buffer_unref(buf); return (GstFlowReturn)own; }
    }
    if (gst_buffer_get_size(buf) < (p->records ? s->frame_bytes / 2 : s->frame_bytes) || !s->frame_bytes) {
        GST_ELEMENT_ERROR(s, STREAM, FORMAT, ("carrier frame of %" G_GSIZE_FORMAT " bytes, need %" G_GSIZE_FORMAT " (caps set?)",
                                              gst_buffer_get_size(buf), s->frame_bytes), (NULL));
        gst_buffer_unref(buf);
        return GST_FLOW_ERROR;
    }
    if (!bf_try_reserve(s, p, buf, &pf, &pos, need, need_pos, &full, &taken)) {
        /* ---- slow path */
        const gint64 ts0 = g_get_monotonic_time();
        g_mutex_lock(&s->lock);
        while (TRUE) {
            guint64 st;
            bf_wait_idle(s);
            if (!bf_ensure_model(s)) { ret = GST_FLOW_ERROR; break; }
            if (bf_try_reserve(s, p, buf, &pf, &pos, need, need_pos, &full, &taken)) break;
            st = bf_state(s);
            if (!(st & BF_ST_FLUSH) && BF_ST_STACKS(st) == 0) {
                /* no room for this frame and at worst three history frames of its stream, and nothing to submit: the
                 * streams' warm-up frames alone fill the slot */
                GST_ELEMENT_ERROR(s, CORE, FAILED, ("batch-size %u is too small for %u streams", s->batch_size, s->pads->len), (NULL));
                ret = GST_FLOW_ERROR;
                break;
            }
            /* a full batch whose last copier has not come back yet is flushed by whoever arrives first */
            if ((ret = bf_flush(s)) != GST_FLOW_OK) break;
        }
        g_mutex_unlock(&s->lock);
        g_atomic_int_add(&s->chain_slow_us, (gint)(g_get_monotonic_time() - ts0));
        g_atomic_int_inc(&s->chain_slow_n);
        if (ret != GST_FLOW_OK) {
            gst_buffer_unref(buf);
            return ret;
        }
    }

    /* metapreprocess copies these bytes (imp.rs:311-312); here they go into the slot as two-byte records (covahip_carrier_pack:
     * what BlobNet keeps of them), so that the host-to-device copy -- the bound of this element -- moves half the bytes */
    gboolean mapped = bf_pack_into(buf, (uint16_t *)(pf + (gsize)pos * (s->frame_bytes / 2)), s->frame_bytes, p->records);
    for (int k = 0; k < BF_TIMESTEP - 1; k++)
        if (need[k]) {
            mapped &= bf_pack_into(need[k], (uint16_t *)(pf + (gsize)need_pos[k] * (s->frame_bytes / 2)), s->frame_bytes, p->records);
            gst_buffer_unref(need[k]);
        }
    __atomic_fetch_add(&s->done, (guint64)taken, __ATOMIC_RELEASE);
    if (!mapped) {
        GST_ELEMENT_WARNING(s, RESOURCE, READ, ("a carrier frame of pad %s could not be mapped for reading", GST_PAD_NAME(pad)),
                            ("its slot was zeroed; the stream gets GST_FLOW_ERROR"));
        ret = GST_FLOW_ERROR;
    }

    if (full) {   /* the frame that completed the batch submits it, unless a later arrival has done so already */
        const gint64 tf0 = g_get_monotonic_time();
        g_mutex_lock(&s->lock);
        bf_wait_idle(s);
        if (BF_ST_STACKS(bf_state(s)) >= (int)s->batch_size) {
            const GstFlowReturn fr = bf_flush(s);
            if (ret == GST_FLOW_OK) ret = fr;
        }
        g_mutex_unlock(&s->lock);
        g_atomic_int_add(&s->chain_flush_us, (gint)(g_get_monotonic_time() - tf0));
    }
    return ret;
}

/* nvstreammux's batched-push-timeout: a batch that does not fill up leaves after this long anyway */
static gpointer bf_timer(gpointer data) {
    GstBlobNetFilter *s = data;
    g_mutex_lock(&s->lock);
    while (!s->stop) {
        if (s->pipe && BF_ST_STACKS(bf_state(s)) > 0 && s->timeout_us > 0) {
            const gint64 due = __atomic_load_n(&s->first_us, __ATOMIC_ACQUIRE) + (gint64)s->timeout_us;
            if (s->flushing) bf_wait_idle(s);
            else if (g_get_monotonic_time() >= due) bf_flush(s);
            else g_cond_wait_until(&s->cond, &s->lock, due);
        } else {
            /* the wake-up of the first stack of a batch is sent without the lock (lock-free path) and may be missed: with a
             * timeout configured the thread looks again after a quarter of it (at most 50 ms) */
            const gint64 nap = s->timeout_us > 0 ? MIN((gint64)s->timeout_us / 4 + 1, 50 * G_TIME_SPAN_MILLISECOND) : 50 * G_TIME_SPAN_MILLISECOND;
            g_cond_wait_until(&s->cond, &s->lock, g_get_monotonic_time() + nap);
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
        gint w = 0, h = 0;
        const gboolean was_records = p->records;
        gst_event_parse_caps(ev, &caps);
        if (gst_structure_has_name(gst_caps_get_structure(caps, 0), "application/x-cova-records")) {
            /* packed records: the grid itself is in the caps */
            const GstStructure *st = gst_caps_get_structure(caps, 0);
            if (!gst_structure_get_int(st, "width-mbs", &w) || !gst_structure_get_int(st, "height-mbs", &h) || w <= 0 || h <= 0) {
                gst_event_unref(ev);
                return FALSE;
            }
            p->records = TRUE;
        } else {
            if (!gst_video_info_from_caps(&vi, caps)) { gst_event_unref(ev); return FALSE; }
            /* metapreprocess' caps arithmetic (imp.rs:262-268): macroblock grid = picture / 16 */
            w = GST_VIDEO_INFO_WIDTH(&vi) / 16; h = GST_VIDEO_INFO_HEIGHT(&vi) / 16;
            p->records = FALSE;
        }
        if (p->records != was_records) {
            /* the pad changed its buffer form mid-stream (serialized with this pad's chain function): the held history frames are
             * in the OLD form -- half or twice the bytes the new mode reads -- so the stream starts its stacks over */
            for (int k = 0; k < BF_TIMESTEP - 1; k++) {
                if (p->hist[k]) gst_buffer_unref(p->hist[k]);
                p->hist[k] = NULL;
                p->hist_pos[k] = -1;
            }
            p->n_seen = 0;
        }
        g_mutex_lock(&s->lock);
        {
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
    case GST_EVENT_FLUSH_STOP:
    case GST_EVENT_STREAM_START:
        /* a new stream / the end of a flushing seek: whatever the last push on this pad returned is history */
        g_atomic_int_set(&p->src_ret, GST_FLOW_OK);
        g_mutex_lock(&s->lock);
        p->eos = FALSE;
        g_mutex_unlock(&s->lock);
        return gst_pad_push_event(p->src, ev);
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
        for (guint i = 0; i < s->pads->len; i++) {   /* a restart begins with clean per-pad flow returns */
            BfPad *bp = g_ptr_array_index(s->pads, i);
            g_atomic_int_set(&bp->src_ret, GST_FLOW_OK);
            bp->eos = FALSE;
        }
        for (guint g = 0; g < BF_PUSHERS; g++)
            if (!s->pusher[g].th) {
                s->pusher[g].s = s;
                s->pusher[g].group = g;
                if (!s->pusher[g].q) s->pusher[g].q = g_async_queue_new();
                s->pusher[g].th = g_thread_new("blobnetfilter-push", bf_pusher, &s->pusher[g]);
            }
        if (!s->timer) s->timer = g_thread_new("blobnetfilter-timeout", bf_timer, s);
        if (!s->submitter) s->submitter = g_thread_new("blobnetfilter-submit", bf_submitter, s);
        if (!s->collector) s->collector = g_thread_new("blobnetfilter-collect", bf_collector, s);
        g_mutex_unlock(&s->lock);
    }
    r = GST_ELEMENT_CLASS(gst_blobnetfilter_parent_class)->change_state(e, t);
    if (t == GST_STATE_CHANGE_PAUSED_TO_READY) {
        GThread *th, *tc, *ts;
        g_mutex_lock(&s->lock);
        s->stop = TRUE;
        g_cond_broadcast(&s->cond);
        g_cond_broadcast(&s->slot_cond);
        g_cond_broadcast(&s->flush_cond);
        th = s->timer;
        tc = s->collector;
        ts = s->submitter;
        s->timer = s->collector = s->submitter = NULL;
        g_mutex_unlock(&s->lock);
        if (th) g_thread_join(th);
        if (ts) g_thread_join(ts);   /* submits what is still queued ... */
        if (tc) g_thread_join(tc);   /* ... and the collector drains it */
        for (guint g = 0; g < BF_PUSHERS; g++)
            if (s->pusher[g].th) {
                g_async_queue_push(s->pusher[g].q, &s->pusher[g]);   /* stop token, behind whatever is still queued */
                g_thread_join(s->pusher[g].th);
                s->pusher[g].th = NULL;
            }
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
    case BF_PROP_TIMING: {
        gchar *t = g_strdup_printf("slot_wait_us=%" G_GINT64_FORMAT " gpu_wait_us=%" G_GINT64_FORMAT " collect_us=%" G_GINT64_FORMAT
                                   " pushers_us=%d chain_slow_us=%d(n=%d) chain_flush_us=%d",
                                   s->slot_wait_us, s->gpu_wait_us, s->push_us, s->pushers_us, s->chain_slow_us, s->chain_slow_n, s->chain_flush_us);
        g_value_take_string(v, t);
        break;
    }
    default: G_OBJECT_WARN_INVALID_PROPERTY_ID(o, id, ps);
    }
}
static void bf_finalize(GObject *o) {
    GstBlobNetFilter *s = (GstBlobNetFilter *)o;
    BfFlight *fl;
    while ((fl = g_queue_pop_head(&s->flights)) != NULL) { g_free(fl->meta); g_free(fl); }
    while ((fl = g_queue_pop_head(&s->submit_q)) != NULL) { g_free(fl->meta); g_free(fl); }
    g_mutex_clear(&s->pipe_lock);
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
    g_cond_clear(&s->cond);
    g_cond_clear(&s->flush_cond);
    g_cond_clear(&s->slot_cond);
    for (guint g = 0; g < BF_PUSHERS; g++) if (s->pusher[g].q) g_async_queue_unref(s->pusher[g].q);
    G_OBJECT_CLASS(gst_blobnetfilter_parent_class)->finalize(o);
}
static void gst_blobnetfilter_init(GstBlobNetFilter *s) {
    g_mutex_init(&s->lock);
    g_cond_init(&s->cond);
    g_cond_init(&s->flush_cond);
    g_cond_init(&s->slot_cond);
    g_queue_init(&s->flights);
    g_queue_init(&s->submit_q);
    g_mutex_init(&s->pipe_lock);
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
    g_object_class_install_property(g, BF_PROP_TIMING, g_param_spec_string("timing", "Timing",
        "Accumulated microseconds: streaming threads waiting for a free slot, collector waiting for the GPU, collecting + pushing results",
        NULL, G_PARAM_READABLE));
    g_object_class_install_property(g, BF_PROP_BATCHES, g_param_spec_uint64("batches", "Batches", "GPU batches submitted so far", 0,
        G_MAXUINT64, 0, G_PARAM_READABLE));
    gst_element_class_set_static_metadata(e, "BlobNet compressed-domain filter (batched)", "Filter/Video",
        "Batches the carrier frames of N streams and runs stacking + BlobNet + connected components on MI355X "
        "(replaces metapreprocess ! nvstreammux ! nvinfer ! nvstreamdemux ! maskcopy ! bboxcc)", "covahip");
    gst_element_class_add_pad_template(e, gst_pad_template_new("sink_%u", GST_PAD_SINK, GST_PAD_REQUEST,
        gst_caps_from_string("video/x-raw, format=(string)I420, width=(int)[16,2147483647], height=(int)[16,2147483647]; "
                             "application/x-cova-records, width-mbs=(int)[1,2147483647], height-mbs=(int)[1,2147483647]")));
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
