/*
 * gst_element_driver.c -- drives the elements of libgstcova.so buffer by buffer (GstHarness /
 * hand-linked pads) so that tests/ can compare their output with the oracle.  Test tool only.
 *
 * Record format used for buffer sequences on disk (little endian):
 *   u8 kind, u64 pts, u32 flags, u32 len, len bytes payload
 * kinds: 'B' buffer (metapreprocess / sorttracker / pipelines), 'E' encoded AU for cova.sink_enc,
 *        'M' bbox buffer for cova.sink_mask, 'e' EOS on sink_enc, 'm' EOS on sink_mask.
 * Output records: 'B' (pts, flags = GstBufferFlags, payload) and for cova 'L' (start of a BufferList).
 */
#define _GNU_SOURCE
#include <gst/check/gstharness.h>
#include <gst/gst.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#define __USE_GNU
#include <sys/resource.h>
#include <unistd.h>

typedef struct { uint8_t kind; uint64_t pts; uint32_t flags, len; uint8_t *data; } rec_t;

static int read_rec(FILE *f, rec_t *r) {
    if (fread(&r->kind, 1, 1, f) != 1) return 0;
    if (fread(&r->pts, 8, 1, f) != 1 || fread(&r->flags, 4, 1, f) != 1 || fread(&r->len, 4, 1, f) != 1) return 0;
    r->data = r->len ? malloc(r->len) : NULL;
    if (r->len && fread(r->data, 1, r->len, f) != r->len) return 0;
    return 1;
}
static void write_rec(FILE *f, uint8_t kind, uint64_t pts, uint32_t flags, const void *data, uint32_t len) {
    fwrite(&kind, 1, 1, f); fwrite(&pts, 8, 1, f); fwrite(&flags, 4, 1, f); fwrite(&len, 4, 1, f);
    if (len) fwrite(data, 1, len, f);
}
static void write_buffer(FILE *f, GstBuffer *b) {
    GstMapInfo m;
    gst_buffer_map(b, &m, GST_MAP_READ);
    write_rec(f, 'B', GST_BUFFER_PTS_IS_VALID(b) ? GST_BUFFER_PTS(b) : UINT64_MAX, GST_BUFFER_FLAGS(b), m.data, (uint32_t)m.size);
    gst_buffer_unmap(b, &m);
}
static GstBuffer *buffer_from(const rec_t *r) {
    GstBuffer *b = gst_buffer_new_allocate(NULL, r->len, NULL);
    if (r->len) gst_buffer_fill(b, 0, r->data, r->len);
    GST_BUFFER_PTS(b) = r->pts;
    if (r->flags & 1) GST_BUFFER_FLAG_SET(b, GST_BUFFER_FLAG_DELTA_UNIT);
    return b;
}

/* ---- generic: launch line + src caps, push 'B' records, write every output buffer ---- */
static int run_harness(const char *launch, const char *srccaps, const char *in_path, const char *out_path) {
    GstHarness *h = gst_harness_new_parse(launch);
    FILE *fi = fopen(in_path, "rb"), *fo = fopen(out_path, "wb");
    rec_t r;
    int pushed = 0, pulled = 0;
    GstBuffer *ob;
    if (!h || !fi || !fo) { fprintf(stderr, "setup failed\n"); return 2; }
    gst_harness_set_src_caps_str(h, srccaps);
    while (read_rec(fi, &r)) {
        if (r.kind == 'B') {
            GstFlowReturn fr = gst_harness_push(h, buffer_from(&r));
            if (fr != GST_FLOW_OK) { fprintf(stderr, "push failed: %s\n", gst_flow_get_name(fr)); return 3; }
            pushed++;
        }
        free(r.data);
        while ((ob = gst_harness_try_pull(h)) != NULL) { write_buffer(fo, ob); gst_buffer_unref(ob); pulled++; }
    }
    gst_harness_push_event(h, gst_event_new_eos());
    while ((ob = gst_harness_try_pull(h)) != NULL) { write_buffer(fo, ob); gst_buffer_unref(ob); pulled++; }
    {
        GstCaps *c = gst_pad_get_current_caps(h->sinkpad);
        gchar *s = c ? gst_caps_to_string(c) : g_strdup("(none)");
        printf("{\"pushed\": %d, \"pulled\": %d, \"out_caps\": \"%s\"}\n", pushed, pulled, s);
        g_free(s);
        if (c) gst_caps_unref(c);
    }
    fclose(fi); fclose(fo);
    gst_harness_teardown(h);
    return 0;
}

/* ---- sink elements (bboxsink, tfrecordsink): push 'B' records (flags bit 0 = DELTA_UNIT), EOS, tear down ---- */
static int run_sink(const char *launch, const char *srccaps, const char *in_path) {
    /* a chain "a ! b ! sink" becomes a bin whose unlinked sink pad is ghosted */
    GstElement *e = strchr(launch, '!') ? gst_parse_bin_from_description(launch, TRUE, NULL) : gst_parse_launch(launch, NULL);
    GstHarness *h = e ? gst_harness_new_with_element(e, "sink", NULL) : NULL;
    FILE *fi = fopen(in_path, "rb");
    rec_t r;
    int pushed = 0;
    if (!h || !fi) { fprintf(stderr, "setup failed\n"); return 2; }
    gst_harness_set_src_caps_str(h, srccaps);
    while (read_rec(fi, &r)) {
        if (r.kind == 'B') {
            GstFlowReturn fr = gst_harness_push(h, buffer_from(&r));
            if (fr != GST_FLOW_OK) { fprintf(stderr, "push failed: %s\n", gst_flow_get_name(fr)); return 3; }
            pushed++;
        }
        free(r.data);
    }
    gst_harness_push_event(h, gst_event_new_eos());
    printf("{\"pushed\": %d}\n", pushed);
    fclose(fi);
    gst_harness_teardown(h);
    gst_object_unref(e);
    return 0;
}

/* ---- cova: two sink pads driven by hand ---- */
static FILE *cova_out;
static GstFlowReturn cova_chain_list(GstPad *pad, GstObject *parent, GstBufferList *list) {
    guint n = gst_buffer_list_length(list);
    write_rec(cova_out, 'L', n, 0, NULL, 0);
    for (guint i = 0; i < n; i++) write_buffer(cova_out, gst_buffer_list_get(list, i));
    gst_buffer_list_unref(list);
    return GST_FLOW_OK;
}
static GstFlowReturn cova_chain(GstPad *pad, GstObject *parent, GstBuffer *b) {
    write_rec(cova_out, 'L', 1, 0, NULL, 0);
    write_buffer(cova_out, b);
    gst_buffer_unref(b);
    return GST_FLOW_OK;
}
static int cova_got_eos = 0;
static gboolean cova_sink_event(GstPad *pad, GstObject *parent, GstEvent *ev) {
    if (GST_EVENT_TYPE(ev) == GST_EVENT_EOS) cova_got_eos = 1;
    gst_event_unref(ev);
    return TRUE;
}
static void start_pad(GstPad *p, const char *stream, const char *caps) {
    GstSegment seg;
    gst_pad_set_active(p, TRUE);
    gst_pad_push_event(p, gst_event_new_stream_start(stream));
    if (caps) { GstCaps *c = gst_caps_from_string(caps); gst_pad_push_event(p, gst_event_new_caps(c)); gst_caps_unref(c); }
    gst_segment_init(&seg, GST_FORMAT_TIME);
    gst_pad_push_event(p, gst_event_new_segment(&seg));
}
static int run_cova(const char *props, const char *in_path, const char *out_path) {
    gchar *desc = g_strdup_printf("cova %s", props);
    GError *err = NULL;
    GstElement *e = gst_parse_launch(desc, &err);
    FILE *fi = fopen(in_path, "rb");
    rec_t r;
    guint64 d = 0, dd = 0, di = 0, held = 0, held_max = 0;
    if (!e || !fi) { fprintf(stderr, "cova setup failed: %s\n", err ? err->message : "?"); return 2; }
    cova_out = fopen(out_path, "wb");
    GstPad *enc = gst_pad_new("enc_src", GST_PAD_SRC), *mask = gst_pad_new("mask_src", GST_PAD_SRC);
    GstPad *sink = gst_pad_new("out_sink", GST_PAD_SINK);
    gst_pad_set_chain_function(sink, cova_chain);
    gst_pad_set_chain_list_function(sink, cova_chain_list);
    gst_pad_set_event_function(sink, cova_sink_event);
    gst_pad_set_active(sink, TRUE);
    GstPad *e_enc = gst_element_get_static_pad(e, "sink_enc"), *e_mask = gst_element_get_static_pad(e, "sink_mask");
    GstPad *e_src = gst_element_get_static_pad(e, "src");
    if (gst_pad_link(enc, e_enc) != GST_PAD_LINK_OK || gst_pad_link(mask, e_mask) != GST_PAD_LINK_OK ||
        gst_pad_link(e_src, sink) != GST_PAD_LINK_OK) { fprintf(stderr, "link failed\n"); return 2; }
    gst_element_set_state(e, GST_STATE_PLAYING);
    start_pad(enc, "enc", "video/x-h264");
    start_pad(mask, "mask", "bbox, width=(int)80, height=(int)45");
    int rc = 0;
    while (read_rec(fi, &r)) {
        GstFlowReturn fr = GST_FLOW_OK;
        if (r.kind == 'E') fr = gst_pad_push(enc, buffer_from(&r));
        else if (r.kind == 'M') fr = gst_pad_push(mask, buffer_from(&r));
        else if (r.kind == 'e') gst_pad_push_event(enc, gst_event_new_eos());
        else if (r.kind == 'm') gst_pad_push_event(mask, gst_event_new_eos());
        free(r.data);
        g_object_get(e, "held-buffers", &held, NULL);
        if (held > held_max) held_max = held;
        if (fr != GST_FLOW_OK) { fprintf(stderr, "flow %s at kind %c pts %llu\n", gst_flow_get_name(fr), r.kind, (unsigned long long)r.pts); rc = 3; break; }
    }
    g_object_get(e, "dropped", &d, "decoded-dependency", &dd, "decoded-inference", &di, NULL);
    g_object_get(e, "held-buffers", &held, NULL);
    printf("{\"dropped\": %llu, \"decoded_dependency\": %llu, \"decoded_inference\": %llu, \"eos\": %d, \"held_max\": %llu, \"held_end\": %llu}\n",
           (unsigned long long)d, (unsigned long long)dd, (unsigned long long)di, cova_got_eos, (unsigned long long)held_max,
           (unsigned long long)held);
    fclose(fi); fclose(cova_out);
    gst_element_set_state(e, GST_STATE_NULL);
    return rc;
}

/* ---- blobnetfilter: N request sink pads / N src pads.  Input records: 'B' with the pad index in flags >> 8, 'e' = EOS on
 * pad flags >> 8.  Output records: 'B', pts, flags = pad index, payload. ---- */
static FILE *mux_out;
static GMutex mux_lock;
static int mux_eos = 0, mux_bufs = 0;
static GstFlowReturn mux_chain(GstPad *pad, GstObject *parent, GstBuffer *b) {
    GstMapInfo m;
    const uint32_t idx = (uint32_t)GPOINTER_TO_UINT(gst_pad_get_element_private(pad));
    gst_buffer_map(b, &m, GST_MAP_READ);
    g_mutex_lock(&mux_lock);
    write_rec(mux_out, 'B', GST_BUFFER_PTS(b), idx, m.data, (uint32_t)m.size);
    mux_bufs++;
    g_mutex_unlock(&mux_lock);
    gst_buffer_unmap(b, &m);
    gst_buffer_unref(b);
    return GST_FLOW_OK;
}
static gboolean mux_sink_event(GstPad *pad, GstObject *parent, GstEvent *ev) {
    if (GST_EVENT_TYPE(ev) == GST_EVENT_EOS) { g_mutex_lock(&mux_lock); mux_eos++; g_mutex_unlock(&mux_lock); }
    gst_event_unref(ev);
    return TRUE;
}
static int run_mux(const char *desc, int n_pads, const char *caps, const char *in_path, const char *out_path, int sleep_ms_at_end) {
    GError *err = NULL;
    GstElement *e = gst_parse_launch(desc, &err);
    FILE *fi = fopen(in_path, "rb");
    rec_t r;
    GstPad *srcs[64];
    guint64 batches = 0;
    int rc = 0;
    if (!e || !fi || n_pads > 64) { fprintf(stderr, "mux setup failed: %s\n", err ? err->message : "?"); return 2; }
    mux_out = fopen(out_path, "wb");
    for (int i = 0; i < n_pads; i++) {
        GstPad *esink = gst_element_get_request_pad(e, "sink_%u");
        gchar *sn = g_strdup_printf("src_%d", i);
        GstPad *esrc = gst_element_get_static_pad(e, sn);
        GstPad *tsink = gst_pad_new("out", GST_PAD_SINK);
        g_free(sn);
        srcs[i] = gst_pad_new("in", GST_PAD_SRC);
        gst_pad_set_element_private(tsink, GUINT_TO_POINTER((guint)i));
        gst_pad_set_chain_function(tsink, mux_chain);
        gst_pad_set_event_function(tsink, mux_sink_event);
        gst_pad_set_active(tsink, TRUE);
        if (!esink || !esrc || gst_pad_link(srcs[i], esink) != GST_PAD_LINK_OK || gst_pad_link(esrc, tsink) != GST_PAD_LINK_OK) {
            fprintf(stderr, "mux link failed on pad %d\n", i);
            return 2;
        }
    }
    gst_element_set_state(e, GST_STATE_PLAYING);
    for (int i = 0; i < n_pads; i++) { gchar *sid = g_strdup_printf("s%d", i); start_pad(srcs[i], sid, caps); g_free(sid); }
    while (read_rec(fi, &r)) {
        const int pad = (int)(r.flags >> 8);
        GstFlowReturn fr = GST_FLOW_OK;
        if (pad >= n_pads) { free(r.data); continue; }
        if (r.kind == 'B') { r.flags &= 0xFF; fr = gst_pad_push(srcs[pad], buffer_from(&r)); }
        else if (r.kind == 'e') gst_pad_push_event(srcs[pad], gst_event_new_eos());
        else if (r.kind == 's') g_usleep(1000 * (gulong)r.pts);   /* pause: lets batched-push-timeout fire */
        free(r.data);
        if (fr != GST_FLOW_OK) { fprintf(stderr, "flow %s on pad %d\n", gst_flow_get_name(fr), pad); rc = 3; break; }
    }
    if (sleep_ms_at_end) g_usleep(1000 * (gulong)sleep_ms_at_end);
    g_object_get(e, "batches", &batches, NULL);
    g_mutex_lock(&mux_lock);
    printf("{\"buffers\": %d, \"eos\": %d, \"batches\": %llu}\n", mux_bufs, mux_eos, (unsigned long long)batches);
    g_mutex_unlock(&mux_lock);
    gst_element_set_state(e, GST_STATE_NULL);
    g_mutex_lock(&mux_lock);
    fclose(mux_out);
    g_mutex_unlock(&mux_lock);
    fclose(fi);
    return rc;
}

/* ---- muxbench: blobnetfilter fed by one thread per stream with in-memory carrier frames; prints frames/s ---- */
typedef struct { GstPad *src; GstBuffer **bufs; int n_bufs, n_frames, first, eos; } bench_feed_t;
static volatile gint feeder_cpu_us;   /* CPU time of the feeder threads that have finished (getrusage) */
static gpointer bench_feeder(gpointer data) {
    bench_feed_t *f = data;
    gint64 tp = 0;
    for (int i = f->first; i < f->first + f->n_frames; i++) {
        GstBuffer *b = gst_buffer_copy(f->bufs[i % f->n_bufs]);   /* shares the memory, own metadata */
        GST_BUFFER_PTS(b) = (GstClockTime)i * (GST_SECOND / 30);
        const gint64 t0 = g_get_monotonic_time();
        const GstFlowReturn fr = gst_pad_push(f->src, b);
        tp += g_get_monotonic_time() - t0;
        if (fr != GST_FLOW_OK) break;
    }
    if (getenv("FEED_TIMING")) fprintf(stderr, "feeder: %d frames, %lld us inside gst_pad_push\n", f->n_frames, (long long)tp);
    {
        struct rusage ru;
        if (!getrusage(RUSAGE_THREAD, &ru))
            g_atomic_int_add(&feeder_cpu_us, (gint)((ru.ru_utime.tv_sec + ru.ru_stime.tv_sec) * 1000000 + ru.ru_utime.tv_usec + ru.ru_stime.tv_usec));
    }
    if (f->eos) gst_pad_push_event(f->src, gst_event_new_eos());
    return NULL;
}
/* per src pad: count, order check and an order-sensitive checksum of (PTS, bytes); a pad is pushed to by one thread at a time */
typedef struct { guint64 n, sum; GstClockTime last_pts; int in_order; int check; } bench_out_t;
static GstFlowReturn bench_chain(GstPad *pad, GstObject *parent, GstBuffer *b) {
    bench_out_t *o = gst_pad_get_element_private(pad);
    if (o->n && GST_BUFFER_PTS(b) <= o->last_pts) o->in_order = 0;
    o->last_pts = GST_BUFFER_PTS(b);
    o->n++;
    if (o->check) {
        GstMapInfo m;
        guint64 h = o->sum ^ GST_BUFFER_PTS(b);
        gst_buffer_map(b, &m, GST_MAP_READ);
        for (gsize i = 0; i < m.size; i++) h = (h ^ m.data[i]) * 1099511628211ull;
        gst_buffer_unmap(b, &m);
        o->sum = h * 1099511628211ull + 1;
    }
    gst_buffer_unref(b);
    return GST_FLOW_OK;
}
static int run_muxbench(const char *desc, int n_pads, int w_px, int h_px, int frames_per_pad) {
    GError *err = NULL;
    GstElement *e = gst_parse_launch(desc, &err);
    bench_feed_t feeds[64];
    static bench_out_t outs[64];
    const int same = getenv("MUXBENCH_SAME") != NULL;   /* every stream carries the same frames: the per-pad checksums must agree */
    GThread *th[64];
    gchar *caps = g_strdup_printf("video/x-raw,format=I420,width=%d,height=%d,framerate=30/1", w_px, h_px);
    const gsize fb = (gsize)(w_px / 16) * (h_px / 16) * 4;
    guint64 batches = 0;
    if (!e || n_pads > 64) { fprintf(stderr, "muxbench setup failed: %s\n", err ? err->message : "?"); return 2; }
    for (int i = 0; i < n_pads; i++) {
        GstPad *esink = gst_element_get_request_pad(e, "sink_%u");
        gchar *sn = g_strdup_printf("src_%d", i);
        GstPad *esrc = gst_element_get_static_pad(e, sn);
        GstPad *tsink = gst_pad_new("out", GST_PAD_SINK);
        g_free(sn);
        feeds[i].src = gst_pad_new("in", GST_PAD_SRC);
        feeds[i].n_bufs = 16;
        feeds[i].n_frames = frames_per_pad;
        feeds[i].bufs = g_new(GstBuffer *, 16);
        for (int k = 0; k < 16; k++) {
            GstMapInfo m;
            unsigned x = 777u * (unsigned)((same ? 0 : i) * 16 + k + 1);
            feeds[i].bufs[k] = gst_buffer_new_allocate(NULL, fb, NULL);
            gst_buffer_map(feeds[i].bufs[k], &m, GST_MAP_WRITE);
            for (gsize q = 0; q < fb; q++) { x = x * 1664525u + 1013904223u; m.data[q] = (x >> 24) < 40 ? (x >> 16) % 7 : 0; }
            gst_buffer_unmap(feeds[i].bufs[k], &m);
        }
        outs[i].in_order = 1;
        outs[i].check = same;
        outs[i].sum = 14695981039346656037ull;
        gst_pad_set_element_private(tsink, &outs[i]);
        gst_pad_set_chain_function(tsink, bench_chain);
        gst_pad_set_event_function(tsink, mux_sink_event);
        gst_pad_set_active(tsink, TRUE);
        if (!esink || !esrc || gst_pad_link(feeds[i].src, esink) != GST_PAD_LINK_OK || gst_pad_link(esrc, tsink) != GST_PAD_LINK_OK) return 2;
    }
    gst_element_set_state(e, GST_STATE_PLAYING);
    for (int i = 0; i < n_pads; i++) { gchar *sid = g_strdup_printf("s%d", i); start_pad(feeds[i].src, sid, caps); g_free(sid); }
    {   /* model load and pipe creation happen with the first frame, code-object load and the scratch allocations with the
         * first batches: a few batches go through before the clock starts */
        const int warm = getenv("MUXBENCH_WARM") ? atoi(getenv("MUXBENCH_WARM")) : 1024 / n_pads + 8;
        for (int i = 0; i < n_pads; i++) { feeds[i].first = 0; feeds[i].n_frames = warm; feeds[i].eos = 0; }
        for (int i = 0; i < n_pads; i++) th[i] = g_thread_new("feed", bench_feeder, &feeds[i]);
        for (int i = 0; i < n_pads; i++) g_thread_join(th[i]);
        g_usleep(200000);
        { gchar *tm = NULL; g_object_get(e, "timing", &tm, NULL); fprintf(stderr, "timing after warm-up: %s\n", tm ? tm : "?"); g_free(tm); }
        for (int i = 0; i < n_pads; i++) { feeds[i].first = warm; feeds[i].n_frames = frames_per_pad; feeds[i].eos = 1; }
    }
    const gint64 t0 = g_get_monotonic_time();
    for (int i = 0; i < n_pads; i++) th[i] = g_thread_new("feed", bench_feeder, &feeds[i]);
    for (int i = 0; i < n_pads; i++) g_thread_join(th[i]);
    const gint64 t1 = g_get_monotonic_time();
    g_object_get(e, "batches", &batches, NULL);
    { gchar *tm = NULL; g_object_get(e, "timing", &tm, NULL); fprintf(stderr, "timing at the end:   %s\n", tm ? tm : "?"); g_free(tm); }
    g_mutex_lock(&mux_lock);
    {
        guint64 n_out = 0;
        int in_order = 1, agree = 1;
        for (int i = 0; i < n_pads; i++) { n_out += outs[i].n; in_order &= outs[i].in_order; agree &= outs[i].sum == outs[0].sum && outs[i].n == outs[0].n; }
        printf("{\"frames_per_s_through_elements\": %.1f, \"streams\": %d, \"frames_in\": %d, \"buffers_out\": %llu, \"eos\": %d, \"batches\": %llu, \"seconds\": %.4f, \"in_order\": %s",
               (double)n_pads * frames_per_pad / ((t1 - t0) * 1e-6), n_pads, n_pads * frames_per_pad, (unsigned long long)n_out, mux_eos,
               (unsigned long long)batches, (t1 - t0) * 1e-6, in_order ? "true" : "false");
        if (same) printf(", \"pads_agree\": %s, \"pad0_sum\": \"%016llx\"", agree ? "true" : "false", (unsigned long long)outs[0].sum);
        printf("}\n");
    }
    g_mutex_unlock(&mux_lock);
    gst_element_set_state(e, GST_STATE_NULL);
    g_free(caps);
    return 0;
}

/* CHAINBENCH_PROF=1: a sampling profile of the whole process without tools the image lacks (no perf / gdb here): ITIMER_PROF
 * delivers SIGPROF to whichever thread is burning CPU, the handler notes the interrupted program counter, and at the end the
 * samples are attributed to the nearest preceding dynamic symbol (dladdr) -- where the host cores of the chain go, by function. */
#define _PROF_MAX 400000
#include <dlfcn.h>
#include <signal.h>
#include <sys/time.h>
#include <ucontext.h>
static void *prof_pc[_PROF_MAX];
static int prof_tid[_PROF_MAX];
static volatile gint prof_n;
#include <execinfo.h>
#include <sys/syscall.h>
/* CHAINBENCH_PROF_STACKS=1 (round 6): up to PROF_DEPTH return addresses per sample (glibc backtrace: primed once before the timer
 * starts so that the unwinder is loaded; the raw file then carries the whole chain, innermost first) */
#define PROF_DEPTH 12
#define PROF_STACK_MAX 20000
static void *prof_stack[PROF_STACK_MAX][PROF_DEPTH];
static int prof_stack_n[PROF_STACK_MAX];
static int prof_stacks_on;
static void prof_handler(int sig, siginfo_t *si, void *uc_) {
    ucontext_t *uc = uc_;
    const gint k = g_atomic_int_add(&prof_n, 1);
    if (k < _PROF_MAX) { prof_pc[k] = (void *)uc->uc_mcontext.gregs[REG_RIP]; prof_tid[k] = (int)syscall(SYS_gettid); }
    if (prof_stacks_on && k < PROF_STACK_MAX) prof_stack_n[k] = backtrace(prof_stack[k], PROF_DEPTH);
}
static void prof_start(void) {
    struct sigaction sa;
    struct itimerval it = {{0, 500}, {0, 500}};   /* 2 kHz of CPU time */
    if (getenv("CHAINBENCH_PROF_STACKS")) { void *prime[4]; backtrace(prime, 4); prof_stacks_on = 1; }
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = prof_handler;
    sa.sa_flags = SA_SIGINFO | SA_RESTART;
    sigaction(SIGPROF, &sa, NULL);
    setitimer(ITIMER_PROF, &it, NULL);
}
static void prof_report(void) {
    struct itimerval off = {{0, 0}, {0, 0}};
    struct { const char *name; const char *lib; int n; } acc[512];
    int na = 0, n = MIN(prof_n, _PROF_MAX);
    setitimer(ITIMER_PROF, &off, NULL);
    for (int i = 0; i < n; i++) {
        Dl_info di;
        const char *nm = "?", *lb = "?";
        if (dladdr(prof_pc[i], &di)) { nm = di.dli_sname ? di.dli_sname : "(static)"; lb = di.dli_fname ? di.dli_fname : "?"; }
        int k = 0;
        for (; k < na; k++) if (!strcmp(acc[k].name, nm) && !strcmp(acc[k].lib, lb)) break;
        if (k == na && na < 512) { acc[na].name = nm; acc[na].lib = lb; acc[na].n = 0; na++; }
        if (k < 512) acc[k].n++;
    }
    if (getenv("CHAINBENCH_PROF_RAW")) {   /* "library offset" per sample: resolved offline with nm (tools/prof_resolve.py) */
        FILE *f = fopen(getenv("CHAINBENCH_PROF_RAW"), "w");
        for (int i = 0; f && i < n; i++) {
            Dl_info di;
            char comm[64] = "?", path[64];
            FILE *cf;
            snprintf(path, sizeof path, "/proc/self/task/%d/comm", prof_tid[i]);   /* (a thread that has exited by now stays "?") */
            if ((cf = fopen(path, "r"))) { if (fgets(comm, sizeof comm, cf)) comm[strcspn(comm, "\n")] = 0; fclose(cf); }
            for (char *c = comm; *c; c++) if (*c == ' ') *c = '_';
            if (dladdr(prof_pc[i], &di) && di.dli_fname) fprintf(f, "%s %lx %s", di.dli_fname, (unsigned long)((char *)prof_pc[i] - (char *)di.dli_fbase), comm);
            else fprintf(f, "? %lx %s", (unsigned long)prof_pc[i], comm);
            if (prof_stacks_on && i < PROF_STACK_MAX)
                for (int d = 0; d < prof_stack_n[i]; d++) {   /* " | lib offset" per frame */
                    Dl_info dj;
                    if (dladdr(prof_stack[i][d], &dj) && dj.dli_fname)
                        fprintf(f, " | %s %lx %s", dj.dli_fname, (unsigned long)((char *)prof_stack[i][d] - (char *)dj.dli_fbase), dj.dli_sname ? dj.dli_sname : "-");
                }
            fprintf(f, "\n");
        }
        if (f) fclose(f);
    }
    fprintf(stderr, "profile: %d samples\n", n);
    for (int r = 0; r < 40; r++) {
        int best = -1;
        for (int k = 0; k < na; k++) if (acc[k].n > 0 && (best < 0 || acc[k].n > acc[best].n)) best = k;
        if (best < 0) break;
        const char *b = strrchr(acc[best].lib, '/');
        fprintf(stderr, "  %5.1f %%  %-48s %s\n", 100.0 * acc[best].n / n, acc[best].name, b ? b + 1 : acc[best].lib);
        acc[best].n = 0;
    }
}
/* CPU seconds per thread name of this process (/proc/self/task): where the host time of a bench run went */
#include <dirent.h>
static void print_thread_cpu(const char *tag) {
    DIR *d = opendir("/proc/self/task");
    struct { char name[32]; double sec; int n; } acc[32];
    int na = 0;
    struct dirent *de;
    const double tick = 1.0 / (double)sysconf(_SC_CLK_TCK);
    if (!d) return;
    while ((de = readdir(d))) {
        char path[128], comm[64] = "?", buf[1024];
        FILE *f;
        unsigned long ut = 0, st = 0;
        if (de->d_name[0] == '.') continue;
        snprintf(path, sizeof path, "/proc/self/task/%s/comm", de->d_name);
        if ((f = fopen(path, "r"))) { if (fgets(comm, sizeof comm, f)) comm[strcspn(comm, "\n")] = 0; fclose(f); }
        snprintf(path, sizeof path, "/proc/self/task/%s/stat", de->d_name);
        if ((f = fopen(path, "r"))) {
            if (fgets(buf, sizeof buf, f)) { char *q = strrchr(buf, ')'); if (q) sscanf(q + 2, "%*c %*d %*d %*d %*d %*d %*u %*u %*u %*u %*u %lu %lu", &ut, &st); }
            fclose(f);
        }
        int k = 0;
        for (; k < na; k++) if (!strcmp(acc[k].name, comm)) break;
        if (k == na && na < 32) { snprintf(acc[na].name, sizeof acc[na].name, "%s", comm); acc[na].sec = 0; acc[na].n = 0; na++; }
        if (k < 32) { acc[k].sec += (ut + st) * tick; acc[k].n++; }
    }
    closedir(d);
    fprintf(stderr, "cpu seconds by thread name (%s):", tag);
    for (int k = 0; k < na; k++) fprintf(stderr, " %s x%d %.2f;", acc[k].name, acc[k].n, acc[k].sec);
    fprintf(stderr, "\n");
}
/* CHAINBENCH_RECORDS=1: the feeders hand over packed two-byte records (caps application/x-cova-records: what `h264entropydec
 * records=true` emits) instead of four-byte carrier regions -- the buffers of a feed are replaced by their packed form */
static void pack_feed(GstBuffer **bufs, int n, int wmb, int hmb) {
    for (int k = 0; k < n; k++) {
        GstMapInfo mi, mo;
        GstBuffer *nb = gst_buffer_new_allocate(NULL, (gsize)wmb * hmb * 2, NULL);
        gst_buffer_map(bufs[k], &mi, GST_MAP_READ);
        gst_buffer_map(nb, &mo, GST_MAP_WRITE);
        /* covahip_carrier_pack's format (include/covahip.h), written out here: the driver does not link the library */
        for (gsize q = 0; q < (gsize)wmb * hmb; q++) {
            const guint8 *r = mi.data + 4 * q;
            ((uint16_t *)mo.data)[q] = (uint16_t)(MIN(r[0], 6) | MIN(r[1], 6) << 3 | MIN(r[2], 6) << 6);
        }
        gst_buffer_unmap(nb, &mo);
        gst_buffer_unmap(bufs[k], &mi);
        gst_buffer_unref(bufs[k]);
        bufs[k] = nb;
    }
}
/* ---- chainbench: BASELINE config 4 as a throughput number.  N streams -> blobnetfilter (metapreprocess + nvstreammux + nvinfer +
 * nvstreamdemux + maskcopy + bboxcc stand-in) -> per stream a `cova` element (embedded SORT + GoP frame filter) whose sink_enc gets
 * the stream's encoded access units (dummy payloads; key frame every 250) -> counting sink.  pipeline/cova/pipeline.py:104-261.
 * Carrier frames: per stream a cycle of 256 frames with a few ellipses of motion vectors bouncing across the grid (periodic in the
 * cycle, so the stream is continuous for as long as it runs) over sparse background noise. ---- */
typedef struct { guint64 bufs, lists; } chain_out_t;
static GstFlowReturn chainb_chain(GstPad *pad, GstObject *parent, GstBuffer *b) {
    chain_out_t *o = gst_pad_get_element_private(pad);
    o->bufs++;
    gst_buffer_unref(b);
    return GST_FLOW_OK;
}
static GstFlowReturn chainb_chain_list(GstPad *pad, GstObject *parent, GstBufferList *l) {
    chain_out_t *o = gst_pad_get_element_private(pad);
    o->bufs += gst_buffer_list_length(l);
    o->lists++;
    gst_buffer_list_unref(l);
    return GST_FLOW_OK;
}
typedef struct { GstPad *src; int first, n_frames, eos; } enc_feed_t;
static gpointer enc_feeder(gpointer data) {
    enc_feed_t *f = data;
    for (int i = f->first; i < f->first + f->n_frames; i++) {
        GstBuffer *b = gst_buffer_new_allocate(NULL, 64, NULL);
        GST_BUFFER_PTS(b) = (GstClockTime)i * (GST_SECOND / 30);
        if (i % 250) GST_BUFFER_FLAG_SET(b, GST_BUFFER_FLAG_DELTA_UNIT);
        if (gst_pad_push(f->src, b) != GST_FLOW_OK) break;
    }
    if (f->eos) gst_pad_push_event(f->src, gst_event_new_eos());
    return NULL;
}
static double tri_wave(double x) {   /* period 2, range [0, 1] */
    x -= 2.0 * floor(x / 2.0);
    return x < 1.0 ? x : 2.0 - x;
}
static void chain_make_frames(GstBuffer **bufs, int n, int wmb, int hmb, unsigned seed) {
    struct { double cx, cy, lx, ly, rx, ry; int mx, my; double px, py; } ob[5];
    unsigned x = seed * 2654435761u + 12345u;
#define RND() (x = x * 1664525u + 1013904223u, (x >> 8) & 0xFFFF)
    const int nob = 3 + RND() % 3;
    for (int j = 0; j < nob; j++) {
        ob[j].rx = 2.0 + RND() % 40 / 10.0; ob[j].ry = 2.0 + RND() % 40 / 10.0;
        ob[j].cx = RND() % wmb; ob[j].cy = RND() % hmb;
        ob[j].lx = wmb - 1; ob[j].ly = hmb - 1;
        ob[j].mx = 1 + RND() % 3; ob[j].my = RND() % 3;
        ob[j].px = RND() % 1000 / 500.0; ob[j].py = RND() % 1000 / 500.0;
    }
    for (int k = 0; k < n; k++) {
        GstMapInfo m;
        bufs[k] = gst_buffer_new_allocate(NULL, (gsize)wmb * hmb * 4, NULL);
        gst_buffer_map(bufs[k], &m, GST_MAP_WRITE);
        for (int q = 0; q < wmb * hmb; q++) {
            const unsigned r = RND();
            m.data[4 * q] = (r & 7) < 2 ? (r >> 4) % 7 : 0;
            m.data[4 * q + 1] = (r >> 8) % 10 == 0 ? 1 + (r >> 3) % 3 : 0;
            m.data[4 * q + 2] = (r >> 12) % 10 == 0 ? 1 + (r >> 5) % 3 : 0;
            m.data[4 * q + 3] = (guint8)r;
        }
        for (int j = 0; j < nob; j++) {
            /* m bounces across the span per cycle of n frames: the position is periodic in n */
            const double cx = ob[j].lx * tri_wave(ob[j].px + 2.0 * ob[j].mx * k / n), cy = ob[j].ly * tri_wave(ob[j].py + 2.0 * ob[j].my * k / n);
            for (int yy = (int)(cy - ob[j].ry) - 1; yy <= (int)(cy + ob[j].ry) + 1; yy++)
                for (int xx = (int)(cx - ob[j].rx) - 1; xx <= (int)(cx + ob[j].rx) + 1; xx++) {
                    if (yy < 0 || yy >= hmb || xx < 0 || xx >= wmb) continue;
                    const double dx = (xx - cx) / ob[j].rx, dy = (yy - cy) / ob[j].ry;
                    if (dx * dx + dy * dy > 1.0) continue;
                    const unsigned r = RND();
                    guint8 *px = m.data + 4 * (yy * wmb + xx);
                    px[0] = 1 + r % 7; px[1] = 1 + (r >> 4) % 12; px[2] = 1 + (r >> 8) % 12;
                }
        }
        gst_buffer_unmap(bufs[k], &m);
    }
#undef RND
}
static int run_chainbench(const char *desc, const char *cova_props, int n_pads, int w_px, int h_px, int frames_per_pad) {
    GError *err = NULL;
    GstElement *e = gst_parse_launch(desc, &err);
    GstElement *cova[64];
    bench_feed_t feeds[64];
    enc_feed_t encs[64];
    static chain_out_t outs[64];
    GThread *th[64], *eth[64];
    const int records = getenv("CHAINBENCH_RECORDS") && atoi(getenv("CHAINBENCH_RECORDS"));
    gchar *caps = records ? g_strdup_printf("application/x-cova-records,width-mbs=%d,height-mbs=%d,framerate=30/1", w_px / 16, h_px / 16)
                          : g_strdup_printf("video/x-raw,format=I420,width=%d,height=%d,framerate=30/1", w_px, h_px);
    gchar *cdesc = g_strdup_printf("cova %s", cova_props);
    guint64 batches = 0;
    const int cycle = 256;
    if (!e || n_pads > 64) { fprintf(stderr, "chainbench setup failed: %s\n", err ? err->message : "?"); return 2; }
    for (int i = 0; i < n_pads; i++) {
        GstPad *esink = gst_element_get_request_pad(e, "sink_%u");
        gchar *sn = g_strdup_printf("src_%d", i);
        GstPad *esrc = gst_element_get_static_pad(e, sn);
        GstPad *tsink = gst_pad_new("out", GST_PAD_SINK);
        g_free(sn);
        cova[i] = gst_parse_launch(cdesc, &err);
        if (!cova[i]) { fprintf(stderr, "cova: %s\n", err ? err->message : "?"); return 2; }
        feeds[i].src = gst_pad_new("in", GST_PAD_SRC);
        feeds[i].n_bufs = cycle;
        feeds[i].bufs = g_new(GstBuffer *, cycle);
        chain_make_frames(feeds[i].bufs, cycle, w_px / 16, h_px / 16, 1000u + (unsigned)i);
        if (records) pack_feed(feeds[i].bufs, cycle, w_px / 16, h_px / 16);
        encs[i].src = gst_pad_new("enc", GST_PAD_SRC);
        gst_pad_set_element_private(tsink, &outs[i]);
        gst_pad_set_chain_function(tsink, chainb_chain);
        gst_pad_set_chain_list_function(tsink, chainb_chain_list);
        gst_pad_set_event_function(tsink, mux_sink_event);
        gst_pad_set_active(tsink, TRUE);
        if (!esink || !esrc || gst_pad_link(feeds[i].src, esink) != GST_PAD_LINK_OK ||
            gst_pad_link(esrc, gst_element_get_static_pad(cova[i], "sink_mask")) != GST_PAD_LINK_OK ||
            gst_pad_link(encs[i].src, gst_element_get_static_pad(cova[i], "sink_enc")) != GST_PAD_LINK_OK ||
            gst_pad_link(gst_element_get_static_pad(cova[i], "src"), tsink) != GST_PAD_LINK_OK) { fprintf(stderr, "chainbench link failed\n"); return 2; }
        gst_element_set_state(cova[i], GST_STATE_PLAYING);
    }
    gst_element_set_state(e, GST_STATE_PLAYING);
    for (int i = 0; i < n_pads; i++) {
        gchar *sid = g_strdup_printf("s%d", i);
        start_pad(encs[i].src, sid, "video/x-h264");
        start_pad(feeds[i].src, sid, caps);
        g_free(sid);
    }
    const int warm = 2048 / n_pads + 8;
    int rc = 0;
    gint64 t0 = 0, t1 = 0;
    for (int phase = 0; phase < 2; phase++) {
        const int first = phase ? warm : 0, n = phase ? frames_per_pad : warm;
        /* the encoded branch runs ahead of the mask branch (an unbounded queue in the reference, pipeline.py:237-253) */
        for (int i = 0; i < n_pads; i++) { encs[i].first = first; encs[i].n_frames = n; encs[i].eos = phase; }
        for (int i = 0; i < n_pads; i++) eth[i] = g_thread_new("enc", enc_feeder, &encs[i]);
        for (int i = 0; i < n_pads; i++) g_thread_join(eth[i]);
        for (int i = 0; i < n_pads; i++) { feeds[i].first = first; feeds[i].n_frames = n; feeds[i].eos = phase; }
        if (phase) t0 = g_get_monotonic_time();
        if (phase && getenv("CHAINBENCH_PROF")) prof_start();
        for (int i = 0; i < n_pads; i++) th[i] = g_thread_new("feed", bench_feeder, &feeds[i]);
        for (int i = 0; i < n_pads; i++) g_thread_join(th[i]);
        if (phase) t1 = g_get_monotonic_time(); else g_usleep(300000);
    }
    g_usleep(200000);   /* the last batches drain through the pusher threads */
    if (getenv("CHAINBENCH_PROF")) prof_report();
    if (getenv("CHAINBENCH_CPU")) {
        print_thread_cpu("live threads at the end");
        fprintf(stderr, "cpu seconds of the carrier-frame feeders (both phases): %.2f; wall seconds of the timed phase %.3f\n", feeder_cpu_us * 1e-6, (t1 - t0) * 1e-6);
    }
    g_object_get(e, "batches", &batches, NULL);
    {
        guint64 fwd = 0, d = 0, dd = 0, di = 0;
        for (int i = 0; i < n_pads; i++) {
            guint64 a = 0, b = 0, c = 0;
            g_object_get(cova[i], "dropped", &a, "decoded-dependency", &b, "decoded-inference", &c, NULL);
            d += a; dd += b; di += c; fwd += outs[i].bufs;
        }
        { gchar *tm = NULL; g_object_get(e, "timing", &tm, NULL); fprintf(stderr, "timing at the end:   %s\n", tm ? tm : "?"); g_free(tm); }
        printf("{\"frames_per_s_full_chain\": %.1f, \"streams\": %d, \"frames_in\": %d, \"batches\": %llu, \"seconds\": %.4f, "
               "\"aus_forwarded\": %llu, \"dropped\": %llu, \"decoded_dependency\": %llu, \"decoded_inference\": %llu, \"eos\": %d}\n",
               (double)n_pads * frames_per_pad / ((t1 - t0) * 1e-6), n_pads, n_pads * frames_per_pad, (unsigned long long)batches,
               (t1 - t0) * 1e-6, (unsigned long long)fwd, (unsigned long long)d, (unsigned long long)dd, (unsigned long long)di, mux_eos);
    }
    for (int i = 0; i < n_pads; i++) gst_element_set_state(cova[i], GST_STATE_NULL);
    gst_element_set_state(e, GST_STATE_NULL);
    g_free(caps); g_free(cdesc);
    return rc;
}

int main(int argc, char **argv) {
    gst_init(&argc, &argv);
    if (argc >= 8 && !strcmp(argv[1], "chainbench")) return run_chainbench(argv[2], argv[3], atoi(argv[4]), atoi(argv[5]), atoi(argv[6]), atoi(argv[7]));
    if (argc >= 7 && !strcmp(argv[1], "muxbench")) return run_muxbench(argv[2], atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), atoi(argv[6]));
    if (argc >= 7 && !strcmp(argv[1], "mux")) return run_mux(argv[2], atoi(argv[3]), argv[4], argv[5], argv[6], argc > 7 ? atoi(argv[7]) : 0);
    if (argc >= 6 && !strcmp(argv[1], "harness")) return run_harness(argv[2], argv[3], argv[4], argv[5]);
    if (argc >= 5 && !strcmp(argv[1], "cova")) return run_cova(argv[2], argv[3], argv[4]);
    if (argc >= 5 && !strcmp(argv[1], "sink")) return run_sink(argv[2], argv[3], argv[4]);
    fprintf(stderr, "usage: %s harness '<launch line>' '<src caps>' in.rec out.rec | cova '<props>' in.rec out.rec | sink '<launch line>' '<src caps>' in.rec\n", argv[0]);
    return 1;
}
