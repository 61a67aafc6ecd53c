/*
 * gstentropydec.c -- `h264entropydec`: stands where the reference runs its patched `avdec_h264 max-threads=1`
 * (pipeline/cova/pipeline.py:84-99, README.md:94-114): H.264 access units in, one frame per picture out whose first
 * (w/16) * (h/16) * 4 bytes are the per-macroblock records `metapreprocess` copies (metapreprocess/imp.rs:233,311-312).
 * Nothing is reconstructed: the macroblock layer is entropy-decoded by covahip_h264_decode_au (include/covahip.h says what
 * the three record bytes are) and the rest of the I420-sized frame is zero.
 *
 *   sink: video/x-h264, stream-format=avc, alignment=au (codec_data = avcC), access units in decode order
 *   src:  video/x-raw, format=I420, width = 16 * width_mbs, height = 16 * height_mbs, frames in OUTPUT order: a picture waits in a
 *         reorder queue until max_num_reorder_frames (VUI; num_ref_frames + 1 without one) later pictures have arrived (or an IDR picture / EOS comes), then the one with
 *         the smallest picture order count leaves; every frame keeps the timestamps of its access unit
 *   property max-threads: accepted for launch-line compatibility with avdec_h264, ignored (one streaming thread per element,
 *         as the reference configures it)
 *   property records (round 5): FALSE (default) = the drop-in form above.  TRUE = src caps application/x-cova-records,
 *         width-mbs, height-mbs: a frame is its macroblocks' PACKED two-byte records (covahip_carrier_pack: everything BlobNet
 *         keeps of a record) and nothing else -- 16 KB per 1080p picture instead of a zero-filled 3 MB I420 frame; `blobnetfilter`
 *         takes them as they are.  Same results: the records are the same, the padding was never read.
 */
#include <gst/gst.h>
#include <string.h>

#include "covahip.h"

GST_DEBUG_CATEGORY_EXTERN(cova_debug);
#define GST_CAT_DEFAULT cova_debug

typedef struct { gint64 key; GstBuffer *buf; } EdHeld;
typedef struct {
    GstElement parent;
    GstPad *sink, *src;
    covahip_h264 *h;
    covahip_h264_info info;
    gsize rec_bytes, frame_bytes;
    GArray *held;          /* EdHeld, unsorted; at most max_num_reorder_frames (or num_ref_frames + 1) + 1 entries */
    guint max_threads;
    gboolean records;      /* property: packed two-byte records out instead of the I420 carrier frame */
    gboolean records_neg;  /* the property's value when the CAPS event arrived: what scratch, the src caps and ed_chain agree on */
    guint8 *scratch;       /* records = TRUE: the four-byte records of the picture being decoded */
    gint fps_n, fps_d;
} GstEntropyDec;
typedef struct { GstElementClass parent_class; } GstEntropyDecClass;
G_DEFINE_TYPE(GstEntropyDec, gst_entropydec, GST_TYPE_ELEMENT)

static GstStaticPadTemplate ed_sink_t = GST_STATIC_PAD_TEMPLATE("sink", GST_PAD_SINK, GST_PAD_ALWAYS,
    GST_STATIC_CAPS("video/x-h264, stream-format=(string)avc, alignment=(string)au"));
static GstStaticPadTemplate ed_src_t = GST_STATIC_PAD_TEMPLATE("src", GST_PAD_SRC, GST_PAD_ALWAYS,
    GST_STATIC_CAPS("video/x-raw, format=(string)I420; application/x-cova-records"));

static GstFlowReturn ed_pop(GstEntropyDec *s, gboolean all) {
    GstFlowReturn ret = GST_FLOW_OK;
    /* how many pictures stay behind: max_num_reorder_frames when the stream's VUI says (E.2.1: no picture is preceded in decoding
     * order and followed in output order by more than that many), num_ref_frames + 1 otherwise */
    const guint depth = s->info.max_num_reorder_frames >= 0 ? (guint)s->info.max_num_reorder_frames : (guint)s->info.num_ref_frames + 1u;
    while (s->held->len > (all ? 0u : depth) && ret == GST_FLOW_OK) {
        guint best = 0;
        for (guint i = 1; i < s->held->len; i++)
            if (g_array_index(s->held, EdHeld, i).key < g_array_index(s->held, EdHeld, best).key) best = i;
        GstBuffer *b = g_array_index(s->held, EdHeld, best).buf;
        g_array_remove_index(s->held, best);
        ret = gst_pad_push(s->src, b);
    }
    return ret;
}
static void ed_drop_all(GstEntropyDec *s) {
    for (guint i = 0; i < s->held->len; i++) gst_buffer_unref(g_array_index(s->held, EdHeld, i).buf);
    g_array_set_size(s->held, 0);
}

static GstFlowReturn ed_chain(GstPad *pad, GstObject *parent, GstBuffer *buf) {
    GstEntropyDec *s = (GstEntropyDec *)parent;
    GstMapInfo in, out;
    covahip_h264_slice hdr;
    gint64 key = 0;
    int rc;
    GstFlowReturn ret = GST_FLOW_OK;
    if (!s->h) {
        GST_ELEMENT_ERROR(s, CORE, NEGOTIATION, ("no caps with codec_data (avcC) before the first access unit"), (NULL));
        gst_buffer_unref(buf);
        return GST_FLOW_NOT_NEGOTIATED;
    }
    GstBuffer *ob = gst_buffer_new_allocate(NULL, s->records_neg ? s->rec_bytes / 2 : s->frame_bytes, NULL);
    if (!ob || !gst_buffer_map(buf, &in, GST_MAP_READ)) {
        if (ob) gst_buffer_unref(ob);
        gst_buffer_unref(buf);
        GST_ELEMENT_ERROR(s, RESOURCE, FAILED, ("cannot allocate / map a %" G_GSIZE_FORMAT "-byte frame", s->frame_bytes), (NULL));
        return GST_FLOW_ERROR;
    }
    if (!gst_buffer_map(ob, &out, GST_MAP_WRITE)) {
        gst_buffer_unmap(buf, &in);
        gst_buffer_unref(ob);
        gst_buffer_unref(buf);
        GST_ELEMENT_ERROR(s, RESOURCE, FAILED, ("cannot map the output frame"), (NULL));
        return GST_FLOW_ERROR;
    }
    if (s->records_neg) {
        rc = covahip_h264_decode_au(s->h, in.data, in.size, s->scratch, s->rec_bytes, &hdr, &key);
        if (rc == COVAHIP_OK) covahip_carrier_pack(s->scratch, s->rec_bytes / 4, (uint16_t *)out.data);
    } else {
        rc = covahip_h264_decode_au(s->h, in.data, in.size, out.data, out.size, &hdr, &key);
        if (rc == COVAHIP_OK) memset(out.data + s->rec_bytes, 0, out.size - s->rec_bytes);
    }
    gst_buffer_unmap(ob, &out);
    gst_buffer_unmap(buf, &in);
    if (rc != COVAHIP_OK) {
        GST_ELEMENT_ERROR(s, STREAM, DECODE, ("covahip_h264_decode_au: %s", covahip_strerror(rc)), (NULL));
        gst_buffer_unref(ob);
        gst_buffer_unref(buf);
        return GST_FLOW_ERROR;
    }
    gst_buffer_copy_into(ob, buf, GST_BUFFER_COPY_TIMESTAMPS, 0, -1);
    GST_BUFFER_DTS(ob) = GST_CLOCK_TIME_NONE;   /* raw frames leave in output order: the access unit's decode time means nothing on them */
    if (hdr.slice_type != 2) GST_BUFFER_FLAG_SET(ob, GST_BUFFER_FLAG_DELTA_UNIT);
    gst_buffer_unref(buf);
    if (hdr.idr) ret = ed_pop(s, TRUE);   /* an IDR picture follows everything before it in output order */
    {
        EdHeld e = {key, ob};
        g_array_append_val(s->held, e);
    }
    if (ret == GST_FLOW_OK) ret = ed_pop(s, FALSE);
    return ret;
}

static gboolean ed_sink_event(GstPad *pad, GstObject *parent, GstEvent *ev) {
    GstEntropyDec *s = (GstEntropyDec *)parent;
    switch (GST_EVENT_TYPE(ev)) {
    case GST_EVENT_CAPS: {
        GstCaps *caps, *out;
        const GValue *cd;
        GstStructure *st;
        GstBuffer *cdb;
        GstMapInfo m;
        int rc;
        gboolean ok;
        gst_event_parse_caps(ev, &caps);
        st = gst_caps_get_structure(caps, 0);
        cd = gst_structure_get_value(st, "codec_data");
        if (!cd || !(cdb = gst_value_get_buffer(cd))) {
            GST_ELEMENT_ERROR(s, CORE, NEGOTIATION, ("video/x-h264 caps without codec_data: put h264parse (stream-format=avc) upstream"), (NULL));
            gst_event_unref(ev);
            return FALSE;
        }
        s->fps_n = 30; s->fps_d = 1;
        gst_structure_get_fraction(st, "framerate", &s->fps_n, &s->fps_d);
        if (s->h) {   /* new parameter sets: what the old stream still holds leaves first, in its own output order */
            ed_pop(s, TRUE);
            covahip_h264_close(s->h);
            s->h = NULL;
        }
        gst_buffer_map(cdb, &m, GST_MAP_READ);
        rc = covahip_h264_open_avcc(m.data, m.size, &s->h);
        gst_buffer_unmap(cdb, &m);
        gst_event_unref(ev);
        if (rc != COVAHIP_OK || covahip_h264_get_info(s->h, &s->info) != COVAHIP_OK) {
            GST_ELEMENT_ERROR(s, STREAM, FORMAT, ("parameter sets not accepted: %s", covahip_strerror(rc)), (NULL));
            return FALSE;
        }
        s->rec_bytes = (gsize)s->info.width_mbs * s->info.height_mbs * 4;
        s->frame_bytes = (gsize)s->info.width_mbs * 16 * s->info.height_mbs * 16 * 3 / 2;
        g_free(s->scratch);
        s->records_neg = s->records;   /* latched: a later g_object_set takes effect with the next CAPS event */
        s->scratch = s->records_neg ? g_malloc0(s->rec_bytes) : NULL;
        if (s->records_neg)
            out = gst_caps_new_simple("application/x-cova-records", "width-mbs", G_TYPE_INT, s->info.width_mbs, "height-mbs", G_TYPE_INT,
                                      s->info.height_mbs, "framerate", GST_TYPE_FRACTION, s->fps_n, s->fps_d, NULL);
        else
            out = gst_caps_new_simple("video/x-raw", "format", G_TYPE_STRING, "I420", "width", G_TYPE_INT, s->info.width_mbs * 16, "height",
                                      G_TYPE_INT, s->info.height_mbs * 16, "framerate", GST_TYPE_FRACTION, s->fps_n, s->fps_d, NULL);
        ok = gst_pad_push_event(s->src, gst_event_new_caps(out));
        gst_caps_unref(out);
        return ok;
    }
    case GST_EVENT_EOS:
        ed_pop(s, TRUE);
        break;
    case GST_EVENT_FLUSH_STOP:
        ed_drop_all(s);
        break;
    default:
        break;
    }
    return gst_pad_event_default(pad, parent, ev);
}

enum { ED_PROP_0, ED_PROP_MAX_THREADS, ED_PROP_RECORDS };
static void ed_set_property(GObject *o, guint id, const GValue *v, GParamSpec *ps) {
    if (id == ED_PROP_MAX_THREADS) ((GstEntropyDec *)o)->max_threads = g_value_get_uint(v);
    else if (id == ED_PROP_RECORDS) ((GstEntropyDec *)o)->records = g_value_get_boolean(v);
}
static void ed_get_property(GObject *o, guint id, GValue *v, GParamSpec *ps) {
    if (id == ED_PROP_MAX_THREADS) g_value_set_uint(v, ((GstEntropyDec *)o)->max_threads);
    else if (id == ED_PROP_RECORDS) g_value_set_boolean(v, ((GstEntropyDec *)o)->records);
}
static void ed_finalize(GObject *o) {
    GstEntropyDec *s = (GstEntropyDec *)o;
    ed_drop_all(s);
    g_array_free(s->held, TRUE);
    if (s->h) covahip_h264_close(s->h);
    g_free(s->scratch);
    G_OBJECT_CLASS(gst_entropydec_parent_class)->finalize(o);
}
static GstStateChangeReturn ed_change_state(GstElement *e, GstStateChange tr) {
    GstStateChangeReturn r = GST_ELEMENT_CLASS(gst_entropydec_parent_class)->change_state(e, tr);
    if (tr == GST_STATE_CHANGE_PAUSED_TO_READY) ed_drop_all((GstEntropyDec *)e);
    return r;
}
static void gst_entropydec_class_init(GstEntropyDecClass *k) {
    GObjectClass *g = G_OBJECT_CLASS(k);
    GstElementClass *e = GST_ELEMENT_CLASS(k);
    g->set_property = ed_set_property;
    g->get_property = ed_get_property;
    g->finalize = ed_finalize;
    e->change_state = ed_change_state;
    g_object_class_install_property(g, ED_PROP_MAX_THREADS,
        g_param_spec_uint("max-threads", "Max threads", "accepted for compatibility with avdec_h264 (pipeline.py:91-92); ignored", 0, 64, 1,
                          G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
    g_object_class_install_property(g, ED_PROP_RECORDS,
        g_param_spec_boolean("records", "Records", "TRUE: packed two-byte macroblock records (application/x-cova-records) instead of "
                             "the I420-sized carrier frame; set before the caps arrive", FALSE,
                             G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS | GST_PARAM_MUTABLE_READY));
    gst_element_class_set_static_metadata(e, "H.264 entropy decoder", "Codec/Decoder/Video",
        "H.264 access units -> per-macroblock records in the first bytes of an I420-sized frame (no reconstruction)", "covahip");
    gst_element_class_add_static_pad_template(e, &ed_sink_t);
    gst_element_class_add_static_pad_template(e, &ed_src_t);
}
static void gst_entropydec_init(GstEntropyDec *s) {
    s->sink = gst_pad_new_from_static_template(&ed_sink_t, "sink");
    s->src = gst_pad_new_from_static_template(&ed_src_t, "src");
    gst_pad_set_chain_function(s->sink, ed_chain);
    gst_pad_set_event_function(s->sink, ed_sink_event);
    gst_element_add_pad(GST_ELEMENT(s), s->sink);
    gst_element_add_pad(GST_ELEMENT(s), s->src);
    s->held = g_array_new(FALSE, FALSE, sizeof(EdHeld));
    s->max_threads = 1;
}
GType gst_entropydec_get_type_public(void) { return gst_entropydec_get_type(); }
