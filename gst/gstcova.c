/*
 * gstcova.c -- GStreamer 1.x plugin "cova": the reference's compressed-domain elements with
 * their names, pads, caps and properties, every one of them a thin marshalling layer over the
 * C-ABI of libcovahip.so (include/covahip.h).
 *
 *   metapreprocess  <- cova-rs/gst-plugins/src/metapreprocess/imp.rs   (BaseTransform, NeverInPlace)
 *   blobnetinfer    <- nvinfer(BlobNet TensorRT engine) + maskcopy:
 *                      config/blobnet/amsterdam_b128.txt, gst-plugins/gst-maskcopy/gstmaskcopy.cpp
 *   blobnetfilter / maskcopy: gstblobnetfilter.c (the batching filter element; maskcopy by name)
 *   bboxcc          <- cova-rs/gst-plugins/src/bboxcc/imp.rs           (BaseTransform, AlwaysInPlace)
 *   sorttracker     <- cova-rs/gst-plugins/src/sorttracker/imp.rs      (BaseTransform, NeverInPlace)
 *   cova            <- cova-rs/gst-plugins/src/cova/imp.rs             (Element, 2 sink pads + src)
 *   bboxsink        <- cova-rs/gst-plugins/src/bboxsink/imp.rs         (BaseSink: bincode boxes -> CSV file)
 *   tfrecordsink    <- cova-rs/gst-plugins/src/tfrecordsink/imp.rs     (BaseSink: RGBA metadata frames + ground
 *                      truth file -> TFRecord file of tf.train.Example, one per frame or per GoP)
 *   h264entropydec  <- the patched avdec_h264 of README.md:94-114 (gstentropydec.c): access units -> record frames
 *
 * The reference elements are Rust (gstreamer-rs); Rust is not available in this build image, so
 * the elements are written in C against the same GStreamer base classes.  All arithmetic and all
 * stream state live behind covahip_* calls; nothing is computed here.
 */
#include <gst/base/gstbasesink.h>
#include <gst/base/gstbasetransform.h>
#include <gst/gst.h>
#include <gst/video/video.h>
#include <arpa/inet.h>
#include <netinet/in.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <unistd.h>

#include "covahip.h"

GST_DEBUG_CATEGORY(cova_debug);
GType gst_blobnetfilter_get_type_public(void);   /* gstblobnetfilter.c */
GType gst_maskcopy_get_type_public(void);
GType gst_entropydec_get_type_public(void);      /* gstentropydec.c */
#define GST_CAT_DEFAULT cova_debug

#define BBOX_CAPS "bbox, width=(int)[0,2147483647], height=(int)[0,2147483647]"
#define MAX_BOXES 4096

/* ===================================================================== metapreprocess */
typedef struct {
    GstBaseTransform parent;
    guint timestep, gamma;
    covahip_stack *stack;
    gsize out_size;
    GMutex lock;
} GstMetaPreprocess;
typedef struct { GstBaseTransformClass parent_class; } GstMetaPreprocessClass;
G_DEFINE_TYPE(GstMetaPreprocess, gst_metapreprocess, GST_TYPE_BASE_TRANSFORM)
enum { MP_PROP_0, MP_PROP_TIMESTEP, MP_PROP_GAMMA };

static void mp_set_property(GObject *o, guint id, const GValue *v, GParamSpec *ps) {
    GstMetaPreprocess *s = (GstMetaPreprocess *)o;
    g_mutex_lock(&s->lock);
    if (id == MP_PROP_TIMESTEP) s->timestep = g_value_get_uint(v);
    else if (id == MP_PROP_GAMMA) s->gamma = g_value_get_uint(v);
    else G_OBJECT_WARN_INVALID_PROPERTY_ID(o, id, ps);
    g_mutex_unlock(&s->lock);
}
static void mp_get_property(GObject *o, guint id, GValue *v, GParamSpec *ps) {
    GstMetaPreprocess *s = (GstMetaPreprocess *)o;
    if (id == MP_PROP_TIMESTEP) g_value_set_uint(v, s->timestep);
    else if (id == MP_PROP_GAMMA) g_value_set_uint(v, s->gamma);
    else G_OBJECT_WARN_INVALID_PROPERTY_ID(o, id, ps);
}
/* imp.rs:247-286: src->sink direction only rewrites the format; sink->src also scales the size */
static GstCaps *mp_transform_caps(GstBaseTransform *bt, GstPadDirection dir, GstCaps *caps, GstCaps *filter) {
    GstMetaPreprocess *s = (GstMetaPreprocess *)bt;
    GstCaps *other = gst_caps_copy(caps);
    for (guint i = 0; i < gst_caps_get_size(other); i++) {
        GstStructure *st = gst_caps_get_structure(other, i);
        if (dir == GST_PAD_SRC) {
            gst_structure_set(st, "format", G_TYPE_STRING, "I420", NULL);
        } else {
            gint w, h, ow, oh;
            gst_structure_set(st, "format", G_TYPE_STRING, "RGBA", NULL);
            if (gst_structure_get_int(st, "width", &w) && gst_structure_get_int(st, "height", &h)) {
                covahip_stack_out_dims(w, h, s->timestep, &ow, &oh);
                gst_structure_set(st, "width", G_TYPE_INT, ow, "height", G_TYPE_INT, oh, NULL);
            }
        }
    }
    if (filter) {
        GstCaps *r = gst_caps_intersect_full(filter, other, GST_CAPS_INTERSECT_FIRST);
        gst_caps_unref(other);
        return r;
    }
    return other;
}
static gboolean mp_get_unit_size(GstBaseTransform *bt, GstCaps *caps, gsize *size) {
    GstVideoInfo info;
    if (!gst_video_info_from_caps(&info, caps)) return FALSE;
    *size = GST_VIDEO_INFO_SIZE(&info);
    return TRUE;
}
static gboolean mp_set_caps(GstBaseTransform *bt, GstCaps *in, GstCaps *out) {
    GstMetaPreprocess *s = (GstMetaPreprocess *)bt;
    GstVideoInfo oi;
    if (!gst_video_info_from_caps(&oi, out)) return FALSE;
    g_mutex_lock(&s->lock);
    if (s->stack) covahip_stack_free(s->stack);
    s->stack = NULL;
    s->out_size = GST_VIDEO_INFO_SIZE(&oi);
    int rc = covahip_stack_new(s->out_size / s->timestep, s->timestep, s->gamma, &s->stack);  /* imp.rs:233 */
    g_mutex_unlock(&s->lock);
    return rc == COVAHIP_OK;
}
static GstFlowReturn mp_transform(GstBaseTransform *bt, GstBuffer *in, GstBuffer *out) {
    GstMetaPreprocess *s = (GstMetaPreprocess *)bt;
    GstMapInfo mi, mo;
    int emitted = 0, rc;
    if (!gst_buffer_map(in, &mi, GST_MAP_READ)) return GST_FLOW_ERROR;
    if (!gst_buffer_map(out, &mo, GST_MAP_WRITE)) { gst_buffer_unmap(in, &mi); return GST_FLOW_ERROR; }
    g_mutex_lock(&s->lock);
    rc = covahip_stack_push(s->stack, mi.data, mi.size, mo.data, mo.size, &emitted);
    g_mutex_unlock(&s->lock);
    gst_buffer_unmap(out, &mo);
    gst_buffer_unmap(in, &mi);
    if (rc != COVAHIP_OK) { GST_ERROR_OBJECT(s, "covahip_stack_push: %s", covahip_strerror(rc)); return GST_FLOW_ERROR; }
    return emitted ? GST_FLOW_OK : GST_BASE_TRANSFORM_FLOW_DROPPED;
}
static gboolean mp_stop(GstBaseTransform *bt) {
    GstMetaPreprocess *s = (GstMetaPreprocess *)bt;
    g_mutex_lock(&s->lock);
    if (s->stack) covahip_stack_free(s->stack);
    s->stack = NULL;
    g_mutex_unlock(&s->lock);
    return TRUE;
}
static void gst_metapreprocess_init(GstMetaPreprocess *s) { s->timestep = 1; s->gamma = 1; g_mutex_init(&s->lock); }
static void gst_metapreprocess_class_init(GstMetaPreprocessClass *k) {
    GObjectClass *g = G_OBJECT_CLASS(k);
    GstElementClass *e = GST_ELEMENT_CLASS(k);
    GstBaseTransformClass *b = GST_BASE_TRANSFORM_CLASS(k);
    g->set_property = mp_set_property;
    g->get_property = mp_get_property;
    g_object_class_install_property(g, MP_PROP_TIMESTEP,
        g_param_spec_uint("timestep", "Time step", "Number of buffers to stack in temporal domain", 1, G_MAXUINT, 1,
                          G_PARAM_READWRITE | GST_PARAM_MUTABLE_READY));
    g_object_class_install_property(g, MP_PROP_GAMMA,
        g_param_spec_uint("gamma", "Gamma", "Value setting how often should frame be passed", 1, G_MAXUINT, 1,
                          G_PARAM_READWRITE | GST_PARAM_MUTABLE_READY));
    gst_element_class_set_static_metadata(e, "Metadata Preprocessor", "Filter/Effect/Converter/Video",
                                          "Preprocess metadatas extracted from avdec (covahip)", "covahip");
    gst_element_class_add_pad_template(e, gst_pad_template_new("src", GST_PAD_SRC, GST_PAD_ALWAYS,
        gst_caps_from_string("video/x-raw, format=(string)RGBA, width=(int)[0,2147483647], height=(int)[0,2147483647]")));
    gst_element_class_add_pad_template(e, gst_pad_template_new("sink", GST_PAD_SINK, GST_PAD_ALWAYS,
        gst_caps_from_string("video/x-raw, format=(string)I420, width=(int)[0,2147483647], height=(int)[0,2147483647]")));
    b->transform_caps = mp_transform_caps;
    b->get_unit_size = mp_get_unit_size;
    b->set_caps = mp_set_caps;
    b->transform = mp_transform;
    b->stop = mp_stop;
    b->passthrough_on_same_caps = FALSE;
}

/* ===================================================================== GPU contexts
 * Every GPU element instance owns its covahip_ctx (HIP stream + model + staging buffers), like an nvinfer
 * instance owns its engine context: two instances with different caps or weights never share state and never
 * serialise on each other.  The element's streaming thread is the only caller of its ctx. */
static covahip_ctx *ctx_new_for_gpu(guint gpu_id) {
    covahip_ctx *c = NULL;
    if (gpu_id >= 16 || covahip_ctx_create((int)gpu_id, &c) != COVAHIP_OK) return NULL;
    return c;
}

/* ===================================================================== blobnetinfer */
typedef struct {
    GstBaseTransform parent;
    gchar *weights;
    guint gpu_id, timestep;
    covahip_ctx *ctx;
    gint w, h;  /* mask size */
    gboolean loaded;
} GstBlobNetInfer;
typedef struct { GstBaseTransformClass parent_class; } GstBlobNetInferClass;
G_DEFINE_TYPE(GstBlobNetInfer, gst_blobnetinfer, GST_TYPE_BASE_TRANSFORM)
enum { BN_PROP_0, BN_PROP_WEIGHTS, BN_PROP_GPU, BN_PROP_TIMESTEP };

static void bn_set_property(GObject *o, guint id, const GValue *v, GParamSpec *ps) {
    GstBlobNetInfer *s = (GstBlobNetInfer *)o;
    if (id == BN_PROP_WEIGHTS) { g_free(s->weights); s->weights = g_value_dup_string(v); }
    else if (id == BN_PROP_GPU) s->gpu_id = g_value_get_uint(v);
    else if (id == BN_PROP_TIMESTEP) s->timestep = g_value_get_uint(v);
    else G_OBJECT_WARN_INVALID_PROPERTY_ID(o, id, ps);
}
static void bn_get_property(GObject *o, guint id, GValue *v, GParamSpec *ps) {
    GstBlobNetInfer *s = (GstBlobNetInfer *)o;
    if (id == BN_PROP_WEIGHTS) g_value_set_string(v, s->weights);
    else if (id == BN_PROP_GPU) g_value_set_uint(v, s->gpu_id);
    else if (id == BN_PROP_TIMESTEP) g_value_set_uint(v, s->timestep);
    else G_OBJECT_WARN_INVALID_PROPERTY_ID(o, id, ps);
}
/* maskcopy caps math (gstmaskcopy.cpp:329-337): GRAY8, same width, height / timestep */
static GstCaps *bn_transform_caps(GstBaseTransform *bt, GstPadDirection dir, GstCaps *caps, GstCaps *filter) {
    GstBlobNetInfer *s = (GstBlobNetInfer *)bt;
    GstCaps *other = gst_caps_copy(caps);
    for (guint i = 0; i < gst_caps_get_size(other); i++) {
        GstStructure *st = gst_caps_get_structure(other, i);
        gint h;
        gst_structure_set(st, "format", G_TYPE_STRING, dir == GST_PAD_SINK ? "GRAY8" : "RGBA", NULL);
        if (gst_structure_get_int(st, "height", &h))
            gst_structure_set(st, "height", G_TYPE_INT, dir == GST_PAD_SINK ? h / (gint)s->timestep : h * (gint)s->timestep, NULL);
    }
    if (filter) {
        GstCaps *r = gst_caps_intersect_full(filter, other, GST_CAPS_INTERSECT_FIRST);
        gst_caps_unref(other);
        return r;
    }
    return other;
}
static gboolean bn_set_caps(GstBaseTransform *bt, GstCaps *in, GstCaps *out) {
    GstBlobNetInfer *s = (GstBlobNetInfer *)bt;
    GstVideoInfo oi;
    gchar *blob = NULL;
    gsize len = 0;
    int rc;
    if (!gst_video_info_from_caps(&oi, out)) return FALSE;
    s->w = GST_VIDEO_INFO_WIDTH(&oi);
    s->h = GST_VIDEO_INFO_HEIGHT(&oi);
    if (!s->weights || !g_file_get_contents(s->weights, &blob, &len, NULL)) {
        GST_ERROR_OBJECT(s, "cannot read weights file '%s'", s->weights ? s->weights : "(unset)");
        return FALSE;
    }
    if (!s->ctx) s->ctx = ctx_new_for_gpu(s->gpu_id);
    if (!s->ctx) { g_free(blob); GST_ERROR_OBJECT(s, "no HIP device %u", s->gpu_id); return FALSE; }
    rc = covahip_blobnet_load(s->ctx, blob, len, s->h, s->w, (int)s->timestep, 1);
    g_free(blob);
    if (rc != COVAHIP_OK) { GST_ERROR_OBJECT(s, "covahip_blobnet_load: %s", covahip_strerror(rc)); return FALSE; }
    s->loaded = TRUE;
    return TRUE;
}
static GstFlowReturn bn_transform(GstBaseTransform *bt, GstBuffer *in, GstBuffer *out) {
    GstBlobNetInfer *s = (GstBlobNetInfer *)bt;
    GstMapInfo mi, mo;
    int rc;
    if (!gst_buffer_map(in, &mi, GST_MAP_READ)) return GST_FLOW_ERROR;
    if (!gst_buffer_map(out, &mo, GST_MAP_WRITE)) { gst_buffer_unmap(in, &mi); return GST_FLOW_ERROR; }
    if (mi.size < (gsize)s->w * s->h * s->timestep * 4 || mo.size < (gsize)s->w * s->h) rc = COVAHIP_ERR_OVERFLOW;
    else {
        rc = covahip_blobnet_forward(s->ctx, mi.data, 1, NULL, mo.data, COVAHIP_MEM_HOST);
    }
    gst_buffer_unmap(out, &mo);
    gst_buffer_unmap(in, &mi);
    if (rc != COVAHIP_OK) { GST_ERROR_OBJECT(s, "covahip_blobnet_forward: %s", covahip_strerror(rc)); return GST_FLOW_ERROR; }
    return GST_FLOW_OK;
}
static void bn_finalize(GObject *o) {
    GstBlobNetInfer *s = (GstBlobNetInfer *)o;
    if (s->ctx) covahip_ctx_destroy(s->ctx);
    g_free(s->weights);
    G_OBJECT_CLASS(gst_blobnetinfer_parent_class)->finalize(o);
}
static void gst_blobnetinfer_init(GstBlobNetInfer *s) { s->timestep = 4; }
static void gst_blobnetinfer_class_init(GstBlobNetInferClass *k) {
    GObjectClass *g = G_OBJECT_CLASS(k);
    GstElementClass *e = GST_ELEMENT_CLASS(k);
    GstBaseTransformClass *b = GST_BASE_TRANSFORM_CLASS(k);
    g->set_property = bn_set_property;
    g->get_property = bn_get_property;
    g->finalize = bn_finalize;
    g_object_class_install_property(g, BN_PROP_WEIGHTS,
        g_param_spec_string("model-weights-file", "Weights", "BlobNet weight blob (cova_amd/weights.py format)", NULL,
                            G_PARAM_READWRITE | GST_PARAM_MUTABLE_READY));
    g_object_class_install_property(g, BN_PROP_GPU,
        g_param_spec_uint("gpu-id", "GPU id", "HIP device to run on", 0, 15, 0, G_PARAM_READWRITE | GST_PARAM_MUTABLE_READY));
    g_object_class_install_property(g, BN_PROP_TIMESTEP,
        g_param_spec_uint("timestep", "Time step", "Stacked frames per input buffer (maskcopy timestep)", 4, 4, 4,
                          G_PARAM_READWRITE | GST_PARAM_MUTABLE_READY));
    gst_element_class_set_static_metadata(e, "BlobNet inference + mask copy", "Filter/Video",
                                          "BlobNet forward on MI355X (replaces nvinfer + maskcopy)", "covahip");
    gst_element_class_add_pad_template(e, gst_pad_template_new("src", GST_PAD_SRC, GST_PAD_ALWAYS,
        gst_caps_from_string("video/x-raw, format=(string)GRAY8, width=(int)[1,2147483647], height=(int)[1,2147483647]")));
    gst_element_class_add_pad_template(e, gst_pad_template_new("sink", GST_PAD_SINK, GST_PAD_ALWAYS,
        gst_caps_from_string("video/x-raw, format=(string)RGBA, width=(int)[1,2147483647], height=(int)[1,2147483647]")));
    b->transform_caps = bn_transform_caps;
    b->get_unit_size = mp_get_unit_size;
    b->set_caps = bn_set_caps;
    b->transform = bn_transform;
    b->passthrough_on_same_caps = FALSE;
}

/* ===================================================================== bboxcc */
typedef struct {
    GstBaseTransform parent;
    guint cc_threshold, gpu_id;
    covahip_ctx *ctx;
    gint w, h;
    covahip_box *boxes;
    covahip_bbox *bboxes;
} GstBboxCc;
typedef struct { GstBaseTransformClass parent_class; } GstBboxCcClass;
G_DEFINE_TYPE(GstBboxCc, gst_bboxcc, GST_TYPE_BASE_TRANSFORM)
enum { CC_PROP_0, CC_PROP_THRESHOLD, CC_PROP_GPU };

static void cc_set_property(GObject *o, guint id, const GValue *v, GParamSpec *ps) {
    GstBboxCc *s = (GstBboxCc *)o;
    if (id == CC_PROP_THRESHOLD) s->cc_threshold = g_value_get_uint(v);
    else if (id == CC_PROP_GPU) s->gpu_id = g_value_get_uint(v);
    else G_OBJECT_WARN_INVALID_PROPERTY_ID(o, id, ps);
}
static void cc_get_property(GObject *o, guint id, GValue *v, GParamSpec *ps) {
    GstBboxCc *s = (GstBboxCc *)o;
    if (id == CC_PROP_THRESHOLD) g_value_set_uint(v, s->cc_threshold);
    else if (id == CC_PROP_GPU) g_value_set_uint(v, s->gpu_id);
    else G_OBJECT_WARN_INVALID_PROPERTY_ID(o, id, ps);
}
/* imp.rs:192-230: sink caps video/x-raw(w,h) -> "bbox", width, height; src direction -> ANY */
static GstCaps *cc_transform_caps(GstBaseTransform *bt, GstPadDirection dir, GstCaps *caps, GstCaps *filter) {
    GstCaps *other;
    if (dir == GST_PAD_SINK) {
        other = gst_caps_new_empty();
        for (guint i = 0; i < gst_caps_get_size(caps); i++) {
            GstStructure *c = gst_caps_get_structure(caps, i);
            GstStructure *o = gst_structure_new_empty("bbox");
            gint w, h;
            if (gst_structure_get_int(c, "width", &w) && gst_structure_get_int(c, "height", &h))
                gst_structure_set(o, "width", G_TYPE_INT, w, "height", G_TYPE_INT, h, NULL);
            gst_caps_append_structure(other, o);
        }
    } else {
        other = gst_caps_new_any();
    }
    if (filter) {
        GstCaps *r = gst_caps_intersect_full(filter, other, GST_CAPS_INTERSECT_FIRST);
        gst_caps_unref(other);
        return r;
    }
    return other;
}
static gboolean cc_set_caps(GstBaseTransform *bt, GstCaps *in, GstCaps *out) {
    GstBboxCc *s = (GstBboxCc *)bt;
    GstVideoInfo ii;
    if (!gst_video_info_from_caps(&ii, in)) return FALSE;
    s->w = GST_VIDEO_INFO_WIDTH(&ii);
    s->h = GST_VIDEO_INFO_HEIGHT(&ii);
    if (!s->ctx) s->ctx = ctx_new_for_gpu(s->gpu_id);
    if (!s->ctx) { GST_ERROR_OBJECT(s, "no HIP device %u", s->gpu_id); return FALSE; }
    return TRUE;
}
static gboolean cc_transform_size(GstBaseTransform *bt, GstPadDirection dir, GstCaps *caps, gsize size, GstCaps *other,
                                  gsize *othersize) {
    *othersize = size;
    return TRUE;
}
static GstFlowReturn cc_transform_ip(GstBaseTransform *bt, GstBuffer *buf) {
    GstBboxCc *s = (GstBboxCc *)bt;
    GstMapInfo m;
    int32_t count = 0;
    int rc, st = 0;
    gsize len;
    if (!gst_buffer_map(buf, &m, GST_MAP_READ)) return GST_FLOW_ERROR;
    if (m.size < (gsize)s->w * s->h) { gst_buffer_unmap(buf, &m); return GST_FLOW_ERROR; }
    rc = covahip_bboxcc(s->ctx, m.data, 1, s->h, s->w, (int)s->cc_threshold, s->boxes, &count, MAX_BOXES, COVAHIP_MEM_HOST);
    gst_buffer_unmap(buf, &m);
    if (rc != COVAHIP_OK || count > MAX_BOXES) { GST_ERROR_OBJECT(s, "covahip_bboxcc failed (%d, %d boxes)", rc, count); return GST_FLOW_ERROR; }
    covahip_boxes_to_bbox(s->boxes, count, s->bboxes);
    len = covahip_bbox_serialize_vec(s->bboxes, (size_t)count, NULL, 0, NULL);
    {   /* imp.rs:252-262: reallocate when the serialised list does not fit, else shrink */
        gsize maxsize = 0;
        gst_buffer_get_sizes(buf, NULL, &maxsize);
        if (maxsize < len) gst_buffer_replace_all_memory(buf, gst_allocator_alloc(NULL, len, NULL));
        else gst_buffer_set_size(buf, len);
    }
    if (!gst_buffer_map(buf, &m, GST_MAP_WRITE)) return GST_FLOW_ERROR;
    covahip_bbox_serialize_vec(s->bboxes, (size_t)count, m.data, m.size, &st);
    gst_buffer_unmap(buf, &m);
    return st == COVAHIP_OK ? GST_FLOW_OK : GST_FLOW_ERROR;
}
static void cc_finalize(GObject *o) {
    GstBboxCc *s = (GstBboxCc *)o;
    if (s->ctx) covahip_ctx_destroy(s->ctx);
    g_free(s->boxes);
    g_free(s->bboxes);
    G_OBJECT_CLASS(gst_bboxcc_parent_class)->finalize(o);
}
static void gst_bboxcc_init(GstBboxCc *s) {
    s->cc_threshold = 30;  /* DEFAULT_CC_THRESHOLD, imp.rs:16 */
    s->boxes = g_new0(covahip_box, MAX_BOXES);
    s->bboxes = g_new0(covahip_bbox, MAX_BOXES);
}
static void gst_bboxcc_class_init(GstBboxCcClass *k) {
    GObjectClass *g = G_OBJECT_CLASS(k);
    GstElementClass *e = GST_ELEMENT_CLASS(k);
    GstBaseTransformClass *b = GST_BASE_TRANSFORM_CLASS(k);
    g->set_property = cc_set_property;
    g->get_property = cc_get_property;
    g->finalize = cc_finalize;
    g_object_class_install_property(g, CC_PROP_THRESHOLD,
        g_param_spec_uint("cc-threshold", "Threshold of Connected Components",
                          "Connected component with area smaller than the threshold is ignored", 0, G_MAXUINT, 30,
                          G_PARAM_READWRITE | GST_PARAM_MUTABLE_PLAYING));
    g_object_class_install_property(g, CC_PROP_GPU,
        g_param_spec_uint("gpu-id", "GPU id", "HIP device to run on", 0, 15, 0, G_PARAM_READWRITE | GST_PARAM_MUTABLE_READY));
    gst_element_class_set_static_metadata(e, "BBox generator with conneted component", "Filter/Video",
                                          "Conneted component algorithm based bounding box generator (covahip)", "covahip");
    gst_element_class_add_pad_template(e, gst_pad_template_new("src", GST_PAD_SRC, GST_PAD_ALWAYS, gst_caps_from_string(BBOX_CAPS)));
    /* the reference's sink template constrains a misspelt field ("formats"), i.e. any raw video */
    gst_element_class_add_pad_template(e, gst_pad_template_new("sink", GST_PAD_SINK, GST_PAD_ALWAYS,
        gst_caps_from_string("video/x-raw, width=(int)[0,2147483647], height=(int)[0,2147483647]")));
    b->transform_caps = cc_transform_caps;
    b->set_caps = cc_set_caps;
    b->transform_size = cc_transform_size;
    b->transform_ip = cc_transform_ip;
    b->passthrough_on_same_caps = FALSE;
    b->transform_ip_on_passthrough = TRUE;
}

/* ===================================================================== sorttracker */
typedef struct {
    GstBaseTransform parent;
    gfloat iou_threshold;
    guint maxage, minhits;
    covahip_sort *sort;
    covahip_bbox *in, *out;
    GMutex lock;
} GstSortTracker;
typedef struct { GstBaseTransformClass parent_class; } GstSortTrackerClass;
G_DEFINE_TYPE(GstSortTracker, gst_sorttracker, GST_TYPE_BASE_TRANSFORM)
enum { ST_PROP_0, ST_PROP_IOU, ST_PROP_MAXAGE, ST_PROP_MINHITS };
#define ST_CAP_BOXES 65536
#define ST_OUT_BYTES (1 << 21) /* "FIXME: constant 2MB for now", imp.rs:317-327 */

static void st_set_property(GObject *o, guint id, const GValue *v, GParamSpec *ps) {
    GstSortTracker *s = (GstSortTracker *)o;
    if (id == ST_PROP_IOU) s->iou_threshold = g_value_get_float(v);
    else if (id == ST_PROP_MAXAGE) s->maxage = g_value_get_uint(v);
    else if (id == ST_PROP_MINHITS) s->minhits = g_value_get_uint(v);
    else G_OBJECT_WARN_INVALID_PROPERTY_ID(o, id, ps);
}
static void st_get_property(GObject *o, guint id, GValue *v, GParamSpec *ps) {
    GstSortTracker *s = (GstSortTracker *)o;
    if (id == ST_PROP_IOU) g_value_set_float(v, s->iou_threshold);
    else if (id == ST_PROP_MAXAGE) g_value_set_uint(v, s->maxage);
    else if (id == ST_PROP_MINHITS) g_value_set_uint(v, s->minhits);
    else G_OBJECT_WARN_INVALID_PROPERTY_ID(o, id, ps);
}
static GstCaps *st_transform_caps(GstBaseTransform *bt, GstPadDirection dir, GstCaps *caps, GstCaps *filter) {
    return gst_caps_copy(caps);  /* imp.rs:291-315 (the filter is ignored there too) */
}
static gboolean st_set_caps(GstBaseTransform *bt, GstCaps *in, GstCaps *out) {
    GstSortTracker *s = (GstSortTracker *)bt;
    int rc;
    g_mutex_lock(&s->lock);
    if (s->sort) covahip_sort_free(s->sort);
    s->sort = NULL;
    rc = covahip_sort_new(s->maxage, s->minhits, s->iou_threshold, &s->sort);  /* imp.rs:209-236 */
    g_mutex_unlock(&s->lock);
    return rc == COVAHIP_OK;
}
static gboolean st_transform_size(GstBaseTransform *bt, GstPadDirection dir, GstCaps *caps, gsize size, GstCaps *other,
                                  gsize *othersize) {
    *othersize = ST_OUT_BYTES;
    return TRUE;
}
static GstFlowReturn st_transform(GstBaseTransform *bt, GstBuffer *in, GstBuffer *out) {
    GstSortTracker *s = (GstSortTracker *)bt;
    GstMapInfo mi, mo;
    size_t n = 0, nd = 0, nt = 0, len;
    int rc, st = 0;
    if (!GST_BUFFER_PTS_IS_VALID(in)) return GST_FLOW_ERROR;
    if (!gst_buffer_map(in, &mi, GST_MAP_READ)) return GST_FLOW_ERROR;
    rc = covahip_bbox_deserialize_vec(mi.data, mi.size, s->in, ST_CAP_BOXES, &n);
    gst_buffer_unmap(in, &mi);
    if (rc != COVAHIP_OK) return GST_FLOW_ERROR;
    g_mutex_lock(&s->lock);
    rc = covahip_sort_update(s->sort, s->in, n, GST_BUFFER_PTS(in), s->out, ST_CAP_BOXES, &nd, NULL, 0, &nt);
    g_mutex_unlock(&s->lock);
    if (rc != COVAHIP_OK) return GST_FLOW_ERROR;
    if (!gst_buffer_map(out, &mo, GST_MAP_WRITE)) return GST_FLOW_ERROR;
    len = covahip_bbox_serialize_vec(s->out, nd, mo.data, mo.size, &st);
    gst_buffer_unmap(out, &mo);
    if (st != COVAHIP_OK) return GST_FLOW_ERROR;
    gst_buffer_set_size(out, len);
    return GST_FLOW_OK;
}
static gboolean st_sink_event(GstBaseTransform *bt, GstEvent *ev) {
    GstSortTracker *s = (GstSortTracker *)bt;
    if (GST_EVENT_TYPE(ev) == GST_EVENT_EOS && s->sort) {  /* imp.rs:271-289: push finalize() before EOS */
        size_t nd = 0, nt = 0, len;
        int st = 0;
        GstBuffer *b;
        GstMapInfo m;
        g_mutex_lock(&s->lock);
        covahip_sort_finalize(s->sort, s->out, ST_CAP_BOXES, &nd, NULL, 0, &nt);
        g_mutex_unlock(&s->lock);
        len = covahip_bbox_serialize_vec(s->out, nd, NULL, 0, NULL);
        b = gst_buffer_new_allocate(NULL, len, NULL);
        gst_buffer_map(b, &m, GST_MAP_WRITE);
        covahip_bbox_serialize_vec(s->out, nd, m.data, m.size, &st);
        gst_buffer_unmap(b, &m);
        gst_pad_push(GST_BASE_TRANSFORM_SRC_PAD(bt), b);
    }
    return GST_BASE_TRANSFORM_CLASS(gst_sorttracker_parent_class)->sink_event(bt, ev);
}
static void st_finalize(GObject *o) {
    GstSortTracker *s = (GstSortTracker *)o;
    if (s->sort) covahip_sort_free(s->sort);
    g_free(s->in);
    g_free(s->out);
    G_OBJECT_CLASS(gst_sorttracker_parent_class)->finalize(o);
}
static void gst_sorttracker_init(GstSortTracker *s) {
    s->iou_threshold = 0.1f; s->maxage = 30; s->minhits = 30;  /* imp.rs:19-21 */
    s->in = g_new0(covahip_bbox, ST_CAP_BOXES);
    s->out = g_new0(covahip_bbox, ST_CAP_BOXES);
    g_mutex_init(&s->lock);
}
static void gst_sorttracker_class_init(GstSortTrackerClass *k) {
    GObjectClass *g = G_OBJECT_CLASS(k);
    GstElementClass *e = GST_ELEMENT_CLASS(k);
    GstBaseTransformClass *b = GST_BASE_TRANSFORM_CLASS(k);
    g->set_property = st_set_property;
    g->get_property = st_get_property;
    g->finalize = st_finalize;
    g_object_class_install_property(g, ST_PROP_IOU,
        g_param_spec_float("iou-threshold", "IoU threshold", "IoU threshold used for matching objects", 0.f, 1.f, 0.1f,
                           G_PARAM_READWRITE | GST_PARAM_MUTABLE_PLAYING));
    g_object_class_install_property(g, ST_PROP_MAXAGE,
        g_param_spec_uint("maxage", "Max age", "Maximum time that the track can be retained without any matches", 0,
                          G_MAXUINT, 30, G_PARAM_READWRITE | GST_PARAM_MUTABLE_PLAYING));
    g_object_class_install_property(g, ST_PROP_MINHITS,
        g_param_spec_uint("minhits", "Min hits", "Minimum number of hits to be considered valid track", 0, G_MAXUINT, 30,
                          G_PARAM_READWRITE | GST_PARAM_MUTABLE_PLAYING));
    gst_element_class_set_static_metadata(e, "SORT tracker", "Filter/Effect/Converter/Video",
                                          "SORT tracking of bboxcc boxes (covahip)", "covahip");
    gst_element_class_add_pad_template(e, gst_pad_template_new("src", GST_PAD_SRC, GST_PAD_ALWAYS, gst_caps_from_string(BBOX_CAPS)));
    gst_element_class_add_pad_template(e, gst_pad_template_new("sink", GST_PAD_SINK, GST_PAD_ALWAYS, gst_caps_from_string(BBOX_CAPS)));
    b->transform_caps = st_transform_caps;
    b->set_caps = st_set_caps;
    b->transform_size = st_transform_size;
    b->transform = st_transform;
    b->sink_event = st_sink_event;
    b->passthrough_on_same_caps = FALSE;
}

/* ===================================================================== cova */
typedef struct {
    GstElement parent;
    GstPad *src, *sink_mask, *sink_enc;
    covahip_gopfilter_cfg cfg;
    guint port;
    gboolean debug;
    covahip_gopfilter *filter;
    GHashTable *bufs;  /* id -> GstBuffer* */
    guint64 next_id;
    gboolean eos[2];
    covahip_bbox *boxes;
    covahip_au_out *out;
    int sock;          /* aggregator connection (`port` != 0; cova/tracker.rs:24-30), -1 when closed */
    GMutex lock;
} GstCova;
typedef struct { GstElementClass parent_class; } GstCovaClass;
G_DEFINE_TYPE(GstCova, gst_cova, GST_TYPE_ELEMENT)
enum { CV_PROP_0, CV_PROP_IOU, CV_PROP_MAXAGE, CV_PROP_MINHITS, CV_PROP_PORT, CV_PROP_INFER_I, CV_PROP_DEBUG, CV_PROP_ALPHA,
       CV_PROP_BETA, CV_PROP_DROPPED, CV_PROP_DEC_DEP, CV_PROP_DEC_INF, CV_PROP_HELD };
#define CV_CAP_OUT 65536

static void cv_ensure_filter(GstCova *s) {
    if (!s->filter) covahip_gopfilter_new(&s->cfg, &s->filter);
}
static void cv_set_property(GObject *o, guint id, const GValue *v, GParamSpec *ps) {
    GstCova *s = (GstCova *)o;
    g_mutex_lock(&s->lock);
    switch (id) {
        case CV_PROP_IOU: s->cfg.sort_iou = g_value_get_float(v); break;
        case CV_PROP_MAXAGE: s->cfg.sort_maxage = g_value_get_uint(v); break;
        case CV_PROP_MINHITS: s->cfg.sort_minhits = g_value_get_uint(v); break;
        case CV_PROP_PORT: s->port = g_value_get_uint(v); break;
        case CV_PROP_INFER_I: s->cfg.infer_i = g_value_get_boolean(v); break;
        case CV_PROP_DEBUG: s->debug = g_value_get_boolean(v); break;
        case CV_PROP_ALPHA: s->cfg.alpha = g_value_get_uint(v); break;
        case CV_PROP_BETA: s->cfg.beta = g_value_get_uint(v); break;
        default: G_OBJECT_WARN_INVALID_PROPERTY_ID(o, id, ps);
    }
    g_mutex_unlock(&s->lock);
}
static void cv_get_property(GObject *o, guint id, GValue *v, GParamSpec *ps) {
    GstCova *s = (GstCova *)o;
    uint64_t d = 0, dd = 0, di = 0;
    g_mutex_lock(&s->lock);
    if (s->filter) covahip_gopfilter_counters(s->filter, &d, &dd, &di);
    switch (id) {
        case CV_PROP_IOU: g_value_set_float(v, s->cfg.sort_iou); break;
        case CV_PROP_MAXAGE: g_value_set_uint(v, s->cfg.sort_maxage); break;
        case CV_PROP_MINHITS: g_value_set_uint(v, s->cfg.sort_minhits); break;
        case CV_PROP_PORT: g_value_set_uint(v, s->port); break;
        case CV_PROP_INFER_I: g_value_set_boolean(v, s->cfg.infer_i); break;
        case CV_PROP_DEBUG: g_value_set_boolean(v, s->debug); break;
        case CV_PROP_ALPHA: g_value_set_uint(v, s->cfg.alpha); break;
        case CV_PROP_BETA: g_value_set_uint(v, s->cfg.beta); break;
        case CV_PROP_DROPPED: g_value_set_uint64(v, d); break;
        case CV_PROP_DEC_DEP: g_value_set_uint64(v, dd); break;
        case CV_PROP_DEC_INF: g_value_set_uint64(v, di); break;
        case CV_PROP_HELD: g_value_set_uint64(v, g_hash_table_size(s->bufs)); break;
        default: G_OBJECT_WARN_INVALID_PROPERTY_ID(o, id, ps);
    }
    g_mutex_unlock(&s->lock);
}
/* Access units the filter has discarded are released here: the reference frees a GoP's buffers when it drops
 * the GoP (imp.rs:268-305); without this the element would hold nearly the whole bitstream until EOS. */
static void cv_release_dropped(GstCova *s) {
    uint64_t ids[1024];
    size_t n;
    do {
        n = 0;
        if (covahip_gopfilter_take_dropped(s->filter, ids, 1024, &n) != COVAHIP_OK) return;
        for (size_t i = 0; i < n; i++) g_hash_table_remove(s->bufs, GSIZE_TO_POINTER((gsize)ids[i]));
    } while (n == 1024);
}
/* Finished tracks go to the aggregator as length-delimited bincode Frames (cova/tracker.rs:59-83,91-118). */
static void cv_send_tracks(GstCova *s) {
    int st = 0;
    const size_t need = covahip_gopfilter_take_track_export(s->filter, NULL, 0, NULL);
    if (!need) return;
    guint8 *wire = g_malloc(need);
    covahip_gopfilter_take_track_export(s->filter, wire, need, &st);
    if (st == COVAHIP_OK && s->port != 0) {
        if (s->sock < 0) {   /* Tracker::new connects to 127.0.0.1:port (tracker.rs:24-30) */
            struct sockaddr_in a;
            memset(&a, 0, sizeof a);
            a.sin_family = AF_INET;
            a.sin_port = htons((guint16)s->port);
            a.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
            s->sock = socket(AF_INET, SOCK_STREAM, 0);
            if (s->sock >= 0 && connect(s->sock, (struct sockaddr *)&a, sizeof a) != 0) { close(s->sock); s->sock = -1; }
            if (s->sock < 0) GST_ELEMENT_WARNING(s, RESOURCE, OPEN_WRITE, ("Socket Creation Failed (port %u)", s->port), (NULL));
        }
        for (size_t off = 0; s->sock >= 0 && off < need;) {
            const ssize_t w = send(s->sock, wire + off, need - off, MSG_NOSIGNAL);
            if (w <= 0) { close(s->sock); s->sock = -1; break; }
            off += (size_t)w;
        }
    }
    g_free(wire);
}
/* Turns the access units the filter released into BufferLists and pushes them (imp.rs:293-303). */
static GstFlowReturn cv_push_out(GstCova *s, size_t n) {
    GstFlowReturn ret = GST_FLOW_OK;
    size_t i = 0;
    while (i < n) {
        GstBufferList *list = gst_buffer_list_new();
        const uint32_t li = s->out[i].list;
        for (; i < n && s->out[i].list == li; i++) {
            GstBuffer *b = g_hash_table_lookup(s->bufs, GSIZE_TO_POINTER((gsize)s->out[i].id));
            if (!b) continue;
            g_hash_table_steal(s->bufs, GSIZE_TO_POINTER((gsize)s->out[i].id));
            b = gst_buffer_make_writable(b);
            if (s->out[i].flags & COVAHIP_AU_DISCONT) GST_BUFFER_FLAG_SET(b, GST_BUFFER_FLAG_DISCONT);
            if (s->out[i].flags & COVAHIP_AU_DROPPABLE) GST_BUFFER_FLAG_SET(b, GST_BUFFER_FLAG_DROPPABLE);
            gst_buffer_list_add(list, b);
        }
        g_mutex_unlock(&s->lock);
        ret = gst_pad_push_list(s->src, list);
        g_mutex_lock(&s->lock);
        if (ret != GST_FLOW_OK) break;
    }
    return ret;
}
static GstFlowReturn cv_sink_enc_chain(GstPad *pad, GstObject *parent, GstBuffer *buf) {
    GstCova *s = (GstCova *)parent;
    int rc;
    if (!GST_BUFFER_PTS_IS_VALID(buf)) { gst_buffer_unref(buf); return GST_FLOW_ERROR; }
    g_mutex_lock(&s->lock);
    cv_ensure_filter(s);
    const guint64 id = ++s->next_id;
    rc = covahip_gopfilter_push_enc(s->filter, id, GST_BUFFER_PTS(buf),
                                    GST_BUFFER_FLAG_IS_SET(buf, GST_BUFFER_FLAG_DELTA_UNIT) ? COVAHIP_AU_DELTA_UNIT : 0);
    if (rc == COVAHIP_OK) g_hash_table_insert(s->bufs, GSIZE_TO_POINTER((gsize)id), buf);
    else gst_buffer_unref(buf);
    g_mutex_unlock(&s->lock);
    return rc == COVAHIP_OK ? GST_FLOW_OK : GST_FLOW_ERROR;
}
static GstFlowReturn cv_sink_mask_chain(GstPad *pad, GstObject *parent, GstBuffer *buf) {
    GstCova *s = (GstCova *)parent;
    GstMapInfo m;
    size_t n = 0, nout = 0;
    int rc;
    GstFlowReturn ret = GST_FLOW_ERROR;
    if (!GST_BUFFER_PTS_IS_VALID(buf) || !gst_buffer_map(buf, &m, GST_MAP_READ)) { gst_buffer_unref(buf); return GST_FLOW_ERROR; }
    rc = covahip_bbox_deserialize_vec(m.data, m.size, s->boxes, MAX_BOXES, &n);
    gst_buffer_unmap(buf, &m);
    if (rc == COVAHIP_OK) {
        g_mutex_lock(&s->lock);
        cv_ensure_filter(s);
        rc = covahip_gopfilter_push_boxes(s->filter, s->boxes, n, GST_BUFFER_PTS(buf), s->out, CV_CAP_OUT, &nout);
        cv_release_dropped(s);
        cv_send_tracks(s);
        ret = rc == COVAHIP_OK ? cv_push_out(s, nout) : GST_FLOW_ERROR;
        g_mutex_unlock(&s->lock);
    }
    gst_buffer_unref(buf);
    return ret;
}
/* imp.rs:361-432: EOS is forwarded once both sinks have seen it, after flushing the GoP lists */
static gboolean cv_both_eos(GstCova *s, GstEvent *ev) {
    size_t nout = 0;
    g_mutex_lock(&s->lock);
    cv_ensure_filter(s);
    covahip_gopfilter_eos(s->filter, s->out, CV_CAP_OUT, &nout);
    cv_release_dropped(s);
    cv_send_tracks(s);
    if (s->sock >= 0) { shutdown(s->sock, SHUT_RDWR); close(s->sock); s->sock = -1; }   /* tracker.rs:120-124 */
    cv_push_out(s, nout);
    g_hash_table_remove_all(s->bufs);
    g_mutex_unlock(&s->lock);
    return gst_pad_push_event(s->src, ev);
}
static gboolean cv_sink_mask_event(GstPad *pad, GstObject *parent, GstEvent *ev) {
    GstCova *s = (GstCova *)parent;
    if (GST_EVENT_TYPE(ev) == GST_EVENT_EOS) {
        s->eos[1] = TRUE;
        if (s->eos[0] && s->eos[1]) return cv_both_eos(s, ev);
    }
    gst_event_unref(ev);
    return TRUE;
}
static gboolean cv_sink_enc_event(GstPad *pad, GstObject *parent, GstEvent *ev) {
    GstCova *s = (GstCova *)parent;
    if (GST_EVENT_TYPE(ev) == GST_EVENT_EOS) {
        s->eos[0] = TRUE;
        if (s->eos[0] && s->eos[1]) return cv_both_eos(s, ev);
        gst_event_unref(ev);
        return TRUE;
    }
    return gst_pad_push_event(s->src, ev);
}
static gboolean cv_sink_query(GstPad *pad, GstObject *parent, GstQuery *q) {
    GstCova *s = (GstCova *)parent;
    if (GST_QUERY_TYPE(q) == GST_QUERY_CAPS || GST_QUERY_TYPE(q) == GST_QUERY_ACCEPT_CAPS) return gst_pad_query_default(pad, parent, q);
    return gst_pad_peer_query(s->src, q);
}
static void cv_finalize(GObject *o) {
    GstCova *s = (GstCova *)o;
    if (s->filter) covahip_gopfilter_free(s->filter);
    if (s->sock >= 0) close(s->sock);
    g_hash_table_destroy(s->bufs);
    g_free(s->boxes);
    g_free(s->out);
    G_OBJECT_CLASS(gst_cova_parent_class)->finalize(o);
}
static void gst_cova_init(GstCova *s) {
    GstElementClass *k = GST_ELEMENT_GET_CLASS(s);
    covahip_gopfilter_default_cfg(&s->cfg);
    s->sock = -1;
    g_mutex_init(&s->lock);
    s->bufs = g_hash_table_new_full(g_direct_hash, g_direct_equal, NULL, (GDestroyNotify)gst_buffer_unref);
    s->boxes = g_new0(covahip_bbox, MAX_BOXES);
    s->out = g_new0(covahip_au_out, CV_CAP_OUT);
    s->sink_mask = gst_pad_new_from_template(gst_element_class_get_pad_template(k, "sink_mask"), "sink_mask");
    gst_pad_set_chain_function(s->sink_mask, cv_sink_mask_chain);
    gst_pad_set_event_function(s->sink_mask, cv_sink_mask_event);
    gst_pad_set_query_function(s->sink_mask, cv_sink_query);
    s->sink_enc = gst_pad_new_from_template(gst_element_class_get_pad_template(k, "sink_enc"), "sink_enc");
    gst_pad_set_chain_function(s->sink_enc, cv_sink_enc_chain);
    gst_pad_set_event_function(s->sink_enc, cv_sink_enc_event);
    gst_pad_set_query_function(s->sink_enc, cv_sink_query);
    s->src = gst_pad_new_from_template(gst_element_class_get_pad_template(k, "src"), "src");
    gst_element_add_pad(GST_ELEMENT(s), s->sink_mask);
    gst_element_add_pad(GST_ELEMENT(s), s->sink_enc);
    gst_element_add_pad(GST_ELEMENT(s), s->src);
}
static void gst_cova_class_init(GstCovaClass *k) {
    GObjectClass *g = G_OBJECT_CLASS(k);
    GstElementClass *e = GST_ELEMENT_CLASS(k);
    const GParamFlags rw = G_PARAM_READWRITE | GST_PARAM_MUTABLE_PLAYING;
    g->set_property = cv_set_property;
    g->get_property = cv_get_property;
    g->finalize = cv_finalize;
    g_object_class_install_property(g, CV_PROP_IOU, g_param_spec_float("sort-iou", "Track IoU", "IoU threshold used by SORT", 0.f, 1.f, 0.1f, rw));
    g_object_class_install_property(g, CV_PROP_MAXAGE, g_param_spec_uint("sort-maxage", "Track Max Age", "Max age parameter used by SORT", 0, G_MAXUINT, 30, rw));
    g_object_class_install_property(g, CV_PROP_MINHITS, g_param_spec_uint("sort-minhits", "Track Min Hits", "Min hits parameter used by SORT", 0, G_MAXUINT, 30, rw));
    g_object_class_install_property(g, CV_PROP_PORT, g_param_spec_uint("port", "Port", "TCP port number for attatching to Aggregator (0: disabled)", 0, G_MAXUINT, 0, rw));
    g_object_class_install_property(g, CV_PROP_INFER_I, g_param_spec_boolean("infer-i", "[DEPRECATED] Infer I frame", "[DEPRECATED] Run inference on I frames", FALSE, rw));
    g_object_class_install_property(g, CV_PROP_DEBUG, g_param_spec_boolean("debug", "Debug", "Run in debug mode", FALSE, rw));
    /* the reference declares 30 as the ParamSpec default of alpha/beta but initialises the struct with 0 */
    g_object_class_install_property(g, CV_PROP_ALPHA, g_param_spec_uint("alpha", "Alpha Parameter", "Parameter setting how many extra frames is sent for decoding", 0, G_MAXUINT, 0, rw));
    g_object_class_install_property(g, CV_PROP_BETA, g_param_spec_uint("beta", "Beta Parameter", "Parameter setting how many extra frames is sent for inferencing", 0, G_MAXUINT, 0, rw));
    g_object_class_install_property(g, CV_PROP_DROPPED, g_param_spec_uint64("dropped", "Dropped frame counts", "Dropped frame counts", 0, G_MAXUINT64, 0, G_PARAM_READABLE));
    g_object_class_install_property(g, CV_PROP_DEC_DEP, g_param_spec_uint64("decoded-dependency", "Decoded for dependency counts", "Number of decoded frames for dependency", 0, G_MAXUINT64, 0, G_PARAM_READABLE));
    g_object_class_install_property(g, CV_PROP_DEC_INF, g_param_spec_uint64("decoded-inference", "Decoded for inference counts", "Number of decoded frames for inference", 0, G_MAXUINT64, 0, G_PARAM_READABLE));
    /* not in the reference: how many encoded access units the element holds right now (bounded by the GoP window) */
    g_object_class_install_property(g, CV_PROP_HELD, g_param_spec_uint64("held-buffers", "Held buffers", "Encoded access units currently buffered", 0, G_MAXUINT64, 0, G_PARAM_READABLE));
    gst_element_class_set_static_metadata(e, "CoVA Filter", "Filter/Video",
                                          "Filter optimal frames to decode using SORT on extracted masks (covahip)", "covahip");
    gst_element_class_add_pad_template(e, gst_pad_template_new("src", GST_PAD_SRC, GST_PAD_ALWAYS, gst_caps_new_any()));
    gst_element_class_add_pad_template(e, gst_pad_template_new("sink_mask", GST_PAD_SINK, GST_PAD_ALWAYS, gst_caps_from_string(BBOX_CAPS)));
    gst_element_class_add_pad_template(e, gst_pad_template_new("sink_enc", GST_PAD_SINK, GST_PAD_ALWAYS, gst_caps_new_any()));
}

/* ===================================================================== bboxsink */
typedef struct {
    GstBaseSink parent;
    gchar *location;
    FILE *f;
    gboolean header_done;
    covahip_bbox *boxes;
} GstBboxSink;
typedef struct { GstBaseSinkClass parent_class; } GstBboxSinkClass;
G_DEFINE_TYPE(GstBboxSink, gst_bboxsink, GST_TYPE_BASE_SINK)
enum { BS_PROP_0, BS_PROP_LOCATION };

static void bs_set_property(GObject *o, guint id, const GValue *v, GParamSpec *ps) {
    GstBboxSink *s = (GstBboxSink *)o;
    if (id == BS_PROP_LOCATION) {
        if (s->f) { g_warning("Changing the `location` property on a started `bboxsink` is not supported"); return; }   /* imp.rs:39-46 */
        g_free(s->location);
        s->location = g_value_dup_string(v);
    } else G_OBJECT_WARN_INVALID_PROPERTY_ID(o, id, ps);
}
static void bs_get_property(GObject *o, guint id, GValue *v, GParamSpec *ps) {
    GstBboxSink *s = (GstBboxSink *)o;
    if (id == BS_PROP_LOCATION) g_value_set_string(v, s->location);
    else G_OBJECT_WARN_INVALID_PROPERTY_ID(o, id, ps);
}
static gboolean bs_start(GstBaseSink *bs) {   /* imp.rs:199-233 */
    GstBboxSink *s = (GstBboxSink *)bs;
    if (!s->location) { GST_ELEMENT_ERROR(s, RESOURCE, SETTINGS, ("File location is not defined"), (NULL)); return FALSE; }
    s->f = fopen(s->location, "wb");
    if (!s->f) { GST_ELEMENT_ERROR(s, RESOURCE, OPEN_WRITE, ("Could not open file %s for writing", s->location), (NULL)); return FALSE; }
    setvbuf(s->f, NULL, _IOFBF, 2 * 1024 * 1024);
    s->header_done = FALSE;
    return TRUE;
}
static gboolean bs_stop(GstBaseSink *bs) {
    GstBboxSink *s = (GstBboxSink *)bs;
    if (s->f) { fclose(s->f); s->f = NULL; }
    return TRUE;
}
static GstFlowReturn bs_render(GstBaseSink *bs, GstBuffer *buf) {   /* imp.rs:252-270 */
    GstBboxSink *s = (GstBboxSink *)bs;
    GstMapInfo m;
    size_t n = 0;
    if (!gst_buffer_map(buf, &m, GST_MAP_READ)) return GST_FLOW_ERROR;
    int rc = covahip_bbox_deserialize_vec(m.data, m.size, s->boxes, MAX_BOXES, &n);
    gst_buffer_unmap(buf, &m);
    if (rc != COVAHIP_OK) { GST_ELEMENT_ERROR(s, STREAM, DECODE, ("bad bbox buffer: %s", covahip_strerror(rc)), (NULL)); return GST_FLOW_ERROR; }
    if (n == 0) return GST_FLOW_OK;   /* csv::Writer emits the header with the first record */
    int st = 0;
    const size_t need = covahip_bbox_csv(s->boxes, n, !s->header_done, NULL, 0, NULL);
    char *text = g_malloc(need);
    covahip_bbox_csv(s->boxes, n, !s->header_done, text, need, &st);
    const gboolean ok = st == COVAHIP_OK && fwrite(text, 1, need, s->f) == need;
    g_free(text);
    s->header_done = TRUE;
    return ok ? GST_FLOW_OK : GST_FLOW_ERROR;
}
static void bs_finalize(GObject *o) {
    GstBboxSink *s = (GstBboxSink *)o;
    g_free(s->location);
    g_free(s->boxes);
    G_OBJECT_CLASS(gst_bboxsink_parent_class)->finalize(o);
}
static void gst_bboxsink_init(GstBboxSink *s) { s->boxes = g_new0(covahip_bbox, MAX_BOXES); gst_base_sink_set_sync(GST_BASE_SINK(s), FALSE); }
static void gst_bboxsink_class_init(GstBboxSinkClass *k) {
    GObjectClass *g = G_OBJECT_CLASS(k);
    GstElementClass *e = GST_ELEMENT_CLASS(k);
    GstBaseSinkClass *b = GST_BASE_SINK_CLASS(k);
    g->set_property = bs_set_property;
    g->get_property = bs_get_property;
    g->finalize = bs_finalize;
    g_object_class_install_property(g, BS_PROP_LOCATION,
        g_param_spec_string("location", "File Location", "Location of the file to write", NULL, G_PARAM_READWRITE));
    gst_element_class_set_static_metadata(e, "Bounding box sink", "Sink/Video", "Writes bounding boxes as CSV (covahip)", "covahip");
    gst_element_class_add_pad_template(e, gst_pad_template_new("sink", GST_PAD_SINK, GST_PAD_ALWAYS, gst_caps_from_string(BBOX_CAPS)));
    b->start = bs_start;
    b->stop = bs_stop;
    b->render = bs_render;
}

/* ===================================================================== tfrecordsink */
typedef struct {
    GstBaseSink parent;
    gchar *location, *gt;
    guint gop;               /* 0: one Example per frame; else frames are stacked until the next key frame */
    FILE *f, *fgt;
    gint width, height;
    guint8 *frames, *gts;    /* the open Example: batch_count frames of w*h*4 / w*h bytes */
    guint batch_count, cap_frames;
    GMutex lock;
} GstTfRecordSink;
typedef struct { GstBaseSinkClass parent_class; } GstTfRecordSinkClass;
G_DEFINE_TYPE(GstTfRecordSink, gst_tfrecordsink, GST_TYPE_BASE_SINK)
enum { TF_PROP_0, TF_PROP_LOCATION, TF_PROP_GT, TF_PROP_GOP };

static void tf_set_property(GObject *o, guint id, const GValue *v, GParamSpec *ps) {
    GstTfRecordSink *s = (GstTfRecordSink *)o;
    g_mutex_lock(&s->lock);
    switch (id) {
    case TF_PROP_LOCATION:
        if (s->f) g_warning("Changing the `location` property on a started `tfrecordsink` is not supported");   /* imp.rs:207-214 */
        else { g_free(s->location); s->location = g_value_dup_string(v); }
        break;
    case TF_PROP_GT:
        if (s->f) g_warning("Changing the `gt` property on a started `tfrecordsink` is not supported");
        else { g_free(s->gt); s->gt = g_value_dup_string(v); }
        break;
    case TF_PROP_GOP: s->gop = g_value_get_uint(v); break;
    default: G_OBJECT_WARN_INVALID_PROPERTY_ID(o, id, ps);
    }
    g_mutex_unlock(&s->lock);
}
static void tf_get_property(GObject *o, guint id, GValue *v, GParamSpec *ps) {
    GstTfRecordSink *s = (GstTfRecordSink *)o;
    switch (id) {
    case TF_PROP_LOCATION: g_value_set_string(v, s->location); break;
    case TF_PROP_GT: g_value_set_string(v, s->gt); break;
    case TF_PROP_GOP: g_value_set_uint(v, s->gop); break;
    default: G_OBJECT_WARN_INVALID_PROPERTY_ID(o, id, ps);
    }
}
static gboolean tf_set_caps(GstBaseSink *bs, GstCaps *caps) {   /* imp.rs:439-447 */
    GstTfRecordSink *s = (GstTfRecordSink *)bs;
    GstVideoInfo vi;
    if (!gst_video_info_from_caps(&vi, caps)) return FALSE;
    s->width = GST_VIDEO_INFO_WIDTH(&vi);
    s->height = GST_VIDEO_INFO_HEIGHT(&vi);
    return TRUE;
}
static gboolean tf_start(GstBaseSink *bs) {   /* imp.rs:449-512 */
    GstTfRecordSink *s = (GstTfRecordSink *)bs;
    if (!s->location) { GST_ELEMENT_ERROR(s, RESOURCE, SETTINGS, ("File location is not defined"), (NULL)); return FALSE; }
    if (!s->gt) { GST_ELEMENT_ERROR(s, RESOURCE, SETTINGS, ("Ground truth file is not defined"), (NULL)); return FALSE; }
    s->f = fopen(s->location, "wb");
    if (!s->f) { GST_ELEMENT_ERROR(s, RESOURCE, OPEN_WRITE, ("Could not open file %s for writing", s->location), (NULL)); return FALSE; }
    setvbuf(s->f, NULL, _IOFBF, 2000000);
    s->fgt = fopen(s->gt, "rb");
    if (!s->fgt) {
        GST_ELEMENT_ERROR(s, RESOURCE, OPEN_READ, ("Could not open ground truth file %s", s->gt), (NULL));
        fclose(s->f); s->f = NULL;
        return FALSE;
    }
    s->batch_count = 0;
    return TRUE;
}
static gboolean tf_stop(GstBaseSink *bs) {   /* imp.rs:514-532: flush only -- an open, partly filled GoP is not written */
    GstTfRecordSink *s = (GstTfRecordSink *)bs;
    if (s->f) { fclose(s->f); s->f = NULL; }
    if (s->fgt) { fclose(s->fgt); s->fgt = NULL; }
    s->batch_count = 0;
    return TRUE;
}
static gboolean tf_write(GstTfRecordSink *s, guint gop) {   /* imp.rs:137-178 */
    int st = 0;
    const int pad = (int)gop;
    const size_t need = covahip_tfrecord_example(s->frames, s->gts, (int)s->batch_count, pad, s->width, s->height, NULL, 0, NULL);
    guint8 *rec = g_malloc(need);
    covahip_tfrecord_example(s->frames, s->gts, (int)s->batch_count, pad, s->width, s->height, rec, need, &st);
    const gboolean ok = st == COVAHIP_OK && fwrite(rec, 1, need, s->f) == need;
    g_free(rec);
    s->batch_count = 0;
    return ok;
}
static GstFlowReturn tf_render(GstBaseSink *bs, GstBuffer *buf) {   /* imp.rs:534-606 */
    GstTfRecordSink *s = (GstTfRecordSink *)bs;
    g_mutex_lock(&s->lock);
    const guint gop = s->gop;
    g_mutex_unlock(&s->lock);
    const size_t hw = (size_t)s->width * s->height;
    /* GoP stacking: a key frame closes the previous Example */
    if (gop != 0 && !GST_BUFFER_FLAG_IS_SET(buf, GST_BUFFER_FLAG_DELTA_UNIT) && s->batch_count != 0)
        if (!tf_write(s, gop)) { GST_ELEMENT_ERROR(s, CORE, FAILED, ("Failed to write TFRecord"), (NULL)); return GST_FLOW_ERROR; }
    if (gop != 0 && s->batch_count >= gop) {   /* the reference underflows `gop - batch_count` here */
        GST_ELEMENT_ERROR(s, CORE, FAILED, ("more than `gop` frames without a key frame"), (NULL));
        return GST_FLOW_ERROR;
    }
    if (s->batch_count + 1 > s->cap_frames) {
        s->cap_frames = s->cap_frames ? 2 * s->cap_frames : 8;
        s->frames = g_realloc(s->frames, (gsize)s->cap_frames * hw * 4);
        s->gts = g_realloc(s->gts, (gsize)s->cap_frames * hw);
    }
    GstMapInfo m;
    if (!gst_buffer_map(buf, &m, GST_MAP_READ)) return GST_FLOW_ERROR;
    if (m.size < hw * 4) { gst_buffer_unmap(buf, &m); GST_ELEMENT_ERROR(s, CORE, FAILED, ("Failed to map input buffer readable"), (NULL)); return GST_FLOW_ERROR; }
    memcpy(s->frames + (gsize)s->batch_count * hw * 4, m.data, hw * 4);   /* RGBA rows are tightly packed (stride = 4 * width) */
    gst_buffer_unmap(buf, &m);
    if (fread(s->gts + (gsize)s->batch_count * hw, 1, hw, s->fgt) != hw) {
        GST_ELEMENT_ERROR(s, RESOURCE, READ, ("Could not read file from ground truth file"), (NULL));
        return GST_FLOW_ERROR;
    }
    s->batch_count++;
    if (gop == 0 && !tf_write(s, 0)) { GST_ELEMENT_ERROR(s, CORE, FAILED, ("Failed to write TFRecord"), (NULL)); return GST_FLOW_ERROR; }
    return GST_FLOW_OK;
}
static void tf_finalize(GObject *o) {
    GstTfRecordSink *s = (GstTfRecordSink *)o;
    g_free(s->location); g_free(s->gt); g_free(s->frames); g_free(s->gts);
    g_mutex_clear(&s->lock);
    G_OBJECT_CLASS(gst_tfrecordsink_parent_class)->finalize(o);
}
static void gst_tfrecordsink_init(GstTfRecordSink *s) { g_mutex_init(&s->lock); gst_base_sink_set_sync(GST_BASE_SINK(s), FALSE); }
static void gst_tfrecordsink_class_init(GstTfRecordSinkClass *k) {
    GObjectClass *g = G_OBJECT_CLASS(k);
    GstElementClass *e = GST_ELEMENT_CLASS(k);
    GstBaseSinkClass *b = GST_BASE_SINK_CLASS(k);
    g->set_property = tf_set_property;
    g->get_property = tf_get_property;
    g->finalize = tf_finalize;
    g_object_class_install_property(g, TF_PROP_LOCATION,
        g_param_spec_string("location", "File Location", "Location of the file to write", NULL, G_PARAM_READWRITE));
    g_object_class_install_property(g, TF_PROP_GT,
        g_param_spec_string("gt", "Ground truth location", "Location of the ground truth file to read", NULL, G_PARAM_READWRITE));
    g_object_class_install_property(g, TF_PROP_GOP,
        g_param_spec_uint("gop", "GoP", "Number of frames to stack per Example (0: one Example per frame)", 0, G_MAXUINT, 0,
                          G_PARAM_READWRITE | GST_PARAM_MUTABLE_PLAYING));
    gst_element_class_set_static_metadata(e, "TFRecord Sink", "Sink/Video", "Pack metadatas extracted from avdec into TFRecord (covahip)", "covahip");
    gst_element_class_add_pad_template(e, gst_pad_template_new("sink", GST_PAD_SINK, GST_PAD_ALWAYS,
        gst_caps_from_string("video/x-raw, format=(string)RGBA, width=(int)[0,2147483647], height=(int)[0,2147483647]")));
    b->set_caps = tf_set_caps;
    b->start = tf_start;
    b->stop = tf_stop;
    b->render = tf_render;
}

/* ===================================================================== plugin */
static gboolean plugin_init(GstPlugin *plugin) {
    GST_DEBUG_CATEGORY_INIT(cova_debug, "cova", 0, "CoVA compressed-domain elements on covahip");
    /* the reference registers every element with Rank::None (cova-rs/gst-plugins/src/<element>/mod.rs) */
    return gst_element_register(plugin, "metapreprocess", GST_RANK_NONE, gst_metapreprocess_get_type()) &&
           gst_element_register(plugin, "blobnetinfer", GST_RANK_NONE, gst_blobnetinfer_get_type()) &&
           gst_element_register(plugin, "blobnetfilter", GST_RANK_NONE, gst_blobnetfilter_get_type_public()) &&
           gst_element_register(plugin, "maskcopy", GST_RANK_NONE, gst_maskcopy_get_type_public()) &&
           gst_element_register(plugin, "bboxcc", GST_RANK_NONE, gst_bboxcc_get_type()) &&
           gst_element_register(plugin, "sorttracker", GST_RANK_NONE, gst_sorttracker_get_type()) &&
           gst_element_register(plugin, "cova", GST_RANK_NONE, gst_cova_get_type()) &&
           gst_element_register(plugin, "bboxsink", GST_RANK_NONE, gst_bboxsink_get_type()) &&
           gst_element_register(plugin, "tfrecordsink", GST_RANK_NONE, gst_tfrecordsink_get_type()) &&
           gst_element_register(plugin, "h264entropydec", GST_RANK_NONE, gst_entropydec_get_type_public());
}
#define PACKAGE "covahip"
GST_PLUGIN_DEFINE(GST_VERSION_MAJOR, GST_VERSION_MINOR, cova, "CoVA compressed-domain filter elements (MI355X / covahip)",
                  plugin_init, "0.1.0", "LGPL", "covahip", "https://github.com/casys-kaist/CoVA")
