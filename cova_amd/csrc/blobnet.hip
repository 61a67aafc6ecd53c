// BlobNet model state: weight blob validation, geometry (pad / crop rules), HBM
// workspace, and the C-ABI entry points covahip_blobnet_* / covahip_filter_forward.
//
// Geometry follows utils/model/encoder.py:58-80 (level i -> ceil(H/2) x ceil(W/2), zero
// row on top / zero column on the left when the pre-pool size is odd) and
// utils/model/decoder.py:43-59 (convT output 2*in+2, crop ceil(p/2) top/left).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "blobnet.h"
#include "covahip_dev.h"

namespace {

constexpr uint32_t W_MAGIC = 0x57485643;  // "CVHW"
constexpr size_t N_PARAMS = 320305;

}  // namespace

static void free_model(covahip_ctx *ctx, covahip_blobnet *m) {
    if (!m) return;
    blobnet_release_mfma(ctx, m);
    for (BnWorkspace &ws : m->ws) {
        for (int i = 0; i <= BN_LEVELS; i++)
            if (ws.act[i]) hipFree(ws.act[i]);
        for (int j = 0; j < BN_LEVELS; j++)
            if (ws.dact[j]) hipFree(ws.dact[j]);
        if (ws.pbuf) hipFree(ws.pbuf);
        if (ws.part) hipFree(ws.part);
        if (ws.d_index) hipFree(ws.d_index);
        if (ws.h_index) hipHostFree(ws.h_index);
        if (ws.ev_index) hipEventDestroy(ws.ev_index);
    }
    delete m;
}

void covahip_blobnet_destroy(covahip_ctx *ctx) {
    if (!ctx->blobnet) return;
    hipSetDevice(ctx->device);
    covahip_sync_all(ctx);
    free_model(ctx, ctx->blobnet);
    ctx->blobnet = nullptr;
}

int covahip_blobnet_geometry(covahip_ctx *ctx, int *h, int *w) {
    if (!ctx->blobnet) return COVAHIP_ERR_NOT_LOADED;
    *h = ctx->blobnet->H;
    *w = ctx->blobnet->W;
    return COVAHIP_OK;
}

// BlobNet forward (+ optionally bboxcc) on device pointers, asynchronous on the ctx stream.
static int ensure_pbuf(covahip_ctx *ctx, covahip_blobnet *m, BnWorkspace &ws, int n_frames);
static int filter_dev(covahip_ctx *ctx, const BnInput &in, int batch, float *d_logits, uint8_t *d_mask,
                      bool with_cc, int area_thresh, covahip_box *d_boxes, int32_t *d_counts, int max_boxes) {
    covahip_blobnet *m = ctx->blobnet;
    if (!m) return COVAHIP_ERR_NOT_LOADED;
    if (batch > m->max_batch) return COVAHIP_ERR_INVALID_ARG;
    if (batch == 0) return COVAHIP_OK;
    if (in.stack) {   // the stacked tensor is batch * T carrier frames (blobnet_mfma.hip)
        const int rc = ensure_pbuf(ctx, m, m->ws[ctx->cur_lane], batch * BN_T);
        if (rc) return rc;
    }
    BnCcTail tail{area_thresh, max_boxes, d_boxes, d_counts};
    bool cc_done = false;
    int rc = blobnet_forward_mfma(ctx, m, m->ws[ctx->cur_lane], in, batch, d_logits, d_mask, with_cc ? &tail : nullptr, &cc_done);
    if (rc) return rc;
    // the fused decoder tail normally runs bboxcc itself; the separate kernel is the fallback for
    // geometries whose frame does not fit its LDS plan
    if (with_cc && !cc_done)
        rc = covahip_bboxcc_launch(ctx, d_mask, batch, m->H, m->W, area_thresh, d_boxes, d_counts, max_boxes);
    return rc;
}

// Carrier-frame input: validates the stack -> frame table on the host (an index outside the frame array would be an
// out-of-bounds read on the GPU) and uploads it.  stack_index == nullptr: one stream in order.  A table equal to the
// one resident in this lane's workspace is not uploaded again.
static int prepare_frames(covahip_ctx *ctx, covahip_blobnet *m, BnWorkspace &ws, const uint8_t *d_frames, int n_frames,
                          const int32_t *stack_index, int batch, BnInput &in) {
    if (n_frames < BN_T || n_frames > BN_T * m->max_batch) return COVAHIP_ERR_INVALID_ARG;
    if (!stack_index && batch != n_frames - (BN_T - 1)) return COVAHIP_ERR_INVALID_ARG;
    std::vector<int32_t> table((size_t)batch * BN_T);
    for (int b = 0; b < batch; b++)
        for (int t = 0; t < BN_T; t++) {
            const int32_t f = stack_index ? stack_index[b * BN_T + t] : b + (BN_T - 1) - t;
            if (f < 0 || f >= n_frames) return COVAHIP_ERR_INVALID_ARG;
            table[(size_t)b * BN_T + t] = f;
        }
    // small batches: the table travels in the level-1 kernel's arguments (blobnet_mfma.hip); the device copy below is for
    // larger ones and for the round-1..3 level-1 kernel
    const bool by_value = batch <= BN_KTAB_STACKS && n_frames <= 65535 && bn_level1_on_enc1(ctx, m);
    if (by_value) {
        ws.last_table = std::move(table);      // (kept alive until the launch has copied it)
        ws.last_n_frames = -1;                 // the device table, if any, is stale
        in.h_index = ws.last_table.data();
        in.frames = d_frames;
        in.n_frames = n_frames;
        return ensure_pbuf(ctx, m, ws, n_frames);
    }
    const bool same = ws.d_index && table == ws.last_table && n_frames == ws.last_n_frames;
    if (!same) {
        const size_t need = table.size();
        if (ws.d_index) COVAHIP_CHECK_HIP(ctx, hipEventSynchronize(ws.ev_index));   // the pinned copy is free again
        if (need > ws.index_ints) {
            if (ws.d_index) {
                COVAHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
                hipFree(ws.d_index);
                hipHostFree(ws.h_index);
                ws.d_index = nullptr; ws.h_index = nullptr; ws.index_ints = 0;
                ws.last_table.clear();
            }
            const size_t cap = std::max(need, (size_t)m->max_batch * BN_T);
            COVAHIP_CHECK_HIP(ctx, hipMalloc((void **)&ws.d_index, cap * sizeof(int32_t)));
            COVAHIP_CHECK_HIP(ctx, hipHostMalloc((void **)&ws.h_index, cap * sizeof(int32_t), hipHostMallocDefault));
            ws.index_ints = cap;
            if (!ws.ev_index) COVAHIP_CHECK_HIP(ctx, hipEventCreateWithFlags(&ws.ev_index, hipEventDisableTiming));
        }
        std::copy(table.begin(), table.end(), ws.h_index);
        COVAHIP_CHECK_HIP(ctx, hipMemcpyAsync(ws.d_index, ws.h_index, need * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
        COVAHIP_CHECK_HIP(ctx, hipEventRecord(ws.ev_index, ctx->stream));
        ws.last_table = std::move(table);
        ws.last_n_frames = n_frames;
    }
    in.frames = d_frames;
    in.n_frames = n_frames;
    in.index = ws.d_index;
    return ensure_pbuf(ctx, m, ws, n_frames);
}

// The tensor P of pooled level-0 values, one slice per carrier frame (the stacked entry has batch * T of them).
static int ensure_pbuf(covahip_ctx *ctx, covahip_blobnet *m, BnWorkspace &ws, int n_frames) {
    if (ws.pbuf_frames < (size_t)n_frames) {
        // grown in whole steps; the pad row / column of P (odd grids) is zeroed here and never written
        const size_t want = std::min((size_t)BN_T * m->max_batch, std::max((size_t)n_frames, 2 * ws.pbuf_frames));
        const size_t bytes = want * m->lv[1].H * m->lv[1].W * m->enc_c[1] * sizeof(__half);
        if (ws.pbuf) {
            COVAHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
            COVAHIP_CHECK_HIP(ctx, hipFree(ws.pbuf));
            ws.pbuf = nullptr;
            ws.pbuf_frames = 0;
        }
        COVAHIP_CHECK_HIP(ctx, hipMalloc((void **)&ws.pbuf, bytes));
        COVAHIP_CHECK_HIP(ctx, hipMemsetAsync(ws.pbuf, 0, bytes, ctx->stream));
        ws.pbuf_frames = want;
    }
    return COVAHIP_OK;
}

int covahip_blobnet_forward_dev(covahip_ctx *ctx, const uint8_t *d_stack, int batch, float *d_logits,
                                uint8_t *d_mask) {
    BnInput in;
    in.stack = d_stack;
    return filter_dev(ctx, in, batch, d_logits, d_mask, false, 0, nullptr, nullptr, 0);
}

// HBM workspace of one lane: activations stay resident; pad rows/columns are zeroed once here and never written
// afterwards.  (The memsets go to the stream of the caller's choice: ctx->stream.)
static int alloc_workspace(covahip_ctx *ctx, covahip_blobnet *m, BnWorkspace &ws) {
    if (ws.ready) return COVAHIP_OK;
    for (int i = 1; i <= BN_LEVELS; i++) {
        const size_t tt = (i == BN_LEVELS) ? 1 : BN_T;
        const size_t n = (size_t)m->max_batch * tt * m->lv[i].H * m->lv[i].W * m->enc_c[i];
        COVAHIP_CHECK_HIP(ctx, hipMalloc((void **)&ws.act[i], n * sizeof(__half)));
        COVAHIP_CHECK_HIP(ctx, hipMemsetAsync(ws.act[i], 0, n * sizeof(__half), ctx->stream));
    }
    for (int j = 0; j < BN_LEVELS - 1; j++) {
        const BnLevelGeom out = m->lv[BN_LEVELS - 1 - j];
        const size_t n = (size_t)m->max_batch * out.H * out.W * m->dec_co[j];
        COVAHIP_CHECK_HIP(ctx, hipMalloc((void **)&ws.dact[j], n * sizeof(__half)));
    }
    COVAHIP_CHECK_HIP(ctx, hipMalloc((void **)&ws.part, (size_t)m->max_batch * (m->lv[1].H + 1) * (m->lv[1].W + 1) * 4 * sizeof(float)));
    ws.ready = true;
    return COVAHIP_OK;
}

int covahip_blobnet_grow_lanes(covahip_ctx *ctx, int n_lanes) {
    covahip_blobnet *m = ctx->blobnet;
    if (!m) return COVAHIP_OK;
    for (int k = 0; k < n_lanes && k < COVAHIP_MAX_LANES; k++) {
        const int rc = alloc_workspace(ctx, m, m->ws[k]);
        if (rc) return rc;
    }
    COVAHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return COVAHIP_OK;
}

// Geometry, HBM workspace and prepared weights of a model.  Every limit of the kernels is checked here (a
// planning-only pass of the forward for batch 1 and max_batch), so an unsupported grid fails at load time.
static int build_model(covahip_ctx *ctx, covahip_blobnet *m, const float *h_w, int h_mb, int w_mb, int max_batch) {
    m->H = h_mb;
    m->W = w_mb;
    m->max_batch = max_batch;
    m->lv[0] = {h_mb, w_mb};
    for (int i = 0; i < BN_LEVELS; i++) m->lv[i + 1] = {(m->lv[i].H + 1) / 2, (m->lv[i].W + 1) / 2};
    for (int j = 0; j < BN_LEVELS; j++) {
        const BnLevelGeom in = m->lv[BN_LEVELS - j], out = m->lv[BN_LEVELS - 1 - j];
        const int ph = 2 * in.H + 2 - out.H, pw = 2 * in.W + 2 - out.W;
        if (ph < 0 || pw < 0) return COVAHIP_ERR_UNSUPPORTED;  // decoder.py zero-pad branch: never at these sizes
        m->dec_cy[j] = ph / 2 + ph % 2;
        m->dec_cx[j] = pw / 2 + pw % 2;
    }
    // algorithmic MACs per frame (conv + convT(full, uncropped as Keras computes it) + tmix + final)
    int64_t macs = 0;
    for (int i = 0; i < BN_LEVELS; i++) {
        macs += (int64_t)BN_T * m->lv[i].H * m->lv[i].W * 9 * m->enc_c[i] * m->enc_c[i + 1];
        macs += (int64_t)m->lv[i + 1].H * m->lv[i + 1].W * m->enc_c[i + 1] * 32;
    }
    for (int j = 0; j < BN_LEVELS; j++) {
        const BnLevelGeom in = m->lv[BN_LEVELS - j];
        macs += (int64_t)in.H * in.W * 16 * m->dec_ci[j] * m->dec_co[j];
    }
    macs += (int64_t)h_mb * w_mb * 16;
    m->macs_per_frame = macs;

    for (int k = 0; k < ctx->n_lanes; k++) {
        const int rc = alloc_workspace(ctx, m, m->ws[k]);
        if (rc) return rc;
    }
    int rc = blobnet_prepare_mfma(ctx, m, h_w);
    if (rc) return rc;
    BnInput plan;   // planning only, both input forms, smallest and largest batch
    plan.dry = true;
    for (int pass = 0; pass < 4 && !rc; pass++) {
        const int b = (pass & 1) ? max_batch : 1;
        plan.n_frames = (pass & 2) ? b + BN_T - 1 : 0;   // both entry points
        rc = blobnet_forward_mfma(ctx, m, m->ws[0], plan, b, nullptr, nullptr, nullptr, nullptr);
    }
    return rc;
}

extern "C" {

int covahip_blobnet_load(covahip_ctx *ctx, const void *weights, size_t weights_bytes, int h_mb, int w_mb, int t,
                         int max_batch) {
    if (!ctx || !weights || h_mb <= 0 || w_mb <= 0 || max_batch <= 0) return COVAHIP_ERR_INVALID_ARG;
    if (t != BN_T) return COVAHIP_ERR_UNSUPPORTED;
    if (h_mb < 16 || w_mb < 16 || h_mb > 1024 || w_mb > 1024) return COVAHIP_ERR_UNSUPPORTED;
    if (weights_bytes < 64) return COVAHIP_ERR_BAD_WEIGHTS;
    uint32_t hdr[16];
    std::memcpy(hdr, weights, 64);
    static const uint32_t want[] = {W_MAGIC, 1, 4, 3, 16, 32, 64, 128, 64, 32, 16, 16, (uint32_t)N_PARAMS};
    for (int i = 0; i < 13; i++)
        if (hdr[i] != want[i]) return COVAHIP_ERR_BAD_WEIGHTS;
    if (weights_bytes != 64 + N_PARAMS * sizeof(float)) return COVAHIP_ERR_BAD_WEIGHTS;
    const float *h_w = reinterpret_cast<const float *>(static_cast<const uint8_t *>(weights) + 64);

    COVAHIP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    covahip_blobnet_destroy(ctx);
    // The model is built on the side and attached to the ctx only when every step succeeded: a failed
    // load leaves the ctx without a model (COVAHIP_ERR_NOT_LOADED afterwards), never with half of one.
    covahip_blobnet *m = new covahip_blobnet();
    int rc = build_model(ctx, m, h_w, h_mb, w_mb, max_batch);
    if (rc == COVAHIP_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = COVAHIP_ERR_HIP;
    if (rc) {
        hipStreamSynchronize(ctx->stream);
        (void)hipGetLastError();
        free_model(ctx, m);
        return rc;
    }
    ctx->blobnet = m;
    return COVAHIP_OK;
}

int covahip_blobnet_macs_per_frame(covahip_ctx *ctx, int64_t *macs) {
    if (!ctx || !macs) return COVAHIP_ERR_INVALID_ARG;
    if (!ctx->blobnet) return COVAHIP_ERR_NOT_LOADED;
    *macs = ctx->blobnet->macs_per_frame;
    return COVAHIP_OK;
}

int covahip_blobnet_set_enc_plan(covahip_ctx *ctx, int level, int nbands, int nbuf) {
    if (!ctx || level < 1 || level > 3 || nbands < 0 || nbands > 64 || (nbands && nbuf != 1 && nbuf != 2)) return COVAHIP_ERR_INVALID_ARG;
    ctx->enc_plan[level].nbands = nbands;
    ctx->enc_plan[level].nbuf = nbands ? nbuf : 0;
    return COVAHIP_OK;
}

int covahip_blobnet_set_impl(covahip_ctx *ctx, int impl) {
    if (!ctx || !ctx->blobnet) return COVAHIP_ERR_NOT_LOADED;
    if (impl != 1 && (impl < 4 || impl > 10)) return COVAHIP_ERR_INVALID_ARG;
    ctx->blobnet->fuse_dec = impl != 4;
    ctx->blobnet->enc1_tile16 = impl != 5;
    ctx->blobnet->enc_rowtiles = impl != 6;
    ctx->blobnet->fuse_enc23 = impl == 7 ? 0 : impl == 8 ? 2 : 1;
    ctx->blobnet->tail_part = impl != 9;
    ctx->blobnet->tail_rows = impl != 10;
    return COVAHIP_OK;
}

// The hot path on device pointers, on ctx->stream with the current lane's workspace (the caller has placed the call).
static int filter_placed(covahip_ctx *ctx, covahip_blobnet *m, const uint8_t *d_src, int n_frames, const int32_t *stack_index,
                         int batch, int area_thresh, covahip_box *d_boxes, int32_t *d_counts, int max_boxes, float *d_logits,
                         uint8_t *d_mask, bool packed = false) {
    if (!d_mask) {
        CtxLane &l = ctx->lane();
        int rc = covahip_ensure_buffer(ctx, &l.cc_scratch, &l.cc_scratch_bytes, (size_t)batch * m->H * m->W);
        if (rc) return rc;
        d_mask = (uint8_t *)l.cc_scratch;
    }
    BnInput in;
    if (n_frames > 0) {
        int rc = prepare_frames(ctx, m, m->ws[ctx->cur_lane], d_src, n_frames, stack_index, batch, in);
        if (rc) return rc;
        in.packed = packed;
    } else {
        in.stack = d_src;
    }
    return filter_dev(ctx, in, batch, d_logits, d_mask, true, area_thresh, d_boxes, d_counts, max_boxes);
}

// Shared body of covahip_filter_forward / covahip_filter_forward_frames: `src` is the stacked tensor (n_frames == 0)
// or the carrier frames.
static int filter_any(covahip_ctx *ctx, const uint8_t *src, int n_frames, const int32_t *stack_index, int batch,
                      int area_thresh, covahip_box *boxes, int32_t *counts, int max_boxes, float *logits, uint8_t *mask,
                      int mem_kind) {
    if (!ctx || batch < 0 || max_boxes < 0 || n_frames < 0) return COVAHIP_ERR_INVALID_ARG;
    covahip_blobnet *m = ctx->blobnet;
    if (!m) return COVAHIP_ERR_NOT_LOADED;
    if (batch == 0) return COVAHIP_OK;
    if (!src || !counts || (!boxes && max_boxes > 0) || batch > m->max_batch) return COVAHIP_ERR_INVALID_ARG;
    COVAHIP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    const bool by_frames = n_frames > 0;
    const size_t hw = (size_t)m->H * m->W;
    const size_t in_bytes = by_frames ? (size_t)n_frames * hw * 4 : (size_t)batch * BN_T * hw * 4;
    const size_t mask_bytes = (size_t)batch * hw;
    const size_t logit_bytes = mask_bytes * sizeof(float);
    const size_t box_bytes = (size_t)batch * max_boxes * sizeof(covahip_box);
    const size_t cnt_bytes = (size_t)batch * sizeof(int32_t);
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    if (mem_kind != COVAHIP_MEM_DEVICE && mem_kind != COVAHIP_MEM_HOST) return COVAHIP_ERR_INVALID_ARG;

    // device pointers: the call goes to the next lane (internal.h, CtxLane) and overlaps the calls before it; host pointers:
    // synchronous, on the primary stream, behind everything in flight
    if (mem_kind == COVAHIP_MEM_DEVICE) {
        LaneScope lane(ctx);
        if (!lane.ok()) return COVAHIP_ERR_HIP;
        return filter_placed(ctx, m, src, n_frames, stack_index, batch, area_thresh, boxes, counts, max_boxes, logits, mask);
    }
    if (int rc = covahip_primary_op(ctx)) return rc;
    const uint8_t *d_src = src;
    int rc = covahip_ensure_buffer(ctx, &ctx->stage_in, &ctx->stage_in_bytes, in_bytes);
    if (rc) return rc;
    rc = covahip_ensure_buffer(ctx, &ctx->stage_out, &ctx->stage_out_bytes,
                               al(mask_bytes) + al(logit_bytes) + al(box_bytes) + al(cnt_bytes));
    if (rc) return rc;
    uint8_t *base = (uint8_t *)ctx->stage_out;
    uint8_t *d_mask = base;
    float *d_logits = logits ? (float *)(base + al(mask_bytes)) : nullptr;
    covahip_box *d_boxes = (covahip_box *)(base + al(mask_bytes) + al(logit_bytes));
    int32_t *d_counts = (int32_t *)(base + al(mask_bytes) + al(logit_bytes) + al(box_bytes));
    COVAHIP_CHECK_HIP(ctx, hipMemcpyAsync(ctx->stage_in, src, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    d_src = (const uint8_t *)ctx->stage_in;
    rc = filter_placed(ctx, m, d_src, n_frames, stack_index, batch, area_thresh, d_boxes, d_counts, max_boxes, d_logits, d_mask);
    if (rc) return rc;
    if (logits) COVAHIP_CHECK_HIP(ctx, hipMemcpyAsync(logits, d_logits, logit_bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (mask) COVAHIP_CHECK_HIP(ctx, hipMemcpyAsync(mask, d_mask, mask_bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (box_bytes) COVAHIP_CHECK_HIP(ctx, hipMemcpyAsync(boxes, d_boxes, box_bytes, hipMemcpyDeviceToHost, ctx->stream));
    COVAHIP_CHECK_HIP(ctx, hipMemcpyAsync(counts, d_counts, cnt_bytes, hipMemcpyDeviceToHost, ctx->stream));
    COVAHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return COVAHIP_OK;
}

int covahip_filter_forward(covahip_ctx *ctx, const uint8_t *rgba_stack, int batch, int area_thresh,
                           covahip_box *boxes, int32_t *counts, int max_boxes, float *logits, uint8_t *mask,
                           int mem_kind) {
    return filter_any(ctx, rgba_stack, 0, nullptr, batch, area_thresh, boxes, counts, max_boxes, logits, mask, mem_kind);
}

int covahip_filter_forward_frames(covahip_ctx *ctx, const uint8_t *frames, int n_frames, const int32_t *stack_index,
                                  int batch, int area_thresh, covahip_box *boxes, int32_t *counts, int max_boxes,
                                  float *logits, uint8_t *mask, int mem_kind) {
    if (n_frames <= 0) return COVAHIP_ERR_INVALID_ARG;
    return filter_any(ctx, frames, n_frames, stack_index, batch, area_thresh, boxes, counts, max_boxes, logits, mask, mem_kind);
}

int covahip_filter_forward_frames_packed(covahip_ctx *ctx, const uint16_t *d_records, int n_frames, const int32_t *stack_index,
                                         int batch, int area_thresh, covahip_box *d_boxes, int32_t *d_counts, int max_boxes,
                                         float *d_logits, uint8_t *d_mask) {
    if (!ctx || batch < 0 || max_boxes < 0 || n_frames <= 0) return COVAHIP_ERR_INVALID_ARG;
    covahip_blobnet *m = ctx->blobnet;
    if (!m) return COVAHIP_ERR_NOT_LOADED;
    if (batch == 0) return COVAHIP_OK;
    if (!d_records || !d_counts || (!d_boxes && max_boxes > 0) || batch > m->max_batch) return COVAHIP_ERR_INVALID_ARG;
    COVAHIP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    LaneScope lane(ctx);
    if (!lane.ok()) return COVAHIP_ERR_HIP;
    return filter_placed(ctx, m, reinterpret_cast<const uint8_t *>(d_records), n_frames, stack_index, batch, area_thresh, d_boxes, d_counts,
                         max_boxes, d_logits, d_mask, true);
}

// Developer probe (include/covahip_dev.h): the same carrier-frame step `iters` times as stream launches and as launches of ONE
// captured HIP graph of it -- what the gaps between the six launches of a step cost on their own.
int covahip_dev_graph_probe(covahip_ctx *ctx, const uint8_t *d_frames, int n_frames, const int32_t *stack_index, int batch,
                            int area_thresh, covahip_box *d_boxes, int32_t *d_counts, int max_boxes, uint8_t *d_mask, int iters,
                            float *ms_direct, float *ms_graph) {
    if (!ctx || !ctx->blobnet || !ms_direct || !ms_graph || iters < 1) return COVAHIP_ERR_INVALID_ARG;
    covahip_blobnet *m = ctx->blobnet;
    COVAHIP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    if (int rc = covahip_primary_op(ctx)) return rc;
    // a captured step must not contain the standalone bboxcc launches: their alternating counter sets are chosen at capture time,
    // a replay would keep adding to ONE set -- the probe is for the fused tail only
    if (!m->fuse_tail) return COVAHIP_ERR_UNSUPPORTED;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    int rc = COVAHIP_OK;
    // every exit goes through `done`: events, graph and exec are destroyed, and a capture that was begun is always ended
    auto hip_ok = [&](hipError_t e, const char *what) {
        if (e == hipSuccess || rc) return e == hipSuccess;
        ctx->last_hip_error = std::string(what) + ": " + hipGetErrorString(e);
        rc = COVAHIP_ERR_HIP;
        return false;
    };
    auto run = [&](int n) {
        for (int i = 0; i < n && !rc; i++)
            rc = filter_placed(ctx, m, d_frames, n_frames, stack_index, batch, area_thresh, d_boxes, d_counts, max_boxes, nullptr, d_mask);
    };
    bool capturing = false;
    do {
        if (!hip_ok(hipEventCreate(&e0), "hipEventCreate") || !hip_ok(hipEventCreate(&e1), "hipEventCreate")) break;
        run(3);
        if (rc || !hip_ok(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize")) break;
        if (!hip_ok(hipEventRecord(e0, ctx->stream), "hipEventRecord")) break;
        run(iters);
        if (rc || !hip_ok(hipEventRecord(e1, ctx->stream), "hipEventRecord") || !hip_ok(hipEventSynchronize(e1), "hipEventSynchronize") ||
            !hip_ok(hipEventElapsedTime(ms_direct, e0, e1), "hipEventElapsedTime"))
            break;
        if (!hip_ok(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeRelaxed), "hipStreamBeginCapture")) break;
        capturing = true;
        run(1);
        capturing = false;
        const hipError_t ee = hipStreamEndCapture(ctx->stream, &graph);   // always: the stream must leave capture mode
        if (rc || !hip_ok(ee, "hipStreamEndCapture")) break;
        if (!hip_ok(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0), "hipGraphInstantiate")) break;
        for (int i = 0; i < 3 && !rc; i++) hip_ok(hipGraphLaunch(exec, ctx->stream), "hipGraphLaunch");
        if (rc || !hip_ok(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize") || !hip_ok(hipEventRecord(e0, ctx->stream), "hipEventRecord")) break;
        for (int i = 0; i < iters && !rc; i++) hip_ok(hipGraphLaunch(exec, ctx->stream), "hipGraphLaunch");
        if (rc || !hip_ok(hipEventRecord(e1, ctx->stream), "hipEventRecord") || !hip_ok(hipEventSynchronize(e1), "hipEventSynchronize")) break;
        hip_ok(hipEventElapsedTime(ms_graph, e0, e1), "hipEventElapsedTime");
    } while (false);
    if (capturing) (void)hipStreamEndCapture(ctx->stream, &graph);
    if (exec) (void)hipGraphExecDestroy(exec);
    if (graph) (void)hipGraphDestroy(graph);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    return rc;
}

int covahip_blobnet_forward(covahip_ctx *ctx, const uint8_t *rgba_stack, int batch, float *logits, uint8_t *mask,
                            int mem_kind) {
    if (!ctx || batch < 0) return COVAHIP_ERR_INVALID_ARG;
    covahip_blobnet *m = ctx->blobnet;
    if (!m) return COVAHIP_ERR_NOT_LOADED;
    if (batch == 0) return COVAHIP_OK;
    if (!rgba_stack || batch > m->max_batch) return COVAHIP_ERR_INVALID_ARG;
    COVAHIP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    if (int prc = covahip_primary_op(ctx)) return prc;   // BlobNet alone runs on the primary stream (lane 0's workspace)
    if (mem_kind == COVAHIP_MEM_DEVICE) return covahip_blobnet_forward_dev(ctx, rgba_stack, batch, logits, mask);
    if (mem_kind != COVAHIP_MEM_HOST) return COVAHIP_ERR_INVALID_ARG;
    const size_t hw = (size_t)m->H * m->W;
    const size_t in_bytes = (size_t)batch * BN_T * hw * 4;
    const size_t mask_bytes = (size_t)batch * hw;
    const size_t logit_bytes = mask_bytes * sizeof(float);
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    int rc = covahip_ensure_buffer(ctx, &ctx->stage_in, &ctx->stage_in_bytes, in_bytes);
    if (rc) return rc;
    rc = covahip_ensure_buffer(ctx, &ctx->stage_out, &ctx->stage_out_bytes, al(mask_bytes) + al(logit_bytes));
    if (rc) return rc;
    uint8_t *d_mask = (uint8_t *)ctx->stage_out;
    float *d_logits = (float *)((uint8_t *)ctx->stage_out + al(mask_bytes));
    COVAHIP_CHECK_HIP(ctx, hipMemcpyAsync(ctx->stage_in, rgba_stack, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    rc = covahip_blobnet_forward_dev(ctx, (const uint8_t *)ctx->stage_in, batch, logits ? d_logits : nullptr,
                                     mask ? d_mask : nullptr);
    if (rc) return rc;
    if (logits) COVAHIP_CHECK_HIP(ctx, hipMemcpyAsync(logits, d_logits, logit_bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (mask) COVAHIP_CHECK_HIP(ctx, hipMemcpyAsync(mask, d_mask, mask_bytes, hipMemcpyDeviceToHost, ctx->stream));
    COVAHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return COVAHIP_OK;
}

}  // extern "C"
