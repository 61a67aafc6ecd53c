// bboxcc on gfx950: 8-connected component labelling + per-component statistics + area
// filter + OpenCV-order compaction, one workgroup per mask frame.
//
// Replaces regionprops() of cova-rs/gst-plugins/src/bboxcc/process.rs:5-49 (OpenCV
// connectedComponentsWithStats, connectivity 8, CC_STAT_* columns, AREA >= threshold).
//
// Algorithm (block-based union-find, all state in LDS):
//   * The frame is cut into 2x2-pixel blocks.  All foreground pixels of one block are
//     mutually 8-adjacent, so a block carries ONE label; block id = by*BW + bx is the
//     raster index of the block.
//   * Rows are bit-packed (one byte = 8 pixels) while they stream in from HBM with
//     8-byte-per-lane coalesced loads; each work item then derives, for the 4 blocks
//     under one packed byte, the foreground nibble and the four "prior neighbour"
//     connections (left, up-left, up, up-right) with shifts/ands on 3x3 bytes.
//   * Horizontal runs need no union at all: the "connected to my left neighbour" bits of a block
//     row form a bit mask, and the first block of the run a block sits in is the highest zero of
//     that mask at or below it (one clz).  Every block starts with the id of its run's first block
//     as label, so only the vertical connections (up-left, up, up-right) go through the union-find.
//   * Blocks are merged with a lock-free min-root union-find (atomicMin on LDS), then
//     flattened.  The root of a component is therefore its SMALLEST block id, i.e. the
//     first block of the component in block-raster order -- exactly the block at which
//     OpenCV's block-based scan (Grana BBDT / Spaghetti) creates the component's first
//     provisional label; flattenL renumbers roots in that order, so ascending root id
//     == OpenCV label order.
//   * Area / min-max extents are accumulated per root with LDS atomics, and surviving
//     roots are compacted in ascending id order with a workgroup prefix sum.
#include <algorithm>
#include <cstdint>
#include <mutex>
#include <vector>

#include "bboxcc_body.h"
#include "internal.h"

namespace {

using namespace ccbody;

__global__ __launch_bounds__(CC_THREADS) void bboxcc_kernel(const uint8_t *__restrict__ masks, CcGeom g,
                                                             int area_thresh, covahip_box *__restrict__ boxes,
                                                             int32_t *__restrict__ counts, int max_boxes) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int frame = blockIdx.x;
    bboxcc_frame(masks + (size_t)frame * g.H * g.W, smem, g, area_thresh, boxes + (size_t)frame * max_boxes,
                 counts + frame, max_boxes, threadIdx.x);
}

}  // namespace

int covahip_bboxcc_launch(covahip_ctx *ctx, const uint8_t *d_mask, int batch, int h, int w, int area_thresh,
                          covahip_box *d_boxes, int32_t *d_counts, int max_boxes) {
    if (batch == 0) return COVAHIP_OK;
    CcGeom g;
    const size_t lds = cc_plan(h, w, g);   // shapes the kernel assumes, checked on the host before any launch
    if (!lds) return COVAHIP_ERR_UNSUPPORTED;
    if (lds > 64 * 1024) {   // the attribute is per device and sticky: set it once per device
        static std::mutex mu;
        static std::vector<int> opened;
        std::lock_guard<std::mutex> lock(mu);
        if (std::find(opened.begin(), opened.end(), ctx->device) == opened.end()) {
            COVAHIP_CHECK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(bboxcc_kernel),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
            opened.push_back(ctx->device);
        }
    }
    ProfScope ps(ctx, "bboxcc_kernel");
    hipLaunchKernelGGL(bboxcc_kernel, dim3(batch), dim3(CC_THREADS), lds, ctx->stream, d_mask, g, area_thresh,
                       d_boxes, d_counts, max_boxes);
    COVAHIP_CHECK_HIP(ctx, hipGetLastError());
    return COVAHIP_OK;
}

extern "C" int covahip_bboxcc(covahip_ctx *ctx, const uint8_t *mask, int batch, int h, int w, int area_thresh,
                              covahip_box *boxes, int32_t *counts, int max_boxes, int mem_kind) {
    if (!ctx || batch < 0 || h <= 0 || w <= 0 || max_boxes < 0) return COVAHIP_ERR_INVALID_ARG;
    if (batch == 0) return COVAHIP_OK;
    if (!mask || !counts || (!boxes && max_boxes > 0)) return COVAHIP_ERR_INVALID_ARG;
    COVAHIP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    if (mem_kind == COVAHIP_MEM_DEVICE)
        return covahip_bboxcc_launch(ctx, mask, batch, h, w, area_thresh, boxes, counts, max_boxes);
    if (mem_kind != COVAHIP_MEM_HOST) return COVAHIP_ERR_INVALID_ARG;

    const size_t mask_bytes = (size_t)batch * h * w;
    const size_t box_bytes = (size_t)batch * max_boxes * sizeof(covahip_box);
    const size_t cnt_bytes = (size_t)batch * sizeof(int32_t);
    int rc = covahip_ensure_buffer(ctx, &ctx->stage_in, &ctx->stage_in_bytes, mask_bytes);
    if (rc) return rc;
    rc = covahip_ensure_buffer(ctx, &ctx->stage_out, &ctx->stage_out_bytes, box_bytes + cnt_bytes + 16);
    if (rc) return rc;
    uint8_t *d_mask = (uint8_t *)ctx->stage_in;
    covahip_box *d_boxes = (covahip_box *)ctx->stage_out;
    int32_t *d_counts = (int32_t *)((uint8_t *)ctx->stage_out + ((box_bytes + 15) & ~(size_t)15));
    COVAHIP_CHECK_HIP(ctx, hipMemcpyAsync(d_mask, mask, mask_bytes, hipMemcpyHostToDevice, ctx->stream));
    rc = covahip_bboxcc_launch(ctx, d_mask, batch, h, w, area_thresh, d_boxes, d_counts, max_boxes);
    if (rc) return rc;
    if (box_bytes)
        COVAHIP_CHECK_HIP(ctx, hipMemcpyAsync(boxes, d_boxes, box_bytes, hipMemcpyDeviceToHost, ctx->stream));
    COVAHIP_CHECK_HIP(ctx, hipMemcpyAsync(counts, d_counts, cnt_bytes, hipMemcpyDeviceToHost, ctx->stream));
    COVAHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return COVAHIP_OK;
}
