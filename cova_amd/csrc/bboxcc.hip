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
#include "bboxcc_wave.h"
#include "covahip_dev.h"
#include "internal.h"

namespace {

using namespace ccbody;

// ---- one 1,024-thread workgroup per frame (any shape that fits LDS; also the overflow pass of the wave kernel)
// list == nullptr: frame = blockIdx.x.  Otherwise the launch is persistent over the *n_list frame indices in list.
// wg.cap > 0: the run-based body (bboxcc_wave.h, frame_wg) with worst-case capacity; else the block-based body.
__global__ __launch_bounds__(CC_THREADS) void bboxcc_kernel(const uint8_t *__restrict__ masks, CcGeom g, ccwave::WvGeom wg,
                                                             int area_thresh, covahip_box *__restrict__ boxes,
                                                             int32_t *__restrict__ counts, int max_boxes,
                                                             const int32_t *__restrict__ list,
                                                             const int32_t *__restrict__ n_list) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    auto one = [&](int frame) {
        const uint8_t *m = masks + (size_t)frame * g.H * g.W;
        if (wg.cap > 0)
            ccwave::frame_wg<CC_THREADS>(m, smem, wg, area_thresh, boxes + (size_t)frame * max_boxes, counts + frame, max_boxes, threadIdx.x);
        else
            bboxcc_frame(m, smem, g, area_thresh, boxes + (size_t)frame * max_boxes, counts + frame, max_boxes, threadIdx.x);
    };
    if (!list) {
        one(blockIdx.x);
        return;
    }
    const int n = *n_list;
    for (int k = blockIdx.x; k < n; k += gridDim.x) {
        one(list[k]);
        __syncthreads();   // the next frame reuses the LDS region
    }
}

// ---- frames whose state does not fit in LDS (4K grids): the same block-based body with its arrays in a slab of global
// memory per resident workgroup; persistent over the batch.  A correctness path, not a fast one.
__global__ __launch_bounds__(CC_THREADS) void bboxcc_big_kernel(const uint8_t *__restrict__ masks, CcGeom g, uint8_t *slabs,
                                                                 size_t slab_bytes, int batch, int area_thresh,
                                                                 covahip_box *__restrict__ boxes, int32_t *__restrict__ counts,
                                                                 int max_boxes) {
    uint8_t *const slab = slabs + (size_t)blockIdx.x * slab_bytes;
    for (int frame = blockIdx.x; frame < batch; frame += gridDim.x) {
        bboxcc_frame(masks + (size_t)frame * g.H * g.W, slab, g, area_thresh, boxes + (size_t)frame * max_boxes, counts + frame,
                     max_boxes, threadIdx.x);
        __syncthreads();   // the next frame reuses the slab
    }
}

// ---- one WAVE per frame (bboxcc_wave.h): WV_WAVES frames per workgroup, no workgroup barrier at all
constexpr int WV_WAVES = 4;
__global__ __launch_bounds__(WV_WAVES * 64) void bboxcc_wave_kernel(const uint8_t *__restrict__ masks, ccwave::WvGeom g, int batch,
                                                                    int area_thresh, covahip_box *__restrict__ boxes,
                                                                    int32_t *__restrict__ counts, int max_boxes,
                                                                    int32_t *__restrict__ ovf_list, int32_t *__restrict__ ovf_n) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int frame = blockIdx.x * WV_WAVES + wave;
    if (frame >= batch) return;
    const bool done = ccwave::frame_wave(masks + (size_t)frame * g.H * g.W, smem + (size_t)wave * g.wave_bytes, g, area_thresh,
                                         boxes + (size_t)frame * max_boxes, counts + frame, max_boxes, lane);
    if (!done && lane == 0) ovf_list[atomicAdd(ovf_n, 1)] = frame;   // more runs than the LDS region holds
}

template <typename K>
int open_lds(covahip_ctx *ctx, K kernel, size_t lds) {
    if (lds <= 64 * 1024) return COVAHIP_OK;
    static std::mutex mu;
    static std::vector<std::pair<int, const void *>> opened;
    const void *fn = reinterpret_cast<const void *>(kernel);
    std::lock_guard<std::mutex> lock(mu);
    for (auto &o : opened)
        if (o.first == ctx->device && o.second == fn) return COVAHIP_OK;
    COVAHIP_CHECK_HIP(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
    opened.emplace_back(ctx->device, fn);
    return COVAHIP_OK;
}

}  // namespace

// Launch plan.  Shapes the wave kernel takes (W a multiple of 8, H and W <= 128, 8-byte aligned frames) go to it:
//   * small batches (a few frames per CU): every wave gets the worst-case run capacity (one run per 2x2 block),
//     nothing can overflow, one launch;
//   * large batches: 256 runs per wave (7 KB of LDS -> about twenty frames in flight per CU); frames with more
//     runs -- noise-like masks -- are collected in an overflow list that a second, persistent launch of the
//     workgroup-per-frame kernel drains (it exits at once when the list is empty).
// Everything else runs the workgroup-per-frame kernel.
int covahip_bboxcc_launch(covahip_ctx *ctx, const uint8_t *d_mask, int batch, int h, int w, int area_thresh,
                          covahip_box *d_boxes, int32_t *d_counts, int max_boxes) {
    if (batch == 0) return COVAHIP_OK;
    CtxLane &ln = ctx->lane();             // scratch of the lane this call runs on (lane 0 on the primary stream)
    CcGeom g;
    const size_t lds = cc_plan(h, w, g);   // shapes the kernel assumes, checked on the host before any launch
    const int num_cu = ctx->props.multiProcessorCount;
    ccwave::WvGeom wg;
    const int nb = ((h + 1) / 2) * ((w + 1) / 2);
    int cap = ctx->cc_wave_cap;            // developer override: > 0 capacity, < 0 wave kernel off, 0 automatic
    // a few frames per CU: the workgroup kernel finishes a frame sooner than a single wave does (9.5 vs 13.5 us at
    // b = 256), and there is nothing to overlap it with
    if (cap == 0) cap = batch <= 3 * num_cu ? -1 : 128;
    cap = std::min(cap, nb);
    const bool aligned = (reinterpret_cast<uintptr_t>(d_mask) & 7) == 0 && (((size_t)h * w) & 7) == 0;
    // the workgroup-per-frame kernel runs the run-based body too when the shape allows it (worst-case capacity)
    ccwave::WvGeom wfull{};
    size_t lds_wg = lds;
    if (ctx->cc_wave_cap >= 0 && aligned && ccwave::wv_plan(h, w, nb, wfull) && (size_t)wfull.wave_bytes <= 160 * 1024 - 256)
        lds_wg = (size_t)wfull.wave_bytes;
    else
        wfull.cap = 0;
    if (cap > 0 && aligned && ccwave::wv_plan(h, w, cap, wg) &&
        (size_t)WV_WAVES * wg.wave_bytes <= 160 * 1024 - 64 && (cap >= nb || lds_wg)) {
        const bool can_overflow = cap < nb;
        if (can_overflow) {
            int rc = covahip_ensure_buffer(ctx, &ln.cc_ovf, &ln.cc_ovf_bytes, ((size_t)batch + 1) * sizeof(int32_t));
            if (rc) return rc;
            COVAHIP_CHECK_HIP(ctx, hipMemsetAsync(ln.cc_ovf, 0, sizeof(int32_t), ctx->stream));
        }
        int32_t *ovf_n = (int32_t *)ln.cc_ovf, *ovf_list = ovf_n ? ovf_n + 1 : nullptr;
        const size_t wlds = (size_t)WV_WAVES * wg.wave_bytes;
        int rc = open_lds(ctx, bboxcc_wave_kernel, wlds);
        if (rc) return rc;
        {
            ProfScope ps(ctx, "bboxcc_wave_kernel");
            hipLaunchKernelGGL(bboxcc_wave_kernel, dim3((batch + WV_WAVES - 1) / WV_WAVES), dim3(WV_WAVES * 64), wlds, ctx->stream,
                               d_mask, wg, batch, area_thresh, d_boxes, d_counts, max_boxes, ovf_list, ovf_n);
            COVAHIP_CHECK_HIP(ctx, hipGetLastError());
        }
        if (can_overflow) {
            rc = open_lds(ctx, bboxcc_kernel, lds_wg);
            if (rc) return rc;
            ProfScope ps(ctx, "bboxcc_kernel");
            hipLaunchKernelGGL(bboxcc_kernel, dim3(std::min(batch, 2 * num_cu)), dim3(CC_THREADS), lds_wg, ctx->stream, d_mask, g, wfull,
                               area_thresh, d_boxes, d_counts, max_boxes, (const int32_t *)ovf_list, (const int32_t *)ovf_n);
            COVAHIP_CHECK_HIP(ctx, hipGetLastError());
        }
        return COVAHIP_OK;
    }
    if (!lds_wg) {
        CcGeom gb;
        const size_t slab = cc_plan_global(h, w, gb);
        if (!slab) return COVAHIP_ERR_UNSUPPORTED;
        const int grid = std::min(batch, 2 * num_cu);
        int rc = covahip_ensure_buffer(ctx, &ln.cc_slab, &ln.cc_slab_bytes, slab * grid);
        if (rc) return rc;
        ProfScope ps(ctx, "bboxcc_big_kernel");
        hipLaunchKernelGGL(bboxcc_big_kernel, dim3(grid), dim3(CC_THREADS), 0, ctx->stream, d_mask, gb, (uint8_t *)ln.cc_slab, slab,
                           batch, area_thresh, d_boxes, d_counts, max_boxes);
        COVAHIP_CHECK_HIP(ctx, hipGetLastError());
        return COVAHIP_OK;
    }
    int rc = open_lds(ctx, bboxcc_kernel, lds_wg);
    if (rc) return rc;
    ProfScope ps(ctx, "bboxcc_kernel");
    hipLaunchKernelGGL(bboxcc_kernel, dim3(batch), dim3(CC_THREADS), lds_wg, ctx->stream, d_mask, g, wfull, area_thresh,
                       d_boxes, d_counts, max_boxes, (const int32_t *)nullptr, (const int32_t *)nullptr);
    COVAHIP_CHECK_HIP(ctx, hipGetLastError());
    return COVAHIP_OK;
}

extern "C" int covahip_bboxcc_set_wave_cap(covahip_ctx *ctx, int cap) {
    if (!ctx) return COVAHIP_ERR_INVALID_ARG;
    ctx->cc_wave_cap = cap;
    return COVAHIP_OK;
}

extern "C" int covahip_bboxcc(covahip_ctx *ctx, const uint8_t *mask, int batch, int h, int w, int area_thresh,
                              covahip_box *boxes, int32_t *counts, int max_boxes, int mem_kind) {
    if (!ctx || batch < 0 || h <= 0 || w <= 0 || max_boxes < 0) return COVAHIP_ERR_INVALID_ARG;
    if (batch == 0) return COVAHIP_OK;
    if (!mask || !counts || (!boxes && max_boxes > 0)) return COVAHIP_ERR_INVALID_ARG;
    COVAHIP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    if (int prc = covahip_primary_op(ctx)) return prc;   // stand-alone bboxcc runs on the primary stream
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
