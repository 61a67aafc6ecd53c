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
                                                             const int32_t *__restrict__ n_list,
                                                             const int32_t *__restrict__ stat_src, int32_t *__restrict__ stat_dst,
                                                             int stat_batch, int stat_cap) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    // last launch of a large-batch call: the call's overflow counters (final: the launches before this one are complete) go to
    // pinned host words that the NEXT call's plan reads -- no copy, no event, no host synchronisation
    if (stat_dst && blockIdx.x == 0 && threadIdx.x < 8)
        stat_dst[threadIdx.x] = threadIdx.x < 5 ? stat_src[threadIdx.x] : threadIdx.x == 5 ? stat_batch : threadIdx.x == 6 ? stat_cap : 0;
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
    const int n = min(*n_list, stat_batch);   // (list mode: stat_batch = the batch = the list's capacity)
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

// ---- one WAVE per frame (bboxcc_wave.h): WV_WAVES frames per workgroup, no workgroup barrier at all.
// list == nullptr: frame = global wave index.  Otherwise the launch is persistent over the *n_list frame indices in list
// (the second-chance pass over the frames that had more runs than the first pass's capacity).  A frame with more than
// g.cap runs goes to ovf_list.
constexpr int WV_WAVES = 4;
// Two instantiations on purpose: the first pass (LIST = false: frame = global wave index, nothing else in the kernel) must
// stay within 64 registers per lane -- eight waves per SIMD are what hides its HBM latency; with the persistent list loop in
// the same kernel hipcc allocated 92 (five waves per SIMD) and the sparse case lost 20 % (round 3's regression).
template <bool LIST>
__global__ __launch_bounds__(WV_WAVES * 64, LIST ? 1 : 8) void bboxcc_wave_kernel(const uint8_t *__restrict__ masks, ccwave::WvGeom g, int batch,
                                                                    int area_thresh, covahip_box *__restrict__ boxes,
                                                                    int32_t *__restrict__ counts, int max_boxes,
                                                                    const int32_t *__restrict__ list, const int32_t *__restrict__ n_list,
                                                                    int32_t *__restrict__ ovf_list, int32_t *__restrict__ ovf_n,
                                                                    int base_cap, int32_t *__restrict__ n_big,
                                                                    int32_t *__restrict__ zero_next) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint8_t *const sm = smem + (size_t)wave * g.wave_bytes;
    if constexpr (!LIST) {
        // the counters of the NEXT call on this lane (the other set of the pair): zeroed here instead of by a memset per call
        if (zero_next && blockIdx.x == 0 && threadIdx.x < 8) zero_next[threadIdx.x] = 0;
        const int frame = blockIdx.x * WV_WAVES + wave;
        if (frame >= batch) return;
        const int n = ccwave::frame_wave(masks + (size_t)frame * g.H * g.W, sm, g, area_thresh, boxes + (size_t)frame * max_boxes,
                                         counts + frame, max_boxes, lane);
        if (lane == 0) {
            if (n > g.cap) {   // more runs than the LDS region holds
                // (the index is clamped to the list's capacity: a counter set that was NOT zero at the start of the call -- a failed
                // call before this one, a replayed graph capture of a step -- must not turn into a write behind the list)
                const int k = atomicAdd(ovf_n, 1);
                if (k < batch) ovf_list[k] = frame;
            }
            // statistics for the next call's plan, SAMPLED: one frame in sixteen -- an atomic per frame on one address cost more
            // than the rest of the kernel when most frames had many runs (52 k same-address atomics = 0.5 ms)
            if (n_big && (frame & 15) == 0 && n > base_cap)   // [0]: 128 < n <= 192, [1]: 192 < n <= 256, [2]: n > 256
                atomicAdd(n_big + (n > 2 * base_cap ? 2 : 2 * n > 3 * base_cap ? 1 : 0), 1);
        }
    } else {
        const int nl = min(*n_list, batch);
        for (int k = blockIdx.x * WV_WAVES + wave; k < nl; k += gridDim.x * WV_WAVES) {
            const int frame = list[k];
            const int n = ccwave::frame_wave(masks + (size_t)frame * g.H * g.W, sm, g, area_thresh, boxes + (size_t)frame * max_boxes,
                                             counts + frame, max_boxes, lane);
            if (lane == 0 && n > g.cap) {
                const int k2 = atomicAdd(ovf_n, 1);
                if (k2 < batch) ovf_list[k2] = frame;
            }
            ccwave::wave_fence();   // the next frame reuses this wave's LDS region
        }
    }
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
//   * small batches (a few frames per CU): the workgroup-per-frame kernel finishes a frame sooner than one wave does;
//   * large batches: pass 1 = one wave per frame with a run capacity of WAVE_CAP (4.2 KB of LDS per wave, about thirty
//     frames in flight per CU); frames with more runs -- many objects, noise -- are collected in an overflow list and get a
//     SECOND CHANCE on the wave kernel at four times the capacity (persistent launch over the list); what overflows that too
//     goes to a persistent launch of the workgroup-per-frame kernel, which exits at once when its list is empty.
//     The first pass's capacity adapts: the kernel counts (on one frame in sixteen) the frames with more than 128 / 192 / 256
//     runs; when a quarter of the last completed call's frames (same lane) had that many, the next call starts at 192 / 256 /
//     512 and skips the wasted first read of those frames (192 runs: 5.7 KB of LDS per wave, 256: 7.2 KB).
//     What a call costs besides pass 1 (round 4; round 3 had added a memset, two persistent launches, a device-to-host copy
//     and an event per call, and a first-pass kernel of 92 registers -- 0.53 -> 0.44 of the HBM peak on sparse masks): ONE
//     more launch.  The counters live in two sets that calls use alternately (pass 1 zeroes the other set), the last kernel of a
//     call stores them to pinned host words (no copy, no event), and the second-chance pass is launched only when the last
//     completed call had frames that overflowed.
// Everything else runs the workgroup-per-frame kernel.
constexpr int WAVE_CAP = 128;
int covahip_bboxcc_launch(covahip_ctx *ctx, const uint8_t *d_mask, int batch, int h, int w, int area_thresh,
                          covahip_box *d_boxes, int32_t *d_counts, int max_boxes) {
    if (batch == 0) return COVAHIP_OK;
    CtxLane &ln = ctx->lane();             // scratch of the lane this call runs on (lane 0 on the primary stream)
    CcGeom g;
    const size_t lds = cc_plan(h, w, g);   // shapes the kernel assumes, checked on the host before any launch
    const int num_cu = ctx->props.multiProcessorCount;
    ccwave::WvGeom wg;
    const int nb = ((h + 1) / 2) * ((w + 1) / 2);
    int cap = ctx->cc_wave_cap;            // developer override: > 0 capacity of pass 1, < 0 wave kernel off, 0 automatic
    // a few frames per CU: the workgroup kernel finishes a frame sooner than a single wave does (9.5 vs 13.5 us at
    // b = 256), and there is nothing to overlap it with
    if (cap == 0) {
        cap = -1;
        if (batch > 3 * num_cu) {
            // the previous call's statistics, when they have arrived (pinned host words behind an event): frames with more
            // than WAVE_CAP runs; until then the last decision stands
            if (ln.cc_stat && ln.cc_stat[5] > 0) {   // (one frame in sixteen is counted; [2]: 128 < runs <= 192, [3]: <= 256, [4]: more; [5]: that call's batch)
                const int64_t c1 = 16 * (int64_t)ln.cc_stat[2], c2 = 16 * (int64_t)ln.cc_stat[3], c3 = 16 * (int64_t)ln.cc_stat[4], sb = ln.cc_stat[5];
                ln.cc_first_cap = 4 * c3 > sb ? 4 * WAVE_CAP : 4 * (c2 + c3) > sb ? 2 * WAVE_CAP : 4 * (c1 + c2 + c3) > sb ? 3 * WAVE_CAP / 2 : WAVE_CAP;
            }
            cap = ln.cc_first_cap ? ln.cc_first_cap : WAVE_CAP;
        }
    }
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
        // second chance: four times the capacity, as long as four such waves still fit a workgroup's LDS
        ccwave::WvGeom wg2{};
        const int cap2 = std::min(4 * cap, nb);
        const bool second = can_overflow && cap2 > cap && ccwave::wv_plan(h, w, cap2, wg2) &&
                            (size_t)WV_WAVES * wg2.wave_bytes <= 160 * 1024 - 64;
        const bool third = can_overflow && (!second || cap2 < nb);
        // [set 0: n1 n2 c1 c2 c3 - - -][set 1: ...][list1: batch][list2: batch]; calls alternate between the two counter sets
        // and pass 1 zeroes the other one, so a call costs no memset (the buffer is zeroed when it is (re)allocated)
        int32_t *ovf = nullptr;
        if (can_overflow) {
            const size_t need = (2 * (size_t)batch + 16) * sizeof(int32_t);
            const bool fresh = need > ln.cc_ovf_bytes;
            int rc = covahip_ensure_buffer(ctx, &ln.cc_ovf, &ln.cc_ovf_bytes, need);
            if (rc) return rc;
            if (fresh) {
                COVAHIP_CHECK_HIP(ctx, hipMemsetAsync(ln.cc_ovf, 0, 16 * sizeof(int32_t), ctx->stream));
                ln.cc_stat_turn = 0;
            }
            ovf = (int32_t *)ln.cc_ovf;
            if (!ln.cc_stat_ring) {   // pinned, device-visible words the last kernel of a call writes: {n1, n2, c1, c2, c3, batch, cap, -}
                COVAHIP_CHECK_HIP(ctx, hipHostMalloc((void **)&ln.cc_stat_ring, 8 * sizeof(int32_t), hipHostMallocMapped));
                for (int k = 0; k < 8; k++) ln.cc_stat_ring[k] = 0;
                ln.cc_stat = ln.cc_stat_ring;
            }
        }
        // (the turn advances only once pass 1 -- which zeroes the OTHER set for the next call -- has been enqueued: a call that
        // fails before that leaves the sets as they were)
        const unsigned set = can_overflow ? (ln.cc_stat_turn & 1) : 0;
        int32_t *cnt = ovf ? ovf + 8 * set : nullptr, *cnt_next = ovf ? ovf + 8 * (set ^ 1) : nullptr;
        int32_t *n1 = cnt, *n2 = cnt ? cnt + 1 : nullptr, *n_big = cnt ? cnt + 2 : nullptr;
        int32_t *list1 = ovf ? ovf + 16 : nullptr, *list2 = ovf ? ovf + 16 + batch : nullptr;
        // the second-chance pass is worth a launch only when frames DO overflow: the last completed call on this lane says so
        // (nothing known yet: it runs); whatever overflows is picked up by the workgroup kernel below either way
        const bool second_now = second && !(ln.cc_stat && ln.cc_stat[5] > 0 && ln.cc_stat[0] == 0);
        const size_t wlds = (size_t)WV_WAVES * wg.wave_bytes;
        int rc = open_lds(ctx, bboxcc_wave_kernel<false>, wlds);
        if (rc) return rc;
        if (second) rc = open_lds(ctx, bboxcc_wave_kernel<true>, (size_t)WV_WAVES * wg2.wave_bytes);
        if (rc) return rc;
        {
            ProfScope ps(ctx, "bboxcc_wave_kernel");
            hipLaunchKernelGGL(bboxcc_wave_kernel<false>, dim3((batch + WV_WAVES - 1) / WV_WAVES), dim3(WV_WAVES * 64), wlds, ctx->stream,
                               d_mask, wg, batch, area_thresh, d_boxes, d_counts, max_boxes, (const int32_t *)nullptr,
                               (const int32_t *)nullptr, list1, n1, WAVE_CAP, n_big, cnt_next);
            COVAHIP_CHECK_HIP(ctx, hipGetLastError());
            if (can_overflow) ln.cc_stat_turn++;
        }
        if (second_now) {
            const size_t wlds2 = (size_t)WV_WAVES * wg2.wave_bytes;
            const int per_cu = std::max(1, (int)((160 * 1024 - 256) / wlds2));
            const int grid = std::min((batch + WV_WAVES - 1) / WV_WAVES, per_cu * num_cu);
            ProfScope ps(ctx, "bboxcc_wave_kernel_2");
            hipLaunchKernelGGL(bboxcc_wave_kernel<true>, dim3(grid), dim3(WV_WAVES * 64), wlds2, ctx->stream, d_mask, wg2, batch, area_thresh,
                               d_boxes, d_counts, max_boxes, (const int32_t *)list1, (const int32_t *)n1, list2, n2, 0, (int32_t *)nullptr,
                               (int32_t *)nullptr);
            COVAHIP_CHECK_HIP(ctx, hipGetLastError());
        }
        if (can_overflow) {
            // the workgroup kernel over what is left (exits at once when the list is empty); it also exports the call's counters
            rc = open_lds(ctx, bboxcc_kernel, lds_wg);
            if (rc) return rc;
            int32_t *stat_dev = nullptr;
            COVAHIP_CHECK_HIP(ctx, hipHostGetDevicePointer((void **)&stat_dev, ln.cc_stat_ring, 0));
            const bool have_list = third || !second_now;   // (second chance at full capacity and run: nothing can be left)
            // a list that was empty in the last completed call is expected to be empty again: a small grid (any grid is correct,
            // the launch is persistent over the list) costs less to start and to drain
            const bool quiet = ln.cc_stat && ln.cc_stat[5] > 0 && ln.cc_stat[0] == 0;
            ProfScope ps(ctx, "bboxcc_kernel");
            hipLaunchKernelGGL(bboxcc_kernel, dim3(!have_list ? 1 : quiet ? 32 : std::min(batch, 2 * num_cu)), dim3(CC_THREADS), lds_wg, ctx->stream, d_mask, g, wfull,
                               area_thresh, d_boxes, d_counts, max_boxes, (const int32_t *)(second_now ? list2 : list1),
                               (const int32_t *)(second_now ? n2 : n1), (const int32_t *)cnt, stat_dev, batch, cap);
            COVAHIP_CHECK_HIP(ctx, hipGetLastError());
            ln.cc_stat_batch = batch;
            ln.cc_stat_cap = cap;
            ln.cc_second_skipped = second && !second_now;
        } else {
            ln.cc_stat_batch = 0;
        }
        return COVAHIP_OK;
    }
    ln.cc_stat_batch = 0;   // the kernels below cannot overflow
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
                       d_boxes, d_counts, max_boxes, (const int32_t *)nullptr, (const int32_t *)nullptr, (const int32_t *)nullptr,
                       (int32_t *)nullptr, 0, 0);
    COVAHIP_CHECK_HIP(ctx, hipGetLastError());
    return COVAHIP_OK;
}

// (covahip_dev.h) overflow statistics of the last large-batch bboxcc call on the primary stream: out = {batch, frames that
// overflowed pass 1, frames that overflowed pass 2 too, capacity of pass 1}; batch = 0 when that call could not overflow.
extern "C" int covahip_dev_bboxcc_overflow(covahip_ctx *ctx, int32_t *out4) {
    if (!ctx || !out4) return COVAHIP_ERR_INVALID_ARG;
    int rc = covahip_sync_all(ctx);
    if (rc) return rc;
    const CtxLane &ln = ctx->lanes[0];
    // (everything has drained: the words are those of the last large-batch call; a frame that overflowed pass 1 while the
    // second-chance pass was skipped counts as having overflowed both)
    out4[0] = ln.cc_stat_batch;
    out4[1] = ln.cc_stat && ln.cc_stat_batch ? ln.cc_stat[0] : 0;
    out4[2] = ln.cc_stat && ln.cc_stat_batch ? ln.cc_stat[ln.cc_second_skipped ? 0 : 1] : 0;
    out4[3] = ln.cc_stat_cap;
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
