/*
 * covahip_dev.h -- developer switches of libcovahip.so.  NOT part of the drop-in boundary
 * (include/covahip.h): used by tools/ and tests/ only, may change or disappear between builds.
 */
#ifndef COVAHIP_DEV_H
#define COVAHIP_DEV_H

#include "covahip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Kernel chain of the loaded model: 1 = default, 4 = decoder blocks 0..2 as three launches instead of one (the form
 * before the fused decoder; measured slower on MI355X, kept because it is the fallback for grids whose three decoder
 * tiles do not fit in LDS together, and tests compare the two bit for bit).  (2 and 3 were single-launch forms of encoder
 * levels 0 + 1, measured slower and removed in round 3; DESIGN.md has their numbers.)  5 = encoder level 1 on the
 * round-1..3 kernel (32x32x16 products, swizzled LDS band, LDS-DMA) instead of enc1_mfma (16x16x32 products, affine band,
 * round 4): the fallback for grids wider than enc1_mfma's fixed LDS row, and the A/B partner.  6 = encoder levels 2 and 3
 * on the general tiles (eight consecutive windows of a band) instead of the row-aligned ones (eight windows of one window
 * row, fragment addresses from per-kernel lane constants; round 4), which are the default where the geometry suits them. */
int covahip_blobnet_set_impl(covahip_ctx *ctx, int impl);

/* bboxcc kernel choice: cap > 0 = run capacity of the wave-per-frame kernel's first pass (frames with more runs get a
 * second chance at four times the capacity, then the workgroup kernel), cap < 0 = workgroup-per-frame kernel only,
 * 0 = automatic (default: 128, or 512 when more than a quarter of the previous call's frames overflowed). */
int covahip_bboxcc_set_wave_cap(covahip_ctx *ctx, int cap);
/* Overflow statistics of the last large-batch covahip_bboxcc call (device pointers): out4 = {batch, frames that overflowed
 * pass 1, frames that overflowed pass 2 too, capacity of pass 1}; batch = 0 when that call ran a kernel that cannot
 * overflow.  Synchronises the ctx.  (tools/bboxcc_sweep.py) */
int covahip_dev_bboxcc_overflow(covahip_ctx *ctx, int32_t *out4);

/* Encoder band plan of level 1..3 (tools/plan_sweep.sh): nbands bands of pool-window rows per frame, nbuf = 1 or 2 LDS
 * buffers (2 = the next band is requested while this one is computed; measured no faster).  nbands = 0
 * restores the automatic plan.  A plan that does not fit in LDS makes the next forward call return
 * COVAHIP_ERR_UNSUPPORTED. */
int covahip_blobnet_set_enc_plan(covahip_ctx *ctx, int level, int nbands, int nbuf);

/* Shader clock (MHz) the chip holds right now: one wave runs a dependent chain for about busy_us microseconds on a stream
 * of its own -- beside whatever the ctx has in flight -- and brackets it with s_memtime / s_memrealtime
 * (MI355X_MICROARCH.md, DVFS give-back).  bench.py prints it so that step times of different boxes can be compared. */
int covahip_dev_clock_mhz(covahip_ctx *ctx, int busy_us, float *mhz);

/* The same carrier-frame step (device pointers, one lane) `iters` times as stream launches and as launches of ONE captured HIP
 * graph of it: total milliseconds of each.  Synchronises the ctx.  (tools/graph_probe.py) */
int covahip_dev_graph_probe(covahip_ctx *ctx, const uint8_t *d_frames, int n_frames, const int32_t *stack_index, int batch,
                            int area_thresh, covahip_box *d_boxes, int32_t *d_counts, int max_boxes, uint8_t *d_mask, int iters,
                            float *ms_direct, float *ms_graph);

/* What covahip_pipe_create's hardware-queue probe decided (pipe.hip, round 6): how many of the ctx's ACTIVE lanes share a hardware
 * queue with the pipe's upload stream / with its result stream.  0 / <= 1 with up to three lanes on the default four queues. */
struct covahip_pipe;
int covahip_dev_pipe_queue_plan(struct covahip_pipe *pipe, int *lanes_on_upload_queue, int *lanes_on_result_queue);

#ifdef __cplusplus
}
#endif
#endif /* COVAHIP_DEV_H */
