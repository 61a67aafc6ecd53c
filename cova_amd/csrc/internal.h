// Internal declarations shared by the translation units of libcovahip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "covahip.h"

struct covahip_blobnet;  // blobnet.hip

// Lanes: batches in flight on one GPU.  Every launch of the hot path has a fixed cost (weights into registers, the first
// band's DMA, the slowest workgroup's tail) during which most of the chip idles; at b = 256 that is a third of a step.
// The reference keeps its GPU busy the same way -- sixteen BlobNet engines with their own batches on one GPU
// (experiment/cova/config.yaml:33-34, pipeline/cova/pipeline.py:139-181).  A ctx therefore owns `n_lanes` HIP streams,
// each with its own activation workspace and bboxcc scratch; consecutive covahip_filter_forward(_frames) calls on device
// pointers go to consecutive lanes and overlap.  Ordering rules (LaneScope / covahip_primary_op, ctx.hip):
//   * a call placed on a lane is ordered behind everything enqueued on the ctx's primary stream before it;
//   * every other entry point (timers, sync, copies, stand-alone bboxcc / blobnet calls) first makes the primary stream
//     wait for all lanes, so it observes every earlier filter call;
//   * two filter calls with nothing in between are NOT ordered with each other: they must not share output buffers.
// n_lanes = 1: everything runs on the primary stream, in call order.
constexpr int COVAHIP_MAX_LANES = 4;
struct CtxLane {
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;     // recorded behind the last call placed on this lane
    bool pending = false;          // `done` has not been joined into the primary stream yet
    uint64_t seen_seq = 0;         // state of the primary stream this lane is ordered behind (covahip_ctx::primary_seq)
    // bboxcc scratch of the fused path (mask when the caller wants none), overflow list of the wave kernel (count + frame
    // indices), state slab of frames too large for LDS (one per resident workgroup)
    void *cc_scratch = nullptr;
    size_t cc_scratch_bytes = 0;
    void *cc_ovf = nullptr;
    size_t cc_ovf_bytes = 0;
    void *cc_slab = nullptr;
    size_t cc_slab_bytes = 0;
    // overflow counters of the last completed large-batch bboxcc call on this lane: pinned, device-visible words
    // {overflowed pass 1, overflowed pass 2, sampled frames with 128 < runs <= 192 / <= 256 / more, batch, capacity of pass 1}
    // written by the call's last kernel (no copy, no event); bboxcc.hip adapts the next call's plan to them
    int32_t *cc_stat = nullptr;
    int32_t *cc_stat_ring = nullptr;
    unsigned cc_stat_turn = 0;           // which of the two device counter sets the next call uses
    bool cc_second_skipped = false;      // the last call ran without the second-chance pass
    int cc_stat_batch = 0, cc_stat_cap = 0;
    int cc_first_cap = 0;                // first-pass capacity the statistics last decided on (0: none yet)
};

struct covahip_ctx {
    int device = 0;
    hipStream_t primary = nullptr;   // the ctx's own stream
    hipStream_t stream = nullptr;    // where launches go NOW: `primary`, or the current lane's stream inside a LaneScope
    CtxLane lanes[COVAHIP_MAX_LANES];
    int n_lanes = 1, next_lane = 0, cur_lane = 0;   // one lane unless the caller opts in (covahip_ctx_set_lanes)
    bool in_lane = false;
    uint64_t primary_seq = 1;        // bumped by every operation enqueued on the primary stream
    hipEvent_t ev_fork = nullptr;
    std::string last_hip_error;
    hipDeviceProp_t props{};
    // timers
    hipEvent_t t_start[16]{};
    hipEvent_t t_stop[16]{};
    // per-kernel profiling
    bool profile = false;
    std::string profile_filter;
    struct ProfEntry {
        std::string name;
        hipEvent_t a, b;
    };
    std::vector<ProfEntry> prof_pending;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_pool;
    std::map<std::string, std::pair<double, int64_t>> prof_acc;
    // staging buffers for host-pointer calls
    void *stage_in = nullptr;
    size_t stage_in_bytes = 0;
    void *stage_out = nullptr;
    size_t stage_out_bytes = 0;
    void *pinned = nullptr;
    size_t pinned_bytes = 0;
    int cc_wave_cap = 0;       // bboxcc wave kernel: developer override of its run capacity
    CtxLane &lane() { return lanes[cur_lane]; }
    struct { int nbands, nbuf; } enc_plan[4] = {};   // developer override of the encoder band plan per level (0 = automatic)
    covahip_blobnet *blobnet = nullptr;
};

#define COVAHIP_CHECK_HIP(ctx, expr)                                                        \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            (ctx)->last_hip_error = std::string(#expr) + ": " + hipGetErrorString(_e);      \
            return COVAHIP_ERR_HIP;                                                         \
        }                                                                                   \
    } while (0)

// RAII-less helpers: bracket a kernel launch with events when profiling is on.
struct ProfScope {
    covahip_ctx *ctx;
    int idx = -1;
    ProfScope(covahip_ctx *c, const char *name);
    ~ProfScope();
};

int covahip_ensure_buffer(covahip_ctx *ctx, void **buf, size_t *cur, size_t need);

// Places what follows on the next lane (see CtxLane above); re-entrant (an inner scope is a no-op).  `ok()` is false when a
// HIP call of the fork failed (ctx->last_hip_error says which).
struct LaneScope {
    covahip_ctx *ctx;
    bool owner = false, failed = false;
    explicit LaneScope(covahip_ctx *c);
    ~LaneScope();
    bool ok() const { return !failed; }
};
// Before anything is enqueued on the primary stream: the primary stream waits for every lane's last call.
int covahip_primary_op(covahip_ctx *ctx);
// Blocks until every lane and the primary stream have drained.
int covahip_sync_all(covahip_ctx *ctx);

// bboxcc.hip
int covahip_bboxcc_launch(covahip_ctx *ctx, const uint8_t *d_mask, int batch, int h, int w, int area_thresh,
                          covahip_box *d_boxes, int32_t *d_counts, int max_boxes);

// blobnet.hip
void covahip_blobnet_destroy(covahip_ctx *ctx);
int covahip_blobnet_forward_dev(covahip_ctx *ctx, const uint8_t *d_stack, int batch, float *d_logits,
                                uint8_t *d_mask);
int covahip_blobnet_geometry(covahip_ctx *ctx, int *h, int *w);
int covahip_blobnet_grow_lanes(covahip_ctx *ctx, int n_lanes);   // workspaces of lanes [0, n_lanes) of the loaded model
