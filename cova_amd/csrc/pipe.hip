// Pipelined host-buffer form of the hot path (covahip_pipe_*): what a batching element needs to keep the GPU fed
// from host-side entropy decoders.
//
//   * every slot owns PINNED host buffers (hipHostMalloc) for its carrier frames, its stack -> frame table and its
//     results; the caller writes the frames of a batch straight into the slot (the copy a batching element makes
//     anyway, nvstreammux-style), so no pageable copy ever blocks the calling thread;
//   * H2D of batch k+1, the kernels of batch k (a lane of the ctx: its own stream and activation workspace, so the
//     kernels of consecutive batches overlap as well), D2H of batch k-1, chained by events -- PCIe traffic in both
//     directions hides behind the kernels;
//   * (round 6) WHERE the copy streams land among the runtime's hardware queues is measured at creation, not left to the order
//     in which the process happened to create its streams: the upload stream takes a queue no active lane uses, and a batch's
//     results leave on the lane that ran it (streams_share_queue below; 2.2 M frames/s PCIe-inclusive instead of 1.4 - 1.9 M);
//   * boxes are compacted on the device (exclusive scan of the per-frame counts -> one packed array + offsets), so
//     the D2H copy carries what exists, not batch x max_boxes slots.
//
// Reference pipeline stage this stands for: nvstreammux -> nvinfer(BlobNet) -> nvstreamdemux -> maskcopy -> bboxcc
// (pipeline/cova/pipeline.py:139-261), with metapreprocess' stacking (imp.rs:288-332) as the GPU-side gather of
// covahip_filter_forward_frames.
#include <algorithm>
#include <chrono>
#include <ctime>
#include <vector>

#include "blobnet.h"
#include "internal.h"

namespace {

// Boxes -> packed array, one launch (round 5; rounds 2-4: a one-workgroup scan kernel + this gather = two launches, 4.6 + 4.2 us on the
// GPU and two launches' worth of host time per batch).  Workgroup b sums min(count, max_boxes) of the frames in front of it itself
// (a wave reduction: at most PACK_SELF_SCAN values from L2), writes offsets[b] -- the last one also offsets[batch] -- and moves its
// frame's boxes.  Batches larger than PACK_SELF_SCAN (the pipe admits 65,536 frames) would read batch^2 / 2 counts that way: their
// offsets come from scan_kernel, one pass over the counts, and `pre` != nullptr.
// Round 6, measured and NOT kept: this kernel storing counts, offsets and boxes straight into the slot's pinned host buffer (no
// copy-engine operation on the result path, the slot's event on the lane's stream) -- DESIGN.md round 6, item 4: equal for a few
// boxes per frame, 10 % slower at 256 boxes per frame (5 MB of the kernel's own stores across the link per batch).
constexpr int PACK_SELF_SCAN = 1024;
__global__ __launch_bounds__(1024) void scan_kernel(const int32_t *__restrict__ counts, int batch, int max_boxes, int32_t *__restrict__ pre) {
    __shared__ int32_t wsum[16];
    __shared__ int32_t carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < batch; base += 1024) {
        const int i = base + tid;
        const int32_t v = i < batch ? min(counts[i], max_boxes) : 0;
        int32_t x = v;   // inclusive scan within the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int32_t y = __shfl_up(x, o, 64);
            if (lane >= o) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        int32_t before = carry;
        for (int w = 0; w < wave; w++) before += wsum[w];
        if (i < batch) pre[i] = before + x - v;
        __syncthreads();
        if (tid == 1023) carry = before + x;
        __syncthreads();
    }
    if (tid == 0) pre[batch] = carry;
}
__global__ __launch_bounds__(64) void pack_kernel(const covahip_box *__restrict__ boxes, const int32_t *__restrict__ counts, int batch,
                                                  int max_boxes, const int32_t *__restrict__ pre, int32_t *__restrict__ offsets,
                                                  covahip_box *__restrict__ packed) {
    const int b = blockIdx.x, lane = threadIdx.x;
    int32_t sum = 0;
    const int n = min(counts[b], max_boxes);
    if (pre) {
        sum = pre[b];   // (scan_kernel wrote offsets[] itself, offsets[batch] included: pre == offsets)
    } else {
        for (int i = lane; i < b; i += 64) sum += min(counts[i], max_boxes);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        if (lane == 0) {
            offsets[b] = sum;
            if (b == batch - 1) offsets[batch] = sum + n;
        }
    }
    // 20-byte boxes as 5 dwords: consecutive lanes move consecutive dwords
    const uint32_t *src = reinterpret_cast<const uint32_t *>(boxes + (size_t)b * max_boxes);
    uint32_t *dst = reinterpret_cast<uint32_t *>(packed + sum);
    for (int i = lane; i < n * 5; i += 64) dst[i] = src[i];
}

// ---- which hardware queue does a stream share?  (round 6; DESIGN.md "The copy streams' hardware queues")
// HIP multiplexes its streams onto a few hardware queues (four by default).  Packets of different streams that share one are
// processed in order, so a copy stream's marker that waits for an 88 us upload -- or for a lane's kernels -- holds up every kernel a
// lane has queued behind it: with the ctx's five streams (primary + four lanes) every queue is taken, and where the pipe's two copy
// streams land decided between 116 and 182 us per batch on one box (creation order alone).  The pipe therefore MEASURES where a
// candidate stream lands: a kernel that spins for PROBE_US on stream A, an empty kernel on stream B right behind it -- B's kernel
// finishes early unless the two share a hardware queue.
constexpr int PROBE_US = 120;
__global__ __launch_bounds__(64) void spin_kernel(unsigned long long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}
__global__ __launch_bounds__(64) void empty_kernel() {}
// true: work on `b` waits for work enqueued earlier on `a` (they share a hardware queue)
static bool streams_share_queue_once(hipStream_t a, hipStream_t b) {
    hipStreamSynchronize(a);
    hipStreamSynchronize(b);
    const auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, a, (unsigned long long)PROBE_US * 100);
    hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, b);
    hipStreamSynchronize(b);
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    hipStreamSynchronize(a);
    return us > 0.6 * PROBE_US;
}
// (a host thread that loses its core between the two launches and the wait reads as "shared": a positive is confirmed once)
static bool streams_share_queue(hipStream_t a, hipStream_t b) { return streams_share_queue_once(a, b) && streams_share_queue_once(a, b); }

struct Slot {
    uint8_t *h_frames = nullptr;
    int32_t *h_index = nullptr;
    int32_t *h_meta = nullptr;       // counts [B] | offsets [B + 1] | pad to 64 B | packed boxes: ONE pinned allocation (and one
    covahip_box *h_packed = nullptr; // device allocation of the same shape), so that one copy brings all of it back
    uint8_t *h_mask = nullptr;
    uint8_t *d_frames = nullptr;
    covahip_box *d_boxes = nullptr, *d_packed = nullptr;
    int32_t *d_meta = nullptr;
    uint8_t *d_mask = nullptr;
    hipEvent_t ev_in = nullptr, ev_done = nullptr, ev_out = nullptr;
    int state = 0;                   // 0 free, 1 acquired (being filled), 2 submitted, 3 collected (results in use)
    int batch = 0, n_frames = 0;
    int spec = 0;                    // packed boxes the pipelined D2H of this submission carries
};

}  // namespace

struct covahip_pipe {
    covahip_ctx *ctx = nullptr;
    size_t meta_bytes = 0;           // bytes of a slot's meta words in front of its packed boxes
    int max_batch = 0, max_frames = 0, max_boxes = 0, n_slots = 0, want_mask = 0, packed = 0;
    size_t frame_bytes = 0, hw = 0;
    int spec = 0;                    // packed boxes copied back by the pipelined D2H; grows to what the stream produces
    hipStream_t s_h2d = nullptr, s_d2h = nullptr;
    std::vector<Slot> slots;
    int next = 0;
    bool sleeping_wait = false;      // covahip_pipe_set_blocking_wait: wait / collect poll + sleep instead of hipEventSynchronize
    bool results_on_lane = false;    // the copy-out of a batch is enqueued on the lane that ran its kernels (see covahip_pipe_submit)
    int queue_plan[2] = {0, 0};      // active lanes that share a hardware queue with the upload / the result stream (covahip_dev_pipe_queue_plan)
};

#define PIPE_CHECK(expr) COVAHIP_CHECK_HIP(p->ctx, expr)

extern "C" {

void covahip_pipe_destroy(covahip_pipe *p) {
    if (!p) return;
    hipSetDevice(p->ctx->device);
    covahip_sync_all(p->ctx);
    if (p->s_h2d) hipStreamSynchronize(p->s_h2d);
    if (p->s_d2h) hipStreamSynchronize(p->s_d2h);
    for (Slot &s : p->slots) {
        if (s.h_frames) hipHostFree(s.h_frames);
        if (s.h_index) hipHostFree(s.h_index);
        if (s.h_meta) hipHostFree(s.h_meta);   // (h_packed / d_packed point into the meta allocations)
        if (s.h_mask) hipHostFree(s.h_mask);
        if (s.d_frames) hipFree(s.d_frames);
        if (s.d_boxes) hipFree(s.d_boxes);
        if (s.d_meta) hipFree(s.d_meta);
        if (s.d_mask) hipFree(s.d_mask);
        if (s.ev_in) hipEventDestroy(s.ev_in);
        if (s.ev_done) hipEventDestroy(s.ev_done);
        if (s.ev_out) hipEventDestroy(s.ev_out);
    }
    if (p->s_h2d) hipStreamDestroy(p->s_h2d);
    if (p->s_d2h) hipStreamDestroy(p->s_d2h);
    delete p;
}

int covahip_pipe_create(covahip_ctx *ctx, int max_batch, int max_frames, int max_boxes, int n_slots, int want_mask,
                        covahip_pipe **out) {
    if (!ctx || !out || max_batch <= 0 || max_frames < BN_T || max_boxes <= 0 || n_slots < 1 || n_slots > 8)
        return COVAHIP_ERR_INVALID_ARG;
    *out = nullptr;
    covahip_blobnet *m = ctx->blobnet;
    if (!m) return COVAHIP_ERR_NOT_LOADED;
    if (max_batch > m->max_batch || max_frames > BN_T * m->max_batch || max_batch > 65536) return COVAHIP_ERR_INVALID_ARG;
    COVAHIP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    covahip_pipe *p = new covahip_pipe();
    p->ctx = ctx;
    p->max_batch = max_batch; p->max_frames = max_frames; p->max_boxes = max_boxes; p->n_slots = n_slots;
    p->want_mask = want_mask != 0;
    p->hw = (size_t)m->H * m->W;
    p->frame_bytes = p->hw * 4;
    p->spec = std::min(max_batch * max_boxes, std::max(1024, max_batch * 16));
    p->slots.resize(n_slots);
    auto fail = [&](int rc) { covahip_pipe_destroy(p); return rc; };
    {
        // the copy streams: among a few candidates, the upload stream is one that shares its hardware queue with as few ACTIVE lanes
        // as possible (none, while the ctx runs at most three of its four lanes: the primary stream's queue is idle while a pipe is
        // fed), the result stream one that does not share the upload stream's queue, again with as few active lanes as possible
        constexpr int NC = 8;
        hipStream_t cand[NC] = {};
        int busy[NC];              // active lanes on the candidate's queue
        int nc = 0;
        covahip_sync_all(ctx);
        for (; nc < NC; nc++) {
            if (hipStreamCreateWithFlags(&cand[nc], hipStreamNonBlocking) != hipSuccess) break;
            busy[nc] = 0;
            if (ctx->n_lanes > 1)
                for (int l = 0; l < ctx->n_lanes; l++) busy[nc] += streams_share_queue(ctx->lanes[l].stream, cand[nc]);
            else
                busy[nc] = streams_share_queue(ctx->primary, cand[nc]);   // one lane: the kernels run on the primary stream
        }
        if (nc < 2) {
            for (int i = 0; i < nc; i++) hipStreamDestroy(cand[i]);
            return fail(COVAHIP_ERR_HIP);
        }
        int up = 0;
        for (int i = 1; i < nc; i++)
            if (busy[i] < busy[up]) up = i;
        int down = -1;
        for (int i = 0; i < nc; i++) {
            if (i == up || streams_share_queue(cand[up], cand[i])) continue;
            if (down < 0 || busy[i] < busy[down]) down = i;
        }
        if (down < 0) down = up == 0 ? 1 : 0;   // (one hardware queue for everything: nothing to choose)
        p->s_h2d = cand[up];
        p->s_d2h = cand[down];
        p->queue_plan[0] = busy[up]; p->queue_plan[1] = busy[down];
        p->results_on_lane = busy[up] == 0 && ctx->n_lanes > 1 && ctx->n_lanes <= 3;
        for (int i = 0; i < nc; i++)
            if (i != up && i != down) hipStreamDestroy(cand[i]);
        (void)hipGetLastError();
    }
    const size_t meta_ints = (size_t)2 * max_batch + 1;
    const size_t meta_bytes = (meta_ints * sizeof(int32_t) + 63) & ~(size_t)63;   // the packed boxes start 64-byte aligned behind it
    p->meta_bytes = meta_bytes;
    for (Slot &s : p->slots) {
        bool ok = hipHostMalloc((void **)&s.h_frames, (size_t)max_frames * p->frame_bytes, hipHostMallocDefault) == hipSuccess &&
                  hipHostMalloc((void **)&s.h_index, (size_t)max_batch * BN_T * sizeof(int32_t), hipHostMallocDefault) == hipSuccess &&
                  hipHostMalloc((void **)&s.h_meta, meta_bytes + (size_t)max_batch * max_boxes * sizeof(covahip_box), hipHostMallocDefault) == hipSuccess &&
                  hipMalloc((void **)&s.d_frames, (size_t)max_frames * p->frame_bytes) == hipSuccess &&
                  hipMalloc((void **)&s.d_boxes, (size_t)max_batch * max_boxes * sizeof(covahip_box)) == hipSuccess &&
                  hipMalloc((void **)&s.d_meta, meta_bytes + (size_t)max_batch * max_boxes * sizeof(covahip_box)) == hipSuccess &&
                  hipEventCreateWithFlags(&s.ev_in, hipEventDisableTiming) == hipSuccess &&
                  hipEventCreateWithFlags(&s.ev_done, hipEventDisableTiming) == hipSuccess &&
                  hipEventCreateWithFlags(&s.ev_out, hipEventDisableTiming) == hipSuccess;
        if (ok) {
            s.h_packed = reinterpret_cast<covahip_box *>(reinterpret_cast<uint8_t *>(s.h_meta) + meta_bytes);
            s.d_packed = reinterpret_cast<covahip_box *>(reinterpret_cast<uint8_t *>(s.d_meta) + meta_bytes);
        }
        if (ok && p->want_mask)
            ok = hipHostMalloc((void **)&s.h_mask, (size_t)max_batch * p->hw, hipHostMallocDefault) == hipSuccess &&
                 hipMalloc((void **)&s.d_mask, (size_t)max_batch * p->hw) == hipSuccess;
        if (!ok) {
            ctx->last_hip_error = "covahip_pipe_create: allocation failed";
            return fail(COVAHIP_ERR_HIP);
        }
    }
    *out = p;
    return COVAHIP_OK;
}

// developer / test view of what the probe at creation found (include/covahip_dev.h)
int covahip_dev_pipe_queue_plan(covahip_pipe *p, int *lanes_on_upload_queue, int *lanes_on_result_queue) {
    if (!p) return COVAHIP_ERR_INVALID_ARG;
    if (lanes_on_upload_queue) *lanes_on_upload_queue = p->queue_plan[0];
    if (lanes_on_result_queue) *lanes_on_result_queue = p->queue_plan[1];
    return COVAHIP_OK;
}

int covahip_pipe_set_packed(covahip_pipe *p, int on) {
    if (!p) return COVAHIP_ERR_INVALID_ARG;
    for (const Slot &s : p->slots)
        if (s.state != 0) return COVAHIP_ERR_INVALID_ARG;   // not with slots acquired or in flight
    p->packed = on != 0;
    p->frame_bytes = p->hw * (p->packed ? 2 : 4);
    return COVAHIP_OK;
}

// hipEventSynchronize spins on the completion signal by default: a collector thread that waits for the GPU most of the time
// then burns a whole core doing so (13 - 16 % of the chain's CPU samples sat in the HSA runtime, tools/prof_resolve.py).  With
// blocking waits the slot's "results are in host memory" event is created with hipEventBlockingSync and the waiting thread
// sleeps until the completion interrupt; several slots are in flight, so the wake-up latency is not on the critical path.
// Round 6: the call-chain profile of the plugin chain (profiles/r6/chain_call_chains_before.txt) put 12.5 % of the process's CPU
// samples inside hipEventSynchronize under covahip_pipe_wait -- WITH hipEventBlockingSync: the runtime spins on the signal for a
// while before it parks, and with a batch every 110 - 200 us it never gets to park.  A sleeping wait is therefore a poll:
// hipEventQuery, then a nanosleep of 20 us (the kernel's timer slack makes that ~70 us), a handful of wake-ups per batch.
static int wait_event_sleeping(hipEvent_t ev) {
    while (true) {
        const hipError_t q = hipEventQuery(ev);
        if (q == hipSuccess) return COVAHIP_OK;
        if (q != hipErrorNotReady) return COVAHIP_ERR_HIP;
        (void)hipGetLastError();
        struct timespec ts = {0, 20000};
        nanosleep(&ts, nullptr);
    }
}

int covahip_pipe_set_blocking_wait(covahip_pipe *p, int on) {
    if (!p) return COVAHIP_ERR_INVALID_ARG;
    for (const Slot &s : p->slots)
        if (s.state != 0) return COVAHIP_ERR_INVALID_ARG;   // not with slots acquired or in flight
    if (hipSetDevice(p->ctx->device) != hipSuccess) return COVAHIP_ERR_HIP;
    p->sleeping_wait = on != 0;
    for (Slot &s : p->slots) {
        hipEvent_t ev = nullptr;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming | (on ? hipEventBlockingSync : 0)) != hipSuccess) return COVAHIP_ERR_HIP;
        if (s.ev_out) hipEventDestroy(s.ev_out);
        s.ev_out = ev;
    }
    return COVAHIP_OK;
}

int covahip_pipe_acquire(covahip_pipe *p, int *slot, uint8_t **frames, int32_t **stack_index) {
    if (!p || !slot || !frames || !stack_index) return COVAHIP_ERR_INVALID_ARG;
    for (int k = 0; k < p->n_slots; k++) {
        const int i = (p->next + k) % p->n_slots;
        if (p->slots[i].state == 0) {
            p->slots[i].state = 1;
            p->next = (i + 1) % p->n_slots;
            *slot = i;
            *frames = p->slots[i].h_frames;
            *stack_index = p->slots[i].h_index;
            return COVAHIP_OK;
        }
    }
    return COVAHIP_ERR_OVERFLOW;   // every slot is acquired or in flight: collect one first
}

int covahip_pipe_submit(covahip_pipe *p, int slot, int n_frames, int batch, int area_thresh) {
    if (!p || slot < 0 || slot >= p->n_slots || p->slots[slot].state != 1) return COVAHIP_ERR_INVALID_ARG;
    if (batch <= 0 || batch > p->max_batch || n_frames < BN_T || n_frames > p->max_frames) return COVAHIP_ERR_INVALID_ARG;
    Slot &s = p->slots[slot];
    covahip_ctx *ctx = p->ctx;
    PIPE_CHECK(hipSetDevice(ctx->device));
    PIPE_CHECK(hipMemcpyAsync(s.d_frames, s.h_frames, (size_t)n_frames * p->frame_bytes, hipMemcpyHostToDevice, p->s_h2d));
    PIPE_CHECK(hipEventRecord(s.ev_in, p->s_h2d));
    // the kernels of this batch go to the ctx's next lane (internal.h, CtxLane): with two lanes the kernels of batch k+1 fill
    // the ramps and tails of batch k's launches
    LaneScope lane(ctx);
    if (!lane.ok()) return COVAHIP_ERR_HIP;
    PIPE_CHECK(hipStreamWaitEvent(ctx->stream, s.ev_in, 0));
    int32_t *d_counts = s.d_meta, *d_offsets = s.d_meta + p->max_batch;
    int rc = p->packed ? covahip_filter_forward_frames_packed(ctx, reinterpret_cast<const uint16_t *>(s.d_frames), n_frames, s.h_index, batch,
                                                              area_thresh, s.d_boxes, d_counts, p->max_boxes, nullptr, s.d_mask)
                       : covahip_filter_forward_frames(ctx, s.d_frames, n_frames, s.h_index, batch, area_thresh, s.d_boxes, d_counts,
                                                       p->max_boxes, nullptr, s.d_mask, COVAHIP_MEM_DEVICE);
    if (rc) return rc;
    {
        ProfScope ps(ctx, "pack_boxes");
        const bool self = batch <= PACK_SELF_SCAN;
        if (!self) hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const int32_t *)d_counts, batch, p->max_boxes, d_offsets);
        hipLaunchKernelGGL(pack_kernel, dim3(batch), dim3(64), 0, ctx->stream, (const covahip_box *)s.d_boxes, (const int32_t *)d_counts, batch,
                           p->max_boxes, self ? (const int32_t *)nullptr : (const int32_t *)d_offsets, d_offsets, s.d_packed);
    }
    PIPE_CHECK(hipGetLastError());
    // the results leave on the LANE's own stream when the upload stream has a hardware queue without a lane (up to three lanes): a
    // separate result stream shares its queue with some lane, and its wait for THIS lane's kernels then holds that other lane's next
    // batch up (3 lanes x 3 slots: 122 vs 147 us per batch); with every queue taken by a lane (four lanes) the separate stream is
    // the better of two evils (120 vs 130)
    hipStream_t s_out = p->results_on_lane ? ctx->stream : p->s_d2h;
    if (!p->results_on_lane) {
        PIPE_CHECK(hipEventRecord(s.ev_done, ctx->stream));
        PIPE_CHECK(hipStreamWaitEvent(p->s_d2h, s.ev_done, 0));
    }
    // counts [batch], offsets [batch + 1] (max_batch apart) and the packed boxes behind them are one allocation: ONE copy brings the
    // meta words and the speculative part of the boxes
    s.spec = std::min(p->spec, batch * p->max_boxes);
    PIPE_CHECK(hipMemcpyAsync(s.h_meta, s.d_meta, p->meta_bytes + (size_t)s.spec * sizeof(covahip_box), hipMemcpyDeviceToHost, s_out));
    if (p->want_mask)
        PIPE_CHECK(hipMemcpyAsync(s.h_mask, s.d_mask, (size_t)batch * p->hw, hipMemcpyDeviceToHost, s_out));
    PIPE_CHECK(hipEventRecord(s.ev_out, s_out));
    s.batch = batch;
    s.n_frames = n_frames;
    s.state = 2;
    return COVAHIP_OK;
}

int covahip_pipe_wait(covahip_pipe *p, int slot) {
    if (!p || slot < 0 || slot >= p->n_slots || p->slots[slot].state != 2) return COVAHIP_ERR_INVALID_ARG;
    if (p->sleeping_wait) return wait_event_sleeping(p->slots[slot].ev_out);
    if (hipEventSynchronize(p->slots[slot].ev_out) != hipSuccess) return COVAHIP_ERR_HIP;
    return COVAHIP_OK;
}

int covahip_pipe_abort(covahip_pipe *p, int slot) {
    if (!p || slot < 0 || slot >= p->n_slots || p->slots[slot].state != 1) return COVAHIP_ERR_INVALID_ARG;
    // a submit that failed half way may have enqueued the slot's upload (and kernels that read its device buffers): nothing may
    // reuse the pinned frames before that work has drained
    if (hipStreamSynchronize(p->s_h2d) != hipSuccess) (void)hipGetLastError();
    covahip_sync_all(p->ctx);
    p->slots[slot].state = 0;
    return COVAHIP_OK;
}

int covahip_pipe_release(covahip_pipe *p, int slot) {
    if (!p || slot < 0 || slot >= p->n_slots || p->slots[slot].state != 3) return COVAHIP_ERR_INVALID_ARG;
    p->slots[slot].state = 0;
    return COVAHIP_OK;
}

int covahip_pipe_collect(covahip_pipe *p, int slot, const int32_t **counts, const int32_t **offsets, const covahip_box **boxes,
                         const uint8_t **mask) {
    if (!p || slot < 0 || slot >= p->n_slots || p->slots[slot].state != 2) return COVAHIP_ERR_INVALID_ARG;
    Slot &s = p->slots[slot];
    PIPE_CHECK(hipSetDevice(p->ctx->device));
    if (p->sleeping_wait) {
        if (wait_event_sleeping(s.ev_out) != COVAHIP_OK) { p->ctx->last_hip_error = "covahip_pipe_collect: hipEventQuery failed"; return COVAHIP_ERR_HIP; }
    } else {
        PIPE_CHECK(hipEventSynchronize(s.ev_out));
    }
    const int32_t *off = s.h_meta + p->max_batch;
    const int total = off[s.batch];
    if (total > s.spec) {   // more boxes than the pipelined copy carried: fetch the rest now, and copy more from the next batch on
        PIPE_CHECK(hipMemcpyAsync(s.h_packed + s.spec, s.d_packed + s.spec, (size_t)(total - s.spec) * sizeof(covahip_box),
                                  hipMemcpyDeviceToHost, p->s_d2h));
        PIPE_CHECK(hipStreamSynchronize(p->s_d2h));
        p->spec = std::min(p->max_batch * p->max_boxes, total + total / 4);
    }
    if (counts) *counts = s.h_meta;
    if (offsets) *offsets = off;
    if (boxes) *boxes = s.h_packed;
    if (mask) *mask = s.h_mask;
    s.state = 3;   // the results belong to the caller until covahip_pipe_release
    return COVAHIP_OK;
}

}  // extern "C"
