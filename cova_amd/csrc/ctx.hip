// GPU context, device memory helpers, HIP-event timers and per-kernel profiling for
// libcovahip.so (see include/covahip.h).
#include <cstdio>
#include <cstring>

#include "internal.h"

extern "C" {

const char *covahip_strerror(int status) {
    switch (status) {
        case COVAHIP_OK: return "ok";
        case COVAHIP_ERR_INVALID_ARG: return "invalid argument";
        case COVAHIP_ERR_NO_DEVICE: return "no usable HIP device";
        case COVAHIP_ERR_HIP: return "HIP runtime call failed";
        case COVAHIP_ERR_NOT_LOADED: return "BlobNet weights not loaded";
        case COVAHIP_ERR_UNSUPPORTED: return "unsupported geometry";
        case COVAHIP_ERR_BAD_WEIGHTS: return "bad weight blob";
        case COVAHIP_ERR_OVERFLOW: return "output buffer too small";
        case COVAHIP_ERR_BAD_DATA: return "malformed input data";
        default: return "unknown status";
    }
}

const char *covahip_version(void) { return "covahip 0.1.0 (gfx950, HIP)"; }

int covahip_device_count(int *count) {
    if (!count) return COVAHIP_ERR_INVALID_ARG;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        *count = 0;
        return COVAHIP_ERR_NO_DEVICE;
    }
    *count = n;
    return COVAHIP_OK;
}

int covahip_device_pci_bus_id(int device_id, char *out, int out_len) {
    if (!out || out_len < 13) return COVAHIP_ERR_INVALID_ARG;
    out[0] = 0;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device_id < 0 || device_id >= n) {
        (void)hipGetLastError();   // (HIP's last error is sticky per thread: a later hipGetLastError() check must not see this one)
        return COVAHIP_ERR_NO_DEVICE;
    }
    if (hipDeviceGetPCIBusId(out, out_len, device_id) != hipSuccess) {
        (void)hipGetLastError();
        out[0] = 0;
        return COVAHIP_ERR_NO_DEVICE;
    }
    return COVAHIP_OK;
}

int covahip_ctx_create(int device_id, covahip_ctx **out) {
    if (!out) return COVAHIP_ERR_INVALID_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return COVAHIP_ERR_NO_DEVICE;
    if (device_id < 0 || device_id >= n) return COVAHIP_ERR_INVALID_ARG;
    if (hipSetDevice(device_id) != hipSuccess) return COVAHIP_ERR_NO_DEVICE;
    covahip_ctx *ctx = new covahip_ctx();
    ctx->device = device_id;
    if (hipGetDeviceProperties(&ctx->props, device_id) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->primary, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return COVAHIP_ERR_NO_DEVICE;
    }
    ctx->stream = ctx->primary;
    bool ok = hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) == hipSuccess;
    for (int k = 0; k < COVAHIP_MAX_LANES && ok; k++)
        ok = hipStreamCreateWithFlags(&ctx->lanes[k].stream, hipStreamNonBlocking) == hipSuccess &&
             hipEventCreateWithFlags(&ctx->lanes[k].done, hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        covahip_ctx_destroy(ctx);
        return COVAHIP_ERR_NO_DEVICE;
    }
    for (int i = 0; i < 16; i++) {
        hipEventCreate(&ctx->t_start[i]);
        hipEventCreate(&ctx->t_stop[i]);
    }
    *out = ctx;
    return COVAHIP_OK;
}

void covahip_ctx_destroy(covahip_ctx *ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    covahip_sync_all(ctx);
    covahip_blobnet_destroy(ctx);
    for (int i = 0; i < 16; i++) {
        if (ctx->t_start[i]) hipEventDestroy(ctx->t_start[i]);
        if (ctx->t_stop[i]) hipEventDestroy(ctx->t_stop[i]);
    }
    for (auto &p : ctx->prof_pending) {
        hipEventDestroy(p.a);
        hipEventDestroy(p.b);
    }
    for (auto &p : ctx->prof_pool) {
        hipEventDestroy(p.first);
        hipEventDestroy(p.second);
    }
    if (ctx->stage_in) hipFree(ctx->stage_in);
    if (ctx->stage_out) hipFree(ctx->stage_out);
    for (CtxLane &l : ctx->lanes) {
        if (l.cc_scratch) hipFree(l.cc_scratch);
        if (l.cc_ovf) hipFree(l.cc_ovf);
        if (l.cc_slab) hipFree(l.cc_slab);
        if (l.cc_stat_ring) hipHostFree(l.cc_stat_ring);
        if (l.done) hipEventDestroy(l.done);
        if (l.stream) hipStreamDestroy(l.stream);
    }
    if (ctx->ev_fork) hipEventDestroy(ctx->ev_fork);
    if (ctx->pinned) hipHostFree(ctx->pinned);
    if (ctx->primary) hipStreamDestroy(ctx->primary);
    delete ctx;
}

int covahip_ctx_sync(covahip_ctx *ctx) {
    if (!ctx) return COVAHIP_ERR_INVALID_ARG;
    return covahip_sync_all(ctx);
}

int covahip_ctx_set_lanes(covahip_ctx *ctx, int n_lanes) {
    if (!ctx || n_lanes < 1 || n_lanes > COVAHIP_MAX_LANES || ctx->in_lane) return COVAHIP_ERR_INVALID_ARG;
    COVAHIP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    int rc = covahip_sync_all(ctx);
    if (rc) return rc;
    if (ctx->blobnet && n_lanes > ctx->n_lanes) {
        rc = covahip_blobnet_grow_lanes(ctx, n_lanes);
        if (rc) return rc;
    }
    ctx->n_lanes = n_lanes;
    ctx->next_lane = 0;
    return COVAHIP_OK;
}

int covahip_ctx_get_lanes(covahip_ctx *ctx, int *n_lanes) {
    if (!ctx || !n_lanes) return COVAHIP_ERR_INVALID_ARG;
    *n_lanes = ctx->n_lanes;
    return COVAHIP_OK;
}

const char *covahip_last_hip_error(covahip_ctx *ctx) { return ctx ? ctx->last_hip_error.c_str() : ""; }

int covahip_device_info(covahip_ctx *ctx, char *name, size_t name_cap, int *num_cu, size_t *hbm_bytes) {
    if (!ctx) return COVAHIP_ERR_INVALID_ARG;
    if (name && name_cap) {
        std::snprintf(name, name_cap, "%s (%s)", ctx->props.name, ctx->props.gcnArchName);
    }
    if (num_cu) *num_cu = ctx->props.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = ctx->props.totalGlobalMem;
    return COVAHIP_OK;
}

int covahip_malloc(covahip_ctx *ctx, size_t bytes, void **dev_ptr) {
    if (!ctx || !dev_ptr) return COVAHIP_ERR_INVALID_ARG;
    COVAHIP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    COVAHIP_CHECK_HIP(ctx, hipMalloc(dev_ptr, bytes ? bytes : 1));
    return COVAHIP_OK;
}

int covahip_free(covahip_ctx *ctx, void *dev_ptr) {
    if (!ctx) return COVAHIP_ERR_INVALID_ARG;
    if (!dev_ptr) return COVAHIP_OK;
    int rc = covahip_sync_all(ctx);
    if (rc) return rc;
    COVAHIP_CHECK_HIP(ctx, hipFree(dev_ptr));
    return COVAHIP_OK;
}

int covahip_memcpy_h2d(covahip_ctx *ctx, void *dev_dst, const void *host_src, size_t bytes) {
    if (!ctx || (!dev_dst && bytes) || (!host_src && bytes)) return COVAHIP_ERR_INVALID_ARG;
    if (int rc = covahip_primary_op(ctx)) return rc;
    COVAHIP_CHECK_HIP(ctx, hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    COVAHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return COVAHIP_OK;
}

int covahip_memcpy_d2h(covahip_ctx *ctx, void *host_dst, const void *dev_src, size_t bytes) {
    if (!ctx || (!host_dst && bytes) || (!dev_src && bytes)) return COVAHIP_ERR_INVALID_ARG;
    if (int rc = covahip_primary_op(ctx)) return rc;
    COVAHIP_CHECK_HIP(ctx, hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    COVAHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return COVAHIP_OK;
}

int covahip_memset(covahip_ctx *ctx, void *dev_ptr, int value, size_t bytes) {
    if (!ctx || (!dev_ptr && bytes)) return COVAHIP_ERR_INVALID_ARG;
    if (int rc = covahip_primary_op(ctx)) return rc;
    COVAHIP_CHECK_HIP(ctx, hipMemsetAsync(dev_ptr, value, bytes, ctx->stream));
    return COVAHIP_OK;
}

int covahip_timer_start(covahip_ctx *ctx, int slot) {
    if (!ctx || slot < 0 || slot >= 16) return COVAHIP_ERR_INVALID_ARG;
    if (int rc = covahip_primary_op(ctx)) return rc;
    COVAHIP_CHECK_HIP(ctx, hipEventRecord(ctx->t_start[slot], ctx->stream));
    return COVAHIP_OK;
}

int covahip_timer_stop(covahip_ctx *ctx, int slot) {
    if (!ctx || slot < 0 || slot >= 16) return COVAHIP_ERR_INVALID_ARG;
    if (int rc = covahip_primary_op(ctx)) return rc;
    COVAHIP_CHECK_HIP(ctx, hipEventRecord(ctx->t_stop[slot], ctx->stream));
    return COVAHIP_OK;
}

int covahip_timer_elapsed_ms(covahip_ctx *ctx, int slot, float *ms) {
    if (!ctx || !ms || slot < 0 || slot >= 16) return COVAHIP_ERR_INVALID_ARG;
    COVAHIP_CHECK_HIP(ctx, hipEventSynchronize(ctx->t_stop[slot]));
    COVAHIP_CHECK_HIP(ctx, hipEventElapsedTime(ms, ctx->t_start[slot], ctx->t_stop[slot]));
    return COVAHIP_OK;
}

int covahip_profile_enable(covahip_ctx *ctx, int on) {
    if (!ctx) return COVAHIP_ERR_INVALID_ARG;
    ctx->profile = on != 0;
    return COVAHIP_OK;
}

int covahip_profile_filter(covahip_ctx *ctx, const char *kernel_name) {
    if (!ctx) return COVAHIP_ERR_INVALID_ARG;
    ctx->profile_filter = kernel_name ? kernel_name : "";
    return COVAHIP_OK;
}

static void prof_drain(covahip_ctx *ctx) {
    for (auto &p : ctx->prof_pending) {
        float ms = 0.f;
        if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            auto &acc = ctx->prof_acc[p.name];
            acc.first += ms;
            acc.second += 1;
        }
        ctx->prof_pool.emplace_back(p.a, p.b);
    }
    ctx->prof_pending.clear();
}

int covahip_profile_reset(covahip_ctx *ctx) {
    if (!ctx) return COVAHIP_ERR_INVALID_ARG;
    prof_drain(ctx);
    ctx->prof_acc.clear();
    return COVAHIP_OK;
}

int covahip_profile_read(covahip_ctx *ctx, covahip_kernel_time *out, int cap, int *n) {
    if (!ctx || !n) return COVAHIP_ERR_INVALID_ARG;
    prof_drain(ctx);
    int i = 0;
    for (auto &kv : ctx->prof_acc) {
        if (out && i < cap) {
            std::memset(out[i].name, 0, sizeof(out[i].name));
            std::strncpy(out[i].name, kv.first.c_str(), sizeof(out[i].name) - 1);
            out[i].total_ms = kv.second.first;
            out[i].launches = kv.second.second;
        }
        i++;
    }
    *n = i;
    return COVAHIP_OK;
}

}  // extern "C"

ProfScope::ProfScope(covahip_ctx *c, const char *name) : ctx(c) {
    if (!ctx->profile) return;
    if (!ctx->profile_filter.empty() && ctx->profile_filter != name) return;
    covahip_ctx::ProfEntry e;
    e.name = name;
    if (!ctx->prof_pool.empty()) {
        e.a = ctx->prof_pool.back().first;
        e.b = ctx->prof_pool.back().second;
        ctx->prof_pool.pop_back();
    } else {
        hipEventCreate(&e.a);
        hipEventCreate(&e.b);
    }
    hipEventRecord(e.a, ctx->stream);
    ctx->prof_pending.push_back(e);
    idx = (int)ctx->prof_pending.size() - 1;
}

ProfScope::~ProfScope() {
    if (idx < 0) return;
    hipEventRecord(ctx->prof_pending[idx].b, ctx->stream);
    // keep the pending list bounded: drain when it grows large
    if (ctx->prof_pending.size() > 4096) {
        // inline drain (synchronises; only in profiling mode)
        for (auto &p : ctx->prof_pending) {
            float ms = 0.f;
            if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
                auto &acc = ctx->prof_acc[p.name];
                acc.first += ms;
                acc.second += 1;
            }
            ctx->prof_pool.emplace_back(p.a, p.b);
        }
        ctx->prof_pending.clear();
    }
}

// ---------------------------------------------------------------- lanes (see CtxLane in internal.h)
int covahip_primary_op(covahip_ctx *ctx) {
    if (ctx->in_lane) return COVAHIP_OK;   // inside a LaneScope `stream` is the lane's: nothing touches the primary stream
    for (int k = 0; k < COVAHIP_MAX_LANES; k++) {
        CtxLane &l = ctx->lanes[k];
        if (l.pending) {
            COVAHIP_CHECK_HIP(ctx, hipStreamWaitEvent(ctx->primary, l.done, 0));
            l.pending = false;
        }
    }
    ctx->primary_seq++;
    return COVAHIP_OK;
}

int covahip_sync_all(covahip_ctx *ctx) {
    for (int k = 0; k < COVAHIP_MAX_LANES; k++) {
        CtxLane &l = ctx->lanes[k];
        if (l.pending && l.stream) {
            COVAHIP_CHECK_HIP(ctx, hipStreamSynchronize(l.stream));
            l.pending = false;
        }
    }
    if (ctx->primary) COVAHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->primary));
    return COVAHIP_OK;
}

LaneScope::LaneScope(covahip_ctx *c) : ctx(c) {
    if (ctx->in_lane || ctx->n_lanes <= 1) {
        // one lane: the call runs on the primary stream, behind whatever an earlier multi-lane configuration left in flight
        if (!ctx->in_lane && covahip_primary_op(ctx)) failed = true;
        return;
    }
    const int k = ctx->next_lane;
    CtxLane &l = ctx->lanes[k];
    if (l.seen_seq != ctx->primary_seq) {
        // something was enqueued on the primary stream since this lane last looked: order the lane behind it
        hipError_t e = hipEventRecord(ctx->ev_fork, ctx->primary);
        if (e == hipSuccess) e = hipStreamWaitEvent(l.stream, ctx->ev_fork, 0);
        if (e != hipSuccess) {
            ctx->last_hip_error = std::string("lane fork: ") + hipGetErrorString(e);
            failed = true;
            return;
        }
        l.seen_seq = ctx->primary_seq;
    }
    ctx->next_lane = (k + 1) % ctx->n_lanes;
    ctx->cur_lane = k;
    ctx->stream = l.stream;
    ctx->in_lane = true;
    owner = true;
}

LaneScope::~LaneScope() {
    if (!owner) return;
    CtxLane &l = ctx->lanes[ctx->cur_lane];
    if (hipEventRecord(l.done, l.stream) == hipSuccess) l.pending = true;
    ctx->stream = ctx->primary;
    ctx->cur_lane = 0;
    ctx->in_lane = false;
}

// ---------------------------------------------------------------- shader clock under load (covahip_dev.h)
// One wave runs a dependent chain for about `busy_us` and brackets it with s_memtime (shader cycles) and s_memrealtime
// (100 MHz): MI355X_MICROARCH.md, DVFS give-back (6).  On a stream of its own, so it shares the chip with whatever the
// ctx has in flight -- that is the clock the caller wants.
__global__ __launch_bounds__(64) void clock_probe_kernel(unsigned long long *out, int iters) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float x = (float)threadIdx.x;
    for (int i = 0; i < iters; i++) {
        x = __builtin_fmaf(x, 1.0001f, 0.5f);
        asm volatile("" : "+v"(x));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        out[0] = t1 - t0;
        out[1] = r1 - r0;
    }
    if (x == 12345.678f) out[2] = 1;   // keeps the chain alive
}

extern "C" int covahip_dev_clock_mhz(covahip_ctx *ctx, int busy_us, float *mhz) {
    if (!ctx || !mhz || busy_us <= 0 || busy_us > 100000) return COVAHIP_ERR_INVALID_ARG;
    COVAHIP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = nullptr;
    unsigned long long *d = nullptr, h[3] = {0, 0, 0};
    COVAHIP_CHECK_HIP(ctx, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipError_t e = hipMalloc((void **)&d, sizeof(h));
    if (e == hipSuccess) {
        hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, st, d, busy_us * 400);   // ~5 cycles per dependent fma
        e = hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    if (d) hipFree(d);
    hipStreamDestroy(st);
    if (e != hipSuccess) {
        ctx->last_hip_error = std::string("clock probe: ") + hipGetErrorString(e);
        return COVAHIP_ERR_HIP;
    }
    *mhz = h[1] ? (float)((double)h[0] / (double)h[1] * 100.0) : 0.f;
    return COVAHIP_OK;
}

int covahip_ensure_buffer(covahip_ctx *ctx, void **buf, size_t *cur, size_t need) {
    if (*cur >= need && *buf) return COVAHIP_OK;
    if (*buf) {
        COVAHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        COVAHIP_CHECK_HIP(ctx, hipFree(*buf));
        *buf = nullptr;
        *cur = 0;
    }
    COVAHIP_CHECK_HIP(ctx, hipMalloc(buf, need));
    *cur = need;
    return COVAHIP_OK;
}
