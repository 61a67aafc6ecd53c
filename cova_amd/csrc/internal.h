// Internal declarations shared by the translation units of libcovahip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "covahip.h"

struct covahip_blobnet;  // blobnet.hip

struct covahip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
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
    // bboxcc scratch for the fused path (mask / boxes / counts on device)
    void *cc_scratch = nullptr;
    size_t cc_scratch_bytes = 0;
    // bboxcc wave kernel: overflow list (count + frame indices) and the developer override of its run capacity
    void *cc_ovf = nullptr;
    size_t cc_ovf_bytes = 0;
    void *cc_slab = nullptr;   // bboxcc state of frames too large for LDS (one slab per resident workgroup)
    size_t cc_slab_bytes = 0;
    int cc_wave_cap = 0;
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

// bboxcc.hip
int covahip_bboxcc_launch(covahip_ctx *ctx, const uint8_t *d_mask, int batch, int h, int w, int area_thresh,
                          covahip_box *d_boxes, int32_t *d_counts, int max_boxes);

// blobnet.hip
void covahip_blobnet_destroy(covahip_ctx *ctx);
int covahip_blobnet_forward_dev(covahip_ctx *ctx, const uint8_t *d_stack, int batch, float *d_logits,
                                uint8_t *d_mask);
int covahip_blobnet_geometry(covahip_ctx *ctx, int *h, int *w);
