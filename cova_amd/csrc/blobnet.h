// BlobNet on gfx950: model state shared by the kernels' translation units.
#pragma once
#include <hip/hip_fp16.h>

#include <vector>

#include "internal.h"

constexpr int BN_T = 4;       // timestep the network is built for (utils/train-blobnet.py:58)
constexpr int BN_LEVELS = 4;
constexpr int BN_KTAB_STACKS = 256;   // stacks whose table (16-bit frame indices, 2 KB) travels in enc1_mfma's kernel arguments

struct BnLevelGeom {
    int H, W;  // spatial size of the tensor at this level (level 0 = network input)
};

// HBM workspace of one lane (internal.h, CtxLane): everything a forward in flight writes.  Lane 0's also serves the calls
// that run on the ctx's primary stream.
struct BnWorkspace {
    bool ready = false;
    // activations (fp16, channels-last)
    __half *act[BN_LEVELS + 1] = {};  // act[i], i=1..3: [B][T][H_i][W_i][C_i]; act[4]: [B][H_4][W_4][128] (t=0)
    __half *dact[BN_LEVELS] = {};     // dact[j], j=0..2: [B][Hd][Wd][Cout_j]
    // carrier-frame path: pooled level-0 values per carrier frame, [frames][H_1][W_1][16] (pad row / column zero)
    __half *pbuf = nullptr;
    size_t pbuf_frames = 0;
    // the level-0 skip connection's share of the logits (round 5): fp32 [B][H_1 + 1][W_1 + 1][4], written by the level-1 kernel, added by the last
    // decoder block -- the skip tensor itself (act[1]) is then neither written nor read
    float *part = nullptr;
    // stack -> frame index table of the call in flight
    int32_t *d_index = nullptr;
    int32_t *h_index = nullptr;       // pinned host copy it is uploaded from
    size_t index_ints = 0;            // capacity of both
    hipEvent_t ev_index = nullptr;    // upload of h_index done (the next call may overwrite it)
    std::vector<int32_t> last_table;  // the table that is resident in d_index (an unchanged table is not uploaded again)
    int last_n_frames = 0;
};

struct covahip_blobnet {
    int H = 0, W = 0, max_batch = 0;
    BnLevelGeom lv[BN_LEVELS + 1];
    int enc_c[BN_LEVELS + 1] = {3, 16, 32, 64, 128};
    int dec_ci[BN_LEVELS] = {128, 128, 64, 32};
    int dec_co[BN_LEVELS] = {64, 32, 16, 16};
    int dec_cy[BN_LEVELS], dec_cx[BN_LEVELS];  // crop offsets (top/left) per decoder block
    BnWorkspace ws[COVAHIP_MAX_LANES];
    // prepared (MFMA path) weights
    void *d_prepared = nullptr;
    size_t prepared_bytes = 0;
    struct Prepared *prep = nullptr;
    int fuse_tail = 1;  // MFMA path, with bboxcc requested: last decoder block + bboxcc in one launch
    int fuse_dec = 1;  // MFMA path: decoder blocks 0..2 as one launch (a frame's three input tiles side by side in LDS) when they fit
    int enc1_tile16 = 1;  // MFMA path: level 1 on 16-position tiles (enc1_mfma) where the row fits its fixed LDS stride
    int tail_rows = 1;    // MFMA path: the fused tail on row tiles with ballots straight into bboxcc's planes (dec3cc_rows_mfma, round 6)
    int tail_part = 1;    // MFMA path: the last decoder block's skip half computed by the level-1 kernel as partial logits (enc1_mfma only)
    int fuse_enc23 = 1;   // MFMA path: encoder levels 2 + 3 in one launch (enc23_mfma: level 2's output stays in LDS as level 3's band) when they fit
                          // and the batch / geometry make it pay; 0: never, 2: whenever they fit
    int enc_rowtiles = 1; // MFMA path: levels 2 and 3 on row-aligned tiles (enc_mfma<.., TSZ>) where the geometry suits them
    int64_t macs_per_frame = 0;
};

// Whether encoder level 1 runs on enc1_mfma (16x16x32 tiles, fixed LDS row: grids of at most BN_E1_MAXW level-1 pixels) or on
// the round-1..3 kernel.  ONE predicate for both translation units: blobnet.hip hands the stack table over by value only to
// enc1_mfma, and blobnet_mfma.hip must then launch exactly that kernel.
constexpr int BN_E1_MAXW = 62;
inline bool bn_level1_on_enc1(const covahip_ctx *ctx, const covahip_blobnet *m) {
    return m->enc1_tile16 && m->enc_c[1] == 16 && m->enc_c[2] == 32 && m->lv[1].W <= BN_E1_MAXW && m->lv[1].H >= 2 &&
           !ctx->enc_plan[1].nbands;
}

// blobnet_mfma.hip
int blobnet_prepare_mfma(covahip_ctx *ctx, covahip_blobnet *m, const float *h_weights);
void blobnet_release_mfma(covahip_ctx *ctx, covahip_blobnet *m);
// cc != nullptr: bboxcc is wanted on the mask; *cc_done tells whether the forward already ran it (fused tail)
struct BnCcTail {
    int area_thresh, max_boxes;
    covahip_box *boxes;   // [batch][max_boxes]
    int32_t *counts;
};
// Input of a forward: either the stacked tensor [batch][T*H][W][4] (metapreprocess output), or carrier frames
// [n_frames][H][W][4] plus, per stack, the indices of its T = 0..3 slices (device pointers all).
// dry: planning only (every geometry / LDS check of the launch sequence, no kernel is launched).
struct BnInput {
    const uint8_t *stack = nullptr;
    const uint8_t *frames = nullptr;
    int n_frames = 0;
    const int32_t *index = nullptr;   // i32 [batch][4] on the device, or null when h_index is used / for the stacked entry
    // the same table on the host (validated): batches of at most BN_KTAB_STACKS stacks hand it to the level-1 kernel BY VALUE, in
    // its kernel arguments -- no copy in front of the kernels, nothing to cache or to order
    const int32_t *h_index = nullptr;
    bool packed = false;   // carrier frames as two-byte records (covahip_carrier_pack) instead of the decoder's four bytes
    bool dry = false;
};
int blobnet_forward_mfma(covahip_ctx *ctx, covahip_blobnet *m, BnWorkspace &ws, const BnInput &in, int batch, float *d_logits,
                         uint8_t *d_mask, const BnCcTail *cc = nullptr, bool *cc_done = nullptr);
