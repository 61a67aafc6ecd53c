// Macroblock-layer parsing of one CABAC slice (h264_cabac.cpp), used by covahip_h264_decode_records (h264_front.cpp).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>

#include "covahip.h"

namespace h264 {

// What temporal direct prediction (8.4.1.2.3) takes from a macroblock of RefPicList1[0], per 8x8 quadrant (the quadrant's corner
// 4x4 block: direct_8x8_inference_flag): the motion vector and reference index of the list the block uses (list 0 first; ref -1:
// intra) and which list that was.
struct ColMb {
    int16_t mv[4][2];
    int8_t ref[4];
    uint8_t list[4];
};
constexpr int16_t DIST_SCALE_NONE = 0x7FFF;   // dist_scale entry: take mvCol as it is, list 1 vector zero (long-term reference or equal counts)

struct SliceParams {
    int slice_type = 2;          // 0 P, 1 B, 2 I
    int first_mb = 0;
    int qp = 26;                 // SliceQPY
    int cabac_init_idc = 0;
    int num_ref_l0 = 1, num_ref_l1 = 1;
    int width_mbs = 0, height_mbs = 0;
    int transform_8x8 = 0;       // PPS transform_8x8_mode_flag
    int direct_8x8_inference = 0;
    int chroma_format = 1;
    int direct_spatial = 1;      // direct_spatial_mv_pred_flag (B slices); 0 = temporal direct (col_motion below)
    // colZeroFlag input (8.4.1.2.2): per macroblock of RefPicList1[0] one bit per 4x4 block (bit y * 4 + x) = "the block does not
    // move" (inter, reference index 0 of the list it uses -- list 0 first --, both vector components within +-1); NULL when the
    // slice is not B, RefPicList1[0] is a long-term picture or its motion is not known: the test then reads "moving" everywhere
    const uint16_t *col_still = nullptr;
    // the same bits of THIS picture, for the pictures that will have it as RefPicList1[0] (may be NULL)
    uint16_t *still_out = nullptr;
    // temporal direct (direct_spatial 0; needs direct_8x8_inference): the co-located picture's motion, per (list, index) of ITS
    // reference lists the index in THIS slice's list 0 of the same picture (-1: not in it), and DistScaleFactor per list-0 index
    // (8.4.1.2.3).  All NULL: temporal direct slices are predicted spatially, without the colZeroFlag test (what the tests then see).
    const ColMb *col_motion = nullptr;
    const int8_t *col_to_l0 = nullptr;     // [2][32]
    const int16_t *dist_scale = nullptr;   // [32]
    ColMb *motion_out = nullptr;           // THIS picture's, for later pictures (may be NULL)
};

// Parses slice_data() of one slice that covers a whole frame picture.  rbsp: the NAL unit's payload behind its header byte
// with the emulation-prevention bytes removed; bit_offset: first bit of slice_data() (covahip_h264_slice.data_bit_offset).
// records (may be NULL): width_mbs * height_mbs entries of 4 bytes,
//   [0] macroblock class: 0 P_Skip / B_Skip, 1 inter 16x16, 2 inter 16x8 / 8x16, 3 inter 8x8 (sub-partitions), 4 B_Direct_16x16,
//       5 intra NxN (4x4 / 8x8), 6 intra 16x16, 7 I_PCM;
//   [1], [2] |mean motion vector| of the macroblock's sixteen 4x4 blocks, x and y, in quarter pixels (<= 255); list 0 where the block
//       uses it, else list 1.  Motion vectors are the standard's: median / directional prediction from the neighbours (8.4.1.3) plus
//       the coded difference, P_Skip inference (8.4.1.1), spatial direct prediction for B_Skip / B_Direct (8.4.1.2.2) including
//       the colZeroFlag test when the caller supplies col_still, temporal direct prediction (8.4.1.2.3) when it supplies
//       col_motion / col_to_l0 / dist_scale (streams with direct_8x8_inference_flag);
//   [3] 0.
// Returns COVAHIP_OK only if exactly width_mbs * height_mbs macroblocks were decoded, end_of_slice_flag came with the last one and
// nothing but trailing bits followed.  *why (may be NULL) names the first inconsistency otherwise.
int parse_slice_cabac(const uint8_t *rbsp, size_t len, size_t bit_offset, const SliceParams &sp, uint8_t *records, std::string *why);

}  // namespace h264
