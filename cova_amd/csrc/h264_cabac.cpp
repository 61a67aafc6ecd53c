// Entropy-decode front end, macroblock layer: CABAC parsing of slice_data() (ITU-T H.264 7.3.4, 7.3.5, 9.3) for
// frame-coded 4:2:0 streams -- what the reference's patched FFmpeg `avdec_h264` does before it writes its
// [mb_type, mv_x, mv_y, -] records (README.md:94-114; consumers metapreprocess/imp.rs:233,311-312,
// tfrecordsink/imp.rs:105-112).  Parsing and motion vector prediction; no reconstruction.
//
// What can be verified in this image, and is (tests/test_host_h264.py): every one of the 1,802 slices of the reference's
// demo/1m.mp4 (High@3.1, CABAC, 8x8 transform, P and B slices, one slice per picture) must decode exactly
// width_mbs * height_mbs macroblocks, see end_of_slice_flag = 1 at the last one and only there, and leave nothing but
// rbsp_trailing_bits behind.  A wrong context-initialisation value, context increment or binarisation desynchronises the
// arithmetic decoder within a few macroblocks and the slice cannot end there, so 6.5 million macroblocks ending on the
// bit are the check of the tables below (transcribed from the standard's tables 9-12 .. 9-23 for the context indices a
// frame-coded 4:2:0 stream uses, cabac_init_idc 0 only: the demo stream uses no other, and an unverifiable table is not
// shipped -- idc 1 / 2 return COVAHIP_ERR_UNSUPPORTED, as do field / MBAFF coding and 4:2:2 / 4:4:4).
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "h264_cabac.h"

namespace h264 {
namespace {

// ---------------------------------------------------------------------------------------------- tables (9.3.1.1, 9.3.3.2)
const uint8_t RANGE_LPS[64][4] = {
    {128, 176, 208, 240}, {128, 167, 197, 227}, {128, 158, 187, 216}, {123, 150, 178, 205}, {116, 142, 169, 195}, {111, 135, 160, 185},
    {105, 128, 152, 175}, {100, 122, 144, 166}, {95, 116, 137, 158},  {90, 110, 130, 150},  {85, 104, 123, 142},  {81, 99, 117, 135},
    {77, 94, 111, 128},   {73, 89, 105, 122},   {69, 85, 100, 116},   {66, 80, 95, 110},    {62, 76, 90, 104},    {59, 72, 86, 99},
    {56, 69, 81, 94},     {53, 65, 77, 89},     {51, 62, 73, 85},     {48, 59, 69, 80},     {46, 56, 66, 76},     {43, 53, 63, 72},
    {41, 50, 59, 69},     {39, 48, 56, 65},     {37, 45, 54, 62},     {35, 43, 51, 59},     {33, 41, 48, 56},     {32, 39, 46, 53},
    {30, 37, 43, 50},     {29, 35, 41, 48},     {27, 33, 39, 45},     {26, 31, 37, 43},     {24, 30, 35, 41},     {23, 28, 33, 39},
    {22, 27, 32, 37},     {21, 26, 30, 35},     {20, 24, 29, 33},     {19, 23, 27, 31},     {18, 22, 26, 30},     {17, 21, 25, 28},
    {16, 20, 23, 27},     {15, 19, 22, 25},     {14, 18, 21, 24},     {14, 17, 20, 23},     {13, 16, 19, 22},     {12, 15, 18, 21},
    {12, 14, 17, 20},     {11, 14, 16, 19},     {11, 13, 15, 18},     {10, 12, 15, 17},     {10, 12, 14, 16},     {9, 11, 13, 15},
    {9, 11, 12, 14},      {8, 10, 12, 14},      {8, 9, 11, 13},       {7, 9, 11, 12},       {7, 9, 10, 12},       {7, 8, 10, 11},
    {6, 8, 9, 11},        {6, 7, 9, 10},        {6, 7, 8, 9},         {2, 2, 2, 2}};
const uint8_t TRANS_LPS[64] = {0,  0,  1,  2,  2,  4,  4,  5,  6,  7,  8,  9,  9,  11, 11, 12, 13, 13, 15, 15, 16, 16,
                               18, 18, 19, 19, 21, 21, 22, 22, 23, 24, 24, 25, 26, 26, 27, 27, 28, 29, 29, 30, 30, 30,
                               31, 32, 32, 33, 33, 33, 34, 34, 35, 35, 35, 36, 36, 36, 37, 37, 37, 38, 38, 63};

struct MN { int8_t m, n; };
struct Range { int first; std::vector<MN> v; };

// (m, n) of tables 9-12 .. 9-23, I slices.  Context indices not listed are not used by I slices of a frame-coded 4:2:0 stream.
const Range INIT_I[] = {
    {0, {{20, -15}, {2, 54}, {3, 74}, {20, -15}, {2, 54}, {3, 74}, {-28, 127}, {-23, 104}, {-6, 53}, {-1, 54}, {7, 51}}},
    {60, {{0, 41}, {0, 63}, {0, 63}, {0, 63}, {-9, 83}, {4, 86}, {0, 97}, {-7, 72}, {13, 41}, {3, 62}}},
    {70, {{0, 11}, {1, 55}, {0, 69}, {-17, 127}, {-13, 102}, {0, 82}, {-7, 74}, {-21, 107}, {-27, 127}, {-31, 127}, {-24, 127}, {-18, 95},
          {-27, 127}, {-21, 114}, {-30, 127}, {-17, 123}, {-12, 115}, {-16, 122},
          {-11, 115}, {-12, 63}, {-2, 68}, {-15, 84}, {-13, 104}, {-3, 70}, {-8, 93}, {-10, 90}, {-30, 127}, {-1, 74}, {-6, 97}, {-7, 91},
          {-20, 127}, {-4, 56}, {-5, 82}, {-7, 76}, {-22, 125}}},
    {105, {{-7, 93}, {-11, 87}, {-3, 77}, {-5, 71}, {-4, 63}, {-4, 68}, {-12, 84}, {-7, 62}, {-7, 65}, {8, 61}, {5, 56}, {-2, 66},
           {1, 64}, {0, 61}, {-2, 78}, {1, 50}, {7, 52}, {10, 35}, {0, 44}, {11, 38}, {1, 45}, {0, 46}, {5, 44}, {31, 17},
           {1, 51}, {7, 50}, {28, 19}, {16, 33}, {14, 62}, {-13, 108}, {-15, 100},
           {-13, 101}, {-13, 91}, {-12, 94}, {-10, 88}, {-16, 84}, {-10, 86}, {-7, 83}, {-13, 87}, {-19, 94}, {1, 70}, {0, 72}, {-5, 74},
           {18, 59}, {-8, 102}, {-15, 100}, {0, 95}, {-4, 75}, {2, 72}, {-11, 75}, {-3, 71}, {15, 46}, {-13, 69}, {0, 62}, {0, 65},
           {21, 37}, {-15, 72}, {9, 57}, {16, 54}, {0, 62}, {12, 72}}},
    {166, {{24, 0}, {15, 9}, {8, 25}, {13, 18}, {15, 9}, {13, 19}, {10, 37}, {12, 18}, {6, 29}, {20, 33}, {15, 30}, {4, 45},
           {1, 58}, {0, 62}, {7, 61}, {12, 38}, {11, 45}, {15, 39}, {11, 42}, {13, 44}, {16, 45}, {12, 41}, {10, 49}, {30, 34},
           {18, 42}, {10, 55}, {17, 51}, {17, 46}, {0, 89}, {26, -19}, {22, -17},
           {26, -17}, {30, -25}, {28, -20}, {33, -23}, {37, -27}, {33, -23}, {40, -28}, {38, -17}, {33, -11}, {40, -15}, {41, -6}, {38, 1},
           {41, 17}, {30, -6}, {27, 3}, {26, 22}, {37, -16}, {35, -4}, {38, -8}, {38, -3}, {37, 3}, {38, 5}, {42, 0}, {35, 16},
           {39, 22}, {14, 48}, {27, 37}, {21, 60}, {12, 68}, {2, 97}}},
    {227, {{-3, 71}, {-6, 42}, {-5, 50}, {-3, 54}, {-2, 62}, {0, 58}, {1, 63}, {-2, 72}, {-1, 74}, {-9, 91}, {-5, 67}, {-5, 27},
           {-3, 39}, {-2, 44}, {0, 46}, {-16, 64}, {-8, 68}, {-10, 78}, {-6, 77}, {-10, 86}, {-12, 92}, {-15, 55}, {-10, 60}, {-6, 62},
           {-4, 65},
           {-12, 73}, {-8, 76}, {-7, 80}, {-9, 88}, {-17, 110}, {-11, 97}, {-20, 84}, {-11, 79}, {-6, 73}, {-4, 74}, {-13, 86}, {-13, 96},
           {-11, 97}, {-19, 117}, {-8, 78}, {-5, 33}, {-4, 48}, {-2, 53}, {-3, 62}, {-13, 71}, {-10, 79}, {-12, 86}, {-13, 90}, {-14, 97}}},
    {399, {{31, 21}, {31, 31}, {25, 50},
           {-17, 120}, {-20, 112}, {-18, 114}, {-11, 85}, {-15, 92}, {-14, 89}, {-26, 71}, {-15, 81}, {-14, 80}, {0, 68}, {-14, 70}, {-24, 56},
           {-23, 68}, {-24, 50}, {-11, 74}, {23, -13}, {26, -13}, {40, -15}, {49, -14}, {44, 3}, {45, 6}, {44, 34}, {33, 54}, {19, 82},
           {-3, 75}, {-1, 23}, {1, 34}, {1, 43}, {0, 54}, {-2, 55}, {0, 61}, {1, 64}, {0, 68}, {-9, 92}}},
};
// P and B slices, cabac_init_idc = 0
const Range INIT_PB0[] = {
    {0, {{20, -15}, {2, 54}, {3, 74}, {20, -15}, {2, 54}, {3, 74}, {-28, 127}, {-23, 104}, {-6, 53}, {-1, 54}, {7, 51},
         {23, 33}, {23, 2}, {21, 0}, {1, 9}, {0, 49}, {-37, 118}, {5, 57}, {-13, 78}, {-11, 65}, {1, 62}, {12, 49}, {-4, 73}, {17, 50},
         {18, 64}, {9, 43}, {29, 0}, {26, 67}, {16, 90}, {9, 104}, {-46, 127}, {-20, 104}, {1, 67}, {-13, 78}, {-11, 65}, {1, 62},
         {-6, 86}, {-17, 95}, {-6, 61}, {9, 45},
         {-3, 69}, {-6, 81}, {-11, 96}, {6, 55}, {7, 67}, {-5, 86}, {2, 88}, {0, 58}, {-3, 76}, {-10, 94}, {5, 54}, {4, 69}, {-3, 81}, {0, 88},
         {-7, 67}, {-5, 74}, {-4, 74}, {-5, 80}, {-7, 72}, {1, 58},
         {0, 41}, {0, 63}, {0, 63}, {0, 63}, {-9, 83}, {4, 86}, {0, 97}, {-7, 72}, {13, 41}, {3, 62},
         {0, 45}, {-4, 78}, {-3, 96}, {-27, 126}, {-28, 98}, {-25, 101}, {-23, 67}, {-28, 82}, {-20, 94}, {-16, 83}, {-22, 110}, {-21, 91},
         {-18, 102}, {-13, 93}, {-29, 127}, {-7, 92}, {-5, 89}, {-7, 96}, {-13, 108}, {-3, 46}, {-1, 65}, {-1, 57}, {-9, 93}, {-3, 74},
         {-9, 92}, {-8, 87}, {-23, 126}, {5, 54}, {6, 60}, {6, 59}, {6, 69}, {-1, 48}, {0, 68}, {-4, 69}, {-8, 88}}},
    {105, {{-2, 85}, {-6, 78}, {-1, 75}, {-7, 77}, {2, 54}, {5, 50}, {-3, 68}, {1, 50}, {6, 42}, {-4, 81}, {1, 63}, {-4, 70},
           {0, 67}, {2, 57}, {-2, 76}, {11, 35}, {4, 64}, {1, 61}, {11, 35}, {18, 25}, {12, 24}, {13, 29}, {13, 36}, {-10, 93},
           {-7, 73}, {-2, 73}, {13, 46}, {9, 49}, {-7, 100}, {9, 53}, {2, 53}, {5, 53}, {-2, 61}, {0, 56}, {0, 56}, {-13, 63},
           {-5, 60}, {-1, 62}, {4, 57}, {-6, 69}, {4, 57}, {14, 39}, {4, 51}, {13, 68}, {3, 64}, {1, 61}, {9, 63}, {7, 50},
           {16, 39}, {5, 44}, {4, 52}, {11, 48}, {-5, 60}, {-1, 59}, {0, 59}, {22, 33}, {5, 44}, {14, 43}, {-1, 78}, {0, 60},
           {9, 69}}},
    {166, {{11, 28}, {2, 40}, {3, 44}, {0, 49}, {0, 46}, {2, 44}, {2, 51}, {0, 47}, {4, 39}, {2, 62}, {6, 46}, {0, 54},
           {3, 54}, {2, 58}, {4, 63}, {6, 51}, {6, 57}, {7, 53}, {6, 52}, {6, 55}, {11, 45}, {14, 36}, {8, 53}, {-1, 82},
           {7, 55}, {-3, 78}, {15, 46}, {22, 31}, {-1, 84}, {25, 7}, {30, -7}, {28, 3}, {28, 4}, {32, 0}, {34, -1}, {30, 6},
           {30, 6}, {32, 9}, {31, 19}, {26, 27}, {26, 30}, {37, 20}, {28, 34}, {17, 70}, {1, 67}, {5, 59}, {9, 67}, {16, 30},
           {18, 32}, {18, 35}, {22, 29}, {24, 31}, {23, 38}, {18, 43}, {20, 41}, {11, 63}, {9, 59}, {9, 64}, {-1, 94}, {-2, 89},
           {-9, 108}}},
    {227, {{-6, 76}, {-2, 44}, {0, 45}, {0, 52}, {-3, 64}, {-2, 59}, {-4, 70}, {-4, 75}, {-8, 82}, {-17, 102}, {-9, 77}, {3, 24},
           {0, 42}, {0, 48}, {0, 55}, {-6, 59}, {-7, 71}, {-12, 83}, {-11, 87}, {-30, 119}, {1, 58}, {-3, 29}, {-1, 36}, {1, 38},
           {2, 43}, {-6, 55}, {0, 58}, {0, 64}, {-3, 74}, {-10, 90}, {0, 70}, {-4, 29}, {5, 31}, {7, 42}, {1, 59}, {-2, 58},
           {-3, 72}, {-3, 81}, {-11, 97}, {0, 58}, {8, 5}, {10, 14}, {14, 18}, {13, 27}, {2, 40}, {0, 58}, {-3, 70}, {-6, 79},
           {-8, 85}}},
    {399, {{12, 40}, {11, 51}, {14, 59},
           {-4, 79}, {-7, 71}, {-5, 69}, {-9, 70}, {-8, 66}, {-10, 68}, {-19, 73}, {-12, 69}, {-16, 70}, {-15, 67}, {-20, 62}, {-19, 70},
           {-16, 66}, {-22, 65}, {-20, 63}, {9, -2}, {26, -9}, {33, -9}, {39, -7}, {41, -2}, {45, 3}, {49, 9}, {45, 27}, {36, 59},
           {-6, 66}, {-7, 35}, {-7, 42}, {-8, 45}, {-5, 48}, {-12, 56}, {-6, 60}, {-5, 62}, {-8, 66}, {-8, 76}}},
};

// ctxIdxInc of significant_coeff_flag / last_significant_coeff_flag for 8x8 blocks, frame coded (table 9-43)
const uint8_t SIG8[63] = {0, 1, 2, 3, 4, 5, 5, 4, 4, 3, 3, 4, 4, 4, 5, 5, 4, 4, 4, 4, 3, 3, 6, 7, 7, 7, 8, 9, 10, 9, 8, 7,
                          7, 6, 11, 12, 13, 11, 6, 7, 8, 9, 14, 10, 9, 8, 6, 11, 12, 13, 11, 6, 9, 14, 10, 9, 11, 12, 13, 11, 14, 10, 12};
const uint8_t LAST8[63] = {0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2,
                           3, 3, 3, 3, 3, 3, 3, 3, 4, 4, 4, 4, 4, 4, 4, 4, 5, 5, 5, 5, 6, 6, 6, 6, 7, 7, 7, 7, 8, 8, 8};
// ctxBlockCat: 0 luma DC of Intra16x16, 1 luma AC of Intra16x16, 2 luma 4x4, 3 chroma DC, 4 chroma AC, 5 luma 8x8
const int CBF_BASE[5] = {85, 89, 93, 97, 101};
const int SIG_BASE[6] = {105, 120, 134, 149, 152, 402};
const int LAST_BASE[6] = {166, 181, 195, 210, 213, 417};
const int ABS_BASE[6] = {227, 237, 247, 257, 266, 426};
const int MAX_COEFF[6] = {16, 15, 16, 4, 15, 64};

// state byte (pStateIdx << 1 | valMPS) after the more ([0]) / less ([1]) probable symbol (9.3.3.2.1.1, table 9-45)
struct NextState {
    uint8_t t[2][128];
    NextState() {
        for (int s = 0; s < 128; s++) {
            const int st = s >> 1, mps = s & 1;
            t[0][s] = (uint8_t)(((st < 62 ? st + 1 : 62) << 1) | mps);
            t[1][s] = (uint8_t)((TRANS_LPS[st] << 1) | (st == 0 ? (mps ^ 1) : mps));
        }
    }
    const uint8_t *operator[](int k) const { return t[k]; }
};
const NextState NEXT_STATE;
// ---------------------------------------------------------------------------------------------- arithmetic decoder (9.3.3.2)
struct Cabac {
    const uint8_t *p = nullptr;
    size_t nbits = 0;     // always whole bytes
    size_t next = 0;      // the next byte the reservoir takes
    uint64_t res = 0;     // its low `nres` bits are the stream bits not consumed yet, the first of them highest
    int nres = 0;
    uint32_t range = 510, offset = 0;
    uint8_t state[460];   // pStateIdx << 1 | valMPS
    size_t pos() const { return next * 8 - (size_t)nres; }      // bits consumed so far
    bool overrun() const { return pos() > nbits; }              // zeros are read behind the end; this tells
    void refill() {
        const size_t nbytes = nbits >> 3;
        if (nres <= 32 && next + 4 <= nbytes) {
            res = (res << 32) | ((uint32_t)p[next] << 24 | (uint32_t)p[next + 1] << 16 | (uint32_t)p[next + 2] << 8 | p[next + 3]);
            next += 4;
            nres += 32;
            return;
        }
        while (nres <= 56) {
            res = (res << 8) | (next < nbytes ? p[next] : 0);
            next++;
            nres += 8;
        }
    }
    void seek(size_t bitpos) {
        next = bitpos >> 3;
        res = 0;
        nres = 0;
        if (bitpos & 7) { refill(); nres -= (int)(bitpos & 7); }
    }
    // n <= 9 bits at once (renormalisation after a less probable symbol shifts up to 7 bits in)
    uint32_t bits(int n) {
        if (nres < n) refill();
        nres -= n;
        return (uint32_t)(res >> nres) & ((1u << n) - 1);
    }
    uint32_t bit() { return bits(1); }
    void start() {
        range = 510;
        offset = bits(9);
    }
    // DecodeDecision (9.3.3.2.1) without data-dependent branches: which of the two symbols comes is close to a coin toss for
    // many contexts, and a mispredicted branch costs more than the arithmetic of both sides
    int decision(int ctx) {
        const uint32_t s = state[ctx];
        const uint32_t lps = RANGE_LPS[s >> 1][(range >> 6) & 3];
        range -= lps;
        const uint32_t less = (uint32_t)-(int32_t)(offset >= range);   // all ones: the less probable symbol
        offset -= range & less;
        range += (lps - range) & less;
        state[ctx] = NEXT_STATE[less & 1][s];
        const int sh = __builtin_clz(range) - 23;   // 0 .. 7: range back into [256, 510]
        range <<= sh;
        offset = (offset << sh) | bits(sh);
        return (int)((s ^ less) & 1);
    }
    int bypass() {
        offset = (offset << 1) | bit();
        if (offset >= range) { offset -= range; return 1; }
        return 0;
    }
    int terminate() {
        range -= 2;
        if (offset >= range) return 1;   // no renormalisation; the last bit read so far is the encoder's closing 1 (9.3.4.5)
        while (range < 256) {
            range <<= 1;
            offset = (offset << 1) | bit();
        }
        return 0;
    }
};

void init_contexts(Cabac &c, const Range *tab, size_t n_ranges, int qp) {
    std::memset(c.state, 0, sizeof c.state);
    qp = std::min(51, std::max(0, qp));
    for (size_t r = 0; r < n_ranges; r++)
        for (size_t k = 0; k < tab[r].v.size(); k++) {
            const MN mn = tab[r].v[k];
            int pre = ((mn.m * qp) >> 4) + mn.n;
            pre = std::min(126, std::max(1, pre));
            c.state[tab[r].first + (int)k] = pre <= 63 ? (uint8_t)((63 - pre) << 1) : (uint8_t)(((pre - 64) << 1) | 1);
        }
}

// ---------------------------------------------------------------------------------------------- per-macroblock state the contexts look at
enum : uint8_t { K_NONE = 0, K_SKIP, K_INTER, K_DIRECT16, K_INXN, K_I16, K_PCM };
struct Mb {
    uint8_t kind = K_NONE;      // K_NONE: not decoded yet in this slice (unavailable)
    uint8_t t8x8 = 0;           // transform_size_8x8_flag
    uint8_t cbp = 0;            // luma bits 0..3, chroma value in bits 4..5
    uint8_t chroma_mode = 0;    // intra_chroma_pred_mode
    uint16_t nz_luma = 0;       // bit (y * 4 + x): the 4x4 block has coefficients (coded_block_flag, or its 8x8 block is coded)
    uint8_t nz_cb = 0, nz_cr = 0;   // bit (y * 2 + x): chroma AC block coded
    uint8_t dc = 0;             // bit 0 luma DC of Intra16x16, bit 1 Cb DC, bit 2 Cr DC coded
    int8_t ref[2][4];           // ref_idx per list and 8x8 block as PARSED (0 for inferred / unused, see direct8)
    uint8_t direct8 = 0;        // bit b8: the 8x8 block is predicted in direct mode (B_Skip, B_Direct_16x16, direct sub-block)
    uint8_t mvd[2][16][2];      // |mvd| per list, 4x4 block, component, clipped to 70
    // motion as predicted + coded (8.4.1): reference index per list and 8x8 block (-1: list not used / intra) and the motion
    // vector of every 4x4 block in quarter pixels
    int8_t aref[2][4];
    int16_t mv[2][16][2];
};

struct SliceCtx {
    const SliceParams &sp;
    Cabac c;
    Mb *mbs;                    // the calling thread's scratch (one slice at a time per thread): only `kind` is reset per slice
    int W, H;
    int last_dqp_nonzero = 0;
    std::string why;
    explicit SliceCtx(const SliceParams &s) : sp(s), W(s.width_mbs), H(s.height_mbs) {
        static thread_local std::vector<Mb> scratch;
        const size_t n = (size_t)W * H;
        if (scratch.size() < n) scratch.resize(n);
        mbs = scratch.data();
        for (size_t i = 0; i < n; i++) mbs[i].kind = K_NONE;
    }
    const Mb *left(int x, int y) const { return x > 0 && mbs[(size_t)y * W + x - 1].kind != K_NONE ? &mbs[(size_t)y * W + x - 1] : nullptr; }
    const Mb *top(int x, int y) const { return y > 0 && mbs[(size_t)(y - 1) * W + x].kind != K_NONE ? &mbs[(size_t)(y - 1) * W + x] : nullptr; }
};

bool is_intra(uint8_t k) { return k == K_INXN || k == K_I16 || k == K_PCM; }

// ---- motion vector prediction (8.4.1.3; neighbours 6.4.11.7)
inline int blk_order(int x, int y) { return ((y >> 1) * 2 + (x >> 1)) * 4 + (y & 1) * 2 + (x & 1); }   // decoding order of the 4x4 blocks
struct Nb { bool avail; int ref, mx, my; };
// 4x4 block (bx, by) relative to macroblock (mbx, mby), bx / by in -1 .. 4; blocks of the current macroblock count only when they
// come before `cur_order` in decoding order
Nb nb_block(const SliceCtx &s, int list, int mbx, int mby, int bx, int by, int cur_order) {
    int nx = mbx, ny = mby;
    if (by < 0) { ny--; by = 3; if (bx < 0) { nx--; bx = 3; } else if (bx > 3) { nx++; bx = 0; } }
    else if (bx < 0) { nx--; bx = 3; }
    else if (bx > 3) return Nb{false, -1, 0, 0};   // the macroblock to the right is not decoded yet
    else if (blk_order(bx, by) >= cur_order) return Nb{false, -1, 0, 0};
    if (nx < 0 || ny < 0 || nx >= s.W) return Nb{false, -1, 0, 0};
    const Mb &n = s.mbs[(size_t)ny * s.W + nx];
    if (n.kind == K_NONE) return Nb{false, -1, 0, 0};
    const int r = n.aref[list][(by >> 1) * 2 + (bx >> 1)];
    if (r < 0) return Nb{true, -1, 0, 0};
    return Nb{true, r, n.mv[list][by * 4 + bx][0], n.mv[list][by * 4 + bx][1]};
}
inline int median3(int a, int b, int c) { return std::max(std::min(a, b), std::min(std::max(a, b), c)); }
// shape: 0 none, 1 / 2 upper / lower 16x8 partition, 3 / 4 left / right 8x16 partition
void predict_mv(const SliceCtx &s, int list, int mbx, int mby, int x, int y, int w, int ref, int shape, int cur_order, int &px, int &py) {
    const Nb A = nb_block(s, list, mbx, mby, x - 1, y, cur_order);
    Nb B = nb_block(s, list, mbx, mby, x, y - 1, cur_order);
    Nb C = nb_block(s, list, mbx, mby, x + w, y - 1, cur_order);
    if (!C.avail) C = nb_block(s, list, mbx, mby, x - 1, y - 1, cur_order);
    if (shape == 1 && B.ref == ref) { px = B.mx; py = B.my; return; }
    if (shape == 2 && A.ref == ref) { px = A.mx; py = A.my; return; }
    if (shape == 3 && A.ref == ref) { px = A.mx; py = A.my; return; }
    if (shape == 4 && C.ref == ref) { px = C.mx; py = C.my; return; }
    if (!B.avail && !C.avail && A.avail) { px = A.mx; py = A.my; return; }
    const int n = (A.ref == ref) + (B.ref == ref) + (C.ref == ref);
    if (n == 1) {
        const Nb &m = A.ref == ref ? A : (B.ref == ref ? B : C);
        px = m.mx; py = m.my;
        return;
    }
    px = median3(A.mx, B.mx, C.mx);
    py = median3(A.my, B.my, C.my);
}
inline void fill_mv(Mb &m, int list, int x, int y, int w, int h, int mx, int my) {
    for (int yy = y; yy < y + h; yy++)
        for (int xx = x; xx < x + w; xx++) { m.mv[list][yy * 4 + xx][0] = (int16_t)mx; m.mv[list][yy * 4 + xx][1] = (int16_t)my; }
}
// The neighbours of a whole macroblock (a 16x16 partition: skip and direct prediction): A = block 3 of the left macroblock,
// B = block 12 of the one above, C = block 12 of the one above to the right or, where that one is missing, D = block 15 of the
// one above to the left -- nb_block / predict_mv for (x, y, w) = (0, 0, 4) without the per-block address arithmetic
struct MbNb { const Mb *a, *b, *c; int cblk; };
inline MbNb mb_neighbours(const SliceCtx &s, int mbx, int mby) {
    MbNb n{nullptr, nullptr, nullptr, 12};
    const Mb *row = s.mbs + (size_t)mby * s.W;
    if (mbx > 0 && row[mbx - 1].kind != K_NONE) n.a = &row[mbx - 1];
    if (mby > 0) {
        const Mb *up = row - s.W;
        if (up[mbx].kind != K_NONE) n.b = &up[mbx];
        if (mbx + 1 < s.W && up[mbx + 1].kind != K_NONE) n.c = &up[mbx + 1];
        else if (mbx > 0 && up[mbx - 1].kind != K_NONE) { n.c = &up[mbx - 1]; n.cblk = 15; }
    }
    return n;
}
inline Nb nb_of(const Mb *m, int list, int blk) {
    if (!m) return Nb{false, -1, 0, 0};
    const int r = m->aref[list][((blk >> 3) << 1) | ((blk & 3) >> 1)];
    if (r < 0) return Nb{true, -1, 0, 0};
    return Nb{true, r, m->mv[list][blk][0], m->mv[list][blk][1]};
}
inline void predict_16x16(const Nb &A, const Nb &B, const Nb &C, int ref, int &px, int &py) {
    if (!B.avail && !C.avail && A.avail) { px = A.mx; py = A.my; return; }
    const int n = (A.ref == ref) + (B.ref == ref) + (C.ref == ref);
    if (n == 1) {
        const Nb &m = A.ref == ref ? A : (B.ref == ref ? B : C);
        px = m.mx; py = m.my;
        return;
    }
    px = median3(A.mx, B.mx, C.mx);
    py = median3(A.my, B.my, C.my);
}
inline void fill_mv_all(Mb &m, int list, int mx, int my) {
    for (int b = 0; b < 16; b++) { m.mv[list][b][0] = (int16_t)mx; m.mv[list][b][1] = (int16_t)my; }
}
// P_Skip (8.4.1.1)
void p_skip_motion(const SliceCtx &s, int mbx, int mby, Mb &m) {
    const MbNb nb = mb_neighbours(s, mbx, mby);
    const Nb A = nb_of(nb.a, 0, 3), B = nb_of(nb.b, 0, 12);
    int mx = 0, my = 0;
    if (A.avail && B.avail && !(A.ref == 0 && A.mx == 0 && A.my == 0) && !(B.ref == 0 && B.mx == 0 && B.my == 0))
        predict_16x16(A, B, nb_of(nb.c, 0, nb.cblk), 0, mx, my);
    for (int b8 = 0; b8 < 4; b8++) { m.aref[0][b8] = 0; m.aref[1][b8] = -1; }
    fill_mv_all(m, 0, mx, my);
}
// B_Skip / B_Direct_16x16 / direct sub-blocks, spatial direct mode (8.4.1.2.2): reference indices and predicted vectors from the
// macroblock's own neighbours; direct_fill applies them per 8x8 block with the colZeroFlag test.  (Temporal direct slices: temporal_fill;
// they come here only when the caller has no co-located motion for them.)
void b_direct_motion(const SliceCtx &s, int mbx, int mby, int ref_out[2], int mv_out[2][2]) {
    const MbNb nb = mb_neighbours(s, mbx, mby);
    Nb A[2], B[2], C[2];
    for (int list = 0; list < 2; list++) {
        A[list] = nb_of(nb.a, list, 3);
        B[list] = nb_of(nb.b, list, 12);
        C[list] = nb_of(nb.c, list, nb.cblk);
        auto minpos = [](int a, int b) { return (a >= 0 && b >= 0) ? std::min(a, b) : std::max(a, b); };
        ref_out[list] = minpos(A[list].ref, minpos(B[list].ref, C[list].ref));
    }
    mv_out[0][0] = mv_out[0][1] = mv_out[1][0] = mv_out[1][1] = 0;
    if (ref_out[0] < 0 && ref_out[1] < 0) { ref_out[0] = ref_out[1] = 0; return; }
    for (int list = 0; list < 2; list++)
        if (ref_out[list] >= 0) predict_16x16(A[list], B[list], C[list], ref_out[list], mv_out[list][0], mv_out[list][1]);
}

// Direct-mode motion of the 8x8 blocks in `mask` from the macroblock-level result of b_direct_motion: list X's vector is zero
// where its reference index is 0 and the co-located block of RefPicList1[0] does not move (colZeroFlag; with
// direct_8x8_inference_flag the co-located block of a quadrant is the macroblock's corner block of that quadrant)
void direct_fill(const SliceCtx &s, Mb &m, int addr, int mask, const int dr[2], const int dm[2][2]) {
    const unsigned still = (s.sp.col_still && s.sp.direct_spatial) ? s.sp.col_still[addr] : 0u;
    if (mask == 0xF && (still == 0 || (dr[0] != 0 && dr[1] != 0) || !(dm[0][0] | dm[0][1] | dm[1][0] | dm[1][1]))) {
        // nothing for the test to change (the usual case: a skipped macroblock of the background predicts a zero vector anyway)
        for (int list = 0; list < 2; list++) {
            for (int b8 = 0; b8 < 4; b8++) m.aref[list][b8] = (int8_t)dr[list];
            if (dr[list] >= 0) fill_mv_all(m, list, dm[list][0], dm[list][1]);
        }
        return;
    }
    for (int b8 = 0; b8 < 4; b8++) {
        if (!((mask >> b8) & 1)) continue;
        const int x0 = 2 * (b8 & 1), y0 = 2 * (b8 >> 1);
        for (int list = 0; list < 2; list++) m.aref[list][b8] = (int8_t)dr[list];
        for (int k = 0; k < 4; k++) {
            const int x = x0 + (k & 1), y = y0 + (k >> 1), blk = y * 4 + x;
            const int colblk = s.sp.direct_8x8_inference ? (y0 ? 12 : 0) + (x0 ? 3 : 0) : blk;
            const bool cz = (still >> colblk) & 1;
            for (int list = 0; list < 2; list++) {
                if (dr[list] < 0) continue;
                const bool zero = dr[list] == 0 && cz;
                m.mv[list][blk][0] = (int16_t)(zero ? 0 : dm[list][0]);
                m.mv[list][blk][1] = (int16_t)(zero ? 0 : dm[list][1]);
            }
        }
    }
}
// Temporal direct prediction (8.4.1.2.3) of the 8x8 blocks in `mask`: reference index 0 of list 1, the list-0 index of the picture
// the co-located block refers to, the co-located vector scaled by the ratio of the picture order count distances.
bool temporal_direct(const SliceCtx &s) {
    return !s.sp.direct_spatial && s.sp.direct_8x8_inference && s.sp.col_motion && s.sp.col_to_l0 && s.sp.dist_scale;
}
void temporal_fill(const SliceCtx &s, Mb &m, int addr, int mask) {
    const ColMb &col = s.sp.col_motion[addr];
    for (int b8 = 0; b8 < 4; b8++) {
        if (!((mask >> b8) & 1)) continue;
        int ref0 = 0, mv0[2] = {0, 0}, mv1[2] = {0, 0};
        if (col.ref[b8] >= 0) {
            const int mapped = s.sp.col_to_l0[(col.list[b8] & 1) * 32 + (col.ref[b8] & 31)];
            ref0 = mapped < 0 ? 0 : mapped;
            const int dsf = s.sp.dist_scale[ref0 & 31];
            for (int k = 0; k < 2; k++) {
                const int mc = col.mv[b8][k];
                if (dsf == DIST_SCALE_NONE) { mv0[k] = mc; mv1[k] = 0; }
                else { mv0[k] = (dsf * mc + 128) >> 8; mv1[k] = mv0[k] - mc; }
            }
        }
        m.aref[0][b8] = (int8_t)ref0;
        m.aref[1][b8] = 0;
        const int x0 = 2 * (b8 & 1), y0 = 2 * (b8 >> 1);
        for (int k = 0; k < 4; k++) {
            const int blk = (y0 + (k >> 1)) * 4 + x0 + (k & 1);
            m.mv[0][blk][0] = (int16_t)mv0[0]; m.mv[0][blk][1] = (int16_t)mv0[1];
            m.mv[1][blk][0] = (int16_t)mv1[0]; m.mv[1][blk][1] = (int16_t)mv1[1];
        }
    }
}
// this macroblock's entry of SliceParams::motion_out
ColMb col_motion_of(const Mb &m) {
    ColMb c;
    for (int b8 = 0; b8 < 4; b8++) {
        const int blk = (b8 >> 1 ? 12 : 0) + (b8 & 1 ? 3 : 0);   // the quadrant's corner block
        const int list = is_intra(m.kind) ? -1 : (m.aref[0][b8] >= 0 ? 0 : (m.aref[1][b8] >= 0 ? 1 : -1));
        c.ref[b8] = (int8_t)(list < 0 ? -1 : m.aref[list][b8]);
        c.list[b8] = (uint8_t)(list < 0 ? 0 : list);
        c.mv[b8][0] = list < 0 ? 0 : m.mv[list][blk][0];
        c.mv[b8][1] = list < 0 ? 0 : m.mv[list][blk][1];
    }
    return c;
}

// "does not move" bits of a decoded macroblock (SliceParams::still_out)
uint16_t still_bits(const Mb &m) {
    if (is_intra(m.kind)) return 0;
    uint16_t bits = 0;
    for (int b8 = 0; b8 < 4; b8++) {
        const int list = m.aref[0][b8] >= 0 ? 0 : (m.aref[1][b8] >= 0 ? 1 : -1);
        if (list < 0 || m.aref[list][b8] != 0) continue;
        const int x0 = 2 * (b8 & 1), y0 = 2 * (b8 >> 1);
        for (int k = 0; k < 4; k++) {
            const int blk = (y0 + (k >> 1)) * 4 + x0 + (k & 1);
            // both components in -1 .. 1
            if ((unsigned)(m.mv[list][blk][0] + 1) <= 2u && (unsigned)(m.mv[list][blk][1] + 1) <= 2u) bits |= (uint16_t)(1u << blk);
        }
    }
    return bits;
}

// ---- mb_type of an intra macroblock (ffmpeg's numbering: 0 I_NxN, 1..24 I_16x16, 25 I_PCM); base 3 in I slices (prefix
// with neighbour context), 17 in P, 32 in B (9.3.2.5, 9.3.3.1.1.3)
int intra_mb_type(SliceCtx &s, int base, bool intra_slice, const Mb *A, const Mb *B) {
    Cabac &c = s.c;
    int st = base;
    if (intra_slice) {
        const int inc = (A && A->kind != K_INXN) + (B && B->kind != K_INXN);
        if (!c.decision(base + inc)) return 0;
        st = base + 2;
    } else if (!c.decision(base)) {
        return 0;
    }
    if (c.terminate()) return 25;
    int t = 1;
    t += 12 * c.decision(st + 1);
    if (c.decision(st + 2)) t += 4 + 4 * c.decision(st + 2 + (intra_slice ? 1 : 0));
    t += 2 * c.decision(st + 3 + (intra_slice ? 1 : 0));
    t += c.decision(st + 3 + (intra_slice ? 2 : 0));
    return t;
}

int p_sub_type(Cabac &c) {
    if (c.decision(21)) return 0;
    if (!c.decision(22)) return 1;
    return c.decision(23) ? 2 : 3;
}
int b_sub_type(Cabac &c) {
    if (!c.decision(36)) return 0;
    if (!c.decision(37)) return 1 + c.decision(39);
    int t = 3;
    if (c.decision(38)) {
        if (c.decision(39)) return 11 + c.decision(39);
        t += 4;
    }
    t += 2 * c.decision(39);
    t += c.decision(39);
    return t;
}
// sub_mb_type of B slices: lists used (bit 0 L0, bit 1 L1; 0 = direct) and number of sub-partitions with their 4x4 geometry
const uint8_t B_SUB_LISTS[13] = {0, 1, 2, 3, 1, 1, 2, 2, 3, 3, 1, 2, 3};
const uint8_t B_SUB_SHAPE[13] = {0, 0, 0, 0, 1, 2, 1, 2, 1, 2, 3, 3, 3};   // 0 8x8, 1 8x4, 2 4x8, 3 4x4
// mb_type of B slices 1..21: lists of partition 0 / 1 (bit 0 L0, bit 1 L1) and shape (0 16x16, 1 16x8, 2 8x16) (table 7-14)
const uint8_t B_MB_L0[23] = {0, 1, 2, 3, 1, 1, 2, 2, 1, 1, 2, 2, 1, 1, 2, 2, 3, 3, 3, 3, 3, 3, 0};
const uint8_t B_MB_L1[23] = {0, 0, 0, 0, 1, 1, 2, 2, 2, 2, 1, 1, 3, 3, 3, 3, 1, 1, 2, 2, 3, 3, 0};
const uint8_t B_MB_SHAPE[23] = {0, 0, 0, 0, 1, 2, 1, 2, 1, 2, 1, 2, 1, 2, 1, 2, 1, 2, 1, 2, 1, 2, 3};

// ---- residual_block_cabac (7.3.5.3.3, 9.3.2.x, 9.3.3.1.3): returns the number of coefficients, -1 on a runaway escape code
int residual_block(SliceCtx &s, int cat, int cbf_ctx_inc, bool has_cbf) {
    Cabac &c = s.c;
    if (has_cbf && !c.decision(CBF_BASE[cat] + cbf_ctx_inc)) return 0;
    const int maxc = MAX_COEFF[cat];
    int n = 0;
    int i = 0;
    for (; i < maxc - 1; i++) {
        const int si = cat == 5 ? SIG8[i] : (cat == 3 ? std::min(i, 2) : i);
        if (c.decision(SIG_BASE[cat] + si)) {
            n++;
            const int li = cat == 5 ? LAST8[i] : (cat == 3 ? std::min(i, 2) : i);
            if (c.decision(LAST_BASE[cat] + li)) break;
        }
    }
    if (i == maxc - 1) n++;   // the last position is significant by inference
    static const uint8_t LEVEL1_CTX[8] = {1, 2, 3, 4, 0, 0, 0, 0}, GT1_CTX[8] = {5, 5, 5, 5, 6, 7, 8, 9};
    static const uint8_t TRANS[2][8] = {{1, 2, 3, 3, 4, 5, 6, 7}, {4, 4, 4, 4, 5, 6, 7, 7}};
    int node = 0;
    for (int k = 0; k < n; k++) {
        if (!c.decision(ABS_BASE[cat] + LEVEL1_CTX[node])) {
            node = TRANS[0][node];
        } else {
            const int ctx = ABS_BASE[cat] + GT1_CTX[node];
            node = TRANS[1][node];
            int a = 2;
            while (a < 15 && c.decision(ctx)) a++;
            if (a >= 15) {
                int j = 0;
                while (c.bypass()) {
                    if (++j > 24) return -1;
                }
                while (j--) c.bypass();
            }
        }
        c.bypass();   // sign
    }
    return n;
}

int mvd_component(SliceCtx &s, int base, int sum, int &abs_out) {
    Cabac &c = s.c;
    if (!c.decision(base + (sum < 3 ? 0 : (sum > 32 ? 2 : 1)))) { abs_out = 0; return 0; }
    int v = 1, ctx = base + 3;
    while (v < 9 && c.decision(ctx)) {
        if (v < 4) ctx++;
        v++;
    }
    if (v >= 9) {
        int k = 3;
        while (c.bypass()) {
            v += 1 << k;
            if (++k > 24) { abs_out = -1; return 0; }
        }
        while (k--) v += c.bypass() << k;
    }
    abs_out = v;
    return c.bypass() ? -v : v;
}

int ref_idx(SliceCtx &s, int refa, bool dira, int refb, bool dirb, bool bslice, int num_ref) {
    Cabac &c = s.c;
    int ctx = 0;
    if (refa > 0 && !(bslice && dira)) ctx++;
    if (refb > 0 && !(bslice && dirb)) ctx += 2;
    int ref = 0;
    while (c.decision(54 + ctx)) {
        ref++;
        ctx = (ctx >> 2) + 4;
        if (ref >= 32) return -1;
    }
    (void)num_ref;
    return ref;
}

// value of neighbour 4x4 luma block's "has coefficients" for coded_block_flag contexts (9.3.3.1.1.9)
int nz_of(const Mb *n, bool cur_intra, int bit) {
    if (!n) return cur_intra ? 1 : 0;
    if (n->kind == K_PCM) return 1;
    return (n->nz_luma >> bit) & 1;
}
int nzc_of(const Mb *n, bool cur_intra, int comp, int bit) {
    if (!n) return cur_intra ? 1 : 0;
    if (n->kind == K_PCM) return 1;
    return ((comp ? n->nz_cr : n->nz_cb) >> bit) & 1;
}
int dc_of(const Mb *n, bool cur_intra, int bit) {
    if (!n) return cur_intra ? 1 : 0;
    if (n->kind == K_PCM) return 1;
    return (n->dc >> bit) & 1;
}

struct MbOut { uint8_t cls; int mvd_max[2]; };

void mb_motion_record(const Mb &m, int &ax, int &ay) {   // mean motion vector of the macroblock (list 0 where it is used, else list 1)
    int sx = 0, sy = 0, n = 0;
    for (int b = 0; b < 16; b++) {
        const int b8 = (b >> 3) * 2 + ((b & 3) >> 1);
        const int list = m.aref[0][b8] >= 0 ? 0 : (m.aref[1][b8] >= 0 ? 1 : -1);
        if (list < 0) continue;
        sx += m.mv[list][b][0]; sy += m.mv[list][b][1]; n++;
    }
    ax = n ? std::abs(sx) / n : 0;
    ay = n ? std::abs(sy) / n : 0;
}

// One macroblock_layer() (7.3.5) that is not skipped.  Fills `m`; returns false on a syntax error.
bool macroblock(SliceCtx &s, int mbx, int mby, Mb &m, MbOut &out) {
    Cabac &c = s.c;
    const SliceParams &sp = s.sp;
    const Mb *A = s.left(mbx, mby), *B = s.top(mbx, mby);
    const bool bslice = sp.slice_type == 1;
    int itype = -1;          // intra mb_type (0..25) or -1
    int shape = 0;           // inter: 0 16x16, 1 16x8, 2 8x16, 3 8x8
    int plist[2] = {1, 1};   // lists of partition 0 / 1 (bit 0 L0, bit 1 L1)
    bool direct16 = false;
    if (sp.slice_type == 2) {
        itype = intra_mb_type(s, 3, true, A, B);
    } else if (sp.slice_type == 0) {
        if (!c.decision(14)) {
            if (!c.decision(15)) shape = c.decision(16) ? 3 : 0;
            else shape = c.decision(17) ? 1 : 2;
        } else {
            itype = intra_mb_type(s, 17, false, A, B);
        }
    } else {
        const int inc = (A && A->kind != K_SKIP && A->kind != K_DIRECT16) + (B && B->kind != K_SKIP && B->kind != K_DIRECT16);
        // (a B_Skip neighbour is stored as K_SKIP: both count as "direct")
        int t;
        if (!c.decision(27 + inc)) t = 0;
        else if (!c.decision(27 + 3)) t = 1 + c.decision(27 + 5);
        else {
            int bits = c.decision(27 + 4) << 3;
            bits |= c.decision(27 + 5) << 2;
            bits |= c.decision(27 + 5) << 1;
            bits |= c.decision(27 + 5);
            if (bits < 8) t = bits + 3;
            else if (bits == 13) t = -1;
            else if (bits == 14) t = 11;
            else if (bits == 15) t = 22;
            else t = ((bits << 1) | c.decision(27 + 5)) - 4;
        }
        if (t < 0) itype = intra_mb_type(s, 32, false, A, B);
        else if (t == 0) direct16 = true;
        else {
            if (t > 22) { s.why = "B mb_type out of range"; return false; }
            shape = B_MB_SHAPE[t];
            plist[0] = B_MB_L0[t];
            plist[1] = B_MB_L1[t];
        }
    }

    std::memset(m.ref, 0, sizeof m.ref);
    std::memset(m.mvd, 0, sizeof m.mvd);
    std::memset(m.aref, -1, sizeof m.aref);
    std::memset(m.mv, 0, sizeof m.mv);
    m.direct8 = 0; m.t8x8 = 0; m.cbp = 0; m.chroma_mode = 0; m.nz_luma = 0; m.nz_cb = m.nz_cr = 0; m.dc = 0;
    out.mvd_max[0] = out.mvd_max[1] = 0;
    bool dct8_ok = sp.transform_8x8 != 0;   // transform_size_8x8_flag may follow the coded_block_pattern

    if (itype == 25) {
        // I_PCM: pcm_alignment_zero_bits, 384 sample bytes, a fresh arithmetic decoder
        m.kind = K_PCM;
        out.cls = 7;
        // (the decoder's register has already taken the encoder's closing 1; the rest of the byte is pcm_alignment_zero_bits)
        const size_t behind = ((c.pos() + 7) & ~(size_t)7) + 384 * 8;
        if (behind > c.nbits) { s.why = "I_PCM: samples beyond the slice"; return false; }
        c.seek(behind);
        c.start();
        m.cbp = 0x2F; m.nz_luma = 0xFFFF; m.nz_cb = m.nz_cr = 0xF; m.dc = 7;
        s.last_dqp_nonzero = 0;
        return true;
    }
    if (itype >= 0) {
        // ---- mb_pred for intra macroblocks (7.3.5.1)
        if (itype == 0) {
            m.kind = K_INXN;
            out.cls = 5;
            if (sp.transform_8x8) m.t8x8 = (uint8_t)c.decision(399 + (A && A->t8x8) + (B && B->t8x8));
            const int n = m.t8x8 ? 4 : 16;
            for (int k = 0; k < n; k++)
                if (!c.decision(68)) { c.decision(69); c.decision(69); c.decision(69); }   // rem_intra4x4_pred_mode / rem_intra8x8_pred_mode
            dct8_ok = false;
        } else {
            m.kind = K_I16;
            out.cls = 6;
            const int t = itype - 1;
            m.cbp = (uint8_t)((t >= 12 ? 15 : 0) | (((t >> 2) % 3) << 4));
            dct8_ok = false;
        }
        {   // intra_chroma_pred_mode
            const int inc = (A && is_intra(A->kind) && A->chroma_mode != 0) + (B && is_intra(B->kind) && B->chroma_mode != 0);
            int v = 0;
            if (c.decision(64 + inc)) {
                v = 1;
                if (c.decision(64 + 3)) { v = 2; if (c.decision(64 + 3)) v = 3; }
            }
            m.chroma_mode = (uint8_t)v;
        }
    } else if (direct16) {
        m.kind = K_DIRECT16;
        out.cls = 4;
        m.direct8 = 0xF;
        {
            int dr[2], dm[2][2];
            if (temporal_direct(s)) {
                temporal_fill(s, m, mby * s.W + mbx, 0xF);
            } else {
                b_direct_motion(s, mbx, mby, dr, dm);
                direct_fill(s, m, mby * s.W + mbx, 0xF, dr, dm);
            }
        }
        dct8_ok = dct8_ok && sp.direct_8x8_inference;
    } else {
        m.kind = K_INTER;
        out.cls = shape == 0 ? 1 : (shape == 3 ? 3 : 2);
        // geometry: per partition its 4x4 origin and size, its 8x8 blocks, its lists
        struct Part { int x, y, w, h, lists; };
        Part parts[16];
        int n_parts = 0;
        int sub_shape[4] = {0, 0, 0, 0}, sub_lists[4] = {1, 1, 1, 1};
        if (shape == 3) {
            for (int b8 = 0; b8 < 4; b8++) {
                if (bslice) {
                    const int t = b_sub_type(c);
                    sub_lists[b8] = B_SUB_LISTS[t];
                    sub_shape[b8] = B_SUB_SHAPE[t];
                    if (t == 0) {
                        m.direct8 |= (uint8_t)(1 << b8);
                        if (!sp.direct_8x8_inference) dct8_ok = false;
                    } else if (sub_shape[b8] != 0) dct8_ok = false;
                } else {
                    sub_shape[b8] = p_sub_type(c);
                    if (sub_shape[b8] != 0) dct8_ok = false;
                }
            }
        }
        if (m.direct8) {   // direct sub-blocks: the macroblock's own neighbours decide (8.4.1.2.2), before anything else of this macroblock
            int dr[2], dm[2][2];
            if (temporal_direct(s)) {
                temporal_fill(s, m, mby * s.W + mbx, m.direct8);
            } else {
                b_direct_motion(s, mbx, mby, dr, dm);
                direct_fill(s, m, mby * s.W + mbx, m.direct8, dr, dm);
            }
        }
        const int nref[2] = {sp.num_ref_l0, sp.num_ref_l1};
        // neighbour lookups on the 4x4 grid, across the macroblock border
        auto ref_at = [&](int list, int bx, int by, bool &dir) -> int {   // 4x4 coordinates, may be -1 (left / top macroblock)
            const Mb *n = &m;
            if (bx < 0) { n = A; bx = 3; } else if (by < 0) { n = B; by = 3; }
            dir = false;
            if (!n || n->kind == K_NONE || is_intra(n->kind)) return -1;
            if (n->kind == K_SKIP) { dir = bslice; return 0; }
            const int b8 = (by >> 1) * 2 + (bx >> 1);
            dir = (n->direct8 >> b8) & 1;
            return n->ref[list][b8];
        };
        auto mvd_at = [&](int list, int bx, int by, int comp) -> int {
            const Mb *n = &m;
            if (bx < 0) { n = A; bx = 3; } else if (by < 0) { n = B; by = 3; }
            if (!n || n->kind == K_NONE) return 0;
            return n->mvd[list][by * 4 + bx][comp];
        };
        // ---- ref_idx_l0 of every partition, then ref_idx_l1 (7.3.5.1, 7.3.5.2)
        for (int list = 0; list < 2; list++) {
            if (list == 1 && !bslice) break;
            const int n8 = shape == 0 ? 1 : (shape == 3 ? 4 : 2);
            for (int p = 0; p < n8; p++) {
                int x4, y4, w8, h8, lists;
                if (shape == 0) { x4 = 0; y4 = 0; w8 = 2; h8 = 2; lists = plist[0]; }
                else if (shape == 1) { x4 = 0; y4 = 2 * p; w8 = 2; h8 = 1; lists = plist[p]; }
                else if (shape == 2) { x4 = 2 * p; y4 = 0; w8 = 1; h8 = 2; lists = plist[p]; }
                else { x4 = 2 * (p & 1); y4 = 2 * (p >> 1); w8 = 1; h8 = 1; lists = sub_lists[p]; }
                int r = 0;
                if ((lists >> list) & 1) {
                    if (nref[list] > 1) {
                        bool da, db;
                        const int ra = ref_at(list, x4 - 1, y4, da), rb = ref_at(list, x4, y4 - 1, db);
                        r = ref_idx(s, ra, da, rb, db, bslice, nref[list]);
                        if (r < 0 || r >= nref[list]) { s.why = "ref_idx out of range"; return false; }
                    }
                }
                for (int yy = 0; yy < h8; yy++)
                    for (int xx = 0; xx < w8; xx++) {
                        const int b8i = ((y4 >> 1) + yy) * 2 + (x4 >> 1) + xx;
                        m.ref[list][b8i] = (int8_t)r;
                        if ((lists >> list) & 1) m.aref[list][b8i] = (int8_t)r;
                    }
            }
        }
        // ---- mvd_l0 of every (sub-)partition, then mvd_l1
        for (int list = 0; list < 2; list++) {
            if (list == 1 && !bslice) break;
            n_parts = 0;
            if (shape == 0) parts[n_parts++] = Part{0, 0, 4, 4, plist[0]};
            else if (shape == 1) { parts[n_parts++] = Part{0, 0, 4, 2, plist[0]}; parts[n_parts++] = Part{0, 2, 4, 2, plist[1]}; }
            else if (shape == 2) { parts[n_parts++] = Part{0, 0, 2, 4, plist[0]}; parts[n_parts++] = Part{2, 0, 2, 4, plist[1]}; }
            else
                for (int b8 = 0; b8 < 4; b8++) {
                    const int x0 = 2 * (b8 & 1), y0 = 2 * (b8 >> 1), l = sub_lists[b8];
                    if ((m.direct8 >> b8) & 1) continue;
                    switch (sub_shape[b8]) {
                    case 0: parts[n_parts++] = Part{x0, y0, 2, 2, l}; break;
                    case 1: parts[n_parts++] = Part{x0, y0, 2, 1, l}; parts[n_parts++] = Part{x0, y0 + 1, 2, 1, l}; break;
                    case 2: parts[n_parts++] = Part{x0, y0, 1, 2, l}; parts[n_parts++] = Part{x0 + 1, y0, 1, 2, l}; break;
                    default:
                        for (int k = 0; k < 4; k++) parts[n_parts++] = Part{x0 + (k & 1), y0 + (k >> 1), 1, 1, l};
                    }
                }
            for (int pi = 0; pi < n_parts; pi++) {
                const Part &p = parts[pi];
                if (!((p.lists >> list) & 1)) continue;
                int ab[2], sd[2];
                for (int comp = 0; comp < 2; comp++) {
                    const int sum = mvd_at(list, p.x - 1, p.y, comp) + mvd_at(list, p.x, p.y - 1, comp);
                    sd[comp] = mvd_component(s, comp ? 47 : 40, sum, ab[comp]);
                    if (ab[comp] < 0) { s.why = "mvd escape code runs away"; return false; }
                    out.mvd_max[comp] = std::max(out.mvd_max[comp], ab[comp]);
                }
                for (int yy = 0; yy < p.h; yy++)
                    for (int xx = 0; xx < p.w; xx++) {
                        m.mvd[list][(p.y + yy) * 4 + p.x + xx][0] = (uint8_t)std::min(ab[0], 70);
                        m.mvd[list][(p.y + yy) * 4 + p.x + xx][1] = (uint8_t)std::min(ab[1], 70);
                    }
                {   // motion vector = prediction from the neighbours + the coded difference
                    int px, py;
                    const int hint = shape == 1 ? (p.y == 0 ? 1 : 2) : (shape == 2 ? (p.x == 0 ? 3 : 4) : 0);
                    predict_mv(s, list, mbx, mby, p.x, p.y, p.w, m.aref[list][(p.y >> 1) * 2 + (p.x >> 1)], hint, blk_order(p.x, p.y), px, py);
                    fill_mv(m, list, p.x, p.y, p.w, p.h, px + sd[0], py + sd[1]);
                }
            }
        }
    }

    // ---- coded_block_pattern (not for Intra16x16: it is part of mb_type) (9.3.2.6, 9.3.3.1.1.4)
    if (m.kind != K_I16) {
        auto luma_bits = [&](const Mb *n) -> int { return !n ? 15 : (n->kind == K_PCM ? 15 : (n->kind == K_SKIP ? 0 : (n->cbp & 15))); };
        auto chroma_val = [&](const Mb *n) -> int { return !n ? 0 : (n->kind == K_PCM ? 2 : (n->kind == K_SKIP ? 0 : (n->cbp >> 4))); };
        const int a = luma_bits(A), b = luma_bits(B);
        int cbp = 0;
        cbp |= c.decision(73 + !(a & 2) + 2 * !(b & 4));
        cbp |= c.decision(73 + !(cbp & 1) + 2 * !(b & 8)) << 1;
        cbp |= c.decision(73 + !(a & 8) + 2 * !(cbp & 1)) << 2;
        cbp |= c.decision(73 + !(cbp & 4) + 2 * !(cbp & 2)) << 3;
        const int ca = chroma_val(A), cb = chroma_val(B);
        int cc = 0;
        if (c.decision(77 + (ca > 0) + 2 * (cb > 0))) cc = 1 + c.decision(77 + 4 + (ca == 2) + 2 * (cb == 2));
        m.cbp = (uint8_t)(cbp | (cc << 4));
        if (dct8_ok && (cbp & 15) && !is_intra(m.kind)) m.t8x8 = (uint8_t)c.decision(399 + (A && A->t8x8) + (B && B->t8x8));
    }

    // ---- mb_qp_delta + residual (7.3.5.3)
    if (m.cbp || m.kind == K_I16) {
        {
            int ctx = s.last_dqp_nonzero ? 1 : 0, val = 0;
            while (c.decision(60 + ctx)) {
                ctx = 2 + (ctx >> 1);
                if (++val > 104) { s.why = "mb_qp_delta runs away"; return false; }
            }
            s.last_dqp_nonzero = val != 0;
        }
        const bool intra = is_intra(m.kind);
        if (m.kind == K_I16) {
            const int n = residual_block(s, 0, dc_of(A, true, 0) + 2 * dc_of(B, true, 0), true);
            if (n < 0) { s.why = "level escape code runs away"; return false; }
            if (n) m.dc |= 1;
        }
        for (int b8 = 0; b8 < 4; b8++) {
            if (!((m.cbp >> b8) & 1)) continue;
            if (m.t8x8) {
                const int n = residual_block(s, 5, 0, false);
                if (n < 0) { s.why = "level escape code runs away"; return false; }
                const int x0 = 2 * (b8 & 1), y0 = 2 * (b8 >> 1);
                for (int k = 0; k < 4; k++) m.nz_luma |= (uint16_t)(1u << ((y0 + (k >> 1)) * 4 + x0 + (k & 1)));
                continue;
            }
            for (int b4 = 0; b4 < 4; b4++) {
                const int x = 2 * (b8 & 1) + (b4 & 1), y = 2 * (b8 >> 1) + (b4 >> 1);
                const int na = x > 0 ? (m.nz_luma >> (y * 4 + x - 1)) & 1 : nz_of(A, intra, y * 4 + 3);
                const int nb = y > 0 ? (m.nz_luma >> ((y - 1) * 4 + x)) & 1 : nz_of(B, intra, 12 + x);
                const int n = residual_block(s, m.kind == K_I16 ? 1 : 2, na + 2 * nb, true);
                if (n < 0) { s.why = "level escape code runs away"; return false; }
                if (n) m.nz_luma |= (uint16_t)(1u << (y * 4 + x));
            }
        }
        if (m.cbp >> 4) {
            for (int comp = 0; comp < 2; comp++) {
                const int n = residual_block(s, 3, dc_of(A, intra, 1 + comp) + 2 * dc_of(B, intra, 1 + comp), true);
                if (n < 0) { s.why = "level escape code runs away"; return false; }
                if (n) m.dc |= (uint8_t)(2 << comp);
            }
            if ((m.cbp >> 4) == 2)
                for (int comp = 0; comp < 2; comp++) {
                    uint8_t &nzc = comp ? m.nz_cr : m.nz_cb;
                    for (int k = 0; k < 4; k++) {
                        const int x = k & 1, y = k >> 1;
                        const int na = x > 0 ? (nzc >> (y * 2)) & 1 : nzc_of(A, intra, comp, y * 2 + 1);
                        const int nb = y > 0 ? (nzc >> x) & 1 : nzc_of(B, intra, comp, 2 + x);
                        const int n = residual_block(s, 4, na + 2 * nb, true);
                        if (n < 0) { s.why = "level escape code runs away"; return false; }
                        if (n) nzc |= (uint8_t)(1 << k);
                    }
                }
        }
    } else {
        s.last_dqp_nonzero = 0;
    }
    return true;
}

}  // namespace

int parse_slice_cabac(const uint8_t *rbsp, size_t len, size_t bit_offset, const SliceParams &sp, uint8_t *records, std::string *why) {
    auto fail = [&](int rc, const std::string &w) {
        if (why) *why = w;
        return rc;
    };
    if (sp.chroma_format != 1) return fail(COVAHIP_ERR_UNSUPPORTED, "chroma format other than 4:2:0");
    if (sp.slice_type != 2 && sp.cabac_init_idc != 0) return fail(COVAHIP_ERR_UNSUPPORTED, "cabac_init_idc 1 / 2: tables not shipped (not verifiable here)");
    if (sp.slice_type < 0 || sp.slice_type > 2) return fail(COVAHIP_ERR_UNSUPPORTED, "SP / SI slice");
    if (sp.first_mb != 0) return fail(COVAHIP_ERR_UNSUPPORTED, "several slices per picture");
    if (bit_offset & 7 || bit_offset / 8 >= len) return fail(COVAHIP_ERR_BAD_DATA, "slice data offset");
    SliceCtx s(sp);
    s.c.p = rbsp;
    s.c.nbits = len * 8;
    s.c.seek(bit_offset);
    if (sp.slice_type == 2) init_contexts(s.c, INIT_I, sizeof INIT_I / sizeof INIT_I[0], sp.qp);
    else init_contexts(s.c, INIT_PB0, sizeof INIT_PB0 / sizeof INIT_PB0[0], sp.qp);
    s.c.start();
    const int n_mbs = s.W * s.H;
    for (int addr = 0; addr < n_mbs; addr++) {
        const int x = addr % s.W, y = addr / s.W;
        Mb &m = s.mbs[addr];
        MbOut out{0, {0, 0}};
        bool skipped = false;
        if (sp.slice_type != 2) {
            const Mb *A = s.left(x, y), *B = s.top(x, y);
            const int inc = (A && A->kind != K_SKIP) + (B && B->kind != K_SKIP);
            skipped = s.c.decision((sp.slice_type == 0 ? 11 : 24) + inc) != 0;
        }
        if (skipped) {
            std::memset((void *)&m, 0, sizeof m);     // kind K_NONE: its own blocks are not neighbours of itself
            if (sp.slice_type == 0) {
                p_skip_motion(s, x, y, m);
            } else {
                int dr[2], dm[2][2];
                if (temporal_direct(s)) {
                    temporal_fill(s, m, addr, 0xF);
                } else {
                    b_direct_motion(s, x, y, dr, dm);
                    direct_fill(s, m, addr, 0xF, dr, dm);
                }
            }
            m.kind = K_SKIP;
            m.direct8 = sp.slice_type == 1 ? 0xF : 0;
            s.last_dqp_nonzero = 0;
            out.cls = 0;
        } else if (!macroblock(s, x, y, m, out)) {
            return fail(COVAHIP_ERR_BAD_DATA, "macroblock " + std::to_string(addr) + ": " + s.why);
        }
        if (s.c.overrun()) return fail(COVAHIP_ERR_BAD_DATA, "macroblock " + std::to_string(addr) + ": slice data exhausted");
        if (sp.motion_out) sp.motion_out[addr] = col_motion_of(m);
        if (sp.still_out) {
            if (skipped && sp.slice_type == 0)   // P_Skip: reference 0 and one vector for all sixteen blocks
                sp.still_out[addr] = ((unsigned)(m.mv[0][0][0] + 1) <= 2u && (unsigned)(m.mv[0][0][1] + 1) <= 2u) ? 0xFFFF : 0;
            else
                sp.still_out[addr] = still_bits(m);
        }
        if (records) {
            // [macroblock class, |mv_x|, |mv_y|, 0]: see h264_cabac.h for what these are and are not
            int ax, ay;
            mb_motion_record(m, ax, ay);
            records[4 * addr] = out.cls;
            records[4 * addr + 1] = (uint8_t)std::min(255, ax);
            records[4 * addr + 2] = (uint8_t)std::min(255, ay);
            records[4 * addr + 3] = 0;
        }
        const int end = s.c.terminate();
        if (end != (addr == n_mbs - 1)) return fail(COVAHIP_ERR_BAD_DATA, end ? "end_of_slice_flag at macroblock " + std::to_string(addr) + " of " + std::to_string(n_mbs)
                                                                                  : "no end_of_slice_flag at the last macroblock");
    }
    // rbsp_slice_trailing_bits: the closing 1 of the arithmetic codeword, zeros to the byte boundary, then only cabac_zero_words
    // rbsp_slice_trailing_bits.  The last bit the arithmetic decoder has read is the encoder's closing 1 = rbsp_stop_one_bit
    // (EncodeFlush writes ten bits behind the terminating bin, the decoder's nine-bit register has taken them all: 9.3.4.5);
    // what is left of that byte is alignment, then only cabac_zero_words.  (x264 puts a signature bit into the alignment
    // bits, so they are not required to be zero.)
    const size_t pos = s.c.pos();
    if (pos == 0 || pos > s.c.nbits || !((rbsp[(pos - 1) >> 3] >> (7 - ((pos - 1) & 7))) & 1))
        return fail(COVAHIP_ERR_BAD_DATA, "rbsp_stop_one_bit");
    for (size_t i = (pos + 7) / 8; i < len; i++)
        if (rbsp[i]) return fail(COVAHIP_ERR_BAD_DATA, "bytes behind the slice's trailing bits");
    return COVAHIP_OK;
}

}  // namespace h264
