// bboxcc of ONE mask frame by ONE wavefront, run based (bboxcc.hip has the overview).
//
// Reference semantics: regionprops() of cova-rs/gst-plugins/src/bboxcc/process.rs:5-49 (OpenCV
// connectedComponentsWithStats, 8-connectivity, label order of the 2x2-block scan, AREA >= threshold).
//
//   A  the frame's H x W mask bytes stream in as 8-byte pieces (64 lanes x 8 B = 512 contiguous bytes per wave
//      instruction) and are bit-packed into two PARITY PLANES per pixel row: E = pixels of even x, O = pixels of
//      odd x, so bit bx of a plane is the pixel of 2x2 block column bx.  With the two rows of a block row that is
//      four 64-bit words a, b / c, d = the four pixel positions of every block of the row.
//   B  lane by owns block row by.  Everything the scan needs is a handful of 64-bit bitwise operations:
//        F  = a|b|c|d                     block has foreground
//        J  = (a|c) & ((b|d) << 1)        block is 8-connected to its left neighbour block
//        S  = F & ~J                      first block of a horizontal RUN of connected blocks
//        cU = (a|b) & (ue|uo), cUL = a & (uo << 1), cUR = b & (ue >> 1)     connections to the three blocks above
//      (ue / uo = parity planes of the pixel row above).  Runs are numbered in block-raster order with a wave
//      prefix sum of popcount(S), so the smallest run id of a component is the run that holds the component's
//      first block in block-raster order -- the block where OpenCV's scan creates its first provisional label.
//   C  per run: extent with ctz, pixel count / extents with popcount / ctz on the run's bit mask, one min-root
//      union (LDS atomicMin) per (run, run above) pair that touches.
//   D  every run adds its statistics to its root's; roots in ascending id order with AREA >= threshold are the
//      boxes in OpenCV label order (wave ballot + prefix count, no workgroup barrier anywhere).
//
// LDS per wave: (2*BH + 2) * 16 B of parity planes + 24 B per run of capacity `cap`.  A frame with more runs than
// `cap` is not processed: its index goes to an overflow list and the workgroup-per-frame kernel labels it.
#pragma once
#include <cstdint>

#include "bboxcc_body.h"

#ifdef PHASE_TIMING
// developer build (tools/build_phase_lib.sh): s_memrealtime ticks per phase of frame_wg, thread 0 of every workgroup, summed
__device__ unsigned long long g_ccph[8];
#define CCWG_MARK(i)                                                        \
    do {                                                                    \
        if (tid == 0) {                                                     \
            const unsigned long long now_ = __builtin_amdgcn_s_memrealtime(); \
            atomicAdd(&g_ccph[i], now_ - cc_last_);                         \
            cc_last_ = now_;                                                \
        }                                                                   \
    } while (0)
#else
#define CCWG_MARK(i) do { } while (0)
#endif

namespace ccwave {

struct WvGeom {
    int H, W, BH, BW, NXB;   // NXB = W / 8 pieces per row
    int cap;                 // run capacity of a wave's LDS region
    int rows_bytes;          // (2*BH + 2) * 16
    int wave_bytes;          // LDS bytes per wave (multiple of 16)
    uint32_t mNXB;           // magic of NXB (division by a launch-time constant)
};

// Shapes the wave kernel takes: one lane per block row, one 64-bit word per parity plane, 8-byte pieces that
// never straddle a row.  Returns false otherwise (the workgroup kernel handles those).
inline bool wv_plan(int h, int w, int cap, WvGeom &g) {
    if (h <= 0 || w <= 0 || h > 128 || w > 128 || (w & 7) || cap <= 0) return false;
    g.H = h; g.W = w;
    g.BH = (h + 1) / 2; g.BW = (w + 1) / 2;
    g.NXB = w / 8;
    g.cap = cap;
    g.rows_bytes = (2 * g.BH + 2) * 16;
    g.wave_bytes = g.rows_bytes + cap * 24;
    g.mNXB = g.NXB <= 1 ? 0u : (uint32_t)(((1ull << 32) / (uint32_t)g.NXB) + 1);
    return true;
}

__device__ __forceinline__ uint32_t nz_bits(uint32_t x) {   // bit 7 of every non-zero byte
    return (((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;
}
// LDS operations of one wave execute in order; this only keeps the compiler from moving them across a phase boundary
__device__ __forceinline__ void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ int ctz64(uint64_t x) { return __ffsll((unsigned long long)x) - 1; }   // x != 0
// position of the k-th (0-based) set bit of x; x has more than k set bits
__device__ __forceinline__ int nth_set_bit64(uint64_t x, int k) {
    uint32_t w = (uint32_t)x;
    int pos = 0;
    int c = __popc(w);
    if (k >= c) { k -= c; w = (uint32_t)(x >> 32); pos = 32; }
    c = __popc(w & 0xFFFFu);
    if (k >= c) { k -= c; w >>= 16; pos += 16; }
    c = __popc(w & 0xFFu);
    if (k >= c) { k -= c; w >>= 8; pos += 8; }
    c = __popc(w & 0xFu);
    if (k >= c) { k -= c; w >>= 4; pos += 4; }
    w &= 0xFu;
    for (int i = 0; i < k; i++) w &= w - 1;   // at most three steps
    return pos + (__ffs((int)w) - 1);
}
__device__ __forceinline__ uint64_t mask_upto(int e) { return e >= 63 ? ~0ull : ((2ull << e) - 1); }   // bits 0..e

// One frame by the calling wave.  m: the frame's mask bytes (8-byte aligned; HBM or LDS); sm: this wave's LDS
// region of g.wave_bytes.  Returns the frame's number of runs n; nothing is written when n > g.cap.
__device__ __forceinline__ int frame_wave(const uint8_t *m, uint8_t *sm, const WvGeom &g, int area_thresh, covahip_box *ob,
                                           int32_t *count_out, int max_boxes, int lane) {
    uint32_t *rows = reinterpret_cast<uint32_t *>(sm);            // [2*BH + 2][4]: E lo, E hi, O lo, O hi; pixel row y at y + 1
    uint32_t *lab = reinterpret_cast<uint32_t *>(sm + g.rows_bytes);
    uint32_t *s_area = lab + g.cap, *s_minx = s_area + g.cap, *s_maxx = s_minx + g.cap, *s_miny = s_maxx + g.cap,
             *s_maxy = s_miny + g.cap;

    // ---- A: parity planes.  LDS operations of one wave execute in order, so the zero fill needs no barrier.
    for (int i = lane; i < g.rows_bytes / 4; i += 64) rows[i] = 0;
    wave_fence();
    const int npieces = g.H * g.NXB;
    // all loads of a group are issued before the first one is consumed: a frame is 8 KB spread over HBM channels,
    // waiting for one piece at a time would expose the miss latency sixteen times
    constexpr int UN = 8;
    for (int q0 = lane; q0 < npieces; q0 += 64 * UN) {
        uint2 v[UN];
#pragma unroll
        for (int k = 0; k < UN; k++) {
            const int q = q0 + 64 * k;
            v[k] = q < npieces ? *reinterpret_cast<const uint2 *>(m + (size_t)q * 8) : make_uint2(0, 0);
        }
#pragma unroll
        for (int k = 0; k < UN; k++) {
            if ((v[k].x | v[k].y) == 0) continue;
            const int q = q0 + 64 * k;
            const int y = g.mNXB ? (int)__umulhi((uint32_t)q, g.mNXB) : q, xc = q - y * g.NXB;
            const uint32_t tl = nz_bits(v[k].x), th = nz_bits(v[k].y);
            // bytes 0 / 2 -> bits 30 / 31 (x 2^23, x 2^8), bytes 1 / 3 -> bits 30 / 31 (x 2^15, x 1); no carries meet
            const uint32_t e = (((tl & 0x00800080u) * 0x00800100u) >> 30) | ((((th & 0x00800080u) * 0x00800100u) >> 30) << 2);
            const uint32_t o = (((tl & 0x80008000u) * 0x00008001u) >> 30) | ((((th & 0x80008000u) * 0x00008001u) >> 30) << 2);
            uint32_t *rw = rows + (y + 1) * 4 + (xc >> 3);
            const int sh = 4 * (xc & 7);
            if (e) atomicOr(rw, e << sh);
            if (o) atomicOr(rw + 2, o << sh);
        }
    }

    wave_fence();
    // ---- B: lane by = block row by
    uint64_t a = 0, b = 0, c = 0, d = 0, ue = 0, uo = 0;
    if (lane < g.BH) {
        const uint4 ru = reinterpret_cast<const uint4 *>(rows)[2 * lane];
        const uint4 r0 = reinterpret_cast<const uint4 *>(rows)[2 * lane + 1];
        const uint4 r1 = reinterpret_cast<const uint4 *>(rows)[2 * lane + 2];
        ue = ru.x | ((uint64_t)ru.y << 32); uo = ru.z | ((uint64_t)ru.w << 32);
        a = r0.x | ((uint64_t)r0.y << 32);  b = r0.z | ((uint64_t)r0.w << 32);
        c = r1.x | ((uint64_t)r1.y << 32);  d = r1.z | ((uint64_t)r1.w << 32);
    }
    const uint64_t F = a | b | c | d;
    const uint64_t J = (a | c) & ((b | d) << 1);
    const uint64_t S = F & ~J;
    const int nr = __popcll((unsigned long long)S);
    int incl = nr;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    const int base = incl - nr;
    const int n = __shfl(incl, 63, 64);
    if (n > g.cap) return n;
    uint64_t S_up = (uint64_t)__shfl_up((unsigned long long)S, 1, 64), J_up = (uint64_t)__shfl_up((unsigned long long)J, 1, 64);
    int base_up = __shfl_up(base, 1, 64);
    if (lane == 0) { S_up = 0; J_up = 0; base_up = 0; }

    for (int i = lane; i < n; i += 64) lab[i] = (uint32_t)i;
    wave_fence();

    // ---- C: runs of a block row.  Round 6: the lanes behind the last block row HELP -- lane BH + r takes the odd-ranked runs of block
    // row r (r < 64 - BH; at 68 rows of pixels 30 of the 34 block rows get a helper), so a frame with many objects walks half as many
    // runs per lane (the longest row sets the pace of the wave).  A helper fetches its row's planes with wave shuffles and derives the
    // masks itself; run ids -- and with them the order of the boxes -- are what they were.
    {
        const int nhelp = min(64 - g.BH, g.BH);
        const bool helper = lane >= g.BH && lane - g.BH < nhelp;
        const int row = helper ? lane - g.BH : lane;
        const bool split = row < nhelp;   // this row's runs are shared between its lane and a helper
        const uint64_t ha = (uint64_t)__shfl((unsigned long long)a, row, 64), hb = (uint64_t)__shfl((unsigned long long)b, row, 64);
        const uint64_t hc = (uint64_t)__shfl((unsigned long long)c, row, 64), hd = (uint64_t)__shfl((unsigned long long)d, row, 64);
        const uint64_t hue = (uint64_t)__shfl((unsigned long long)ue, row, 64), huo = (uint64_t)__shfl((unsigned long long)uo, row, 64);
        const uint64_t hSu = (uint64_t)__shfl((unsigned long long)S_up, row, 64), hJu = (uint64_t)__shfl((unsigned long long)J_up, row, 64);
        const int hbase = __shfl(base, row, 64), hbase_up = __shfl(base_up, row, 64);
        const uint64_t hJ = (ha | hc) & ((hb | hd) << 1);
        const uint64_t hS = (ha | hb | hc | hd) & ~hJ;
        const uint64_t hcU = (ha | hb) & (hue | huo), hcUL = ha & (huo << 1), hcUR = hb & (hue >> 1);
        // P: bit i = parity of the number of run starts at or below block i -> hS & P = runs of even rank, hS & ~P = odd rank
        uint64_t P = hS;
        P ^= P << 1; P ^= P << 2; P ^= P << 4; P ^= P << 8; P ^= P << 16; P ^= P << 32;
        uint64_t Sr = lane >= g.BH && !helper ? 0 : !split ? hS : helper ? hS & ~P : hS & P;
        while (__any(Sr != 0)) {
            uint64_t Tu = 0;
            int idx = 0;
            if (Sr) {
                const int s = ctz64(Sr);
                Sr &= Sr - 1;
                idx = hbase + __popcll((unsigned long long)(hS & ((1ull << s) - 1)));
                const uint64_t jr = (hJ >> s) >> 1;                      // bit k: block s+1+k is joined to its left neighbour
                const int e = s + (~jr ? ctz64(~jr) : 0);                // jr has zeros above BW, so ~jr != 0
                const uint64_t mk = mask_upto(e) & ~((1ull << s) - 1);   // the run's blocks
                const uint64_t ra = ha & mk, rb = hb & mk, rc = hc & mk, rd = hd & mk;
                s_area[idx] = (uint32_t)(__popcll((unsigned long long)ra) + __popcll((unsigned long long)rb) +
                                         __popcll((unsigned long long)rc) + __popcll((unsigned long long)rd));
                s_minx[idx] = (uint32_t)(2 * s + (((ha | hc) >> s) & 1 ? 0 : 1));
                s_maxx[idx] = (uint32_t)(2 * e + (((hb | hd) >> e) & 1 ? 1 : 0));
                s_miny[idx] = (uint32_t)(2 * row + ((ra | rb) ? 0 : 1));
                s_maxy[idx] = (uint32_t)(2 * row + ((rc | rd) ? 1 : 0));
                Tu = (hcU & mk) | ((hcUL & mk) >> 1) | ((hcUR & mk) << 1);   // blocks of the row above this run touches
            }
            while (__any(Tu != 0)) {
                if (Tu) {
                    const int p = ctz64(Tu);
                    const int rank = __popcll((unsigned long long)(hSu << (63 - p))) - 1;   // run of the row above that holds block p
                    const uint64_t jr = (hJu >> p) >> 1;
                    const int e_up = p + (~jr ? ctz64(~jr) : 0);
                    Tu &= ~mask_upto(e_up);
                    ccbody::uf_union(lab, (uint32_t)idx, (uint32_t)(hbase_up + rank));
                }
            }
        }
    }

    wave_fence();
    // ---- D: statistics to the roots
    for (int i = lane; i < n; i += 64) {
        const uint32_t root = ccbody::uf_find(lab, (uint32_t)i);
        if (root != (uint32_t)i) {
            lab[i] = root;
            atomicAdd(&s_area[root], s_area[i]);
            atomicMin(&s_minx[root], s_minx[i]);
            atomicMax(&s_maxx[root], s_maxx[i]);
            atomicMin(&s_miny[root], s_miny[i]);
            atomicMax(&s_maxy[root], s_maxy[i]);
        }
    }
    wave_fence();
    // ---- E: surviving roots in ascending id order
    int total = 0;
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        const bool keep = i < n && lab[i] == (uint32_t)i && (int)s_area[i] >= area_thresh;
        const uint64_t bal = __ballot(keep);
        if (keep) {
            const int pos = total + __popcll((unsigned long long)(bal & ((1ull << lane) - 1)));
            if (pos < max_boxes) {
                covahip_box bx;
                bx.left = (int32_t)s_minx[i];
                bx.top = (int32_t)s_miny[i];
                bx.width = (int32_t)(s_maxx[i] - s_minx[i] + 1);
                bx.height = (int32_t)(s_maxy[i] - s_miny[i] + 1);
                bx.area_px = (int32_t)s_area[i];
                ob[pos] = bx;
            }
        }
        total += __popcll((unsigned long long)bal);
    }
    if (lane == 0) *count_out = total;
    return n;
}

// ---- the same algorithm by ONE WORKGROUP of NTH threads per frame (the fused decoder tail: one frame per CU, the frame's
// mask bytes already in LDS; latency matters, not LDS footprint).  Phases A, D, E spread over all threads; every wave derives
// the row masks of phase B for itself (a few dozen instructions); in phase C wave w takes the runs w, w + NTH / 64, ... of
// every block row.  Run capacity is the worst case (one run per block): nothing overflows.
// m: H x W mask bytes (LDS or HBM, 8-byte aligned); sm: g.wave_bytes of LDS; called by all NTH threads.
// PLANES (round 6, dec3cc_rows_mfma): the parity planes are there already (`planes`, the layout of phase A: [2 BH + 2][E lo, E hi,
// O lo, O hi], pixel row y at y + 1, complete and behind a workgroup barrier); phase A is skipped and `m` is not read.
template <int NTH, bool PLANES = false>
__device__ __forceinline__ void frame_wg(const uint8_t *m, uint8_t *sm, const WvGeom &g, int area_thresh, covahip_box *ob,
                                         int32_t *count_out, int max_boxes, int tid, uint32_t *planes = nullptr) {
    constexpr int NWV = NTH / 64;
    const int lane = tid & 63, wave = tid >> 6;
    uint32_t *rows = PLANES ? planes : reinterpret_cast<uint32_t *>(sm);
    uint32_t *lab = reinterpret_cast<uint32_t *>(sm + g.rows_bytes);
    uint32_t *s_area = lab + g.cap, *s_minx = s_area + g.cap, *s_maxx = s_minx + g.cap, *s_miny = s_maxx + g.cap,
             *s_maxy = s_miny + g.cap;
    __shared__ uint32_t wave_tot[NWV];
#ifdef PHASE_TIMING
    unsigned long long cc_last_ = __builtin_amdgcn_s_memrealtime();
#endif

    // ---- A
    if constexpr (!PLANES) {
    for (int i = tid; i < g.rows_bytes / 4; i += NTH) rows[i] = 0;
    __syncthreads();
    const int npieces = g.H * g.NXB;
    for (int q = tid; q < npieces; q += NTH) {
        const uint2 v = *reinterpret_cast<const uint2 *>(__builtin_assume_aligned(m + (size_t)q * 8, 8));   // one 8-byte load, LDS or HBM
        if ((v.x | v.y) == 0) continue;
        const int y = g.mNXB ? (int)__umulhi((uint32_t)q, g.mNXB) : q, xc = q - y * g.NXB;
        const uint32_t tl = nz_bits(v.x), th = nz_bits(v.y);
        const uint32_t e = (((tl & 0x00800080u) * 0x00800100u) >> 30) | ((((th & 0x00800080u) * 0x00800100u) >> 30) << 2);
        const uint32_t o = (((tl & 0x80008000u) * 0x00008001u) >> 30) | ((((th & 0x80008000u) * 0x00008001u) >> 30) << 2);
        uint32_t *rw = rows + (y + 1) * 4 + (xc >> 3);
        const int sh = 4 * (xc & 7);
        if (e) atomicOr(rw, e << sh);
        if (o) atomicOr(rw + 2, o << sh);
    }
    __syncthreads();
    }
    CCWG_MARK(0);   // A: bit planes
    // ---- B (every wave for itself)
    uint64_t a = 0, b = 0, c = 0, d = 0, ue = 0, uo = 0;
    if (lane < g.BH) {
        const uint4 ru = reinterpret_cast<const uint4 *>(rows)[2 * lane];
        const uint4 r0 = reinterpret_cast<const uint4 *>(rows)[2 * lane + 1];
        const uint4 r1 = reinterpret_cast<const uint4 *>(rows)[2 * lane + 2];
        ue = ru.x | ((uint64_t)ru.y << 32); uo = ru.z | ((uint64_t)ru.w << 32);
        a = r0.x | ((uint64_t)r0.y << 32);  b = r0.z | ((uint64_t)r0.w << 32);
        c = r1.x | ((uint64_t)r1.y << 32);  d = r1.z | ((uint64_t)r1.w << 32);
    }
    const uint64_t F = a | b | c | d;
    const uint64_t J = (a | c) & ((b | d) << 1);
    const uint64_t S = F & ~J;
    const uint64_t cU = (a | b) & (ue | uo), cUL = a & (uo << 1), cUR = b & (ue >> 1);
    const int nr = __popcll((unsigned long long)S);
    int incl = nr;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    const int base = incl - nr;
    const int n = __shfl(incl, 63, 64);
    uint64_t S_up = (uint64_t)__shfl_up((unsigned long long)S, 1, 64), J_up = (uint64_t)__shfl_up((unsigned long long)J, 1, 64);
    int base_up = __shfl_up(base, 1, 64);
    if (lane == 0) { S_up = 0; J_up = 0; base_up = 0; }
    for (int i = tid; i < n; i += NTH) lab[i] = (uint32_t)i;
    __syncthreads();
    CCWG_MARK(1);   // B: row masks, prefix sums
    // ---- C: wave w takes the runs w, w + NWV, ... of every block row (lane = block row): the runs of a row are spread
    // evenly over the waves, a dense row costs ceil(runs / NWV) passes instead of one pass per run
    for (int k = wave; __any(k < nr); k += NWV) {
        uint64_t Tu = 0;
        const int idx = base + k;
        if (k < nr) {
            const int s = nth_set_bit64(S, k);
            const uint64_t jr = (J >> s) >> 1;
            const int e = s + (~jr ? ctz64(~jr) : 0);
            const uint64_t mk = mask_upto(e) & ~((1ull << s) - 1);
            const uint64_t ra = a & mk, rb = b & mk, rc = c & mk, rd = d & mk;
            s_area[idx] = (uint32_t)(__popcll((unsigned long long)ra) + __popcll((unsigned long long)rb) +
                                     __popcll((unsigned long long)rc) + __popcll((unsigned long long)rd));
            s_minx[idx] = (uint32_t)(2 * s + (((a | c) >> s) & 1 ? 0 : 1));
            s_maxx[idx] = (uint32_t)(2 * e + (((b | d) >> e) & 1 ? 1 : 0));
            s_miny[idx] = (uint32_t)(2 * lane + ((ra | rb) ? 0 : 1));
            s_maxy[idx] = (uint32_t)(2 * lane + ((rc | rd) ? 1 : 0));
            Tu = (cU & mk) | ((cUL & mk) >> 1) | ((cUR & mk) << 1);
        }
        while (__any(Tu != 0)) {
            if (Tu) {
                const int p = ctz64(Tu);
                const int rank = __popcll((unsigned long long)(S_up << (63 - p))) - 1;
                const uint64_t jr = (J_up >> p) >> 1;
                const int e_up = p + (~jr ? ctz64(~jr) : 0);
                Tu &= ~mask_upto(e_up);
                ccbody::uf_union(lab, (uint32_t)idx, (uint32_t)(base_up + rank));
            }
        }
    }
    __syncthreads();
    CCWG_MARK(2);   // C: runs, unions
    // ---- D
    for (int i = tid; i < n; i += NTH) {
        const uint32_t root = ccbody::uf_find(lab, (uint32_t)i);
        if (root != (uint32_t)i) {
            lab[i] = root;
            atomicAdd(&s_area[root], s_area[i]);
            atomicMin(&s_minx[root], s_minx[i]);
            atomicMax(&s_maxx[root], s_maxx[i]);
            atomicMin(&s_miny[root], s_miny[i]);
            atomicMax(&s_maxy[root], s_maxy[i]);
        }
    }
    __syncthreads();
    CCWG_MARK(3);   // D: statistics to the roots
    // ---- E: ordered compaction, `per` consecutive runs per thread
    const int per = (n + NTH - 1) / NTH;
    const int i0 = tid * per;
    uint32_t cnt = 0;
    for (int k = 0; k < per; k++) {
        const int i = i0 + k;
        if (i < n && lab[i] == (uint32_t)i && (int)s_area[i] >= area_thresh) cnt++;
    }
    uint32_t inc2 = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(inc2, o, 64);
        if (lane >= o) inc2 += t;
    }
    if (lane == 63) wave_tot[wave] = inc2;
    __syncthreads();
    uint32_t pbase = 0, total = 0;
#pragma unroll
    for (int k = 0; k < NWV; k++) {
        const uint32_t t = wave_tot[k];
        if (k < wave) pbase += t;
        total += t;
    }
    uint32_t pos = pbase + inc2 - cnt;
    for (int k = 0; k < per; k++) {
        const int i = i0 + k;
        if (i < n && lab[i] == (uint32_t)i && (int)s_area[i] >= area_thresh) {
            if ((int)pos < max_boxes) {
                covahip_box bx;
                bx.left = (int32_t)s_minx[i];
                bx.top = (int32_t)s_miny[i];
                bx.width = (int32_t)(s_maxx[i] - s_minx[i] + 1);
                bx.height = (int32_t)(s_maxy[i] - s_miny[i] + 1);
                bx.area_px = (int32_t)s_area[i];
                ob[pos] = bx;
            }
            pos++;
        }
    }
    if (tid == 0) *count_out = (int32_t)total;
    __syncthreads();   // wave_tot and the frame's LDS region are free for the next frame
    CCWG_MARK(4);   // E: compaction, box stores
}

}  // namespace ccwave
