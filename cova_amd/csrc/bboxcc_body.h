// bboxcc of ONE mask frame by one 1,024-thread workgroup, all state in LDS: shared by the standalone
// kernel (bboxcc.hip, mask read from HBM) and the fused decoder tail (blobnet_mfma.hip, mask still
// in LDS).  See bboxcc.hip for the algorithm.
#pragma once
#include <cstdint>

#include "internal.h"

namespace ccbody {

constexpr int CC_THREADS = 1024;  // 16 waves: the union-find phases are LDS-latency bound, so run 4 waves per SIMD
constexpr uint32_t NONE = 0xFFFFFFFFu;

__device__ __forceinline__ uint32_t uf_find(const volatile uint32_t *lab, uint32_t i) {
    uint32_t p = lab[i];
    while (p != i) {
        i = p;
        p = lab[i];
    }
    return i;
}

// min-root union; safe under concurrent unions from other lanes/waves.
__device__ __forceinline__ void uf_union(uint32_t *lab, uint32_t a, uint32_t b) {
    while (true) {
        a = uf_find(lab, a);
        b = uf_find(lab, b);
        if (a == b) return;
        if (a < b) {
            uint32_t t = a;
            a = b;
            b = t;
        }
        // a > b: hang a under b unless somebody re-parented a meanwhile
        uint32_t old = atomicMin(&lab[a], b);
        if (old == a) return;
        a = old;
    }
}

struct CcGeom {
    int H, W, BH, BW, NB;
    int RS;     // packed-row stride in bytes (1 pad byte left, >=1 right)
    int NXB;    // packed bytes per row = ceil(W/8)
    int rowl_off;  // byte offset of the per-block-row "joined to the left" masks in LDS
    int wt_off;    // byte offset of the CC_THREADS / 64 wave totals of the compaction scan
};

// LDS bytes one frame needs and the offsets inside it (host side).  Returns 0 when the shape is unsupported.
inline size_t cc_plan(int h, int w, CcGeom &g) {
    g.H = h;
    g.W = w;
    g.BH = (h + 1) / 2;
    g.BW = (w + 1) / 2;
    g.NB = g.BH * g.BW;
    g.NXB = (w + 7) / 8;
    g.RS = g.NXB + 2;
    const size_t rb_bytes = (((size_t)(h + 3) * g.RS) + 15) & ~(size_t)15;
    const size_t rowl_off = rb_bytes + (size_t)g.NB * 4 * 6 + (((size_t)g.NB + 15) & ~(size_t)15);
    g.rowl_off = (int)rowl_off;
    g.wt_off = (int)(rowl_off + (size_t)g.BH * 16);
    const size_t lds = (size_t)g.wt_off + 64;
    if (lds + 64 > 160 * 1024 || g.BW > 128) return 0;
    return lds;
}
// The same carve-up for a frame whose state does not fit in LDS (4K grids): bytes of a slab in GLOBAL memory that
// bboxcc_frame takes instead (generic pointers, flat atomics; slower, any height, width up to 256 blocks' worth of the
// rowL bit rows = 128 blocks).  0: not even that (wider than 256 pixels).
inline size_t cc_plan_global(int h, int w, CcGeom &g) {
    const size_t fits = cc_plan(h, w, g);
    if (fits) return fits;
    if (g.BW > 128) return 0;
    return (((size_t)g.wt_off + 64) + 255) & ~(size_t)255;
}

// m: the frame's H x W mask bytes (HBM or LDS); smem: this frame's LDS region of cc_plan() bytes;
// ob: the frame's box slots; *count_out: its box count.  Called by all CC_THREADS threads.
__device__ __forceinline__ void bboxcc_frame(const uint8_t *m, uint8_t *smem, const CcGeom &g, int area_thresh,
                                             covahip_box *ob, int32_t *count_out, int max_boxes, int tid) {
    const int H = g.H, W = g.W, BW = g.BW, NB = g.NB, RS = g.RS, NXB = g.NXB;

    // LDS carve-up (all offsets multiples of 16)
    const int rb_bytes = ((H + 3) * RS + 15) & ~15;
    uint8_t *rb = smem;                                   // packed rows, row y at (y+1)*RS + 1
    uint32_t *lab = (uint32_t *)(smem + rb_bytes);        // [NB]
    uint32_t *s_area = lab + NB;
    uint32_t *s_minx = s_area + NB;
    uint32_t *s_maxx = s_minx + NB;
    uint32_t *s_miny = s_maxx + NB;
    uint32_t *s_maxy = s_miny + NB;
    uint8_t *binfo = (uint8_t *)(s_maxy + NB);            // [NB] fg nibble | conn nibble << 4
    uint32_t *rowL = (uint32_t *)(smem + g.rowl_off);     // [BH][4]: bit bx = block bx is joined to block bx-1 (BW <= 128)
    uint32_t *wave_tot = (uint32_t *)(smem + g.wt_off);   // [CC_THREADS / 64]

    // ---- phase 0: clear packed rows (pads must be zero)
    for (int i = tid; i < rb_bytes / 4; i += CC_THREADS) ((uint32_t *)rb)[i] = 0;
    for (int i = tid; i < g.BH * 4; i += CC_THREADS) rowL[i] = 0;
    __syncthreads();

    // ---- phase 1: stream the mask in, 8 pixels per work item, and bit-pack it
    const bool fast = (W % 8 == 0) && ((((uintptr_t)m) & 7) == 0);
    const int n_chunks = H * NXB;
    for (int q = tid; q < n_chunks; q += CC_THREADS) {
        const int y = q / NXB, xc = q - y * NXB;
        uint32_t bits = 0;
        if (fast) {
            const uint2 v = *reinterpret_cast<const uint2 *>(m + (size_t)y * W + xc * 8);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                bits |= (((v.x >> (8 * k)) & 0xFF) ? 1u : 0u) << k;
                bits |= (((v.y >> (8 * k)) & 0xFF) ? 1u : 0u) << (4 + k);
            }
        } else {
            for (int k = 0; k < 8; k++) {
                const int x = xc * 8 + k;
                if (x < W && m[(size_t)y * W + x]) bits |= 1u << k;
            }
        }
        rb[(y + 1) * RS + 1 + xc] = (uint8_t)bits;
    }
    __syncthreads();

    // ---- phase 2: per packed byte of a block row -> 4 blocks: nibble + connections, parked as one
    // byte per block (fg nibble | conn nibble << 4) so that the union / statistics phases run one
    // block per work item.
    const int n_units = g.BH * NXB;
    for (int q = tid; q < n_units; q += CC_THREADS) {
        const int by = q / NXB, xc = q - by * NXB;
        const int r = 2 * by;
        // 24-bit windows: bit 8+k = pixel 8*xc+k; bits 7 / 16 = neighbours across bytes
        const uint8_t *p0 = rb + (r + 0) * RS + xc;  // row r-1 (stored at index r), byte xc-1
        const uint8_t *p1 = p0 + RS;                 // row r
        const uint8_t *p2 = p1 + RS;                 // row r+1
        const uint32_t up = p0[0] | (p0[1] << 8) | (p0[2] << 16);
        const uint32_t ra = p1[0] | (p1[1] << 8) | (p1[2] << 16);
        const uint32_t rc = p2[0] | (p2[1] << 8) | (p2[2] << 16);
        uint32_t lbits = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int bx = xc * 4 + j;
            const int s = 8 + 2 * j;  // bit of pixel column c = 2*bx
            const uint32_t a = (ra >> s) & 1, b = (ra >> (s + 1)) & 1;
            const uint32_t c = (rc >> s) & 1, d = (rc >> (s + 1)) & 1;
            const uint32_t fg = a | (b << 1) | (c << 2) | (d << 3);
            const uint32_t u_l = (up >> (s - 1)) & 1, u_0 = (up >> s) & 1, u_1 = (up >> (s + 1)) & 1,
                           u_r = (up >> (s + 2)) & 1;
            const uint32_t l_a = (ra >> (s - 1)) & 1, l_c = (rc >> (s - 1)) & 1;
            const uint32_t cL = (a | c) & (l_a | l_c);
            const uint32_t cUL = a & u_l;
            const uint32_t cU = (a | b) & (u_0 | u_1);
            const uint32_t cUR = b & u_r;
            const uint32_t conn = cL | (cUL << 1) | (cU << 2) | (cUR << 3);
            const uint32_t blk = by * BW + bx;
            if (bx < BW) {
                lab[blk] = fg ? blk : NONE;
                s_area[blk] = 0;
                s_minx[blk] = 0x7FFFFFFF;
                s_maxx[blk] = 0;
                s_miny[blk] = 0x7FFFFFFF;
                s_maxy[blk] = 0;
                binfo[blk] = (uint8_t)(fg | (conn << 4));
                lbits |= cL << j;
            }
        }
        if (lbits) atomicOr(&rowL[by * 4 + (xc >> 3)], lbits << ((xc & 7) * 4));   // 4 blocks per unit, 8 units per word
    }
    __syncthreads();

    // ---- phase 2b: label of a foreground block = id of the first block of its horizontal run
    for (int blk = tid; blk < NB; blk += CC_THREADS) {
        if (!(binfo[blk] & 0xF)) continue;
        const int by = blk / BW, bx = blk - by * BW;
        // highest zero of the "joined to the left" mask at or below bx (bit 0 of a row is always zero)
        int w = bx >> 5;
        uint32_t z = ~rowL[by * 4 + w] & (0xFFFFFFFFu >> (31 - (bx & 31)));
        while (!z) z = ~rowL[by * 4 + --w];
        lab[blk] = by * BW + w * 32 + (31 - __clz(z));
    }
    __syncthreads();

    // ---- phase 3: unions with the three neighbours in the block row above.  A union is skipped when
    // it is implied by one that is made anyway: by this block (the up neighbour is joined to the
    // up-left / up-right one inside the upper row) or by the left neighbour of the same run (it
    // reaches the same upper block, or one joined to it).
    for (int blk = tid; blk < NB; blk += CC_THREADS) {
        const uint32_t me = binfo[blk];
        const uint32_t conn = me >> 5 << 1;
        if (!conn) continue;
        const int by = blk / BW, bx = blk - by * BW;
        const uint32_t *upL = rowL + (by - 1) * 4;                       // conn != 0 implies by >= 1
        const bool up_j0 = (upL[bx >> 5] >> (bx & 31)) & 1;              // upper bx joined to upper bx-1
        const bool up_j1 = bx + 1 < BW && ((upL[(bx + 1) >> 5] >> ((bx + 1) & 31)) & 1);   // upper bx+1 joined to upper bx
        const uint32_t left = (me >> 4) & 1 ? binfo[blk - 1] >> 4 : 0;   // connections of the left block when it is in my run
        const bool cUL = conn & 2, cU = conn & 4, cUR = conn & 8;
        const bool lU = left & 4, lUR = left & 8;
        if (cU && !(lUR || (lU && up_j0))) uf_union(lab, blk, blk - BW);
        if (cUL && !(cU && up_j0) && !lU) uf_union(lab, blk, blk - BW - 1);
        if (cUR && !(cU && up_j1)) uf_union(lab, blk, blk - BW + 1);
    }
    __syncthreads();

    // ---- phase 4+5: flatten and accumulate statistics on the root
    // (one set of atomics per horizontal run instead of per block was measured: 25 % faster on
    //  half-full masks, 10 % slower on sparse blobs because the run's first block walks it serially)
    for (int blk = tid; blk < NB; blk += CC_THREADS) {
        const uint32_t fg = binfo[blk] & 0xF;
        if (!fg) continue;
        const uint32_t by = (uint32_t)blk / (uint32_t)BW, bx = blk - by * BW;
        const uint32_t root = uf_find(lab, blk);
        const uint32_t x0 = 2 * bx + ((fg & 5) ? 0 : 1), x1 = 2 * bx + ((fg & 10) ? 1 : 0);
        const uint32_t y0 = 2 * by + ((fg & 3) ? 0 : 1), y1 = 2 * by + ((fg & 12) ? 1 : 0);
        atomicAdd(&s_area[root], __popc(fg));
        atomicMin(&s_minx[root], x0);
        atomicMax(&s_maxx[root], x1);
        atomicMin(&s_miny[root], y0);
        atomicMax(&s_maxy[root], y1);
    }
    __syncthreads();

    // ---- phase 6: ordered compaction of surviving roots (ascending block id)
    const int per = (NB + CC_THREADS - 1) / CC_THREADS;
    const int i0 = tid * per;
    uint32_t cnt = 0;
    for (int k = 0; k < per; k++) {
        const int i = i0 + k;
        if (i < NB && lab[i] == (uint32_t)i && (int)s_area[i] >= area_thresh) cnt++;
    }
    // wave inclusive scan
    const int lane = tid & 63, wv = tid >> 6;
    uint32_t incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wave_tot[wv] = incl;
    __syncthreads();
    uint32_t base = 0, total = 0;
#pragma unroll
    for (int k = 0; k < CC_THREADS / 64; k++) {
        const uint32_t t = wave_tot[k];
        if (k < wv) base += t;
        total += t;
    }
    uint32_t pos = base + incl - cnt;
    for (int k = 0; k < per; k++) {
        const int i = i0 + k;
        if (i < NB && lab[i] == (uint32_t)i && (int)s_area[i] >= area_thresh) {
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
}

}  // namespace ccbody
