/*
 * oracle/ccl_ref.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * Plain-C restatement of what CoVA's bboxcc element computes:
 *   cova-rs/gst-plugins/src/bboxcc/process.rs:5-49  regionprops():
 *     connectedComponentsWithStats(connectivity=8, CV_32S) on the H x W u8 mask
 *     (non-zero = foreground), labels 1..n-1 in OpenCV label order, keep
 *     CC_STAT_AREA >= area_thresh, emit (LEFT, TOP, WIDTH, HEIGHT).
 *
 * The algorithm itself lives in a third-party dependency that is NOT under
 * /root/reference: OpenCV (crate opencv 0.53.2, cova-rs/Cargo.lock:1278-1279,
 * binding the system OpenCV of the DeepStream 6.0 image; version not pinned in
 * the repo).  PARITY STATUS: "parity unpinned" -- the reference holds no test or
 * golden vector at this boundary.  What is restated here is the published
 * algorithm OpenCV uses for 8-connectivity (Grana et al., block-based decision
 * tree "BBDT"; the later "Spaghetti" default keeps the same scan and label
 * numbering): the image is scanned in raster order of 2x2 blocks, a block that
 * touches no already-labelled neighbour block gets the next provisional label,
 * equivalences are merged with a min-root union-find, and flattenL() renumbers
 * the roots consecutively in increasing provisional-label order.  The decision
 * tree only prunes redundant pixel tests; the scan below evaluates all of the
 * block-connectivity conditions directly and therefore yields the same labels.
 * tests/ additionally checks the component SET against scipy.ndimage.label
 * (an independent pixel-based implementation).
 *
 * Statistics follow OpenCV's CCStatsOp: LEFT/TOP = min x/y, WIDTH/HEIGHT =
 * max-min+1, AREA = pixel count.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int32_t left, top, width, height, area;
} cova_ref_box;

static int uf_find(const int *P, int i) {
    while (P[i] < i) i = P[i];
    return i;
}
static int uf_union(int *P, int i, int j) {
    i = uf_find(P, i);
    j = uf_find(P, j);
    if (i < j) { P[j] = i; return i; }
    P[i] = j;
    return j;
}

/* mask: u8 [H][W]; labels (optional, may be NULL): i32 [H][W] final labels
 * (0 = background); boxes: up to max_boxes entries in label order after the
 * area filter.  Returns the number of boxes that pass the filter (may exceed
 * max_boxes; only the first max_boxes are written); *n_labels gets the total
 * number of components before filtering. */
int cova_ref_regionprops(const uint8_t *mask, int H, int W, int area_thresh, cova_ref_box *boxes,
                         int max_boxes, int32_t *labels, int *n_labels) {
    const int BH = (H + 1) / 2, BW = (W + 1) / 2;
    int *blab = (int *)calloc((size_t)BH * BW, sizeof(int));
    int *P = (int *)malloc(sizeof(int) * ((size_t)BH * BW + 1));
    int next = 1;
    P[0] = 0;
#define PX(y, x) ((y) >= 0 && (y) < H && (x) >= 0 && (x) < W && mask[(y) * W + (x)] != 0)
    for (int by = 0; by < BH; by++)
        for (int bx = 0; bx < BW; bx++) {
            const int r = 2 * by, c = 2 * bx;
            const int a = PX(r, c), b = PX(r, c + 1), cc = PX(r + 1, c), d = PX(r + 1, c + 1);
            if (!(a | b | cc | d)) continue;
            int lab = 0;
#define MERGE(cond, yy, xx)                                            \
    if (cond) {                                                        \
        int nb = blab[(yy) * BW + (xx)];                               \
        lab = lab ? uf_union(P, lab, nb) : uf_find(P, nb);             \
    }
            /* neighbour blocks already scanned: up-left, up, up-right, left */
            MERGE(by > 0 && bx > 0 && a && PX(r - 1, c - 1), by - 1, bx - 1)
            MERGE(by > 0 && (a | b) && (PX(r - 1, c) | PX(r - 1, c + 1)), by - 1, bx)
            MERGE(by > 0 && bx + 1 < BW && b && PX(r - 1, c + 2), by - 1, bx + 1)
            MERGE(bx > 0 && (a | cc) && (PX(r, c - 1) | PX(r + 1, c - 1)), by, bx - 1)
#undef MERGE
            if (!lab) { lab = next; P[next] = next; next++; }
            blab[by * BW + bx] = lab;
        }
    /* flattenL: consecutive final labels in increasing provisional order */
    int k = 1;
    for (int i = 1; i < next; i++) {
        if (P[i] < i) P[i] = P[P[i]];
        else P[i] = k++;
    }
    const int n = k; /* labels 0..n-1, 0 = background */
    int *minx = (int *)malloc(sizeof(int) * n), *miny = (int *)malloc(sizeof(int) * n);
    int *maxx = (int *)malloc(sizeof(int) * n), *maxy = (int *)malloc(sizeof(int) * n);
    int *area = (int *)calloc(n, sizeof(int));
    for (int i = 0; i < n; i++) { minx[i] = W; miny[i] = H; maxx[i] = -1; maxy[i] = -1; }
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            int l = 0;
            if (mask[y * W + x]) l = P[blab[(y >> 1) * BW + (x >> 1)]];
            if (labels) labels[y * W + x] = l;
            if (!l) continue;
            area[l]++;
            if (x < minx[l]) minx[l] = x;
            if (x > maxx[l]) maxx[l] = x;
            if (y < miny[l]) miny[l] = y;
            if (y > maxy[l]) maxy[l] = y;
        }
    int out = 0;
    for (int l = 1; l < n; l++) {
        if (area[l] < area_thresh) continue; /* process.rs:39  AREA >= thresh */
        if (out < max_boxes) {
            boxes[out].left = minx[l];
            boxes[out].top = miny[l];
            boxes[out].width = maxx[l] - minx[l] + 1;
            boxes[out].height = maxy[l] - miny[l] + 1;
            boxes[out].area = area[l];
        }
        out++;
    }
    if (n_labels) *n_labels = n - 1;
    free(minx); free(miny); free(maxx); free(maxy); free(area); free(blab); free(P);
#undef PX
    return out;
}

/* Batch helper used by the cpu_baseline leg and the parity tests:
 * masks [B][H][W] -> boxes [B][max_boxes], counts [B]. */
void cova_ref_regionprops_batch(const uint8_t *masks, int B, int H, int W, int area_thresh,
                                cova_ref_box *boxes, int32_t *counts, int max_boxes) {
    for (int b = 0; b < B; b++)
        counts[b] = cova_ref_regionprops(masks + (size_t)b * H * W, H, W, area_thresh,
                                         boxes + (size_t)b * max_boxes, max_boxes, NULL, NULL);
}

/*
 * metapreprocess temporal stacking (cova-rs/gst-plugins/src/metapreprocess/
 * imp.rs:288-332).  frames: u8 [N][frame_stride] carrier frames of which only
 * the first size_per_buf bytes matter; out: u8 [n_out][T*size_per_buf].
 * Output k (emitted for input i >= T-1 when gamma_idx == 0) = input i followed
 * by inputs i-1 ... i-T+1.  Returns the number of outputs; out_src_index[k] =
 * index i of the input that produced output k (its PTS is inherited).
 */
int cova_ref_metapreprocess(const uint8_t *frames, int N, size_t frame_stride, size_t size_per_buf,
                            int T, int gamma, uint8_t *out, int32_t *out_src_index) {
    int n_prev = 0, gamma_idx = 0, n_out = 0;
    for (int i = 0; i < N; i++) {
        if (n_prev < T - 1) { n_prev++; continue; } /* stored, FLOW_DROPPED */
        if (gamma_idx == 0) {
            uint8_t *dst = out + (size_t)n_out * T * size_per_buf;
            for (int t = 0; t < T; t++)
                memcpy(dst + (size_t)t * size_per_buf, frames + (size_t)(i - t) * frame_stride, size_per_buf);
            if (out_src_index) out_src_index[n_out] = i;
            n_out++;
            gamma_idx = gamma - 1;
        } else {
            gamma_idx--;
        }
    }
    return n_out;
}
