/* Developer helper (CPU only): host time per frame of the `cova` element's state machine -- covahip_gopfilter_push_enc +
 * covahip_gopfilter_push_boxes (embedded SORT + GoP frame filter) at the experiment's parameters (maxage 60, minhits 30, iou 0.1,
 * experiment/cova/config.yaml:59,67) on blob-like detections: K objects that drift and turn plus a few spurious one-macroblock
 * boxes per frame (what cc-threshold 1 lets through).  One stream, one thread.
 *   gcc -O2 -Iinclude tools/host_chain_bench.c -o tools/host_chain_bench -Lcova_amd -lcovahip -Wl,-rpath,$PWD/cova_amd -lm
 *   tools/host_chain_bench [frames] [objects] [spurious per frame] */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "covahip.h"

static unsigned long long rs = 88172645463325252ull;
static double rnd(void) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (rs >> 11) * (1.0 / 9007199254740992.0); }

int main(int argc, char **argv) {
    const int frames = argc > 1 ? atoi(argv[1]) : 20000, K = argc > 2 ? atoi(argv[2]) : 4, spurious = argc > 3 ? atoi(argv[3]) : 3;
    covahip_gopfilter_cfg cfg;
    covahip_gopfilter_default_cfg(&cfg);
    cfg.sort_iou = 0.1f; cfg.sort_maxage = 60; cfg.sort_minhits = 30;
    covahip_gopfilter *g;
    if (covahip_gopfilter_new(&cfg, &g)) return 1;
    double ox[64], oy[64], vx[64], vy[64], ow[64], oh[64];
    for (int k = 0; k < K; k++) { ox[k] = rnd() * 100; oy[k] = rnd() * 50; vx[k] = rnd() - 0.5; vy[k] = rnd() * 0.6 - 0.3; ow[k] = 3 + rnd() * 8; oh[k] = 2 + rnd() * 6; }
    covahip_bbox *bb = calloc(256, sizeof *bb);
    covahip_au_out *out = calloc(4096, sizeof *out);
    uint64_t dropped[1024];
    struct timespec t0, t1;
    long boxes = 0, outs = 0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int f = 0; f < frames; f++) {
        const uint64_t pts = (uint64_t)f * (1000000000ull / 30);
        covahip_gopfilter_push_enc(g, (uint64_t)f + 1, pts, f % 250 ? COVAHIP_AU_DELTA_UNIT : 0);
        int n = 0;
        for (int k = 0; k < K; k++) {
            ox[k] += vx[k]; oy[k] += vy[k];
            if (ox[k] < 0 || ox[k] > 110) vx[k] = -vx[k];
            if (oy[k] < 0 || oy[k] > 60) vy[k] = -vy[k];
            if (rnd() < 0.02) { vx[k] = rnd() - 0.5; vy[k] = rnd() * 0.6 - 0.3; }
            if (rnd() < 0.9) {   /* an object is missed now and then */
                memset(&bb[n], 0, sizeof bb[n]);
                bb[n].left = (float)floor(ox[k]); bb[n].top = (float)floor(oy[k]); bb[n].width = (float)floor(ow[k] + rnd() * 2); bb[n].height = (float)floor(oh[k] + rnd() * 2);
                bb[n].area = bb[n].width * bb[n].height;
                n++;
            }
        }
        for (int s = 0; s < spurious; s++)
            if (rnd() < 0.7) {
                memset(&bb[n], 0, sizeof bb[n]);
                bb[n].left = (float)floor(rnd() * 118); bb[n].top = (float)floor(rnd() * 66); bb[n].width = 1 + (float)floor(rnd() * 2); bb[n].height = 1 + (float)floor(rnd() * 2);
                bb[n].area = bb[n].width * bb[n].height;
                n++;
            }
        boxes += n;
        size_t no = 0, nd = 0;
        const int rc = covahip_gopfilter_push_boxes(g, bb, (size_t)n, pts, out, 4096, &no);
        if (rc) { fprintf(stderr, "push_boxes: %d at frame %d\n", rc, f); return 2; }
        outs += (long)no;
        covahip_gopfilter_take_dropped(g, dropped, 1024, &nd);
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    const double s = (t1.tv_sec - t0.tv_sec) + (t1.tv_nsec - t0.tv_nsec) * 1e-9;
    printf("%d frames, %.2f boxes per frame, %ld access units forwarded: %.2f us per frame (%.0f frames/s on one thread)\n", frames,
           (double)boxes / frames, outs, s / frames * 1e6, frames / s);
    covahip_gopfilter_free(g);
    free(bb);
    free(out);
    return 0;
}
