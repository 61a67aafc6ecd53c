/* pipe_bench.c -- the pipelined host path (covahip_pipe_*) driven from C: carrier frames of 8 streams in pinned slots,
 * packed boxes out, b = 256 at 68x120.  Usage: pipe_bench <weights blob> <steps> [fill]
 * Prints frames/s.  (Build: cc -O2 -I include tools/pipe_bench.c -L cova_amd -lcovahip -Wl,-rpath,cova_amd) */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "covahip.h"

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main(int argc, char **argv) {
    const int H = 68, W = 120, B = 256, S = 8, PER = B / S, NF = S * (PER + 3);
    if (argc < 3) { fprintf(stderr, "usage: %s weights.bin steps [fill]\n", argv[0]); return 1; }
    const int steps = atoi(argv[2]), fill = argc > 3;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    void *blob = malloc(n);
    if (fread(blob, 1, n, f) != (size_t)n) return 2;
    fclose(f);
    covahip_ctx *ctx; covahip_pipe *pipe;
    if (covahip_ctx_create(0, &ctx) || covahip_ctx_set_lanes(ctx, 2) || covahip_blobnet_load(ctx, blob, n, H, W, 4, B)) return 3;
    if (covahip_pipe_create(ctx, B, NF, 2048, 3, 0, &pipe)) return 4;
    const size_t fb = (size_t)H * W * 4;
    uint8_t *src = malloc(NF * fb);
    unsigned x = 12345;
    for (size_t i = 0; i < NF * fb; i++) { x = x * 1664525u + 1013904223u; src[i] = (x >> 24) % 7; }
    int32_t *idx = malloc(sizeof(int32_t) * B * 4);
    for (int s = 0; s < S; s++)
        for (int j = 0; j < PER; j++)
            for (int k = 0; k < 4; k++) idx[(j * S + s) * 4 + k] = s * (PER + 3) + j + 3 - k;
    int inflight[8], nin = 0;
    long boxes = 0;
    double t0 = 0;
    for (int k = -6; k < steps; k++) {
        if (k == 0) t0 = now();
        int slot; uint8_t *pf; int32_t *pi;
        while (covahip_pipe_acquire(pipe, &slot, &pf, &pi) == COVAHIP_ERR_OVERFLOW) {
            const int32_t *off;
            if (covahip_pipe_collect(pipe, inflight[0], NULL, &off, NULL, NULL)) return 5;
            boxes += off[B];
            covahip_pipe_release(pipe, inflight[0]);
            memmove(inflight, inflight + 1, sizeof(int) * --nin);
        }
        if (fill || k < -3) { memcpy(pf, src, NF * fb); memcpy(pi, idx, sizeof(int32_t) * B * 4); }
        if (covahip_pipe_submit(pipe, slot, NF, B, 1)) return 6;
        inflight[nin++] = slot;
    }
    for (int i = 0; i < nin; i++) { covahip_pipe_collect(pipe, inflight[i], NULL, NULL, NULL, NULL); covahip_pipe_release(pipe, inflight[i]); }
    const double dt = now() - t0;
    printf("{\"frames_per_s\": %.1f, \"fill\": %d, \"us_per_batch\": %.1f}\n", steps * (double)B / dt, fill, dt / steps * 1e6);
    covahip_pipe_destroy(pipe);
    covahip_ctx_destroy(ctx);
    return 0;
}
