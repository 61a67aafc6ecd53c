/* pipe_host_cost.c -- where a batch's time goes on the HOST side of the pinned pipeline (covahip_pipe_*), driven from C so
 * that no interpreter is in the loop: frames/s and the mean time the submitting thread spends inside covahip_pipe_submit, inside
 * the blocking part of covahip_pipe_collect and in acquire / release, at b = 256 / 68x120 with packed records.
 * Usage: pipe_host_cost <weights blob> <steps> <lanes> <slots>      (slots = 0: the same batch resident in HBM, no pipe)
 * Build: gcc -O2 -I include tools/pipe_host_cost.c -o tools/pipe_host_cost -L cova_amd -lcovahip -Wl,-rpath,$PWD/cova_amd */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "covahip.h"

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main(int argc, char **argv) {
    const int H = 68, W = 120, B = 256, S = 8, PER = B / S, NF = S * (PER + 3);
    if (argc < 5) { fprintf(stderr, "usage: %s weights.bin steps lanes slots\n", argv[0]); return 1; }
    const int steps = atoi(argv[2]), lanes = atoi(argv[3]), slots = atoi(argv[4]);
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    void *blob = malloc(n);
    if (fread(blob, 1, n, f) != (size_t)n) return 2;
    fclose(f);
    covahip_ctx *ctx; covahip_pipe *pipe;
    if (covahip_ctx_create(0, &ctx) || covahip_ctx_set_lanes(ctx, lanes) || covahip_blobnet_load(ctx, blob, n, H, W, 4, B)) return 3;
    if (slots > 0 && (covahip_pipe_create(ctx, B, NF, 256, slots, 0, &pipe) || covahip_pipe_set_packed(pipe, 1))) return 4;
    const size_t fb = (size_t)H * W * 4;
    uint8_t *src = malloc(NF * fb);
    uint16_t *packed = malloc(NF * fb / 2);
    unsigned x = 12345;
    for (size_t i = 0; i < NF * fb; i++) { x = x * 1664525u + 1013904223u; src[i] = (x >> 24) % 7; }
    covahip_carrier_pack(src, NF * (size_t)H * W, packed);
    int32_t *idx = malloc(sizeof(int32_t) * B * 4);
    for (int s = 0; s < S; s++)
        for (int j = 0; j < PER; j++)
            for (int k = 0; k < 4; k++) idx[(j * S + s) * 4 + k] = s * (PER + 3) + j + 3 - k;
    if (slots == 0) {   /* the same batch RESIDENT in HBM through covahip_filter_forward_frames_packed: what the kernels alone give on this input */
        void *d_fr, *d_bx[4], *d_ct[4];
        if (covahip_malloc(ctx, NF * fb / 2, &d_fr) || covahip_memcpy_h2d(ctx, d_fr, packed, NF * fb / 2)) return 7;
        for (int l = 0; l < 4; l++)
            if (covahip_malloc(ctx, (size_t)B * 256 * sizeof(covahip_box), &d_bx[l]) || covahip_malloc(ctx, B * 4, &d_ct[l])) return 7;
        double r0 = 0;
        for (int k = -200; k < steps; k++) {
            if (k == 0) { covahip_ctx_sync(ctx); r0 = now(); }
            if (covahip_filter_forward_frames_packed(ctx, d_fr, NF, idx, B, 1, d_bx[k & 3], d_ct[k & 3], 256, NULL, NULL)) return 8;
            if ((k & 15) == 15 && k < 0) covahip_ctx_sync(ctx);
        }
        covahip_ctx_sync(ctx);
        const double rt = now() - r0;
        printf("{\"lanes\": %d, \"resident\": true, \"frames_per_s\": %.1f, \"us_per_batch\": %.1f}\n", lanes, steps * (double)B / rt, rt / steps * 1e6);
        return 0;
    }
    int inflight[16], nin = 0;
    long boxes = 0;
    double t0 = 0, t_submit = 0, t_collect = 0, t_other = 0;
    for (int k = -2 * slots; k < steps; k++) {
        if (k == 0) { t0 = now(); t_submit = t_collect = t_other = 0; }
        int slot; uint8_t *pf; int32_t *pi;
        double a = now();
        while (covahip_pipe_acquire(pipe, &slot, &pf, &pi) == COVAHIP_ERR_OVERFLOW) {
            const int32_t *off;
            const double c0 = now();
            if (covahip_pipe_collect(pipe, inflight[0], NULL, &off, NULL, NULL)) return 5;
            t_collect += now() - c0;
            boxes += off[B];
            covahip_pipe_release(pipe, inflight[0]);
            memmove(inflight, inflight + 1, sizeof(int) * --nin);
        }
        if (k < -slots) { memcpy(pf, packed, NF * fb / 2); memcpy(pi, idx, sizeof(int32_t) * B * 4); }
        const double s0 = now();
        if (covahip_pipe_submit(pipe, slot, NF, B, 1)) return 6;
        const double s1 = now();
        t_submit += s1 - s0;
        t_other += s0 - a;
        inflight[nin++] = slot;
    }
    for (int i = 0; i < nin; i++) { covahip_pipe_collect(pipe, inflight[i], NULL, NULL, NULL, NULL); covahip_pipe_release(pipe, inflight[i]); }
    const double dt = now() - t0;
    printf("{\"lanes\": %d, \"slots\": %d, \"frames_per_s\": %.1f, \"us_per_batch\": %.1f, \"us_in_submit\": %.1f, \"us_waiting_in_collect\": %.1f, "
           "\"us_acquire_release_incl_collect\": %.1f, \"boxes_per_batch\": %.0f}\n",
           lanes, slots, steps * (double)B / dt, dt / steps * 1e6, t_submit / steps * 1e6, t_collect / steps * 1e6, t_other / steps * 1e6,
           (double)boxes / (steps + slots));
    covahip_pipe_destroy(pipe);
    covahip_ctx_destroy(ctx);
    return 0;
}
