/* stream_probe.c -- how much of a step is ramp / tail of its launches?  N contexts (N HIP streams, N host threads) on ONE
 * GPU, each running the carrier-frame hot path on B/N of the batch back to back; prints the aggregate frames/s.
 * Every context keeps `lanes` batches in flight (covahip_ctx_set_lanes).  Usage: stream_probe <weights blob> <B total> <N ctx> <steps> [lanes]
 * (Build: cc -O2 -pthread -I include tools/stream_probe.c -L cova_amd -lcovahip -Wl,-rpath,$PWD/cova_amd -o tools/stream_probe) */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "covahip_dev.h"

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

enum { H = 68, W = 120, MAXB = 2048 };
static void *g_blob;
static long g_blob_bytes;
static int g_b, g_steps, g_lanes = 2;
static pthread_barrier_t g_bar;

static void *worker(void *arg) {
    const int id = (int)(long)arg;
    const int B = g_b, S = B >= 8 ? 8 : 1, PER = B / S, NF = S * (PER + 3);
    covahip_ctx *ctx;
    if (covahip_ctx_create(0, &ctx) || covahip_blobnet_load(ctx, g_blob, g_blob_bytes, H, W, 4, B)) { fprintf(stderr, "load failed\n"); exit(3); }
    const size_t fb = (size_t)H * W * 4;
    uint8_t *src = malloc(NF * fb);
    unsigned x = 12345 + id;
    for (size_t i = 0; i < NF * fb; i++) { x = x * 1664525u + 1013904223u; src[i] = (x >> 24) % 7; }
    int32_t *idx = malloc(sizeof(int32_t) * B * 4);
    for (int s = 0; s < S; s++)
        for (int j = 0; j < PER; j++)
            for (int k = 0; k < 4; k++) idx[(j * S + s) * 4 + k] = s * (PER + 3) + j + 3 - k;
    if (covahip_ctx_set_lanes(ctx, g_lanes)) exit(4);
    if (getenv("PROBE_IMPL") && covahip_blobnet_set_impl(ctx, atoi(getenv("PROBE_IMPL")))) exit(4);   /* developer switch: 4 = three-launch decoder */
    void *d_frames, *d_boxes[4], *d_counts[4];   // one output set per lane: concurrent calls must not share them
    if (covahip_malloc(ctx, NF * fb, &d_frames)) exit(4);
    for (int l = 0; l < 4; l++)
        if (covahip_malloc(ctx, (size_t)B * 2048 * 20, &d_boxes[l]) || covahip_malloc(ctx, B * 4, &d_counts[l])) exit(4);
    covahip_memcpy_h2d(ctx, d_frames, src, NF * fb);
    for (int k = 0; k < 12; k++)
        if (covahip_filter_forward_frames(ctx, d_frames, NF, idx, B, 1, d_boxes[k % g_lanes], d_counts[k % g_lanes], 2048, NULL, NULL, COVAHIP_MEM_DEVICE)) exit(5);
    covahip_ctx_sync(ctx);
    pthread_barrier_wait(&g_bar);
    for (int k = 0; k < g_steps; k++)
        if (covahip_filter_forward_frames(ctx, d_frames, NF, idx, B, 1, d_boxes[k % g_lanes], d_counts[k % g_lanes], 2048, NULL, NULL, COVAHIP_MEM_DEVICE)) exit(5);
    covahip_ctx_sync(ctx);
    pthread_barrier_wait(&g_bar);
    covahip_ctx_destroy(ctx);
    return NULL;
}

int main(int argc, char **argv) {
    if (argc < 5) { fprintf(stderr, "usage: %s weights.bin B N steps\n", argv[0]); return 1; }
    const int Btot = atoi(argv[2]), N = atoi(argv[3]);
    g_steps = atoi(argv[4]);
    if (argc > 5) g_lanes = atoi(argv[5]);
    g_b = Btot / N;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    fseek(f, 0, SEEK_END); g_blob_bytes = ftell(f); fseek(f, 0, SEEK_SET);
    g_blob = malloc(g_blob_bytes);
    if (fread(g_blob, 1, g_blob_bytes, f) != (size_t)g_blob_bytes) return 2;
    fclose(f);
    pthread_barrier_init(&g_bar, NULL, N + 1);
    pthread_t th[16];
    for (int i = 0; i < N; i++) pthread_create(&th[i], NULL, worker, (void *)(long)i);
    pthread_barrier_wait(&g_bar);
    const double t0 = now();
    pthread_barrier_wait(&g_bar);
    const double dt = now() - t0;
    for (int i = 0; i < N; i++) pthread_join(th[i], NULL);
    printf("{\"B_total\": %d, \"contexts\": %d, \"lanes\": %d, \"B_per_ctx\": %d, \"frames_per_s\": %.1f, \"us_per_B_total\": %.2f}\n", Btot, N, g_lanes, g_b,
           (double)g_steps * g_b * N / dt, dt / g_steps * 1e6);
    return 0;
}
