/* native_chain_bench.c -- BASELINE config 4 WITHOUT GStreamer: what the C-ABI itself costs the host per frame.
 * S streams of packed two-byte records -> covahip_pipe (pinned slots, stacking as a GPU gather, BlobNet + bboxcc, packed boxes) ->
 * per stream covahip_gopfilter (embedded SORT + GoP frame filter, the experiment's parameters: maxage 60 / minhits 30 / iou 0.1,
 * experiment/cova/config.yaml:59,67), access units fed ahead of the masks as in pipeline/cova/pipeline.py:237-253.
 * An element written against include/covahip.h (the reference's Rust elements) pays this plus its own framework's
 * per-buffer work; the GStreamer plugin of this repo pays ~11 us of core time per frame, of which this is the library's share.
 *
 * One OpenMP team of T threads fills a slot (a stream's new frames are copied by one thread), then thread 0 submits it while the
 * others track the batch that has just come back (a stream's boxes go through its gopfilter on one thread, in order).
 *   gcc -O2 -fopenmp -Iinclude tools/native_chain_bench.c -o tools/native_chain_bench -Lcova_amd -lcovahip -Wl,-rpath,$PWD/cova_amd -lm
 *   tools/native_chain_bench <weights blob> [batches] [streams] [threads]
 * Prints one JSON line: frames/s, CPU microseconds per frame (process CPU time / frames) split into fill / track / submit+wait. */
#define _GNU_SOURCE
#include <math.h>
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/resource.h>
#include <time.h>

#include "covahip.h"

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static double cpu_now(void) { struct rusage r; getrusage(RUSAGE_SELF, &r); return r.ru_utime.tv_sec + r.ru_stime.tv_sec + 1e-6 * (r.ru_utime.tv_usec + r.ru_stime.tv_usec); }
static double tcpu(void) { struct timespec t; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

enum { H = 68, W = 120, B = 256, CYCLE = 256, MAXB = 256 };

/* a stream's cycle of frames: a few ellipses of motion bouncing across the grid over sparse noise (periodic in the cycle), as the
 * GStreamer chain bench feeds (gst/gst_element_driver.c, chain_make_frames), packed to two bytes per macroblock */
static void make_stream(uint16_t *dst, unsigned seed) {
    struct { double lx, ly, rx, ry, px, py; int mx, my; } ob[5];
    unsigned x = seed * 2654435761u + 12345u;
#define RND() (x = x * 1664525u + 1013904223u, (x >> 8) & 0xFFFF)
    const int nob = 3 + RND() % 3;
    for (int j = 0; j < nob; j++) {
        ob[j].rx = 2.0 + RND() % 40 / 10.0; ob[j].ry = 2.0 + RND() % 40 / 10.0;
        ob[j].lx = W - 1; ob[j].ly = H - 1;
        ob[j].mx = 1 + RND() % 3; ob[j].my = RND() % 3;
        ob[j].px = RND() % 1000 / 500.0; ob[j].py = RND() % 1000 / 500.0;
    }
    for (int k = 0; k < CYCLE; k++) {
        uint16_t *f = dst + (size_t)k * H * W;
        for (int q = 0; q < H * W; q++) {
            const unsigned r = RND();
            const unsigned c = (r & 7) < 2 ? (r >> 4) % 7 : 0, mx = (r >> 8) % 10 == 0 ? 1 + (r >> 3) % 3 : 0, my = (r >> 12) % 10 == 0 ? 1 + (r >> 5) % 3 : 0;
            f[q] = (uint16_t)((c > 6 ? 6 : c) | mx << 3 | my << 6);
        }
        for (int j = 0; j < nob; j++) {
            double tx = ob[j].px + 2.0 * ob[j].mx * k / CYCLE, ty = ob[j].py + 2.0 * ob[j].my * k / CYCLE;
            tx -= 2.0 * floor(tx / 2.0); ty -= 2.0 * floor(ty / 2.0);
            const double cx = ob[j].lx * (tx < 1.0 ? tx : 2.0 - tx), cy = ob[j].ly * (ty < 1.0 ? ty : 2.0 - ty);
            for (int yy = (int)(cy - ob[j].ry) - 1; yy <= (int)(cy + ob[j].ry) + 1; yy++)
                for (int xx = (int)(cx - ob[j].rx) - 1; xx <= (int)(cx + ob[j].rx) + 1; xx++) {
                    if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                    const double dx = (xx - cx) / ob[j].rx, dy = (yy - cy) / ob[j].ry;
                    if (dx * dx + dy * dy > 1.0) continue;
                    const unsigned r = RND();
                    unsigned c = 1 + r % 7, mx = 1 + (r >> 4) % 12, my = 1 + (r >> 8) % 12;
                    f[yy * W + xx] = (uint16_t)((c > 6 ? 6 : c) | (mx > 6 ? 6 : mx) << 3 | (my > 6 ? 6 : my) << 6);
                }
        }
    }
#undef RND
}

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s weights.bin [batches] [streams] [threads]\n", argv[0]); return 1; }
    const int batches = argc > 2 ? atoi(argv[2]) : 2000, S = argc > 3 ? atoi(argv[3]) : 16, T = argc > 4 ? atoi(argv[4]) : 16;
    if (S < 1 || S > 64 || B % S) { fprintf(stderr, "streams must divide %d\n", B); return 1; }
    const int PER = B / S, NF = S * (PER + 3), NSLOT = 6;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    void *blob = malloc(n);
    if (fread(blob, 1, n, f) != (size_t)n) return 2;
    fclose(f);
    covahip_ctx *ctx; covahip_pipe *pipe;
    if (covahip_ctx_create(0, &ctx) || covahip_ctx_set_lanes(ctx, 3) || covahip_blobnet_load(ctx, blob, n, H, W, 4, B)) return 3;
    if (covahip_pipe_create(ctx, B, NF, MAXB, NSLOT, 0, &pipe) || covahip_pipe_set_packed(pipe, 1) || covahip_pipe_set_blocking_wait(pipe, 1)) return 4;
    omp_set_num_threads(T);
    uint16_t **src = malloc(sizeof(*src) * S);
    covahip_gopfilter **gf = malloc(sizeof(*gf) * S);
    covahip_gopfilter_cfg cfg;
    covahip_gopfilter_default_cfg(&cfg);
    cfg.sort_iou = 0.1f; cfg.sort_maxage = 60; cfg.sort_minhits = 30;
#pragma omp parallel for schedule(static)
    for (int s = 0; s < S; s++) {
        src[s] = malloc((size_t)CYCLE * H * W * 2);
        make_stream(src[s], 1000u + (unsigned)s);
        if (covahip_gopfilter_new(&cfg, &gf[s])) exit(7);
    }
    const uint64_t clk = 1000000000ull / 30;
    const long total_frames = (long)batches * PER;           /* per stream */
    long enc_pushed = 0;                                      /* access units fed so far, per stream (the same for all) */
    const int LEAD = 600;                                     /* the encoded branch runs this far ahead of the masks */
    int inflight[8], infirst[8], nin = 0;
    long aus = 0, nboxes = 0;
    double cpu_fill = 0, cpu_track = 0, t0 = 0, c0 = 0, w_submit = 0, w_collect = 0, w_track = 0, w_fill = 0;
    const int warm = 8;
    for (int k = -warm; k < batches + NSLOT; k++) {
        if (k == 0) { t0 = now(); c0 = cpu_now(); cpu_fill = cpu_track = 0; aus = nboxes = 0; w_submit = w_collect = w_track = w_fill = 0; }
        const int feeding = k < batches;                     /* the last NSLOT rounds only drain */
        const long first = (long)(k + warm) * PER;           /* first new frame of this batch, per stream */
        /* the oldest batch comes back when NSLOT - 2 are in flight (or nothing is left to feed): its boxes are tracked below, while
         * thread 0 submits the batch that is filled now */
        const int32_t *cnt = NULL, *off = NULL; const covahip_box *bx = NULL;
        int have = 0, got_slot = -1;
        long bfirst = 0;
        if (nin && (nin >= NSLOT - 2 || !feeding)) {
            const double wc0 = now();
            if (covahip_pipe_collect(pipe, inflight[0], &cnt, &off, &bx, NULL)) return 5;
            w_collect += now() - wc0;
            have = 1; got_slot = inflight[0]; bfirst = (long)infirst[0];
            memmove(inflight, inflight + 1, sizeof(int) * --nin);
            memmove(infirst, infirst + 1, sizeof(int) * nin);
        }
        int slot = -1; uint8_t *pf = NULL; int32_t *pi = NULL;
        const double wf0 = now();
        if (feeding) {
            if (covahip_pipe_acquire(pipe, &slot, &pf, &pi)) return 9;
            /* access units up to LEAD frames ahead of this batch; this stream's frames of the batch: three of history + PER new ones */
            const long want = first + PER + LEAD < total_frames + (long)warm * PER ? first + PER + LEAD : total_frames + (long)warm * PER;
            double cf = 0;
#pragma omp parallel for schedule(static) reduction(+ : cf)
            for (int s = 0; s < S; s++) {
                const double c1 = tcpu();
                for (long i = enc_pushed; i < want; i++) covahip_gopfilter_push_enc(gf[s], (uint64_t)i + 1, (uint64_t)i * clk, i % 250 ? COVAHIP_AU_DELTA_UNIT : 0);
                uint16_t *dst = (uint16_t *)pf + (size_t)s * (PER + 3) * H * W;
                for (int j = 0; j < PER + 3; j++) {
                    const long fr = first - 3 + j;
                    if (fr < 0) memset(dst + (size_t)j * H * W, 0, (size_t)H * W * 2);
                    else memcpy(dst + (size_t)j * H * W, src[s] + (size_t)(fr % CYCLE) * H * W, (size_t)H * W * 2);
                }
                for (int j = 0; j < PER; j++)
                    for (int t = 0; t < 4; t++) pi[(j * S + s) * 4 + t] = s * (PER + 3) + j + 3 - t;
                cf += tcpu() - c1;
            }
            cpu_fill += cf;
            enc_pushed = want;
        }
        const double ws0 = now();
        w_fill += ws0 - wf0;
        double ct = 0, wsub = 0;
        long a2 = 0, b2 = 0;
        int rc_submit = 0;
#pragma omp parallel reduction(+ : ct, a2, b2)
        {
#pragma omp master
            if (feeding) {
                const double w0 = now();
                rc_submit = covahip_pipe_submit(pipe, slot, NF, B, 1);
                wsub = now() - w0;
            }
            if (have) {
#pragma omp for schedule(static) nowait   /* a stream stays on its thread: its tracker state (~130 young trackers) stays in that core's cache -- a dynamic schedule tripled the tracking time */
                for (int s = 0; s < S; s++) {
                    const double c1 = tcpu();
                    covahip_bbox bb[MAXB];
                    covahip_au_out out[1024];
                    uint64_t dropped[1024];
                    for (int j = 0; j < PER; j++) {
                        const int b = j * S + s;                 /* stacks are interleaved stream by stream */
                        const int nb = cnt[b] < MAXB ? cnt[b] : MAXB;
                        covahip_boxes_to_bbox(bx + off[b], nb, bb);
                        size_t no = 0, nd = 0;
                        if (covahip_gopfilter_push_boxes(gf[s], bb, (size_t)nb, (uint64_t)(bfirst + j) * clk, out, 1024, &no)) exit(8);
                        a2 += (long)no; b2 += nb;
                        do { covahip_gopfilter_take_dropped(gf[s], dropped, 1024, &nd); } while (nd == 1024);
                    }
                    ct += tcpu() - c1;
                }
            }
        }
        if (rc_submit) { fprintf(stderr, "submit: %s\n", covahip_last_hip_error(ctx)); return 6; }
        w_submit += wsub;
        w_track += now() - ws0;
        cpu_track += ct; aus += a2; nboxes += b2;
        if (have) covahip_pipe_release(pipe, got_slot);
        if (feeding) { inflight[nin] = slot; infirst[nin++] = (int)first; }
    }
    const double dt = now() - t0, dc = cpu_now() - c0;
    const double frames = (double)batches * B;
    printf("{\"frames_per_s_native_chain\": %.1f, \"streams\": %d, \"threads\": %d, \"batches\": %d, \"seconds\": %.4f, "
           "\"cpu_us_per_frame\": %.2f, \"cpu_us_per_frame_fill\": %.2f, \"cpu_us_per_frame_track\": %.2f, \"boxes_per_frame\": %.2f, "
           "\"wall_us_per_batch\": {\"collect_wait\": %.1f, \"submit_and_track_together\": %.1f, \"fill\": %.1f, \"submit_alone\": %.1f}, \"aus_forwarded\": %ld, \"note\": \"C-ABI only (covahip_pipe + covahip_gopfilter), no GStreamer; cpu = process CPU time of the timed batches\"}\n",
           frames / dt, S, T, batches, dt, dc / frames * 1e6, cpu_fill / frames * 1e6, cpu_track / frames * 1e6, (double)nboxes / ((double)batches * B), w_collect / batches * 1e6, w_track / batches * 1e6, w_fill / batches * 1e6, w_submit / batches * 1e6, aus);
    return 0;
}
