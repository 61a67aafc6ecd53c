/*
 * oracle/blobnet_ref.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * Plain-C fp32 restatement of CoVA's BlobNet forward pass.  Nothing in the
 * product path (cova_amd/, libcovahip.so) may link, import or call this file;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * PARITY STATUS: "parity unpinned" against the reference itself -- the
 * reference model is Keras/TensorFlow (absent here), ships no weights and has
 * no tests or golden vectors for this path.  The restatement is pinned instead
 * against an independent torch.nn.functional composition of the same graph
 * (tests/golden/gen_blobnet_golden.py, run in the build container) whose
 * outputs are committed under tests/golden/.
 *
 * Reference files restated (paths under /root/reference):
 *   utils/model/preprocessing.py:6-7    clip(x,0,6)/6
 *   utils/model/encoder.py:30-80        Conv3D(1,3,3,same,bias,relu) -> BN(axis=1)
 *                                       -> MaxPool3D((1,2,2),valid) -> ZeroPadding
 *                                       top/left when the pre-pool dim is odd
 *                                       -> PointWiseTN
 *   utils/model/pointwise.py:8-26       T-as-channel 4->4->4 MLP (relu, no bias),
 *                                       residual, relu
 *   utils/model/blobnet.py:32           skips = level outputs, T index 0 only
 *   utils/model/decoder.py:5-75,122-134 ReLU -> Conv3DTranspose((1,4,4),s=(1,2,2),
 *                                       valid,bias) -> Cropping3D -> BN -> concat;
 *                                       last block: no BN/concat; Conv3D(1,1);
 *                                       sigmoid
 *   utils/train-blobnet.py:57-69        channels 16/32/64/128, tmix [4,4],
 *                                       decoder 64/32/16/16
 *   utils/train-blobnet.py:113-116      export reshape [3, T*H, W] -> [3,T,H,W]
 *   cova-rs/gst-plugins/src/tfrecordsink/imp.rs:105-112  byte 0/1/2 = mb_type/mv_x/mv_y
 *   config/blobnet/amsterdam_b128.txt:7-9,20-26 + gst-plugins/gst-maskcopy/
 *   gstmaskcopy.cpp:226-230             mask = (p > 0.5) ? 1 : 0   (== logit > 0)
 *
 * Weight blob layout (all fp32, this build's own format; the reference has no
 * weight files): see cova_amd/weights.py.  Order:
 *   for i in 0..3: enc_i conv kernel [3][3][Cin][Cout], bias [Cout],
 *                  bn gamma/beta/mean/var [Cout] x4, tmix w1 [4][4] (tin,tout),
 *                  tmix w2 [4][4]
 *   for j in 0..3: dec_j convT kernel [4][4][Cout][Cin] (Keras layout), bias
 *                  [Cout], (j<3) bn gamma/beta/mean/var [Cout] x4
 *   final kernel [16], final bias [1]
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NLEV 4
#define TT 4
#define BN_EPS 1e-3f /* Keras BatchNormalization default epsilon */

static const int ENC_C[NLEV + 1] = {3, 16, 32, 64, 128};
static const int DEC_CO[NLEV] = {64, 32, 16, 16};
static const int DEC_CI[NLEV] = {128, 128, 64, 32};

typedef struct {
    const float *k, *b, *gamma, *beta, *mean, *var, *w1, *w2;
} enc_w;
typedef struct {
    const float *k, *b, *gamma, *beta, *mean, *var;
} dec_w;

static size_t parse_weights(const float *w, enc_w *e, dec_w *d, const float **fk, const float **fb) {
    const float *p = w;
    for (int i = 0; i < NLEV; i++) {
        int ci = ENC_C[i], co = ENC_C[i + 1];
        e[i].k = p; p += 9 * ci * co;
        e[i].b = p; p += co;
        e[i].gamma = p; p += co;
        e[i].beta = p; p += co;
        e[i].mean = p; p += co;
        e[i].var = p; p += co;
        e[i].w1 = p; p += 16;
        e[i].w2 = p; p += 16;
    }
    for (int j = 0; j < NLEV; j++) {
        int ci = DEC_CI[j], co = DEC_CO[j];
        d[j].k = p; p += 16 * ci * co;
        d[j].b = p; p += co;
        if (j < NLEV - 1) {
            d[j].gamma = p; p += co;
            d[j].beta = p; p += co;
            d[j].mean = p; p += co;
            d[j].var = p; p += co;
        } else {
            d[j].gamma = d[j].beta = d[j].mean = d[j].var = NULL;
        }
    }
    *fk = p; p += 16;
    *fb = p; p += 1;
    return (size_t)(p - w);
}

size_t cova_ref_blobnet_num_params(void) {
    enc_w e[NLEV]; dec_w d[NLEV]; const float *fk, *fb;
    return parse_weights((const float *)0, e, d, &fk, &fb);
}

/* One encoder level on one frame.  in: [Cin][T][H][W]; out: [Cout][T][Ho][Wo]
 * with Ho = ceil(H/2), Wo = ceil(W/2) (encoder.py:58-80). */
static void enc_level(const enc_w *w, int ci, int co, int H, int W, const float *in, float *out,
                      float *conv /* scratch [H][W] */) {
    const int Hp = H / 2, Wp = W / 2;        /* MaxPool valid */
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const int oy = H & 1, ox = W & 1;        /* zero row on top / zero col on left */
    memset(out, 0, sizeof(float) * (size_t)co * TT * Ho * Wo);
    for (int c = 0; c < co; c++) {
        const float scale = w->gamma[c] / sqrtf(w->var[c] + BN_EPS);
        const float shift = w->beta[c] - w->mean[c] * scale;
        for (int t = 0; t < TT; t++) {
            for (int i = 0; i < H * W; i++) conv[i] = w->b[c];
            for (int k = 0; k < ci; k++) {
                const float *src = in + ((size_t)k * TT + t) * H * W;
                for (int ky = 0; ky < 3; ky++)
                    for (int kx = 0; kx < 3; kx++) {
                        const float wv = w->k[((ky * 3 + kx) * ci + k) * co + c];
                        const int y0 = ky == 0 ? 1 : 0, y1 = ky == 2 ? H - 1 : H;
                        const int x0 = kx == 0 ? 1 : 0, x1 = kx == 2 ? W - 1 : W;
                        for (int y = y0; y < y1; y++) {
                            const float *s = src + (y + ky - 1) * W + (kx - 1);
                            float *dst = conv + y * W;
                            for (int x = x0; x < x1; x++) dst[x] += wv * s[x];
                        }
                    }
            }
            /* relu -> BN affine -> 2x2 max pool -> shifted write */
            float *o = out + ((size_t)c * TT + t) * Ho * Wo;
            for (int y = 0; y < Hp; y++)
                for (int x = 0; x < Wp; x++) {
                    float m = -INFINITY;
                    for (int dy = 0; dy < 2; dy++)
                        for (int dx = 0; dx < 2; dx++) {
                            float v = conv[(2 * y + dy) * W + 2 * x + dx];
                            v = v > 0.f ? v : 0.f;
                            v = v * scale + shift;
                            m = v > m ? v : m;
                        }
                    o[(y + oy) * Wo + x + ox] = m;
                }
        }
        /* PointWiseTN over the T axis (pointwise.py:16-26) */
        for (int i = 0; i < Ho * Wo; i++) {
            float p[TT], u[TT], v[TT];
            for (int t = 0; t < TT; t++) p[t] = out[((size_t)c * TT + t) * Ho * Wo + i];
            for (int j = 0; j < TT; j++) {
                float a = 0.f;
                for (int t = 0; t < TT; t++) a += w->w1[t * TT + j] * p[t];
                u[j] = a > 0.f ? a : 0.f;
            }
            for (int j = 0; j < TT; j++) {
                float a = 0.f;
                for (int t = 0; t < TT; t++) a += w->w2[t * TT + j] * u[t];
                v[j] = a > 0.f ? a : 0.f;
            }
            for (int t = 0; t < TT; t++) {
                float a = v[t] + p[t];
                out[((size_t)c * TT + t) * Ho * Wo + i] = a > 0.f ? a : 0.f;
            }
        }
    }
}

/* One decoder up-sampling block on one frame.  in: [Cin][Hi][Wi] -> out rows
 * [0,Cout) of a [Ctot][Hd][Wd] tensor (decoder.py:5-75). */
static void dec_level(const dec_w *w, int ci, int co, int Hi, int Wi, int Hd, int Wd, const float *in,
                      float *out, float *full /* scratch [(2Hi+2)][(2Wi+2)] */) {
    const int Hf = 2 * Hi + 2, Wf = 2 * Wi + 2;
    const int ph = Hf - Hd, pw = Wf - Wd;
    const int cy = ph / 2 + ph % 2, cx = pw / 2 + pw % 2; /* crop top/left = ceil */
    for (int c = 0; c < co; c++) {
        for (int i = 0; i < Hf * Wf; i++) full[i] = w->b[c];
        for (int k = 0; k < ci; k++) {
            const float *src = in + (size_t)k * Hi * Wi;
            for (int ky = 0; ky < 4; ky++)
                for (int kx = 0; kx < 4; kx++) {
                    const float wv = w->k[((ky * 4 + kx) * co + c) * ci + k];
                    for (int y = 0; y < Hi; y++)
                        for (int x = 0; x < Wi; x++) {
                            float v = src[y * Wi + x];
                            v = v > 0.f ? v : 0.f; /* leading ReLU of the block */
                            full[(2 * y + ky) * Wf + 2 * x + kx] += wv * v;
                        }
                }
        }
        float scale = 1.f, shift = 0.f;
        if (w->gamma) {
            scale = w->gamma[c] / sqrtf(w->var[c] + BN_EPS);
            shift = w->beta[c] - w->mean[c] * scale;
        }
        float *o = out + (size_t)c * Hd * Wd;
        for (int y = 0; y < Hd; y++)
            for (int x = 0; x < Wd; x++) o[y * Wd + x] = full[(y + cy) * Wf + x + cx] * scale + shift;
    }
}

/* rgba_stack: u8 [B][T*H][W][4] (metapreprocess output, row block t = frame k-t)
 * logits:     f32 [B][H][W]; mask (optional): u8 [B][H][W] in {0,1}
 * returns 0 on success. */
int cova_ref_blobnet_forward(const float *weights, int H, int W, const uint8_t *rgba_stack, int B,
                             float *logits, uint8_t *mask) {
    enc_w e[NLEV]; dec_w d[NLEV]; const float *fk, *fb;
    parse_weights(weights, e, d, &fk, &fb);
    int hs[NLEV + 1], ws[NLEV + 1];
    hs[0] = H; ws[0] = W;
    for (int i = 0; i < NLEV; i++) { hs[i + 1] = (hs[i] + 1) / 2; ws[i + 1] = (ws[i] + 1) / 2; }

    int rc = 0;
#pragma omp parallel for schedule(dynamic)
    for (int b = 0; b < B; b++) {
        float *act[NLEV + 1];
        size_t maxhw = (size_t)(2 * hs[1] + 2) * (2 * ws[1] + 2);
        if (maxhw < (size_t)H * W) maxhw = (size_t)H * W;
        float *scratch = (float *)malloc(sizeof(float) * maxhw);
        for (int i = 0; i <= NLEV; i++)
            act[i] = (float *)malloc(sizeof(float) * (size_t)ENC_C[i] * TT * hs[i] * ws[i]);
        /* B0/B1: [T*H][W][4] u8 -> [3][T][H][W], clip(x,0,6)/6 */
        const uint8_t *src = rgba_stack + (size_t)b * TT * H * W * 4;
        for (int c = 0; c < 3; c++)
            for (int t = 0; t < TT; t++)
                for (int i = 0; i < H * W; i++) {
                    float v = (float)src[((size_t)t * H * W + i) * 4 + c];
                    v = v < 0.f ? 0.f : (v > 6.f ? 6.f : v);
                    act[0][((size_t)c * TT + t) * H * W + i] = v / 6.0f;
                }
        for (int i = 0; i < NLEV; i++)
            enc_level(&e[i], ENC_C[i], ENC_C[i + 1], hs[i], ws[i], act[i], act[i + 1], scratch);

        /* decoder: x = skip_3 (T index 0); for j: up -> bn -> concat skip */
        int ch = ENC_C[NLEV];
        int Hi = hs[NLEV], Wi = ws[NLEV];
        float *x = (float *)malloc(sizeof(float) * (size_t)ch * Hi * Wi);
        for (int c = 0; c < ch; c++)
            memcpy(x + (size_t)c * Hi * Wi, act[NLEV] + ((size_t)c * TT + 0) * Hi * Wi, sizeof(float) * Hi * Wi);
        for (int j = 0; j < NLEV; j++) {
            const int lvl = NLEV - 1 - j;          /* spatial level of the output */
            const int Hd = hs[lvl], Wd = ws[lvl];
            const int co = DEC_CO[j];
            const int cskip = j < NLEV - 1 ? ENC_C[lvl] : 0;
            float *y = (float *)malloc(sizeof(float) * (size_t)(co + cskip) * Hd * Wd);
            dec_level(&d[j], DEC_CI[j], co, Hi, Wi, Hd, Wd, x, y, scratch);
            for (int c = 0; c < cskip; c++)
                memcpy(y + (size_t)(co + c) * Hd * Wd, act[lvl] + ((size_t)c * TT + 0) * Hd * Wd,
                       sizeof(float) * Hd * Wd);
            free(x);
            x = y; Hi = Hd; Wi = Wd;
        }
        /* final 1x1 conv 16 -> 1 (decoder.py:118,132) */
        for (int i = 0; i < H * W; i++) {
            float a = fb[0];
            for (int c = 0; c < DEC_CO[NLEV - 1]; c++) a += fk[c] * x[(size_t)c * H * W + i];
            logits[(size_t)b * H * W + i] = a;
            if (mask) mask[(size_t)b * H * W + i] = a > 0.f ? 1 : 0;
        }
        free(x);
        for (int i = 0; i <= NLEV; i++) free(act[i]);
        free(scratch);
    }
    return rc;
}

/* Debug hook for layer-level parity tests: returns encoder level `lvl` output
 * (0..3) of frame 0 as [Cout][T][Ho][Wo]. */
int cova_ref_blobnet_encoder_level(const float *weights, int H, int W, const uint8_t *rgba_stack, int lvl,
                                   float *out) {
    enc_w e[NLEV]; dec_w d[NLEV]; const float *fk, *fb;
    parse_weights(weights, e, d, &fk, &fb);
    int hs[NLEV + 1], ws[NLEV + 1];
    hs[0] = H; ws[0] = W;
    for (int i = 0; i < NLEV; i++) { hs[i + 1] = (hs[i] + 1) / 2; ws[i + 1] = (ws[i] + 1) / 2; }
    if (lvl < 0 || lvl >= NLEV) return -1;
    float *scratch = (float *)malloc(sizeof(float) * (size_t)H * W);
    float *cur = (float *)malloc(sizeof(float) * (size_t)3 * TT * H * W);
    for (int c = 0; c < 3; c++)
        for (int t = 0; t < TT; t++)
            for (int i = 0; i < H * W; i++) {
                float v = (float)rgba_stack[((size_t)t * H * W + i) * 4 + c];
                v = v > 6.f ? 6.f : v;
                cur[((size_t)c * TT + t) * H * W + i] = v / 6.0f;
            }
    for (int i = 0; i <= lvl; i++) {
        float *nxt = (float *)malloc(sizeof(float) * (size_t)ENC_C[i + 1] * TT * hs[i + 1] * ws[i + 1]);
        enc_level(&e[i], ENC_C[i], ENC_C[i + 1], hs[i], ws[i], cur, nxt, scratch);
        free(cur);
        cur = nxt;
    }
    memcpy(out, cur, sizeof(float) * (size_t)ENC_C[lvl + 1] * TT * hs[lvl + 1] * ws[lvl + 1]);
    free(cur);
    free(scratch);
    return 0;
}
