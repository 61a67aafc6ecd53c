// Direct (non-MFMA) HIP kernels for BlobNet: one thread per output element, fp32
// weights straight from the blob, fp16 channels-last activations in HBM.  This is the
// on-GPU bring-up / debugging path (COVAHIP_BLOBNET_IMPL=naive or
// covahip_blobnet_set_impl(ctx, 0)); the production path is blobnet_mfma.hip and both
// write the same buffers in the same layouts, so they can be compared layer by layer.
#include "blobnet.h"

namespace {

constexpr float BN_EPS = 1e-3f;  // Keras BatchNormalization default

// Encoder level (encoder.py:58-80): conv3x3 same + bias + relu -> BN -> maxpool 2x2 ->
// top/left zero pad (odd sizes) -> temporal 4->4->4 MLP + residual + relu.
// One thread per (b, pooled y, pooled x, cout); all 4 T slices.
//   FIRST: input is the u8 RGBA stack [B][T*H][W][4] with clip(x,0,6)/6 fused in.
//   else : input is fp16 [B][T][H][W][Cin].
// out: fp16 [B][To][Ho][Wo][Cout], To = 4 (or 1 when only_t0).
template <bool FIRST>
__global__ void enc_naive(const void *__restrict__ in, __half *__restrict__ out, EncParams p, int B, int H, int W,
                          int Cin, int Cout, int only_t0) {
    const int Hp = H / 2, Wp = W / 2, Ho = (H + 1) / 2, Wo = (W + 1) / 2, oy = H & 1, ox = W & 1;
    const long long total = (long long)B * Hp * Wp * Cout;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int co = (int)(gid % Cout);
    long long r = gid / Cout;
    const int wx = (int)(r % Wp);
    r /= Wp;
    const int wy = (int)(r % Hp);
    const int b = (int)(r / Hp);
    const float scale = p.gamma[co] * rsqrtf(p.var[co] + BN_EPS);
    const float shift = p.beta[co] - p.mean[co] * scale;
    const float bias = p.b[co];
    float pooled[BN_T];
    for (int t = 0; t < BN_T; t++) {
        float mx = -INFINITY;
        for (int dy = 0; dy < 2; dy++)
            for (int dx = 0; dx < 2; dx++) {
                const int y = 2 * wy + dy, x = 2 * wx + dx;
                float acc = bias;
                for (int ky = 0; ky < 3; ky++) {
                    const int yy = y + ky - 1;
                    if (yy < 0 || yy >= H) continue;
                    for (int kx = 0; kx < 3; kx++) {
                        const int xx = x + kx - 1;
                        if (xx < 0 || xx >= W) continue;
                        const float *wk = p.k + (size_t)((ky * 3 + kx) * Cin) * Cout + co;
                        if (FIRST) {
                            const uint8_t *px = (const uint8_t *)in + (((size_t)b * BN_T + t) * H * W + (size_t)yy * W + xx) * 4;
                            for (int ci = 0; ci < 3; ci++) {
                                float v = (float)px[ci];
                                v = fminf(v, 6.f) / 6.0f;
                                acc += v * wk[(size_t)ci * Cout];
                            }
                        } else {
                            const __half *px = (const __half *)in + (((size_t)b * BN_T + t) * H * W + (size_t)yy * W + xx) * Cin;
                            for (int ci = 0; ci < Cin; ci++) acc += __half2float(px[ci]) * wk[(size_t)ci * Cout];
                        }
                    }
                }
                acc = fmaxf(acc, 0.f) * scale + shift;
                mx = fmaxf(mx, acc);
            }
        pooled[t] = mx;
    }
    float u[BN_T], v[BN_T];
    for (int j = 0; j < BN_T; j++) {
        float a = 0.f;
        for (int t = 0; t < BN_T; t++) a += p.w1[t * BN_T + j] * pooled[t];
        u[j] = fmaxf(a, 0.f);
    }
    for (int j = 0; j < BN_T; j++) {
        float a = 0.f;
        for (int t = 0; t < BN_T; t++) a += p.w2[t * BN_T + j] * u[t];
        v[j] = fmaxf(a, 0.f);
    }
    const int To = only_t0 ? 1 : BN_T;
    for (int t = 0; t < To; t++) {
        const float o = fmaxf(v[t] + pooled[t], 0.f);
        out[((((size_t)b * To + t) * Ho + (wy + oy)) * Wo + (wx + ox)) * Cout + co] = __float2half(o);
    }
}

// Decoder block (decoder.py:5-75,122-131): relu -> convT 4x4 s2 valid + bias -> crop
// -> (BN).  out[y][x] = sum_{a,b in {0,1}} relu(in[(Y>>1)-a][(X>>1)-b]) * w[(Y&1)+2a][(X&1)+2b],
// (Y,X) = (y+cy, x+cx).  Input = concat(up [B][Hi][Wi][C1], skip [B][skipT][Hi][Wi][C2] (t=0)).
__device__ __forceinline__ float dec_point(const __half *up, const __half *skip, int skip_t, const float *k, int b,
                                           int Hi, int Wi, int C1, int C2, int Cout, int co, int Y, int X) {
    const int Cin = C1 + C2;
    float acc = 0.f;
    for (int a = 0; a < 2; a++) {
        const int iy = (Y >> 1) - a, ky = (Y & 1) + 2 * a;
        if (iy < 0 || iy >= Hi) continue;
        for (int bb = 0; bb < 2; bb++) {
            const int ix = (X >> 1) - bb, kx = (X & 1) + 2 * bb;
            if (ix < 0 || ix >= Wi) continue;
            const float *wk = k + ((size_t)(ky * 4 + kx) * Cout + co) * Cin;
            if (C1) {
                const __half *px = up + (((size_t)b * Hi + iy) * Wi + ix) * C1;
                for (int ci = 0; ci < C1; ci++) acc += fmaxf(__half2float(px[ci]), 0.f) * wk[ci];
            }
            const __half *ps = skip + ((((size_t)b * skip_t) * Hi + iy) * Wi + ix) * C2;
            for (int ci = 0; ci < C2; ci++) acc += fmaxf(__half2float(ps[ci]), 0.f) * wk[C1 + ci];
        }
    }
    return acc;
}

__global__ void dec_naive(const __half *__restrict__ up, const __half *__restrict__ skip, int skip_t,
                          __half *__restrict__ out, DecParams p, int B, int Hi, int Wi, int C1, int C2, int Cout,
                          int Hd, int Wd, int cy, int cx) {
    const long long total = (long long)B * Hd * Wd * Cout;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int co = (int)(gid % Cout);
    long long r = gid / Cout;
    const int x = (int)(r % Wd);
    r /= Wd;
    const int y = (int)(r % Hd);
    const int b = (int)(r / Hd);
    float acc = p.b[co] + dec_point(up, skip, skip_t, p.k, b, Hi, Wi, C1, C2, Cout, co, y + cy, x + cx);
    const float scale = p.gamma[co] * rsqrtf(p.var[co] + BN_EPS);
    const float shift = p.beta[co] - p.mean[co] * scale;
    // stored after the consumer's leading ReLU (its only consumer, decoder.py:9-11): same buffer
    // semantics as the MFMA path
    out[(((size_t)b * Hd + y) * Wd + x) * Cout + co] = __float2half(fmaxf(acc * scale + shift, 0.f));
}

// Last block + final 1x1 conv (decoder.py:131-134) + threshold (nvinfer segmentation
// threshold 0.5 == logit > 0; gstmaskcopy.cpp:226-230 class_map+1).
__global__ void final_naive(const __half *__restrict__ up, const __half *__restrict__ skip, int skip_t, DecParams p,
                            const float *__restrict__ fk, const float *__restrict__ fb, float *__restrict__ logits,
                            uint8_t *__restrict__ mask, int B, int Hi, int Wi, int C1, int C2, int Cout, int Hd,
                            int Wd, int cy, int cx) {
    const long long total = (long long)B * Hd * Wd;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    long long r = gid;
    const int x = (int)(r % Wd);
    r /= Wd;
    const int y = (int)(r % Hd);
    const int b = (int)(r / Hd);
    float logit = fb[0];
    for (int co = 0; co < Cout; co++) {
        const float v = p.b[co] + dec_point(up, skip, skip_t, p.k, b, Hi, Wi, C1, C2, Cout, co, y + cy, x + cx);
        logit += fk[co] * v;
    }
    if (logits) logits[gid] = logit;
    if (mask) mask[gid] = logit > 0.f ? 1 : 0;
}

}  // namespace

int blobnet_forward_naive(covahip_ctx *ctx, covahip_blobnet *m, const uint8_t *d_stack, int batch, float *d_logits,
                          uint8_t *d_mask) {
    const int TPB = 256;
    for (int i = 0; i < BN_LEVELS; i++) {
        const int H = m->lv[i].H, W = m->lv[i].W;
        const long long total = (long long)batch * (H / 2) * (W / 2) * m->enc_c[i + 1];
        const unsigned grid = (unsigned)((total + TPB - 1) / TPB);
        const int only_t0 = (i == BN_LEVELS - 1);
        ProfScope ps(ctx, "enc_naive");
        if (i == 0)
            hipLaunchKernelGGL(enc_naive<true>, dim3(grid), dim3(TPB), 0, ctx->stream, (const void *)d_stack, m->act[1],
                               m->enc[0], batch, H, W, 3, m->enc_c[1], only_t0);
        else
            hipLaunchKernelGGL(enc_naive<false>, dim3(grid), dim3(TPB), 0, ctx->stream, (const void *)m->act[i],
                               m->act[i + 1], m->enc[i], batch, H, W, m->enc_c[i], m->enc_c[i + 1], only_t0);
        COVAHIP_CHECK_HIP(ctx, hipGetLastError());
    }
    for (int j = 0; j < BN_LEVELS; j++) {
        const BnLevelGeom in = m->lv[BN_LEVELS - j], out = m->lv[BN_LEVELS - 1 - j];
        // sources: j == 0: skip = act[4] (t0 only, 128 ch), no up branch;
        //          j >= 1: up = dact[j-1], skip = act[4-j] (T slices, use t=0)
        const __half *up = j == 0 ? nullptr : m->dact[j - 1];
        const int C1 = j == 0 ? 0 : m->dec_co[j - 1];
        const __half *skip = m->act[BN_LEVELS - j];
        const int C2 = m->enc_c[BN_LEVELS - j];
        const int skip_t = j == 0 ? 1 : BN_T;
        if (j < BN_LEVELS - 1) {
            const long long total = (long long)batch * out.H * out.W * m->dec_co[j];
            const unsigned grid = (unsigned)((total + TPB - 1) / TPB);
            ProfScope ps(ctx, "dec_naive");
            hipLaunchKernelGGL(dec_naive, dim3(grid), dim3(TPB), 0, ctx->stream, up, skip, skip_t, m->dact[j], m->dec[j],
                               batch, in.H, in.W, C1, C2, m->dec_co[j], out.H, out.W, m->dec_cy[j], m->dec_cx[j]);
        } else {
            const long long total = (long long)batch * out.H * out.W;
            const unsigned grid = (unsigned)((total + TPB - 1) / TPB);
            ProfScope ps(ctx, "final_naive");
            hipLaunchKernelGGL(final_naive, dim3(grid), dim3(TPB), 0, ctx->stream, up, skip, skip_t, m->dec[j],
                               m->final_k, m->final_b, d_logits, d_mask, batch, in.H, in.W, C1, C2, m->dec_co[j], out.H,
                               out.W, m->dec_cy[j], m->dec_cx[j]);
        }
        COVAHIP_CHECK_HIP(ctx, hipGetLastError());
    }
    return COVAHIP_OK;
}
