// Developer probe: does PointWiseTN computed on v_mfma_f32_16x16x16_f16 with block-diagonal weights (data in its natural
// [T][pixel][channel] layout, no 16-bit transposes) give the SAME BITS as the v_mfma_f32_4x4x4_16b_f16 form (tmix_core)?
// Also: pipe cycles of the two instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// form A: one element (4 T values) per lane
__global__ void formA(const _Float16 *x /*[N][4]*/, const float *w /*w1[16] w2[16]*/, float *out /*[N][4]*/, int N) {
    const int lane = threadIdx.x & 63, i = lane & 3;
    half4 a1, a2, id;
    for (int t = 0; t < 4; t++) { a1[t] = (_Float16)w[t * 4 + i]; a2[t] = (_Float16)w[16 + t * 4 + i]; id[t] = (_Float16)(t == i ? 1.f : 0.f); }
    const f32x4 z = {0, 0, 0, 0};
    const half4 hz = {0, 0, 0, 0};
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < N; e += gridDim.x * blockDim.x) {
        half4 p = *reinterpret_cast<const half4 *>(x + 4 * e);
        f32x4 u = __builtin_amdgcn_mfma_f32_4x4x4f16(a1, p, z, 0, 0, 0);
        f32x4 pv = __builtin_amdgcn_mfma_f32_4x4x4f16(id, p, z, 0, 0, 0);
        half4 ub = __builtin_elementwise_max(__builtin_convertvector(u, half4), hz);
        f32x4 s = __builtin_amdgcn_mfma_f32_4x4x4f16(a2, ub, pv, 0, 0, 0);
        for (int t = 0; t < 4; t++) out[4 * e + t] = fmaxf(fmaxf(s[t], pv[t]), 0.f);
    }
}
// form B: 16x16x16, K = (T, c), M = (T', c'), N = 16 pixels; lane (n = l & 15, g = l >> 4): B halves = channels 0..3 of T = g at
// pixel n; A[i = (T', c')][k = (T, c)] = W[T][T'] * (c == c'): lane (i = l & 15, g = l >> 4) holds k = 4 g + j -> T = g, c = j
__global__ void formB(const _Float16 *xt /*[4 T][NP pixels][4 ch]*/, const float *w, float *out /*[4 T'][NP][4 ch]*/, int NP) {
    const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
    const int Tp = (lane & 15) >> 2, cp = lane & 3;   // row i = (T', c')
    half4 a1, a2, id;
    for (int j = 0; j < 4; j++) {   // k = (T = g, c = j)
        a1[j] = (_Float16)(j == cp ? w[g * 4 + Tp] : 0.f);          // W1[T][T']
        a2[j] = (_Float16)(j == cp ? w[16 + g * 4 + Tp] : 0.f);
        id[j] = (_Float16)((j == cp && g == Tp) ? 1.f : 0.f);
    }
    const f32x4 z = {0, 0, 0, 0};
    const half4 hz = {0, 0, 0, 0};
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
    for (int p0 = wave * 16; p0 < NP; p0 += nw * 16) {
        half4 p = *reinterpret_cast<const half4 *>(xt + ((size_t)g * NP + p0 + n) * 4);
        f32x4 u = __builtin_amdgcn_mfma_f32_16x16x16f16(a1, p, z, 0, 0, 0);     // D: col n, row 4 g + r = (T' = g, c' = r)
        f32x4 pv = __builtin_amdgcn_mfma_f32_16x16x16f16(id, p, z, 0, 0, 0);
        half4 ub = __builtin_elementwise_max(__builtin_convertvector(u, half4), hz);
        f32x4 s = __builtin_amdgcn_mfma_f32_16x16x16f16(a2, ub, pv, 0, 0, 0);
        for (int r = 0; r < 4; r++) out[((size_t)g * NP + p0 + n) * 4 + r] = fmaxf(fmaxf(s[r], pv[r]), 0.f);
    }
}
template <int KIND>
__global__ void rate(unsigned long long *o) {
    half4 a = {1, 2, 3, 4}, b = {1, 1, 1, 1};
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < 1000; t++)
#pragma unroll
        for (int k = 0; k < 8; k++)
            acc[k & 3] = KIND == 0 ? __builtin_amdgcn_mfma_f32_4x4x4f16(a, b, acc[k & 3], 0, 0, 0) : __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, acc[k & 3], 0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { o[0] = t1 - t0; o[1] = (unsigned long long)(acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0]); }
}
int main() {
    const int NP = 1 << 16, N = NP * 4;   // N elements (pixel, channel); 4 T each
    std::vector<_Float16> x(N * 4), xt(N * 4);
    std::vector<float> w(32);
    srand(7);
    for (auto &v : w) v = (rand() / (float)RAND_MAX - 0.5f) * 1.7f;
    for (int e = 0; e < N; e++)
        for (int t = 0; t < 4; t++) {
            const float v = (rand() % 7 == 0) ? 0.f : (rand() / (float)RAND_MAX) * 3.f;   // pooled post-BN values, some zeros
            x[4 * e + t] = (_Float16)(v - 0.4f);
            const int pix = e >> 2, c = e & 3;
            xt[((size_t)t * NP + pix) * 4 + c] = x[4 * e + t];
        }
    _Float16 *dx, *dxt; float *dw, *oa, *ob; unsigned long long *dr;
    hipMalloc(&dx, N * 8); hipMalloc(&dxt, N * 8); hipMalloc(&dw, 128); hipMalloc(&oa, N * 16); hipMalloc(&ob, N * 16); hipMalloc(&dr, 64);
    hipMemcpy(dx, x.data(), N * 8, hipMemcpyHostToDevice); hipMemcpy(dxt, xt.data(), N * 8, hipMemcpyHostToDevice);
    hipMemcpy(dw, w.data(), 128, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(formA, dim3(256), dim3(256), 0, 0, dx, dw, oa, N);
    hipLaunchKernelGGL(formB, dim3(256), dim3(256), 0, 0, dxt, dw, ob, NP);
    std::vector<float> ha(N * 4), hb(N * 4);
    hipMemcpy(ha.data(), oa, N * 16, hipMemcpyDeviceToHost); hipMemcpy(hb.data(), ob, N * 16, hipMemcpyDeviceToHost);
    long diff = 0; double maxd = 0;
    for (int e = 0; e < N; e++)
        for (int t = 0; t < 4; t++) {
            const int pix = e >> 2, c = e & 3;
            const float a = ha[4 * e + t], b = hb[((size_t)t * NP + pix) * 4 + c];
            if (a != b) { diff++; maxd = fmax(maxd, fabs(a - b)); }
        }
    printf("elements %d x 4 T: %ld values differ (max |d| %g); sample %g %g\n", N, diff, maxd, ha[5], hb[5]);
    for (int k = 0; k < 2; k++) {
        if (k == 0) hipLaunchKernelGGL(rate<0>, dim3(1), dim3(64), 0, 0, dr); else hipLaunchKernelGGL(rate<1>, dim3(1), dim3(64), 0, 0, dr);
        unsigned long long h[2]; hipMemcpy(h, dr, 16, hipMemcpyDeviceToHost);
        printf("%s: %.1f cycles per instruction (one wave)\n", k == 0 ? "v_mfma_f32_4x4x4_16b_f16" : "v_mfma_f32_16x16x16_f16", h[0] / 8000.0);
    }
    return 0;
}
