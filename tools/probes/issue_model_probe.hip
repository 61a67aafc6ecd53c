// Developer probe: what one SIMD of gfx950 issues per cycle when SEVERAL waves share it.
// One workgroup on one CU; W waves per SIMD (W = 1, 2, 4, 8); every wave runs the same loop of independent instructions of one
// mix; wave 0 stamps s_memtime around its loop.  Reported: cycles per loop trip of wave 0 (all waves run concurrently, so the
// SIMD's throughput is W x instructions-per-trip / that).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/issue_model_probe.hip -o tools/probes/issue_model_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define VMAX3(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define VCVT(i) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define VADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define DSR(i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[i & 3]) : "v"(lds_a), "n"((i) * 1024));

template <int KIND>
__global__ __launch_bounds__(1024) void probe(unsigned long long *out, unsigned seed, int trips) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    float a[8];
    float b = __uint_as_float(seed | 0x3f800000u), c = 0.5f;
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 0.001f + i;
    half8 ha, hb;
    for (int j = 0; j < 8; j++) { ha[j] = (_Float16)(threadIdx.x * 0.01f + j); hb[j] = (_Float16)(j * 0.1f); }
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    f32x16 big[2];
    for (int q = 0; q < 16; q++) { big[0][q] = 0; big[1][q] = 0; }
    f32x4 r[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    const unsigned lds_a = (threadIdx.x & 63) * 16;
    for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) reinterpret_cast<float *>(lds)[i] = 0.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < trips; t++) {
        if (KIND == 0) { REP8(VMAX3) REP8(VMAX3) }                       // 16 VALU
        if (KIND == 1) { REP8(VCVT) REP8(VCVT) }                         // 16 VALU (cvt_pk)
        if (KIND == 2) {                                                 // 8 MFMA 16x16x32, independent accumulators
#pragma unroll
            for (int k = 0; k < 8; k++) acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[k & 3], 0, 0, 0);
        }
        if (KIND == 3) {                                                 // 8 MFMA + 16 VALU interleaved
#pragma unroll
            for (int k = 0; k < 8; k++) {
                acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[k & 3], 0, 0, 0);
                asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
                asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[(k + 4) & 7]) : "v"(b), "v"(c));
            }
        }
        if (KIND == 4) { REP8(DSR) asm volatile("s_waitcnt lgkmcnt(0)"); }   // 8 ds_read_b128
        if (KIND == 5) {                                                 // 8 ds_read_b128 + 8 MFMA
#pragma unroll
            for (int k = 0; k < 8; k++) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[k & 3]) : "v"(lds_a), "n"(k * 1024));
                acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[k & 3], 0, 0, 0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)");
        }
        if (KIND == 6) {                                                 // 8 ds_read_b128 + 16 VALU
#pragma unroll
            for (int k = 0; k < 8; k++) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[k & 3]) : "v"(lds_a), "n"(k * 1024));
                asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
                asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[(k + 4) & 7]) : "v"(b), "v"(c));
            }
            asm volatile("s_waitcnt lgkmcnt(0)");
        }
        if (KIND == 7) {                                                 // 4 MFMA 32x32x16
#pragma unroll
            for (int k = 0; k < 4; k++) big[k & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, big[k & 1], 0, 0, 0);
        }
        if (KIND == 8) {                                                 // 16 v_add_f32 (VOP2)
            REP8(VADD) REP8(VADD)
        }
        if (KIND == 9) {                                                 // 16 SALU
#pragma unroll
            for (int k = 0; k < 16; k++) asm volatile("s_add_u32 %0, %0, 1" : "+s"(seed));
        }
        if (KIND == 10) {                                                // 8 MFMA + 16 VALU + 8 ds_read
#pragma unroll
            for (int k = 0; k < 8; k++) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[k & 3]) : "v"(lds_a), "n"(k * 1024));
                acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[k & 3], 0, 0, 0);
                asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
                asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[(k + 4) & 7]) : "v"(b), "v"(c));
            }
            asm volatile("s_waitcnt lgkmcnt(0)");
        }
        if (KIND >= 11 && KIND <= 16) {                                  // 8 MFMA 16x16x32 + NV v_max3 (NV = 8, 24, 32, 48, 64, 96)
            constexpr int NV = KIND == 11 ? 1 : KIND == 12 ? 3 : KIND == 13 ? 4 : KIND == 14 ? 6 : KIND == 15 ? 8 : 12;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[k & 3], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < NV; q++) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[(k + q) & 7]) : "v"(b), "v"(c));
            }
        }
        if (KIND >= 17 && KIND <= 20) {                                  // 4 MFMA 32x32x16 + NV v_max3 per MFMA (2, 6, 8, 12)
            constexpr int NV = KIND == 17 ? 2 : KIND == 18 ? 6 : KIND == 19 ? 8 : 12;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                big[k & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, big[k & 1], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < NV; q++) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[(k + q) & 7]) : "v"(b), "v"(c));
            }
        }
        if (KIND == 21) {                                                // 16 MFMA 4x4x4
            typedef _Float16 half4 __attribute__((ext_vector_type(4)));
            half4 a4 = {ha[0], ha[1], ha[2], ha[3]}, b4 = {hb[0], hb[1], hb[2], hb[3]};
#pragma unroll
            for (int k = 0; k < 16; k++) acc[k & 3] = __builtin_amdgcn_mfma_f32_4x4x4f16(a4, b4, acc[k & 3], 0, 0, 0);
        }
        if (KIND == 22) {                                                // 16 MFMA 4x4x4 + 32 v_max3
            typedef _Float16 half4 __attribute__((ext_vector_type(4)));
            half4 a4 = {ha[0], ha[1], ha[2], ha[3]}, b4 = {hb[0], hb[1], hb[2], hb[3]};
#pragma unroll
            for (int k = 0; k < 16; k++) {
                acc[k & 3] = __builtin_amdgcn_mfma_f32_4x4x4f16(a4, b4, acc[k & 3], 0, 0, 0);
                asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[k & 7]) : "v"(b), "v"(c));
                asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[(k + 4) & 7]) : "v"(b), "v"(c));
            }
        }
        if (KIND == 23) {                                                // 16 v_cndmask (VOP2, vcc)
#pragma unroll
            for (int k = 0; k < 16; k++) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k & 7]) : "v"(b) : "vcc");
        }
        if (KIND == 24) {                                                // 16 v_pk_max_f16
#pragma unroll
            for (int k = 0; k < 16; k++) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(a[k & 7]) : "v"(b));
        }
        if (KIND == 25) {                                                // 16 v_perm_b32
#pragma unroll
            for (int k = 0; k < 16; k++) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[k & 7]) : "v"(b), "v"(c));
        }
        if (KIND == 26) {                                                // 16 v_fma_f32 (VOP3, 3 operands) 
#pragma unroll
            for (int k = 0; k < 16; k++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k & 7]) : "v"(b), "v"(c));
        }
        if (KIND == 27) {                                                // 16 v_max_f32 (VOP2)
#pragma unroll
            for (int k = 0; k < 16; k++) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[k & 7]) : "v"(b));
        }
        if (KIND == 28) {                                                // 16 v_mov_b32
#pragma unroll
            for (int k = 0; k < 16; k++) asm volatile("v_mov_b32 %0, %1" : "=v"(a[k & 7]) : "v"(b));
        }
        if (KIND == 29) {                                                // 16 ds_write_b16
#pragma unroll
            for (int k = 0; k < 16; k++) asm volatile("ds_write_b16 %0, %1 offset:%2" :: "v"(lds_a), "v"(a[k & 7]), "n"(k * 128));
            asm volatile("s_waitcnt lgkmcnt(0)");
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    for (int i = 0; i < 4; i++) s += acc[i][0] + r[i][0];
    s += big[0][0] + big[1][0] + (float)seed;
    if ((threadIdx.x & 63) == 0) { out[2 * (threadIdx.x >> 6)] = t1 - t0; out[2 * (threadIdx.x >> 6) + 1] = (unsigned long long)s; }
}

template <int KIND>
void run(const char *name, int per_trip, unsigned long long *d) {
    const int trips = 2000;
    printf("%-36s", name);
    for (int wps : {1, 2, 4}) {
        const int threads = wps * 256;
        hipLaunchKernelGGL(probe<KIND>, dim3(1), dim3(threads), 0, 0, d, 12345u, trips);
        hipLaunchKernelGGL(probe<KIND>, dim3(1), dim3(threads), 0, 0, d, 12345u, trips);
        if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) { printf(" launch failed"); continue; }
        unsigned long long h[32];
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        double mx = 0;
        for (int w = 0; w < wps * 4; w++) mx = h[2 * w] > mx ? h[2 * w] : mx;
        printf("  %dw: %7.1f/trip (%5.2f/instr/SIMD)", wps, mx / trips, mx / trips / (per_trip * wps));
    }
    printf("\n");
}

int main() {
    unsigned long long *d;
    hipMalloc(&d, 4096);
    run<0>("16 v_max3_f32", 16, d);
    run<8>("16 v_add_f32", 16, d);
    run<1>("16 v_cvt_pk_f16_f32", 16, d);
    run<9>("16 s_add_u32", 16, d);
    run<2>("8 mfma 16x16x32", 8, d);
    run<7>("4 mfma 32x32x16", 4, d);
    run<3>("8 mfma16 + 16 v_max3", 24, d);
    run<4>("8 ds_read_b128", 8, d);
    run<5>("8 ds_read_b128 + 8 mfma16", 16, d);
    run<6>("8 ds_read_b128 + 16 v_max3", 24, d);
    run<10>("8 ds_read + 8 mfma16 + 16 v_max3", 32, d);
    run<11>("8 mfma16 + 8 v_max3", 16, d);
    run<12>("8 mfma16 + 24 v_max3", 32, d);
    run<13>("8 mfma16 + 32 v_max3", 40, d);
    run<14>("8 mfma16 + 48 v_max3", 56, d);
    run<15>("8 mfma16 + 64 v_max3", 72, d);
    run<16>("8 mfma16 + 96 v_max3", 104, d);
    run<17>("4 mfma32 + 8 v_max3", 12, d);
    run<18>("4 mfma32 + 24 v_max3", 28, d);
    run<19>("4 mfma32 + 32 v_max3", 36, d);
    run<20>("4 mfma32 + 48 v_max3", 52, d);
    run<21>("16 mfma 4x4x4", 16, d);
    run<22>("16 mfma 4x4x4 + 32 v_max3", 48, d);
    run<23>("16 v_cndmask_b32 (vcc)", 16, d);
    run<24>("16 v_pk_max_f16", 16, d);
    run<25>("16 v_perm_b32", 16, d);
    run<26>("16 v_fma_f32", 16, d);
    run<27>("16 v_max_f32 (VOP2)", 16, d);
    run<28>("16 v_mov_b32", 16, d);
    run<29>("16 ds_write_b16", 16, d);
    return 0;
}
