// Developer probe: what the board draws while EVERY CU runs one kind of instruction stream, and at which clock -- the inputs of
// an energy-per-operation table (tools/power_probe_kinds.sh samples rocm-smi while this runs).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/power_probe.hip -o tools/probes/power_probe
//   power_probe <kind> <seconds>      kind: 0 resident waves that sleep, 1 v_mfma_f32_32x32x16_f16, 2 v_mfma_f32_16x16x32_f16,
//                                           3 v_max3_f32, 4 ds_read_b128, 5 HBM read stream, 6 mfma32 + ds_read_b128 (one per product),
//                                           7 v_pk_fma_f16, 8 v_mfma_f32_4x4x4_16B_f16, 9 HBM write stream
// Prints: operations per second over the run and the shader clock (s_memtime / s_memrealtime of one wave).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND>
__global__ __launch_bounds__(256) void burn(unsigned long long *out, const uint4 *src, size_t n16, int trips, unsigned seed) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[32768];
    float a[8];
    const float b = __uint_as_float(seed | 0x3f800000u), c = 0.5f;
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 0.001f + i;
    half8 ha, hb;
    for (int j = 0; j < 8; j++) { ha[j] = (_Float16)(threadIdx.x * 0.01f + j); hb[j] = (_Float16)(j * 0.1f); }
    f32x4 acc4[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    f32x16 big[2];
    for (int q = 0; q < 16; q++) { big[0][q] = 0; big[1][q] = 0; }
    f32x4 r[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    uint4 g = {0, 0, 0, 0};
    const unsigned lds_a = (threadIdx.x & 63) * 16;
    for (int i = threadIdx.x; i < 32768 / 4; i += blockDim.x) reinterpret_cast<float *>(lds)[i] = 0.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
    for (int t = 0; t < trips; t++) {
        if (KIND == 0) __builtin_amdgcn_s_sleep(64);
        if (KIND == 1) {
#pragma unroll
            for (int k = 0; k < 8; k++) big[k & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, big[k & 1], 0, 0, 0);
        }
        if (KIND == 2) {
#pragma unroll
            for (int k = 0; k < 16; k++) acc4[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc4[k & 3], 0, 0, 0);
        }
        if (KIND == 3) {
#pragma unroll
            for (int k = 0; k < 32; k++) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[k & 7]) : "v"(b), "v"(c));
        }
        if (KIND == 4) {
#pragma unroll
            for (int k = 0; k < 16; k++)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[k & 3]) : "v"(lds_a), "n"((k & 15) * 1024));
            asm volatile("s_waitcnt lgkmcnt(0)");
        }
        if (KIND == 5) {
            // every workgroup walks the whole buffer with a stride of the grid: 16 bytes per lane and trip
            const size_t i = (((size_t)t * gridDim.x + blockIdx.x) * 8) * blockDim.x + threadIdx.x;
            uint4 v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = src[(i + (size_t)k * blockDim.x) & (n16 - 1)];   // n16 is a power of two
#pragma unroll
            for (int k = 0; k < 8; k++) { g.x ^= v[k].x; g.y ^= v[k].y; g.z ^= v[k].z; g.w ^= v[k].w; }
        }
        if (KIND == 6) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[k & 3]) : "v"(lds_a), "n"(k * 1024));
                big[k & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, big[k & 1], 0, 0, 0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)");
        }
        if (KIND == 7) {
#pragma unroll
            for (int k = 0; k < 32; k++) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(a[k & 7]) : "v"(b), "v"(c));
        }
        if (KIND == 8) {
            typedef _Float16 half4 __attribute__((ext_vector_type(4)));
            const half4 h4a = {ha[0], ha[1], ha[2], ha[3]}, h4b = {hb[0], hb[1], hb[2], hb[3]};
#pragma unroll
            for (int k = 0; k < 16; k++) acc4[k & 3] = __builtin_amdgcn_mfma_f32_4x4x4f16(h4a, h4b, acc4[k & 3], 0, 0, 0);
        }
        if (KIND == 9) {
            const size_t i = (((size_t)t * gridDim.x + blockIdx.x) * 8) * blockDim.x + threadIdx.x;
#pragma unroll
            for (int k = 0; k < 8; k++) const_cast<uint4 *>(src)[(i + (size_t)k * blockDim.x) & (n16 - 1)] = make_uint4(t, k, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    for (int i = 0; i < 4; i++) s += acc4[i][0] + r[i][0];
    s += big[0][0] + big[1][0] + (float)(g.x ^ g.y ^ g.z ^ g.w);
    if (s == 12345.678f) out[3] = 1;
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; }
}

template <int KIND>
static void run(const char *name, double ops_per_trip_per_wg, double seconds, unsigned long long *d, const uint4 *src, size_t n16) {
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int grid = prop.multiProcessorCount * 8;   // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    int trips = 20000;
    // calibrate: one short launch, then size the long one
    auto launch = [&](int tr) {
        auto a = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(burn<KIND>, dim3(grid), dim3(256), 0, 0, d, src, n16, tr, 0x1234u);
        (void)hipDeviceSynchronize();
        return std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count();
    };
    launch(1000);
    const double t_cal = launch(trips);
    trips = (int)(trips * seconds / t_cal);
    const double t = launch(trips);
    unsigned long long h[4];
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-40s %.3f s  %.4g ops/s  shader clock %.0f MHz\n", name, t, ops_per_trip_per_wg * trips * grid / t, (double)h[0] / (double)h[1] * 100.0);
    fflush(stdout);
}

int main(int argc, char **argv) {
    const int kind = argc > 1 ? atoi(argv[1]) : 1;
    const double seconds = argc > 2 ? atof(argv[2]) : 3.0;
    unsigned long long *d;
    (void)hipMalloc(&d, 4096);
    // 4 GB: larger than the last-level cache; a third argument (MB, power of two) makes the read / write streams walk a smaller
    // buffer (128 MB: misses the 32 MB of L2, fits the 256 MB Infinity Cache)
    const size_t bytes = argc > 3 ? (size_t)atoi(argv[3]) << 20 : (size_t)4 << 30;
    uint4 *src = nullptr;
    if (kind == 5 || kind == 9) { (void)hipMalloc(&src, bytes); (void)hipMemset(src, 1, bytes); }
    const size_t n16 = bytes / 16;
    // operations per trip and WORKGROUP (4 waves): flop for the matrix kinds, lane-operations for v_max3 / v_pk_fma, bytes for LDS / HBM
    switch (kind) {
    case 0: run<0>("resident waves, s_sleep", 0, seconds, d, src, n16); break;
    case 1: run<1>("v_mfma_f32_32x32x16_f16 [flop]", 4.0 * 8 * 32768, seconds, d, src, n16); break;
    case 2: run<2>("v_mfma_f32_16x16x32_f16 [flop]", 4.0 * 16 * 16384, seconds, d, src, n16); break;
    case 3: run<3>("v_max3_f32 [lane ops]", 4.0 * 32 * 64, seconds, d, src, n16); break;
    case 4: run<4>("ds_read_b128 [bytes]", 4.0 * 16 * 1024, seconds, d, src, n16); break;
    case 5: run<5>("HBM read stream [bytes]", 256.0 * 16 * 8, seconds, d, src, n16); break;
    case 6: run<6>("mfma 32x32x16 + ds_read_b128 each [flop]", 4.0 * 8 * 32768, seconds, d, src, n16); break;
    case 7: run<7>("v_pk_fma_f16 [lane ops]", 4.0 * 32 * 64, seconds, d, src, n16); break;
    case 8: run<8>("v_mfma_f32_4x4x4_16B_f16 [instructions]", 4.0 * 16, seconds, d, src, n16); break;
    case 9: run<9>("HBM write stream [bytes]", 256.0 * 16 * 8, seconds, d, src, n16); break;
    }
    return 0;
}
