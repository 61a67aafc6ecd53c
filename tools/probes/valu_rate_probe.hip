// Developer probe: issue cost of integer vector instructions on gfx950, one wave per SIMD and four waves per SIMD.
// Each kernel runs N independent instructions of one kind per loop trip (8 accumulators), stamped with s_memtime.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/valu_rate_probe.hip -o tools/probes/valu_rate_probe && tools/probes/valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int KIND>
__global__ __launch_bounds__(256) void probe(unsigned long long *out, unsigned seed, int trips) {
    unsigned a[8], b = seed | 1u, c = threadIdx.x + 3u;
    unsigned long long w[8];
    for (int i = 0; i < 8; i++) { a[i] = threadIdx.x * 7u + i; w[i] = a[i]; }
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < trips; t++) {
#define MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define MULHI(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define MUL24(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define MAD24(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define ADD(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define MAD64(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(b), "v"(c) : "vcc");
#define LSHLADD64(i) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(w[i]) : "v"(w[(i + 1) & 7]));
#define ADD3(i) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define MAX3(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define CVTPK(i) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define PKMAX(i) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a[i]) : "v"(b));
#define BITOP(i) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a[i]) : "v"(b), "v"(c));
        if (KIND == 0) { REP8(ADD) REP8(ADD) }
        if (KIND == 1) { REP8(MULLO) REP8(MULLO) }
        if (KIND == 2) { REP8(MULHI) REP8(MULHI) }
        if (KIND == 3) { REP8(MUL24) REP8(MUL24) }
        if (KIND == 4) { REP8(MAD24) REP8(MAD24) }
        if (KIND == 5) { REP8(MAD64) REP8(MAD64) }
        if (KIND == 6) { REP8(LSHLADD64) REP8(LSHLADD64) }
        if (KIND == 7) { REP8(ADD3) REP8(ADD3) }
        if (KIND == 8) { REP8(PERM) REP8(PERM) }
        if (KIND == 9) { REP8(MAX3) REP8(MAX3) }
        if (KIND == 10) { REP8(CVTPK) REP8(CVTPK) }
        if (KIND == 11) { REP8(PKMAX) REP8(PKMAX) }
        if (KIND == 12) { REP8(LSHLADD) REP8(LSHLADD) }
        if (KIND == 13) { REP8(BITOP) REP8(BITOP) }
    }
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + (unsigned)w[i];
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = s; }
}

template <int KIND>
void run(const char *name, unsigned long long *d) {
    const int trips = 2000;
    for (int threads : {64, 256, 1024}) {   // one wave (one SIMD), one wave per SIMD, four waves per SIMD
        hipLaunchKernelGGL(probe<KIND>, dim3(1), dim3(threads), 0, 0, d, 12345u, trips);
        hipLaunchKernelGGL(probe<KIND>, dim3(1), dim3(threads), 0, 0, d, 12345u, trips);
        hipDeviceSynchronize();
        unsigned long long h[2];
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("%-16s %4d threads: %6.2f cycles per wave-instruction (wave 0's clock / its own instructions)\n", name, threads,
               (double)h[0] / (trips * 16.0));
    }
}

int main() {
    unsigned long long *d;
    hipMalloc(&d, 4096);
    run<0>("v_add_u32", d);
    run<1>("v_mul_lo_u32", d);
    run<2>("v_mul_hi_u32", d);
    run<3>("v_mul_u32_u24", d);
    run<4>("v_mad_u32_u24", d);
    run<5>("v_mad_u64_u32", d);
    run<6>("v_lshl_add_u64", d);
    run<7>("v_add3_u32", d);
    run<8>("v_perm_b32", d);
    run<9>("v_max3_f32", d);
    run<10>("v_cvt_pk_f16_f32", d);
    run<11>("v_pk_max_f16", d);
    run<12>("v_lshl_add_u32", d);
    run<13>("v_bitop3_b32", d);
    return 0;
}
