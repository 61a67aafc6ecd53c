// Developer probe (run ON THE GPU BOX): what the host link gives the pinned pipeline -- H2D of one batch of packed / four-byte
// carrier frames, D2H of a batch's packed boxes, alone and together (two streams).
// build: hipcc -O2 --offload-arch=gfx950 tools/probes/pcie_probe.hip -o tools/probes/pcie_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t MAXB = 16u << 20;
    void *h_in, *h_out, *d_in, *d_out;
    CK(hipHostMalloc(&h_in, MAXB, hipHostMallocDefault));
    CK(hipHostMalloc(&h_out, MAXB, hipHostMallocDefault));
    CK(hipMalloc(&d_in, MAXB));
    CK(hipMalloc(&d_out, MAXB));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    const size_t sizes[] = {280 * 68 * 120 * 2, 280 * 68 * 120 * 4, 256 * 500 * 20, 64 << 10};
    const char *names[] = {"H2D packed batch (4.57 MB)", "H2D four-byte batch (9.14 MB)", "H2D 2.56 MB", "H2D 64 KB"};
    const int N = 200;
    for (int k = 0; k < 4; k++) {
        for (int i = 0; i < 10; i++) CK(hipMemcpyAsync(d_in, h_in, sizes[k], hipMemcpyHostToDevice, s1));
        CK(hipStreamSynchronize(s1));
        double t0 = now();
        for (int i = 0; i < N; i++) CK(hipMemcpyAsync(d_in, h_in, sizes[k], hipMemcpyHostToDevice, s1));
        CK(hipStreamSynchronize(s1));
        double dt = (now() - t0) / N;
        printf("%-34s %8.1f us  %6.1f GB/s\n", names[k], dt * 1e6, sizes[k] / dt / 1e9);
    }
    {
        const size_t sz = 256 * 500 * 20;
        double t0 = now();
        for (int i = 0; i < N; i++) CK(hipMemcpyAsync(h_out, d_out, sz, hipMemcpyDeviceToHost, s2));
        CK(hipStreamSynchronize(s2));
        double dt = (now() - t0) / N;
        printf("%-34s %8.1f us  %6.1f GB/s\n", "D2H packed boxes (2.56 MB)", dt * 1e6, sz / dt / 1e9);
        t0 = now();
        for (int i = 0; i < N; i++) {
            CK(hipMemcpyAsync(d_in, h_in, sizes[0], hipMemcpyHostToDevice, s1));
            CK(hipMemcpyAsync(h_out, d_out, sz, hipMemcpyDeviceToHost, s2));
        }
        CK(hipStreamSynchronize(s1));
        CK(hipStreamSynchronize(s2));
        dt = (now() - t0) / N;
        printf("%-34s %8.1f us per pair\n", "H2D packed + D2H boxes together", dt * 1e6);
    }
    return 0;
}
