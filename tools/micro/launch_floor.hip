// Developer micro-benchmark: duration of an empty persistent-style launch vs workgroup size and dynamic LDS.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void empty_k(int *p) { extern __shared__ char s[]; if (p && threadIdx.x == 9999) p[0] = s[0]; }
int main() {
    hipFuncSetAttribute(reinterpret_cast<const void *>(empty_k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int threads[] = {256, 512, 1024};
    const int ldskb[] = {0, 32, 64, 80, 100, 128, 150, 158};
    for (int t : threads) for (int l : ldskb) for (int grid : {256, 512}) {
        for (int i = 0; i < 5; i++) hipLaunchKernelGGL(empty_k, dim3(grid), dim3(t), l * 1024, 0, nullptr);
        hipDeviceSynchronize();
        hipEventRecord(a);
        for (int i = 0; i < 100; i++) hipLaunchKernelGGL(empty_k, dim3(grid), dim3(t), l * 1024, 0, nullptr);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("threads %4d lds %3d KB grid %3d: %.2f us/launch\n", t, l, grid, ms * 10);
    }
    return 0;
}
