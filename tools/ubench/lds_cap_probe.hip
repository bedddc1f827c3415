// Does a dynamic-LDS request cap the number of resident workgroups per CU (occupancy control without touching the kernel)?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void burn(float* out, int iters) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    for (int i = 0; i < iters; ++i) a = a * b + 0.5f;     // one dependent chain per wave: throughput scales with resident waves
    if (a == 12345.f) out[0] = a;
}
int main() {
    float* d; (void)hipMalloc(&d, 4);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(burn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipStream_t s; (void)hipStreamCreate(&s);
    for (int lds : {0, 20000, 40000, 52000, 80000}) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(burn, dim3(256 * 8), dim3(256), lds, s, d, 1000);
        (void)hipEventRecord(e0, s);
        hipLaunchKernelGGL(burn, dim3(256 * 8), dim3(256), lds, s, d, 400000);
        (void)hipEventRecord(e1, s);
        (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("dynamic LDS %6d B: 2048 blocks x 256 threads: %.3f ms  (%s)\n", lds, ms, hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
