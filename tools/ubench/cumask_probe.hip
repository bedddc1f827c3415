// Does hipExtStreamCreateWithCUMask restrict kernels to a CU subset on this GPU, and do two masked streams run side by side?
// Build: hipcc --offload-arch=gfx950 -O2 tools/ubench/cumask_probe.hip -o tools/ubench/cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <chrono>
__global__ void burn(float* out, int iters) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    for (int i = 0; i < iters; ++i) a = a * b + 0.5f;
    if (a == 12345.f) out[0] = a;
}
static double run(hipStream_t s, int blocks, int iters, float* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, s);
    hipLaunchKernelGGL(burn, dim3(blocks), dim3(256), 0, s, d, iters);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    int ncu = p.multiProcessorCount;
    printf("CUs reported: %d\n", ncu);
    float* d; hipMalloc(&d, 4);
    hipStream_t full; hipStreamCreateWithFlags(&full, hipStreamNonBlocking);
    const int words = (ncu + 31) / 32;
    std::vector<uint32_t> m_heavy(words, 0), m_light(words, 0);
    for (int cu = 0; cu < ncu; ++cu) { if (cu % 8 == 7) m_light[cu / 32] |= 1u << (cu % 32); else m_heavy[cu / 32] |= 1u << (cu % 32); }
    hipStream_t heavy, light;
    hipError_t e1 = hipExtStreamCreateWithCUMask(&heavy, words, m_heavy.data());
    hipError_t e2 = hipExtStreamCreateWithCUMask(&light, words, m_light.data());
    printf("create masked streams: %s / %s\n", hipGetErrorString(e1), hipGetErrorString(e2));
    if (e1 != hipSuccess || e2 != hipSuccess) return 1;
    const int blocks = ncu * 8 * 4, iters = 200000;
    run(full, blocks, 1000, d);
    printf("full   : %.3f ms\n", run(full, blocks, iters, d));
    printf("heavy  (7/8 of the CUs): %.3f ms   (expect x 8/7 = %.3f)\n", run(heavy, blocks, iters, d), 0.0);
    printf("light  (1/8 of the CUs): %.3f ms\n", run(light, blocks / 8, iters, d));
    // side by side: heavy long kernel + light short kernels
    hipEvent_t a0, a1, b0, b1; hipEventCreate(&a0); hipEventCreate(&a1); hipEventCreate(&b0); hipEventCreate(&b1);
    hipEventRecord(a0, heavy); hipLaunchKernelGGL(burn, dim3(blocks), dim3(256), 0, heavy, d, iters); hipEventRecord(a1, heavy);
    hipEventRecord(b0, light); for (int k = 0; k < 10; ++k) hipLaunchKernelGGL(burn, dim3(ncu), dim3(256), 0, light, d, iters / 40); hipEventRecord(b1, light);
    hipEventSynchronize(a1); hipEventSynchronize(b1);
    float ta, tb, off; hipEventElapsedTime(&ta, a0, a1); hipEventElapsedTime(&tb, b0, b1); hipEventElapsedTime(&off, a0, b1);
    printf("side by side: heavy %.3f ms, 10 light kernels %.3f ms (finished %.3f ms after the heavy start)\n", ta, tb, off);
    // the same light kernels on an UNMASKED second stream while the heavy kernel runs unmasked (today's situation)
    hipStream_t s2; hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    hipEventRecord(a0, full); hipLaunchKernelGGL(burn, dim3(blocks), dim3(256), 0, full, d, iters); hipEventRecord(a1, full);
    hipEventRecord(b0, s2); for (int k = 0; k < 10; ++k) hipLaunchKernelGGL(burn, dim3(ncu), dim3(256), 0, s2, d, iters / 40); hipEventRecord(b1, s2);
    hipEventSynchronize(a1); hipEventSynchronize(b1);
    hipEventElapsedTime(&ta, a0, a1); hipEventElapsedTime(&tb, b0, b1); hipEventElapsedTime(&off, a0, b1);
    printf("unmasked    : heavy %.3f ms, 10 light kernels %.3f ms (finished %.3f ms after the heavy start)\n", ta, tb, off);
    return 0;
}
