// graph_latency.hip -- is a captured hipGraph of K short dependent kernels faster end to end than K direct launches on this stack?
// (the small proof / commitment shapes of the reference's benches are ~10 dependent kernels of 5-20 us each: launch-bound)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 graph_latency.hip -o graph_latency ; run: ./graph_latency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_work(unsigned* p, int iters) {       // ~iters x 8 dependent integer ops on one wave: a few microseconds
    unsigned v = p[threadIdx.x];
    for (int i = 0; i < iters; ++i) v = v * 1664525u + 1013904223u;
    p[threadIdx.x] = v;
}
int main() {
    unsigned* d = nullptr;
    CK(hipMalloc(&d, 4096));
    CK(hipMemset(d, 0, 4096));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (int K : {2, 5, 10, 20}) {
        for (int iters : {200, 2000}) {
            auto direct = [&]() { for (int k = 0; k < K; ++k) hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, st, d, iters); return hipStreamSynchronize(st); };
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            for (int k = 0; k < K; ++k) hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, st, d, iters);
            CK(hipStreamEndCapture(st, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            auto graph = [&]() { hipError_t e = hipGraphLaunch(ge, st); if (e != hipSuccess) return e; return hipStreamSynchronize(st); };
            double td = 0, tg = 0;
            for (int which = 0; which < 2; ++which) {
                std::vector<double> ts;
                for (int r = 0; r < 300; ++r) {
                    const auto t0 = std::chrono::steady_clock::now();
                    CK(which ? graph() : direct());
                    ts.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
                }
                std::sort(ts.begin(), ts.end());
                (which ? tg : td) = ts[ts.size() / 2];
            }
            printf("K = %2d kernels of %4d iterations: direct %7.1f us, graph %7.1f us (median of 300, enqueue + synchronise)\n", K, iters, td, tg);
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        }
    }
    return 0;
}
