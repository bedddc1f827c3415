// Micro-benchmark: ONE dependent chain of field multiplications (mul, sqr, mul: 206 + 170 + 206 instructions) per wave, at 1 / 2 / 4 waves
// per SIMD: how much slower does a LONE wave -- the regime of the bucket reductions -- issue than a shared SIMD?
// Measured (MI355X): 456 ns per multiplication alone, 365 ns per multiplication and SIMD at two waves, 349 at four: a lone wave already
// issues at ~77 % of the saturated rate.  Two variants built to give a lone wave independent instructions did NOT help: the column sums
// in separate accumulators ahead of the reduction chain, list-scheduled for a two-slot latency (222 instructions: 460 ns), and s[..] instead
// of vcc as the carry-out of every v_mad_i64_i32 (no change).  The reductions are bound by the instruction COUNT on their critical path.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 mul_latency.hip -o mul_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../rust-kzg-bn254_amd/csrc/curve.h"
using namespace kzg;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int VAR, int WAVES>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
k_chain(uint32_t* out, int iters) {
    Fq a, b;
    for (int j = 0; j < NL; ++j) { a.l[j] = (int32_t)((threadIdx.x * 2654435761u + j * 40503u) & LMASK); b.l[j] = (int32_t)((blockIdx.x * 40503u + j * 2654435761u + 7) & LMASK); }
    a.l[8] &= 0x1FFFFF; b.l[8] &= 0x1FFFFF;
    for (int i = 0; i < iters; ++i) {
        fe_mul(a, a, b); fe_sqr(b, a); fe_mul(b, b, a);
    }
    uint32_t x = 0; for (int j = 0; j < NL; ++j) x ^= (uint32_t)a.l[j] ^ (uint32_t)b.l[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
template <class K> int run(const char* name, K kern, int waves, int iters, uint32_t* check) {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int blocks = prop.multiProcessorCount * waves; uint32_t* d; CHECK(hipMalloc(&d, (size_t)blocks * 256 * 4));
    hipEvent_t t0, t1; CHECK(hipEventCreate(&t0)); CHECK(hipEventCreate(&t1));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, iters); CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) { CHECK(hipEventRecord(t0)); hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, iters); CHECK(hipEventRecord(t1)); CHECK(hipEventSynchronize(t1)); float ms; CHECK(hipEventElapsedTime(&ms, t0, t1)); if (ms < best) best = ms; }
    CHECK(hipMemcpy(check, d, 4, hipMemcpyDeviceToHost));
    printf("%-12s waves/SIMD=%d  %.3f ms  -> %.1f ns per multiplication of a chain; %.1f ns per multiplication and SIMD\n", name, waves, best, best * 1e6 / (iters * 3.0), best * 1e6 / (iters * 3.0 * waves));
    CHECK(hipFree(d)); return 0;
}
int main() {
    uint32_t c0, c1;
    run("fe_mul", k_chain<0, 1>, 1, 2000, &c0);
    run("fe_mul", k_chain<0, 2>, 2, 2000, &c0);
    run("fe_mul", k_chain<0, 4>, 4, 2000, &c0);
    (void)c1;
    return 0;
}
