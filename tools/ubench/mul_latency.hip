// Micro-benchmark: latency of ONE dependent chain of field multiplications on a lone wave per SIMD (the regime of the bucket reductions):
//   A  fe_mul (fe_asm.h: one accumulator, 206 dependent instructions)
//   B  fe_mul_ilp (field29.h: the 17 column sums of a b as independent chains, then the Montgomery reduction)
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
        if (VAR == 0) { fe_mul(a, a, b); fe_mul(b, b, a); }
        else { fe_mul_ilp(a, a, b); fe_mul_ilp(b, b, a); }
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
    printf("%-12s waves/SIMD=%d  %.3f ms  -> %.1f ns per multiplication of a chain; %.1f ns per multiplication and SIMD\n", name, waves, best, best * 1e6 / (iters * 2.0), best * 1e6 / (iters * 2.0 * waves));
    CHECK(hipFree(d)); return 0;
}
int main() {
    uint32_t c0, c1;
    run("fe_mul", k_chain<0, 1>, 1, 2000, &c0); run("fe_mul_ilp", k_chain<1, 1>, 1, 2000, &c1);
    printf("same canonical result is not implied (ranges may differ); raw xor %08x %08x\n", c0, c1);
    run("fe_mul", k_chain<0, 2>, 2, 2000, &c0); run("fe_mul_ilp", k_chain<1, 2>, 2, 2000, &c1);
    run("fe_mul", k_chain<0, 4>, 4, 2000, &c0); run("fe_mul_ilp", k_chain<1, 4>, 4, 2000, &c1);
    return 0;
}
