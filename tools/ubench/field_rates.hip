// Micro-benchmark: sustained rate of the real field / curve routines (field29.h, curve.h) in registers only,
// at 1..4 waves per SIMD.  Compares with the per-instruction rates of valu_rates.hip to see whether the MSM
// accumulate loop is issue bound.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../rust-kzg-bn254_amd/csrc/curve.h"
using namespace kzg;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int WAVES>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
k_mul_chain(uint32_t* out, int iters) {
    Fq a, b;
    for (int j = 0; j < NL; ++j) { a.l[j] = (int32_t)((threadIdx.x * 2654435761u + j * 40503u) & LMASK); b.l[j] = (int32_t)((blockIdx.x * 40503u + j * 2654435761u + 7) & LMASK); }
    a.l[8] &= 0x1FFFFF; b.l[8] &= 0x1FFFFF;
    for (int i = 0; i < iters; ++i) { fe_mul(a, a, b); fe_mul(b, b, a); }
    uint32_t x = 0; for (int j = 0; j < NL; ++j) x ^= (uint32_t)a.l[j] ^ (uint32_t)b.l[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
template <int WAVES>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
k_madd_chain(uint32_t* out, int iters) {
    Affine p;
    for (int j = 0; j < NL; ++j) { p.x.l[j] = (int32_t)((threadIdx.x * 2654435761u + j * 40503u) & LMASK); p.y.l[j] = (int32_t)((blockIdx.x * 40503u + j * 2654435761u + 7) & LMASK); }
    p.x.l[8] &= 0x1FFFFF; p.y.l[8] &= 0x1FFFFF;
    Xyzz acc; xyzz_from_affine(acc, p, 0);
    p.x.l[0] ^= 5;
    for (int i = 0; i < iters; ++i) { xyzz_madd<true>(acc, p, i & 1); p.x.l[1] = (p.x.l[1] + 3) & (int32_t)LMASK; }
    uint32_t x = 0; for (int j = 0; j < NL; ++j) x ^= (uint32_t)acc.x.l[j] ^ (uint32_t)acc.y.l[j] ^ (uint32_t)acc.zz.l[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
template <class K> int run(const char* name, K kern, int waves, int iters, double per_iter_ops, const char* unit) {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int blocks = prop.multiProcessorCount * waves; uint32_t* d; CHECK(hipMalloc(&d, (size_t)blocks * 256 * 4));
    hipEvent_t t0, t1; CHECK(hipEventCreate(&t0)); CHECK(hipEventCreate(&t1));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, iters); CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) { CHECK(hipEventRecord(t0)); hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, iters); CHECK(hipEventRecord(t1)); CHECK(hipEventSynchronize(t1)); float ms; CHECK(hipEventElapsedTime(&ms, t0, t1)); if (ms < best) best = ms; }
    double per_simd_ns = best * 1e6 / ((double)iters * per_iter_ops * waves);
    printf("%-10s waves/SIMD=%d  %.3f ms  -> %.1f ns per %s per SIMD  (chip: %.3g %s/s per lane-wave => %.3g lane-%s/s)\n", name, waves, best, per_simd_ns, unit,
           1e9 / per_simd_ns * 1024, unit, 1e9 / per_simd_ns * 1024 * 64, unit);
    CHECK(hipFree(d)); return 0;
}
int main() {
    run("fe_mul", k_mul_chain<1>, 1, 2000, 2, "mul"); run("fe_mul", k_mul_chain<2>, 2, 2000, 2, "mul");
    run("fe_mul", k_mul_chain<3>, 3, 2000, 2, "mul"); run("fe_mul", k_mul_chain<4>, 4, 2000, 2, "mul");
    run("xyzz_madd", k_madd_chain<1>, 1, 400, 1, "madd"); run("xyzz_madd", k_madd_chain<2>, 2, 400, 1, "madd");
    run("xyzz_madd", k_madd_chain<3>, 3, 400, 1, "madd"); run("xyzz_madd", k_madd_chain<4>, 4, 400, 1, "madd");
    return 0;
}
