// Micro-benchmark: does the random 64-byte gather of k_msm_accumulate survive a 16 GiB table?
// (One table per BIT position of the scalar -- 255 x 64 MiB at 2^20 points -- would let the sort use a width-w NAF:
//  254 / (w + 1) entries per scalar instead of 255 / c.)  Two loops per table span:
//   G  pure gather: 4 x 128-bit loads per iteration, two iterations in flight, xor-reduced
//   M  the accumulate loop of tools/ubench/acc_variants.hip V4 (gather + xyzz_madd), 2 and 3 waves per SIMD
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 gather_span.hip -o gather_span
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../rust-kzg-bn254_amd/csrc/curve.h"
using namespace kzg;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint32_t next_idx(uint32_t& s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }

__global__ void __launch_bounds__(256) k_gather(uint32_t* out, const uint4* __restrict__ points, uint32_t idx_mask, int iters) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s = t * 2654435761u + 12345u;
    uint4 acc = make_uint4(0, 0, 0, 0);
    const uint4* src = points + 4 * (size_t)(next_idx(s) & idx_mask);
    uint4 q0 = src[0], q1 = src[1], q2 = src[2], q3 = src[3];
    for (int i = 0; i < iters; ++i) {
        const uint4* nsrc = points + 4 * (size_t)(next_idx(s) & idx_mask);
        uint4 n0 = nsrc[0], n1 = nsrc[1], n2 = nsrc[2], n3 = nsrc[3];
        acc.x ^= q0.x ^ q1.y ^ q2.z ^ q3.w; acc.y += q0.y + q1.z + q2.w + q3.x;
        q0 = n0; q1 = n1; q2 = n2; q3 = n3;
    }
    out[t] = acc.x ^ acc.y ^ q0.x;
}

template <int WAVES>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
k_madd(uint32_t* out, const uint4* __restrict__ points, uint32_t idx_mask, int iters) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s = t * 2654435761u + 12345u;
    Affine p;
    for (int j = 0; j < NL; ++j) { p.x.l[j] = (int32_t)((threadIdx.x * 2654435761u + j * 40503u) & LMASK); p.y.l[j] = (int32_t)((blockIdx.x * 40503u + j * 2654435761u + 7) & LMASK); }
    p.x.l[8] &= 0x1FFFFF; p.y.l[8] &= 0x1FFFFF;
    Xyzz acc; xyzz_from_affine(acc, p, 0);
    uint32_t v = next_idx(s);
    const uint4* src = points + 4 * (size_t)(v & idx_mask);
    uint4 q0 = src[0], q1 = src[1], q2 = src[2], q3 = src[3];
    for (int i = 0; i < iters; ++i) {
        const uint32_t neg = v >> 31;
        uint32_t wx[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
        uint32_t wy[8] = {q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
        Affine q;
        fe_unpack(q.x, wx);
        fe_unpack(q.y, wy);
        v = next_idx(s);
        src = points + 4 * (size_t)(v & idx_mask);
        q0 = src[0]; q1 = src[1]; q2 = src[2]; q3 = src[3];
        xyzz_madd<true>(acc, q, neg);
    }
    uint32_t x = 0; for (int j = 0; j < NL; ++j) x ^= (uint32_t)acc.x.l[j] ^ (uint32_t)acc.y.l[j] ^ (uint32_t)acc.zz.l[j];
    out[t] = x;
}

template <class K> int run(const char* name, K kern, int blocks, uint32_t* d_out, const uint4* pts, uint32_t mask, int iters) {
    hipEvent_t t0, t1; CHECK(hipEventCreate(&t0)); CHECK(hipEventCreate(&t1));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, pts, mask, iters); CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) { CHECK(hipEventRecord(t0)); hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, pts, mask, iters); CHECK(hipEventRecord(t1)); CHECK(hipEventSynchronize(t1)); float ms; CHECK(hipEventElapsedTime(&ms, t0, t1)); if (ms < best) best = ms; }
    const double gathers = (double)blocks * 256 * iters;
    printf("%-44s %.3f ms  %.1f G gathers/s  %.2f TB/s\n", name, best, gathers / best * 1e-6, gathers * 64 / best * 1e-9);
    fflush(stdout);
    return 0;
}

int main(int argc, char** argv) {
    const int max_log = argc > 1 ? atoi(argv[1]) : 28;            // points of 64 B: 2^28 = 16 GiB
    const size_t pts = (size_t)1 << max_log;
    uint4* d_pts;
    if (getenv("GATHER_CONTIGUOUS")) { CHECK(hipExtMallocWithFlags(reinterpret_cast<void**>(&d_pts), pts * 64, hipDeviceMallocContiguous)); printf("hipDeviceMallocContiguous\n"); }
    else CHECK(hipMalloc(&d_pts, pts * 64));
    std::vector<uint32_t> h(1 << 22);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)(i * 2654435761u + 12345u) & 0x0FFFFFFFu;
    CHECK(hipMemcpy(d_pts, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    for (size_t done = h.size() * 4; done < pts * 64; done *= 2)   // doubling device-to-device fill: every page touched
        CHECK(hipMemcpy((char*)d_pts + done, d_pts, done, hipMemcpyDeviceToDevice));
    uint32_t* d_out; CHECK(hipMalloc(&d_out, (size_t)2048 * 256 * 4));
    for (int lg = 24; lg <= max_log; lg += 2) {
        const uint32_t mask = (uint32_t)(((size_t)1 << lg) - 1);
        char name[96];
        snprintf(name, sizeof name, "G pure gather, 2^%d points (%.0f GiB), 2048 wg", lg, (double)(((size_t)64 << lg) >> 20) / 1024.0);
        run(name, k_gather, 2048, d_out, d_pts, mask, 200);
        snprintf(name, sizeof name, "M gather + madd, 2^%d points, 2 waves/SIMD", lg);
        run(name, k_madd<2>, 512, d_out, d_pts, mask, 100);
        snprintf(name, sizeof name, "M gather + madd, 2^%d points, 3 waves/SIMD", lg);
        run(name, k_madd<3>, 768, d_out, d_pts, mask, 100);
    }
    return 0;
}
