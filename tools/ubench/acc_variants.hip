// Micro-benchmark: where does k_msm_accumulate lose issue rate against the bare xyzz_madd chain of field_rates.hip?
// One feature of the real loop is added per variant (3 waves per SIMD, 3072 waves, ITER mixed additions per lane):
//   V0  bare chain, point in registers (= field_rates k_madd_chain)
//   V1  + per-lane sign and fe_unpack of 16 packed words per iteration (registers only)
//   V2  + the 64-byte point comes from global memory, prefetched one iteration ahead, from a 16 KiB table (L1/L2 hits)
//   V3  + the index comes from a per-lane stream sorted[t * ITER + i], prefetched two ahead (table still 16 KiB)
//   V4  = V3 with a 960 MiB table and uniformly random indices (the real gather)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 acc_variants.hip -o acc_variants
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../rust-kzg-bn254_amd/csrc/curve.h"
using namespace kzg;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int V>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
k_variant(uint32_t* out, const uint4* __restrict__ points, const uint32_t* __restrict__ sorted, uint32_t idx_mask, int iters) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    Affine p;
    for (int j = 0; j < NL; ++j) { p.x.l[j] = (int32_t)((threadIdx.x * 2654435761u + j * 40503u) & LMASK); p.y.l[j] = (int32_t)((blockIdx.x * 40503u + j * 2654435761u + 7) & LMASK); }
    p.x.l[8] &= 0x1FFFFF; p.y.l[8] &= 0x1FFFFF;
    Xyzz acc; xyzz_from_affine(acc, p, 0);
    p.x.l[0] ^= 5;
    if (V == 0) {
        for (int i = 0; i < iters; ++i) { xyzz_madd<true>(acc, p, i & 1); p.x.l[1] = (p.x.l[1] + 3) & (int32_t)LMASK; }
    } else if (V == 1) {
        uint32_t w[16];
        for (int j = 0; j < 16; ++j) w[j] = t * 2654435761u + j * 40503u;
        w[7] &= 0x0FFFFFFF; w[15] &= 0x0FFFFFFF;
        for (int i = 0; i < iters; ++i) {
            Affine q;
            fe_unpack(q.x, w);
            fe_unpack(q.y, w + 8);
            xyzz_madd<true>(acc, q, (w[0] >> 3) & 1);
            w[0] += 0x9E3779B9u; w[9] ^= w[0];
        }
    } else {
        const uint32_t begin = t * (uint32_t)iters, end = begin + (uint32_t)iters, last = end - 1;
        uint32_t v = V >= 3 ? sorted[begin] : (t * 7u) & idx_mask;
        uint32_t v1 = V >= 3 ? sorted[begin + 1 < end ? begin + 1 : last] : 0u;
        const uint4* src = points + 4 * (size_t)(v & idx_mask);
        uint4 q0 = src[0], q1 = src[1], q2 = src[2], q3 = src[3];
        for (uint32_t e = begin; e < end; ++e) {
            const uint32_t neg = v >> 31;
            uint32_t wx[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
            uint32_t wy[8] = {q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
            Affine q;
            fe_unpack(q.x, wx);
            fe_unpack(q.y, wy);
            if (V >= 3) { v = v1; } else { v = (v * 5u + 1u); }
            src = points + 4 * (size_t)(v & idx_mask);
            q0 = src[0]; q1 = src[1]; q2 = src[2]; q3 = src[3];
            if (V >= 3) v1 = sorted[e + 2 < end ? e + 2 : last];
            xyzz_madd<true>(acc, q, neg);
        }
    }
    uint32_t x = 0; for (int j = 0; j < NL; ++j) x ^= (uint32_t)acc.x.l[j] ^ (uint32_t)acc.y.l[j] ^ (uint32_t)acc.zz.l[j];
    out[t] = x;
}

template <class K> int run(const char* name, K kern, uint32_t* d_out, const uint4* pts, const uint32_t* sorted, uint32_t mask, int iters) {
    const int blocks = 768;
    hipEvent_t t0, t1; CHECK(hipEventCreate(&t0)); CHECK(hipEventCreate(&t1));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, pts, sorted, mask, iters); CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) { CHECK(hipEventRecord(t0)); hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, pts, sorted, mask, iters); CHECK(hipEventRecord(t1)); CHECK(hipEventSynchronize(t1)); float ms; CHECK(hipEventElapsedTime(&ms, t0, t1)); if (ms < best) best = ms; }
    printf("%-34s %.3f ms for %d additions per lane, 3 waves/SIMD -> %.0f ns per madd per SIMD\n", name, best, iters, best * 1e6 / (iters * 3.0));
    return 0;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 80;
    const size_t lanes = 768 * 256;
    const size_t big_pts = (size_t)15 << 20;                       // 960 MiB of 64-byte points
    uint4* d_pts; CHECK(hipMalloc(&d_pts, big_pts * 64));
    // any canonical-looking words do: only the arithmetic cost matters here
    std::vector<uint32_t> h(1 << 20);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)(i * 2654435761u + 12345u) & 0x0FFFFFFFu;
    for (size_t off = 0; off < big_pts * 64; off += h.size() * 4) CHECK(hipMemcpy((char*)d_pts + off, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    std::vector<uint32_t> s(lanes * iters);
    uint64_t x = 88172645463325252ull;
    for (auto& v : s) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (uint32_t)(x >> 20); }
    uint32_t* d_sorted; CHECK(hipMalloc(&d_sorted, s.size() * 4));
    CHECK(hipMemcpy(d_sorted, s.data(), s.size() * 4, hipMemcpyHostToDevice));
    uint32_t* d_out; CHECK(hipMalloc(&d_out, lanes * 4));
    // index masks keep bit 31 (the sign) and select the table size
    const uint32_t small = 0x800000FFu, big = 0x80000000u | (uint32_t)(big_pts - 1);
    (void)small;
    run("V0 bare chain", k_variant<0>, d_out, d_pts, d_sorted, 0xFFu, iters);
    run("V1 + sign, unpack", k_variant<1>, d_out, d_pts, d_sorted, 0xFFu, iters);
    run("V2 + point loads (16 KiB table)", k_variant<2>, d_out, d_pts, d_sorted, 0xFFu, iters);
    run("V3 + index stream (16 KiB table)", k_variant<3>, d_out, d_pts, d_sorted, 0xFFu, iters);
    run("V4 960 MiB random gather", k_variant<4>, d_out, d_pts, d_sorted, (uint32_t)(big_pts - 1), iters);
    (void)big;
    return 0;
}
