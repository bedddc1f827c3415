// Device-vs-host check of naf_for_digits (csrc/naf.h) on a mix of scalars per wave (divergent trip counts).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 naf_device_check.hip -o naf_device_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../rust-kzg-bn254_amd/csrc/naf.h"
using namespace kzg;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_recode(const uint32_t* __restrict__ scalars, int w, uint32_t* __restrict__ out, uint32_t* __restrict__ cnt, uint32_t n, uint32_t* __restrict__ out_hist) {
    __shared__ uint32_t hist[128];
    if (threadIdx.x < 128) hist[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        uint32_t k[8];
        for (int j = 0; j < 8; ++j) k[j] = scalars[8 * (size_t)i + j];
        uint32_t m = 0;
        naf_for_digits(k, w, [&](uint32_t pos, uint32_t key, uint32_t neg) {
            atomicAdd(&hist[key >> 7 & 127], 1u);
            if (m < 20) out[(size_t)i * 20 + m] = (neg << 31) | (pos << 20) | key;
            ++m;
        });
        cnt[i] = m;
    }
    __syncthreads();
    if (threadIdx.x < 128) atomicAdd(&out_hist[threadIdx.x], hist[threadIdx.x]);
}

// the lambda shape of k_sort2_scalars<false, true>: an LDS atomic without a return value, nothing else
__global__ void k_count_only(const uint32_t* __restrict__ scalars, int w, uint32_t n, uint32_t* __restrict__ out_hist) {
    extern __shared__ uint32_t lds_u32[];
    for (uint32_t b = threadIdx.x; b < 128; b += blockDim.x) lds_u32[b] = 0;
    __syncthreads();
    const uint32_t lo = blockIdx.x * 1024, hi = lo + 1024 < n ? lo + 1024 : n;
    for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        uint32_t k[8];
        for (int j = 0; j < 8; ++j) k[j] = scalars[8 * (size_t)i + j];
        naf_for_digits(k, w, [&](uint32_t, uint32_t key, uint32_t) { atomicAdd(&lds_u32[key >> 7], 1u); });
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < 128; b += blockDim.x) if (lds_u32[b]) atomicAdd(&out_hist[b], lds_u32[b]);
}

// the LDS form (naf_for_digits_lds): digits to out, as k_recode does
__global__ void k_recode_lds(const uint32_t* __restrict__ scalars, int w, uint32_t* __restrict__ out, uint32_t* __restrict__ cnt, uint32_t n) {
    extern __shared__ uint32_t lds_u32[];
    uint32_t* col = lds_u32 + threadIdx.x;
    col[8 * blockDim.x] = 0; col[9 * blockDim.x] = 0; col[10 * blockDim.x] = 0;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        for (int j = 0; j < 8; ++j) col[j * blockDim.x] = scalars[8 * (size_t)i + j];
        uint32_t m = 0;
        naf_for_digits_lds(col, blockDim.x, w, [&](uint32_t pos, uint32_t key, uint32_t neg) {
            if (m < 20) out[(size_t)i * 20 + m] = (neg << 31) | (pos << 20) | key;
            ++m;
        });
        cnt[i] = m;
    }
}

int main() {
    const uint32_t n = 4096;
    std::vector<uint32_t> sc(8 * n, 0);
    auto set_ones = [&](uint32_t i, int bits) { for (int b = 0; b < bits; ++b) sc[8 * i + b / 32] |= 1u << (b % 32); };
    uint64_t x = 88172645463325252ull;
    for (uint32_t i = 0; i < n; ++i) {
        switch (i % 5) {
            case 0: set_ones(i, 253); break;                 // 2^253 - 1
            case 1: set_ones(i, 36); break;                  // 2^36 - 1
            case 2: sc[8 * i + 7] = 0x20000000u; break;       // 2^253
            case 3: sc[8 * i] = 1; break;
            default: for (int j = 0; j < 8; ++j) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; sc[8 * i + j] = (uint32_t)(x >> 16); } sc[8 * i + 7] &= 0x1FFFFFFFu; break;
        }
    }
    uint32_t *d_sc, *d_out, *d_cnt, *d_hist;
    CHECK(hipMalloc(&d_hist, 128 * 4));
    CHECK(hipMalloc(&d_sc, sc.size() * 4)); CHECK(hipMalloc(&d_out, (size_t)n * 20 * 4)); CHECK(hipMalloc(&d_cnt, n * 4));
    CHECK(hipMemcpy(d_sc, sc.data(), sc.size() * 4, hipMemcpyHostToDevice));
    int bad = 0;
    for (int w : {16, 18}) {
        CHECK(hipMemset(d_out, 0, (size_t)n * 20 * 4));
        CHECK(hipMemset(d_hist, 0, 128 * 4));
        hipLaunchKernelGGL(k_recode, dim3(n / 256), dim3(256), 0, 0, d_sc, w, d_out, d_cnt, n, d_hist);
        CHECK(hipDeviceSynchronize());
        std::vector<uint32_t> out((size_t)n * 20), cnt(n);
        CHECK(hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(cnt.data(), d_cnt, n * 4, hipMemcpyDeviceToHost));
        {   // the LDS form must give the same digits
            CHECK(hipMemset(d_out, 0, (size_t)n * 20 * 4));
            hipLaunchKernelGGL(k_recode_lds, dim3(n / 256), dim3(256), 11 * 256 * 4, 0, d_sc, w, d_out, d_cnt, n);
            CHECK(hipDeviceSynchronize());
            std::vector<uint32_t> out2((size_t)n * 20), cnt2(n);
            CHECK(hipMemcpy(out2.data(), d_out, out2.size() * 4, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(cnt2.data(), d_cnt, n * 4, hipMemcpyDeviceToHost));
            for (uint32_t i = 0; i < n; ++i) {
                bool same = cnt2[i] == cnt[i];
                for (uint32_t m = 0; same && m < cnt[i] && m < 20; ++m) same = out2[(size_t)i * 20 + m] == out[(size_t)i * 20 + m];
                if (!same) { if (bad < 10) printf("w=%d scalar %u (kind %u): LDS form %u digits, register form %u\n", w, i, i % 5, cnt2[i], cnt[i]); ++bad; }
            }
        }
        std::vector<uint32_t> dh(128), hh(128, 0);
        CHECK(hipMemcpy(dh.data(), d_hist, 128 * 4, hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < n; ++i) {
            uint32_t k[8]; memcpy(k, &sc[8 * i], 32);
            naf_for_digits(k, w, [&](uint32_t, uint32_t key, uint32_t) { hh[key >> 7 & 127]++; });
        }
        {   // count-only kernel (w = 16: keys < 2^14 -> 128 bins)
            if (w == 16) {
                CHECK(hipMemset(d_hist, 0, 128 * 4));
                hipLaunchKernelGGL(k_count_only, dim3(n / 1024), dim3(256), 128 * 4, 0, d_sc, w, n, d_hist);
                CHECK(hipDeviceSynchronize());
                std::vector<uint32_t> dc(128), hc(128, 0);
                CHECK(hipMemcpy(dc.data(), d_hist, 128 * 4, hipMemcpyDeviceToHost));
                for (uint32_t i = 0; i < n; ++i) {
                    uint32_t k[8]; memcpy(k, &sc[8 * i], 32);
                    naf_for_digits(k, w, [&](uint32_t, uint32_t key, uint32_t) { hc[key >> 7]++; });
                }
                for (int b = 0; b < 128; ++b) if (dc[b] != hc[b]) { if (bad < 10) printf("count-only bin %d: device %u host %u\n", b, dc[b], hc[b]); ++bad; }
            }
        }
        for (int b = 0; b < 128; ++b) if (dh[b] != hh[b]) { if (bad < 10) printf("w=%d hist bin %d: device %u host %u\n", w, b, dh[b], hh[b]); ++bad; }
        for (uint32_t i = 0; i < n; ++i) {
            uint32_t k[8]; memcpy(k, &sc[8 * i], 32);
            uint32_t m = 0; bool ok = true;
            naf_for_digits(k, w, [&](uint32_t pos, uint32_t key, uint32_t neg) {
                if (m < 20 && out[(size_t)i * 20 + m] != ((neg << 31) | (pos << 20) | key)) ok = false;
                ++m;
            });
            if (m != cnt[i]) ok = false;
            if (!ok) { if (bad < 10) printf("w=%d scalar %u (kind %u): device %u digits, host %u; first device word %08x\n", w, i, i % 5, cnt[i], m, out[(size_t)i * 20]); ++bad; }
        }
    }
    printf("naf device check: %d mismatches\n", bad);
    return bad != 0;
}
