// msm_kernels.h — hand-written gfx950 kernels for the G1 multi-scalar multiplication
//     sum_i s_i * P_i
// that backs `G1Projective::msm` at prover/src/kzg.rs:100,121 and primitives/src/helpers.rs:332.
//
// Pipeline (Pippenger, signed c-bit windows, sort-by-bucket):
//   k_msm_digits      scalars (wire) -> canonical integer -> W signed digits; bucket histogram
//   k_scan_*          exclusive scan of (count, #segments) per bucket            (3 small kernels)
//   k_msm_scatter     entries (point index | sign) grouped by bucket             (counting sort)
//   k_msm_segments    heavy buckets are cut into segments of <= L entries (load balance, degenerate inputs)
//   k_msm_accumulate  one lane per segment: XYZZ mixed adds of 64-byte affine points (4 x 128-bit loads)
//   k_msm_bucket_fin  segment partials -> bucket sums
//   k_red_*           per window sum_k (k+1) * B_k by chunked running sums + block suffix scan + tree
// The W window sums leave the device as wire-format XYZZ; the O(W*c) Horner doublings and the single
// field inversion of `into_affine()` run on the host (host_curve.h) next to the D2H copy.
//
// All kernels are integer/VALU bound (no MFMA: there is no dense contraction here).
#pragma once
#include <hip/hip_runtime.h>
#include "curve.h"

namespace kzg {

constexpr uint32_t DIGIT_NONE = 0xFFFFFFFFu;
constexpr int RED_T = 512;          // chunks (= threads of the per-window scan block) per window

// -------------------------------------------------------------------------------------------------
// 1. scalars -> signed digits + histogram
// -------------------------------------------------------------------------------------------------
// digits[w * n + i] = (|d| - 1) | (d < 0) << 31, or DIGIT_NONE for d == 0.
__global__ void __launch_bounds__(256)
k_msm_digits(const uint4* __restrict__ scalars, uint32_t n, int c, int W, uint32_t B,
             uint32_t* __restrict__ digits, uint32_t* __restrict__ count) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint4 lo = scalars[2 * (size_t)i], hi = scalars[2 * (size_t)i + 1];
    uint32_t w32[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    uint32_t k[8];
    fe_wire_to_canonical_words<FrParams>(k, w32);           // `into_bigint()`
    const uint32_t mask = (1u << c) - 1u, half = 1u << (c - 1);
    uint32_t carry = 0;
    for (int w = 0; w < W; ++w) {
        uint32_t raw = (k[0] & mask) + carry;
#pragma unroll
        for (int j = 0; j < 7; ++j) k[j] = (k[j] >> c) | (k[j + 1] << (32 - c));
        k[7] >>= c;
        uint32_t neg = raw > half;
        uint32_t mag = neg ? (1u << c) - raw : raw;          // |digit| in [0, 2^(c-1)]
        carry = neg;
        uint32_t v = DIGIT_NONE;
        if (mag != 0) {
            v = (mag - 1) | (neg << 31);
            atomicAdd(&count[(size_t)w * B + (mag - 1)], 1u);
        }
        digits[(size_t)w * n + i] = v;
    }
}

// -------------------------------------------------------------------------------------------------
// 2. exclusive scan over G buckets of the packed pair (count, ceil(count / L)); out has G + 1 entries
// -------------------------------------------------------------------------------------------------
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

__device__ __forceinline__ unsigned long long scan_pack(uint32_t cnt, uint32_t L) {
    return (unsigned long long)cnt | ((unsigned long long)((cnt + L - 1) / L) << 32);
}
// block-wide exclusive scan of one u64 per thread; returns the exclusive prefix, *total = block sum
__device__ __forceinline__ unsigned long long block_excl_scan(unsigned long long v, unsigned long long* total,
                                                              unsigned long long* lds /* SCAN_THREADS */) {
    int t = threadIdx.x;
    lds[t] = v;
    __syncthreads();
    for (int d = 1; d < SCAN_THREADS; d <<= 1) {
        unsigned long long x = (t >= d) ? lds[t - d] : 0ull;
        __syncthreads();
        lds[t] += x;
        __syncthreads();
    }
    unsigned long long incl = lds[t];
    *total = lds[SCAN_THREADS - 1];
    __syncthreads();
    return incl - v;
}
__global__ void __launch_bounds__(SCAN_THREADS)
k_scan_block_sums(const uint32_t* __restrict__ count, uint32_t G, uint32_t L, unsigned long long* __restrict__ block_sums) {
    __shared__ unsigned long long lds[SCAN_THREADS];
    size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
    unsigned long long s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) if (base + j < G) s += scan_pack(count[base + j], L);
    unsigned long long total;
    block_excl_scan(s, &total, lds);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}
// single block: exclusive scan of nb block sums in place (nb <= SCAN_TILE)
__global__ void __launch_bounds__(SCAN_THREADS)
k_scan_top(unsigned long long* __restrict__ block_sums, uint32_t nb) {
    __shared__ unsigned long long lds[SCAN_THREADS];
    size_t base = (size_t)threadIdx.x * SCAN_ITEMS;
    unsigned long long v[SCAN_ITEMS], s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) { v[j] = (base + j < nb) ? block_sums[base + j] : 0ull; s += v[j]; }
    unsigned long long total;
    unsigned long long pre = block_excl_scan(s, &total, lds);
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) { if (base + j < nb) block_sums[base + j] = pre; pre += v[j]; }
}
__global__ void __launch_bounds__(SCAN_THREADS)
k_scan_final(const uint32_t* __restrict__ count, uint32_t G, uint32_t L, const unsigned long long* __restrict__ block_sums,
             unsigned long long* __restrict__ offs /* G + 1 */) {
    __shared__ unsigned long long lds[SCAN_THREADS];
    size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
    unsigned long long v[SCAN_ITEMS], s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) { v[j] = (base + j < G) ? scan_pack(count[base + j], L) : 0ull; s += v[j]; }
    unsigned long long total;
    unsigned long long pre = block_excl_scan(s, &total, lds) + block_sums[blockIdx.x];
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) { if (base + j < G) offs[base + j] = pre; pre += v[j]; }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == SCAN_THREADS - 1) offs[G] = pre;   // grand total
}

// -------------------------------------------------------------------------------------------------
// 3. scatter entries into bucket order
// -------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_msm_scatter(const uint32_t* __restrict__ digits, uint32_t n, int W, uint32_t B,
              const unsigned long long* __restrict__ offs, uint32_t* __restrict__ cursor, uint32_t* __restrict__ sorted) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int w = 0; w < W; ++w) {
        uint32_t v = digits[(size_t)w * n + i];
        if (v == DIGIT_NONE) continue;
        size_t g = (size_t)w * B + (v & 0x7FFFFFFFu);
        uint32_t pos = (uint32_t)offs[g] + atomicAdd(&cursor[g], 1u);
        sorted[pos] = i | (v & 0x80000000u);
    }
}

// -------------------------------------------------------------------------------------------------
// 4. segment -> bucket map
// -------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_msm_segments(const unsigned long long* __restrict__ offs, uint32_t G, uint32_t* __restrict__ seg_bucket) {
    uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    uint32_t s0 = (uint32_t)(offs[g] >> 32), s1 = (uint32_t)(offs[g + 1] >> 32);
    for (uint32_t s = s0; s < s1; ++s) seg_bucket[s] = g;
}

// -------------------------------------------------------------------------------------------------
// 5. bucket accumulation: one lane per segment
// -------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_msm_accumulate(const uint4* __restrict__ points, const uint32_t* __restrict__ sorted,
                 const uint32_t* __restrict__ seg_bucket, const unsigned long long* __restrict__ offs, uint32_t G, uint32_t L,
                 int32_t* __restrict__ segsum, size_t seg_stride) {
    uint32_t sid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t nseg = (uint32_t)(offs[G] >> 32);
    if (sid >= nseg) return;
    uint32_t g = seg_bucket[sid];
    unsigned long long o0 = offs[g], o1 = offs[g + 1];
    uint32_t s = sid - (uint32_t)(o0 >> 32);
    uint32_t begin = (uint32_t)o0 + s * L, end = (uint32_t)o1;
    if (end - begin > L) end = begin + L;
    Xyzz acc;
    xyzz_set_inf(acc);
    for (uint32_t e = begin; e < end; ++e) {
        uint32_t v = sorted[e];
        Affine p;
        if (!affine_load(p, points + 4 * (size_t)(v & 0x7FFFFFFFu))) continue;      // identity base
        xyzz_madd(acc, p, v >> 31);
    }
    xyzz_store(segsum, seg_stride, sid, acc);
}

// Bucket g of the G = W * B buckets is stored at a transposed position so that the reduction kernels,
// where lane t walks chunk t (buckets t*m .. t*m+m-1), read consecutive addresses across lanes.
__device__ __forceinline__ size_t bucket_pos(uint32_t g, uint32_t m, uint32_t n_chunks) {
    return (size_t)(g % m) * n_chunks + (g / m);
}
__global__ void __launch_bounds__(256)
k_msm_bucket_fin(const unsigned long long* __restrict__ offs, uint32_t G, uint32_t m, uint32_t n_chunks,
                 const int32_t* __restrict__ segsum, size_t seg_stride, int32_t* __restrict__ bucket, size_t bucket_stride) {
    uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    uint32_t s0 = (uint32_t)(offs[g] >> 32), s1 = (uint32_t)(offs[g + 1] >> 32);
    Xyzz acc;
    xyzz_set_inf(acc);
    for (uint32_t s = s0; s < s1; ++s) {
        Xyzz v, t;
        xyzz_load(v, segsum, seg_stride, s);
        xyzz_add(t, acc, v);
        acc = t;
    }
    xyzz_store(bucket, bucket_stride, bucket_pos(g, m, n_chunks), acc);
}

// -------------------------------------------------------------------------------------------------
// 6. bucket reduction per window: sum_{k=0}^{B-1} (k+1) * bucket[k]
//    chunks of m = B / T buckets; T = min(B, RED_T) chunks per window
// -------------------------------------------------------------------------------------------------
// (a) chunk sums S_t
__global__ void __launch_bounds__(256)
k_red_chunk_sums(const int32_t* __restrict__ bucket, size_t bucket_stride, uint32_t n_chunks, uint32_t m,
                 int32_t* __restrict__ chunkS, size_t chunk_stride) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_chunks) return;
    Xyzz acc;
    xyzz_set_inf(acc);
    for (uint32_t k = 0; k < m; ++k) {
        Xyzz v, r;
        xyzz_load(v, bucket, bucket_stride, (size_t)k * n_chunks + t);
        xyzz_add(r, acc, v);
        acc = r;
    }
    xyzz_store(chunkS, chunk_stride, t, acc);
}
// (b) one block per window: inclusive suffix scan of the T chunk sums (Hillis-Steele through global memory)
//     result: chunkS[w*T + t] = sum_{u >= t} S_u
__global__ void __launch_bounds__(RED_T)
k_red_suffix_scan(int32_t* __restrict__ a, int32_t* __restrict__ b, size_t stride, uint32_t T) {
    uint32_t t = threadIdx.x;
    size_t base = (size_t)blockIdx.x * T;
    int32_t* src = a;
    int32_t* dst = b;
    for (uint32_t d = 1; d < T; d <<= 1) {
        if (t < T) {
            Xyzz v;
            xyzz_load(v, src, stride, base + t);
            if (t + d < T) {
                Xyzz u, r;
                xyzz_load(u, src, stride, base + t + d);
                xyzz_add(r, v, u);
                v = r;
            }
            xyzz_store(dst, stride, base + t, v);
        }
        __syncthreads();
        int32_t* tmp = src; src = dst; dst = tmp;
    }
    // make sure the result sits in `a`
    if (src != a && t < T) {
        Xyzz v;
        xyzz_load(v, src, stride, base + t);
        xyzz_store(a, stride, base + t, v);
    }
}
// (c) chunk running sums: A_t = sum_{k in chunk} (weight within window), seeded with the suffix of later chunks
__global__ void __launch_bounds__(256)
k_red_chunk_running(const int32_t* __restrict__ bucket, size_t bucket_stride, const int32_t* __restrict__ suffix, size_t chunk_stride,
                    uint32_t n_chunks, uint32_t T, uint32_t m, int32_t* __restrict__ chunkA) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_chunks) return;
    Xyzz run, acc;
    if ((t % T) + 1 < T) xyzz_load(run, suffix, chunk_stride, t + 1);      // sum of all later chunks of this window
    else xyzz_set_inf(run);
    xyzz_set_inf(acc);
    for (uint32_t k = m; k-- > 0;) {
        Xyzz v, r;
        xyzz_load(v, bucket, bucket_stride, (size_t)k * n_chunks + t);
        xyzz_add(r, run, v); run = r;
        xyzz_add(r, acc, run); acc = r;
    }
    xyzz_store(chunkA, chunk_stride, t, acc);
}
// (d) one block per window: tree-sum of the T chunk results, emitted as wire-format XYZZ (32 u32)
__global__ void __launch_bounds__(RED_T)
k_red_window_sum(int32_t* __restrict__ a, size_t stride, uint32_t T, uint32_t* __restrict__ out_wire) {
    uint32_t t = threadIdx.x;
    size_t base = (size_t)blockIdx.x * T;
    for (uint32_t d = T >> 1; d >= 1; d >>= 1) {
        if (t < d) {
            Xyzz v, u, r;
            xyzz_load(v, a, stride, base + t);
            xyzz_load(u, a, stride, base + t + d);
            xyzz_add(r, v, u);
            xyzz_store(a, stride, base + t, r);
        }
        __syncthreads();
    }
    if (t == 0) {
        Xyzz v;
        xyzz_load(v, a, stride, base);
        uint32_t w[32];
        xyzz_to_wire(w, v);
        for (int j = 0; j < 32; ++j) out_wire[(size_t)blockIdx.x * 32 + j] = w[j];
    }
}

// -------------------------------------------------------------------------------------------------
// SRS / bases upload: wire affine (x || y, radix 2^256) -> device affine format
// -------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_points_wire_to_device(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint4 a = in[4 * i], b = in[4 * i + 1], c = in[4 * i + 2], d = in[4 * i + 3];
    uint32_t w[16] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
    uint32_t o[16];
    affine_wire_to_device(o, w);
    out[4 * i] = make_uint4(o[0], o[1], o[2], o[3]);
    out[4 * i + 1] = make_uint4(o[4], o[5], o[6], o[7]);
    out[4 * i + 2] = make_uint4(o[8], o[9], o[10], o[11]);
    out[4 * i + 3] = make_uint4(o[12], o[13], o[14], o[15]);
}

}  // namespace kzg
