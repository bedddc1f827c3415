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
// `n_total` scalars = batch * n; scalar j belongs to MSM j / n, whose digit rows are (j / n) * W + w.
__global__ void __launch_bounds__(256)
k_msm_digits(const uint4* __restrict__ scalars, uint32_t n_total, uint32_t n, int c, int W, uint32_t* __restrict__ digits) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_total) return;
    const uint32_t msm = j / n, i = j - msm * n;
    uint4 lo = scalars[2 * (size_t)j], hi = scalars[2 * (size_t)j + 1];
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
        if (mag != 0) v = (mag - 1) | (neg << 31);
        digits[((size_t)msm * W + w) * n + i] = v;
    }
}

// -------------------------------------------------------------------------------------------------
// 2. exclusive scan over G buckets of the packed pair (count, ceil(count / L)); out has G + 1 entries
// -------------------------------------------------------------------------------------------------
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

// segments of a bucket with cnt entries: round(cnt / L) (>= 1 if cnt > 0); the accumulate kernel cuts the bucket into
// that many EQUAL parts, so every lane of a wave runs nearly the same trip count (<= 1.5 L) and the segment total
// stays ~ entries / L (a ceil() here made 6 % extra, tiny segments and a ragged second round of waves)
__device__ __forceinline__ uint32_t seg_count(uint32_t cnt, uint32_t L) {
    if (cnt == 0) return 0;
    uint32_t s = (cnt + L / 2) / L;
    return s ? s : 1u;
}
__device__ __forceinline__ unsigned long long scan_pack(uint32_t cnt, uint32_t L) {
    return (unsigned long long)cnt | ((unsigned long long)seg_count(cnt, L) << 32);
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
// 3. counting sort of the digit array by bucket, histograms and cursors privatised in LDS
// -------------------------------------------------------------------------------------------------
// The digit array is cut into `sets` independent key spaces of `set_len` entries each (sets = W windows of n
// entries, or ONE set of W*n entries when the SRS carries precomputed window tables); every workgroup owns one
// tile of one set.  Random global atomics ran at ~19 G/s (v0: 2.5 ms for 2^20 x 19 entries); LDS atomics plus one
// contiguous flush per tile remove that cost.
__global__ void __launch_bounds__(256)
k_sort_hist(const uint32_t* __restrict__ digits, uint32_t set_len, uint32_t tile_len, uint32_t tiles_per_set, uint32_t B,
            uint32_t* __restrict__ count, uint32_t* __restrict__ blockbase) {
    extern __shared__ uint32_t lds_u32[];
    const uint32_t set = blockIdx.x / tiles_per_set, tile = blockIdx.x % tiles_per_set;
    const uint32_t lo = tile * tile_len;
    const uint32_t hi = (set_len - lo < tile_len) ? set_len : lo + tile_len;
    for (uint32_t b = threadIdx.x; b < B; b += blockDim.x) lds_u32[b] = 0;
    __syncthreads();
    const uint32_t* d = digits + (size_t)set * set_len;
    for (uint32_t e = lo + threadIdx.x; e < hi; e += blockDim.x) {
        uint32_t v = d[e];
        if (v != DIGIT_NONE) atomicAdd(&lds_u32[v & 0x7FFFFFFFu], 1u);
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < B; b += blockDim.x) {
        uint32_t h = lds_u32[b];
        uint32_t base = h ? atomicAdd(&count[(size_t)set * B + b], h) : 0u;     // contiguous addresses across lanes
        blockbase[(size_t)blockIdx.x * B + b] = base;
    }
}
// entry written for digit e of a set: point index | sign << 31.
//   table_stride == 0 : index = i                       (bases used as given, one bucket set per window)
//   table_stride  > 0 : index = w * table_stride + i    (precomputed tables T_w[i] = 2^(c w) P_i, one bucket set)
__global__ void __launch_bounds__(256)
k_sort_scatter(const uint32_t* __restrict__ digits, uint32_t n, uint32_t set_len, uint32_t tile_len, uint32_t tiles_per_set, uint32_t B,
               const unsigned long long* __restrict__ offs, const uint32_t* __restrict__ blockbase, uint32_t table_stride,
               uint32_t windows_per_msm, uint32_t* __restrict__ sorted) {
    extern __shared__ uint32_t lds_u32[];
    const uint32_t set = blockIdx.x / tiles_per_set, tile = blockIdx.x % tiles_per_set;
    const uint32_t base_idx = table_stride ? 0u : (set / windows_per_msm) * n;      // batched MSMs: bases are concatenated
    const uint32_t lo = tile * tile_len;
    const uint32_t hi = (set_len - lo < tile_len) ? set_len : lo + tile_len;
    for (uint32_t b = threadIdx.x; b < B; b += blockDim.x)
        lds_u32[b] = (uint32_t)offs[(size_t)set * B + b] + blockbase[(size_t)blockIdx.x * B + b];
    __syncthreads();
    const uint32_t* d = digits + (size_t)set * set_len;
    for (uint32_t e = lo + threadIdx.x; e < hi; e += blockDim.x) {
        uint32_t v = d[e];
        if (v == DIGIT_NONE) continue;
        uint32_t pos = atomicAdd(&lds_u32[v & 0x7FFFFFFFu], 1u);
        uint32_t idx;
        if (table_stride) { uint32_t w = e / n; idx = w * table_stride + (e - w * n); }
        else idx = base_idx + e;
        sorted[pos] = idx | (v & 0x80000000u);
    }
}

// Small problems (< 2^18 entries): the LDS-privatised sort above is sized by the BUCKET count (every tile zeroes and flushes B
// counters: 57 + 64 us for a 2048-coefficient commitment against a 2^19-point SRS, B = 16384), while a few 10^4 global atomics
// take microseconds.  One thread per entry; the order inside a bucket is arbitrary (the bucket sum does not depend on it).
__global__ void __launch_bounds__(256)
k_sort_small_hist(const uint32_t* __restrict__ digits, uint32_t n_entries, uint32_t set_len, uint32_t B, uint32_t* __restrict__ count) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_entries) return;
    const uint32_t v = digits[e];
    if (v != DIGIT_NONE) atomicAdd(&count[(size_t)(e / set_len) * B + (v & 0x7FFFFFFFu)], 1u);
}
__global__ void __launch_bounds__(256)
k_sort_small_scatter(const uint32_t* __restrict__ digits, uint32_t n_entries, uint32_t n, uint32_t set_len, uint32_t B,
                     const unsigned long long* __restrict__ offs, uint32_t* __restrict__ cursor /* G zeros */, uint32_t table_stride,
                     uint32_t windows_per_msm, uint32_t* __restrict__ sorted) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_entries) return;
    const uint32_t v = digits[e];
    if (v == DIGIT_NONE) return;
    const uint32_t set = e / set_len, es = e - set * set_len;
    const size_t gb = (size_t)set * B + (v & 0x7FFFFFFFu);
    const uint32_t pos = (uint32_t)offs[gb] + atomicAdd(&cursor[gb], 1u);
    uint32_t idx;
    if (table_stride) { const uint32_t w = es / n; idx = w * table_stride + (es - w * n); }
    else idx = (set / windows_per_msm) * n + es;
    sorted[pos] = idx | (v & 0x80000000u);
}

// -------------------------------------------------------------------------------------------------
// 3b. two-level counting sort (table mode): coarse bins first, then the low key bits inside every bin
// -------------------------------------------------------------------------------------------------
// The single-pass scatter above writes 4-byte entries to ~2^15 different bucket regions: rocprofv3 counted 486 MiB
// written for 64 MiB of payload (partial-line writes).  Here pass 1 groups entries by the high key bits (<= 512
// bins: every tile writes runs of >= 256 B per bin), pass 2 sorts each bin's entries by the low 6 key bits (runs of
// ~512 B per bucket).  Between the passes an entry is one u32: sign << 31 | low key << 25 | point index (< 2^25).
constexpr int SORT2_LO_BITS = 6;
constexpr uint32_t SORT2_LO = 1u << SORT2_LO_BITS;
constexpr uint32_t SORT2_IDX_MASK = (1u << 25) - 1u;
constexpr uint32_t SORT2_CHUNK = 16384;         // entries per pass-2 tile

__global__ void __launch_bounds__(256)
k_sort2_hist1(const uint32_t* __restrict__ digits, uint32_t E, uint32_t tile_len, uint32_t Hb,
              uint32_t* __restrict__ ccount, uint32_t* __restrict__ blockbase1) {
    extern __shared__ uint32_t lds_u32[];
    const uint32_t lo = blockIdx.x * tile_len;
    const uint32_t hi = (E - lo < tile_len) ? E : lo + tile_len;
    for (uint32_t b = threadIdx.x; b < Hb; b += blockDim.x) lds_u32[b] = 0;
    __syncthreads();
    for (uint32_t e = lo + threadIdx.x; e < hi; e += blockDim.x) {
        uint32_t v = digits[e];
        if (v != DIGIT_NONE) atomicAdd(&lds_u32[(v & 0x7FFFFFFFu) >> SORT2_LO_BITS], 1u);
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < Hb; b += blockDim.x) {
        uint32_t h = lds_u32[b];
        blockbase1[(size_t)blockIdx.x * Hb + b] = h ? atomicAdd(&ccount[b], h) : 0u;
    }
}
// single block: cstart[h] = exclusive scan of the coarse counts, tstart[h] = exclusive scan of ceil(count / CHUNK)
__global__ void __launch_bounds__(512)
k_sort2_scan(const uint32_t* __restrict__ ccount, uint32_t Hb, uint32_t* __restrict__ cstart, uint32_t* __restrict__ tstart) {
    __shared__ uint32_t a[512], b[512];
    const uint32_t t = threadIdx.x;
    uint32_t c = t < Hb ? ccount[t] : 0u;
    uint32_t k = (c + SORT2_CHUNK - 1) / SORT2_CHUNK;
    a[t] = c; b[t] = k;
    __syncthreads();
    for (uint32_t d = 1; d < 512; d <<= 1) {
        uint32_t x = t >= d ? a[t - d] : 0u, y = t >= d ? b[t - d] : 0u;
        __syncthreads();
        a[t] += x; b[t] += y;
        __syncthreads();
    }
    if (t < Hb) { cstart[t] = a[t] - c; tstart[t] = b[t] - k; }
    if (t == Hb - 1) { cstart[Hb] = a[t]; tstart[Hb] = b[t]; }
}
__global__ void __launch_bounds__(256)
k_sort2_scatter1(const uint32_t* __restrict__ digits, uint32_t n, uint32_t E, uint32_t tile_len, uint32_t Hb,
                 const uint32_t* __restrict__ cstart, const uint32_t* __restrict__ blockbase1, uint32_t table_stride,
                 uint32_t* __restrict__ tmp1) {
    extern __shared__ uint32_t lds_u32[];
    const uint32_t lo = blockIdx.x * tile_len;
    const uint32_t hi = (E - lo < tile_len) ? E : lo + tile_len;
    for (uint32_t b = threadIdx.x; b < Hb; b += blockDim.x) lds_u32[b] = cstart[b] + blockbase1[(size_t)blockIdx.x * Hb + b];
    __syncthreads();
    for (uint32_t e = lo + threadIdx.x; e < hi; e += blockDim.x) {
        uint32_t v = digits[e];
        if (v == DIGIT_NONE) continue;
        uint32_t key = v & 0x7FFFFFFFu;
        uint32_t pos = atomicAdd(&lds_u32[key >> SORT2_LO_BITS], 1u);
        uint32_t w = e / n;
        uint32_t idx = w * table_stride + (e - w * n);
        tmp1[pos] = (v & 0x80000000u) | ((key & (SORT2_LO - 1)) << 25) | idx;
    }
}
// pass-2 tile -> coarse bin
__global__ void __launch_bounds__(256)
k_sort2_tiles(const uint32_t* __restrict__ tstart, uint32_t Hb, uint32_t* __restrict__ tile_bin) {
    uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= Hb) return;
    for (uint32_t t = tstart[h]; t < tstart[h + 1]; ++t) tile_bin[t] = h;
}
__global__ void __launch_bounds__(256)
k_sort2_hist2(const uint32_t* __restrict__ tmp1, const uint32_t* __restrict__ cstart, const uint32_t* __restrict__ tstart,
              const uint32_t* __restrict__ tile_bin, uint32_t Hb, uint32_t* __restrict__ count, uint32_t* __restrict__ blockbase2) {
    __shared__ uint32_t hist[SORT2_LO];
    const uint32_t tile = blockIdx.x;
    if (tile >= tstart[Hb]) return;
    const uint32_t h = tile_bin[tile];
    const uint32_t lo = cstart[h] + (tile - tstart[h]) * SORT2_CHUNK;
    const uint32_t hi = (cstart[h + 1] - lo < SORT2_CHUNK) ? cstart[h + 1] : lo + SORT2_CHUNK;
    if (threadIdx.x < SORT2_LO) hist[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t e = lo + threadIdx.x; e < hi; e += blockDim.x) atomicAdd(&hist[(tmp1[e] >> 25) & (SORT2_LO - 1)], 1u);
    __syncthreads();
    if (threadIdx.x < SORT2_LO) {
        uint32_t c = hist[threadIdx.x];
        blockbase2[(size_t)tile * SORT2_LO + threadIdx.x] = c ? atomicAdd(&count[(size_t)h * SORT2_LO + threadIdx.x], c) : 0u;
    }
}
__global__ void __launch_bounds__(256)
k_sort2_scatter2(const uint32_t* __restrict__ tmp1, const uint32_t* __restrict__ cstart, const uint32_t* __restrict__ tstart,
                 const uint32_t* __restrict__ tile_bin, uint32_t Hb, const unsigned long long* __restrict__ offs,
                 const uint32_t* __restrict__ blockbase2, uint32_t* __restrict__ sorted) {
    __shared__ uint32_t cur[SORT2_LO];
    const uint32_t tile = blockIdx.x;
    if (tile >= tstart[Hb]) return;
    const uint32_t h = tile_bin[tile];
    const uint32_t lo = cstart[h] + (tile - tstart[h]) * SORT2_CHUNK;
    const uint32_t hi = (cstart[h + 1] - lo < SORT2_CHUNK) ? cstart[h + 1] : lo + SORT2_CHUNK;
    if (threadIdx.x < SORT2_LO)
        cur[threadIdx.x] = (uint32_t)offs[(size_t)h * SORT2_LO + threadIdx.x] + blockbase2[(size_t)tile * SORT2_LO + threadIdx.x];
    __syncthreads();
    for (uint32_t e = lo + threadIdx.x; e < hi; e += blockDim.x) {
        uint32_t v = tmp1[e];
        uint32_t pos = atomicAdd(&cur[(v >> 25) & (SORT2_LO - 1)], 1u);
        sorted[pos] = v & (0x80000000u | SORT2_IDX_MASK);
    }
}

// -------------------------------------------------------------------------------------------------
// 4. segment -> bucket map
// -------------------------------------------------------------------------------------------------
// One thread per SEGMENT: binary search of its id in the per-bucket segment offsets (high words of offs[]).  A thread per
// bucket looping over its segments serialised ~40 dependent stores when few buckets hold many segments (shard-sized MSMs:
// 77 us at 2^17 pairs against 6 us at 2^20).
__global__ void __launch_bounds__(256)
k_msm_segments(const unsigned long long* __restrict__ offs, uint32_t G, uint32_t* __restrict__ seg_bucket) {
    const uint32_t sid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t nseg = (uint32_t)(offs[G] >> 32);
    if (sid >= nseg) return;
    uint32_t lo = 0, hi = G;                       // invariant: segstart(lo) <= sid < segstart(hi)
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if ((uint32_t)(offs[mid] >> 32) <= sid) lo = mid; else hi = mid;
    }
    seg_bucket[sid] = lo;
}

// -------------------------------------------------------------------------------------------------
// 5. bucket accumulation: one lane per segment
// -------------------------------------------------------------------------------------------------

__device__ __forceinline__ void xyzz_shfl_down(Xyzz& r, const Xyzz& v, int d) {
    const Fq* s[4] = {&v.x, &v.y, &v.zz, &v.zzz};
    Fq* t[4] = {&r.x, &r.y, &r.zz, &r.zzz};
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < NL; ++j) t[q]->l[j] = __shfl_down(s[q]->l[j], d, 64);
    r.inf = __shfl_down((int)v.inf, d, 64) != 0;
}

// One lane per segment (<= L entries of one bucket).  Segments of a bucket are consecutive, so after the serial part
// the lanes of a wave that share a bucket fold their partials with a SEGMENTED suffix scan over ds_bpermute
// shuffles (all lanes busy; ~log2(segments per bucket) extra adds).  Only the first lane of each run stores:
// the surviving partials of bucket g sit at segment ids  s0(g)  and the multiples of 64 inside (s0, s1).
// Waves per SIMD of the accumulate kernel: 3 (<= 168 VGPRs, 4 spilled dwords) or 4 (128 VGPRs, ~50 spilled dwords).  Measured back to
// back on one box they are equal (1.499 / 1.494 ms per 2^20-pair launch at L = 96), as are variants without the point prefetch: the
// kernel is bound by instruction issue, and the GPU boxes of the pool differ by +-6 % among themselves (1.33 .. 1.50 ms for the same
// binary), which is more than any of these variants.  3 is kept for the smaller scratch traffic (1.29 vs 1.63 GB per launch, PMC).
// experiment switch: -DKZG_ACC_NOGATHER makes every lane read the same few table points (no HBM gather) to time the arithmetic alone
#ifdef KZG_ACC_NOGATHER
#define KZG_ACC_IDX(v) ((v) & 0xFFu)
#else
#define KZG_ACC_IDX(v) ((v) & 0x7FFFFFFFu)
#endif
#ifndef KZG_ACC_WAVES
#define KZG_ACC_WAVES 3
#endif
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(KZG_ACC_WAVES, KZG_ACC_WAVES)))
k_msm_accumulate(const uint4* __restrict__ points, const uint32_t* __restrict__ sorted,
                 const uint32_t* __restrict__ seg_bucket, const unsigned long long* __restrict__ offs, uint32_t G, uint32_t L,
                 int32_t* __restrict__ segsum, size_t seg_stride, uint32_t do_scan) {
    const uint32_t sid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t nseg = (uint32_t)(offs[G] >> 32);
    const bool active = sid < nseg;
    uint32_t g = 0xFFFFFFFFu, begin = 0, end = 0;
    if (active) {
        g = seg_bucket[sid];
        unsigned long long o0 = offs[g], o1 = offs[g + 1];
        const uint32_t s = sid - (uint32_t)(o0 >> 32), ns = (uint32_t)(o1 >> 32) - (uint32_t)(o0 >> 32);
        const uint32_t cnt = (uint32_t)o1 - (uint32_t)o0;
        begin = (uint32_t)o0 + (uint32_t)(((unsigned long long)s * cnt) / ns);
        end = (uint32_t)o0 + (uint32_t)(((unsigned long long)(s + 1) * cnt) / ns);
    }
    Xyzz acc;
    xyzz_set_inf(acc);
    if (begin < end) {
        // software pipeline: the 64-byte point of entry e+1 is requested before the ~2 700-instruction mixed add of
        // entry e, so the random gather (HBM-resident tables) is hidden behind arithmetic
        // two-deep pipeline: index e+2 and point e+1 are in flight while entry e is added, so neither the index
        // load (dependent address) nor the 64-byte gather is waited for inside an iteration
        uint32_t v = sorted[begin];
        uint32_t v1 = (begin + 1 < end) ? sorted[begin + 1] : 0u;
        const uint4* src = points + 4 * (size_t)(KZG_ACC_IDX(v));
        uint4 q0 = src[0], q1 = src[1], q2 = src[2], q3 = src[3];
        for (uint32_t e = begin; e < end; ++e) {
            const uint32_t neg = v >> 31;
            uint32_t wx[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
            uint32_t wy[8] = {q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
            const uint32_t any = q0.x | q0.y | q0.z | q0.w | q1.x | q1.y | q1.z | q1.w | q2.x | q2.y | q2.z | q2.w | q3.x | q3.y | q3.z | q3.w;
            if (e + 1 < end) {
                v = v1;
                src = points + 4 * (size_t)(KZG_ACC_IDX(v));
                q0 = src[0]; q1 = src[1]; q2 = src[2]; q3 = src[3];
                if (e + 2 < end) v1 = sorted[e + 2];
            }
            if (any == 0) continue;                                                       // identity base
            Affine p;
            fe_unpack(p.x, wx);
            fe_unpack(p.y, wy);
            xyzz_madd<true>(acc, p, neg);
        }
    }
    if (!do_scan) {                                      // fold mode: every segment's partial goes to memory, k_msm_fold combines them
        if (active) xyzz_store(segsum, seg_stride, sid, acc);
        return;
    }
    // segmented suffix scan: acc_lane = sum of the partials of lanes lane .. end of its bucket run in this wave
#pragma unroll 1
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t gd = __shfl_down(g, d, 64);
        const bool join = active && (lane + d < 64) && gd == g;
        if (!__any(join)) continue;                      // wave-uniform: nobody has a partner at this distance
        Xyzz u;
        xyzz_shfl_down(u, acc, d);
        if (join) {
            Xyzz r;
            xyzz_add<true>(r, acc, u);
            acc = r;
        }
    }
    const uint32_t gprev = __shfl_up(g, 1, 64);
    if (active && (lane == 0 || gprev != g)) xyzz_store(segsum, seg_stride, sid, acc);
}

// Bucket g of the G = W * B buckets is stored at a transposed position so that the reduction kernels,
// where lane t walks chunk t (buckets t*m .. t*m+m-1), read consecutive addresses across lanes.
__device__ __forceinline__ size_t bucket_pos(uint32_t g, uint32_t m, uint32_t n_chunks) {
    return (size_t)(g % m) * n_chunks + (g / m);
}
// surviving partial k of a bucket whose segments are [s0, s1): k = 0 -> s0, k >= 1 -> the k-th multiple of 64 above s0
__device__ __forceinline__ uint32_t partial_count(uint32_t s0, uint32_t s1) {
    if (s1 == s0) return 0;
    return 1 + ((s1 - 1) / 64 - s0 / 64);
}
__device__ __forceinline__ uint32_t partial_sid(uint32_t s0, uint32_t k) { return k == 0 ? s0 : (s0 / 64 + k) * 64; }

constexpr uint32_t FIN_SERIAL_MAX = 4;
constexpr uint32_t FOLD_F = 8;            // partials one lane of k_msm_fold sums serially
constexpr uint32_t FOLD_SERIAL_MAX = 8;   // fold mode: bucket_fin sums up to 8 folded partials serially (64 segments)

// ---- fold mode (many segments per bucket: shard-sized MSMs) -------------------------------------------------------------
// The in-wave suffix scan costs log2(run) full additions on EVERY lane (5-6 steps x 14 multiplies against ~24 mixed adds of
// 10: +30-55 % work when a bucket spans 27-43 lanes).  Here each lane stores its partial; a second, tiny launch lets one lane
// sum FOLD_F consecutive partials of one bucket (one addition per partial: work-efficient, longer dependent chain), and
// bucket_fin adds the <= 8 results of a normal bucket.
// fold_start[g] = first folded partial of bucket g (exclusive prefix sum of ceil(ns / FOLD_F)), fold_start[G] = total
__global__ void __launch_bounds__(1024)
k_fold_offsets(const unsigned long long* __restrict__ offs, uint32_t G, uint32_t* __restrict__ fold_start) {
    __shared__ uint32_t sums[1024];
    const uint32_t t = threadIdx.x;
    const uint32_t per = (G + 1023) / 1024;
    const uint32_t lo = min(G, t * per), hi = min(G, lo + per);
    uint32_t acc = 0;
    for (uint32_t g = lo; g < hi; ++g) {
        const uint32_t ns = (uint32_t)(offs[g + 1] >> 32) - (uint32_t)(offs[g] >> 32);
        acc += (ns + FOLD_F - 1) / FOLD_F;
    }
    sums[t] = acc;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {           // inclusive scan of the per-thread totals
        uint32_t v = t >= d ? sums[t - d] : 0;
        __syncthreads();
        sums[t] += v;
        __syncthreads();
    }
    uint32_t run = sums[t] - acc;                       // exclusive prefix of this thread's chunk
    for (uint32_t g = lo; g < hi; ++g) {
        fold_start[g] = run;
        const uint32_t ns = (uint32_t)(offs[g + 1] >> 32) - (uint32_t)(offs[g] >> 32);
        run += (ns + FOLD_F - 1) / FOLD_F;
    }
    if (t == 1023) fold_start[G] = sums[1023];
}
// one lane per folded partial: sum of FOLD_F consecutive segment partials of one bucket
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_msm_fold(const unsigned long long* __restrict__ offs, const uint32_t* __restrict__ fold_start, uint32_t G,
           const int32_t* __restrict__ segsum, size_t seg_stride, int32_t* __restrict__ foldsum, size_t fold_stride) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= fold_start[G]) return;
    uint32_t lo = 0, hi = G;                             // largest g with fold_start[g] <= t (empty buckets have equal starts)
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (fold_start[mid] <= t) lo = mid; else hi = mid;
    }
    const uint32_t s0 = (uint32_t)(offs[lo] >> 32), s1 = (uint32_t)(offs[lo + 1] >> 32);
    const uint32_t a = s0 + (t - fold_start[lo]) * FOLD_F, b = min(a + FOLD_F, s1);
    Xyzz acc;
    xyzz_load(acc, segsum, seg_stride, a);
    for (uint32_t i = a + 1; i < b; ++i) {
        Xyzz v, r;
        xyzz_load(v, segsum, seg_stride, i);
        xyzz_add<true>(r, acc, v);
        acc = r;
    }
    xyzz_store(foldsum, fold_stride, t, acc);
}

// thread per bucket: buckets with <= 4 surviving partials (the normal case: 1-2) are summed serially; heavier ones
// (skewed scalars: few distinct digits) are queued for k_msm_bucket_fin_heavy.
__global__ void __launch_bounds__(256)
k_msm_bucket_fin(const unsigned long long* __restrict__ offs, uint32_t G, uint32_t m, uint32_t n_chunks,
                 const int32_t* __restrict__ segsum, size_t seg_stride, int32_t* __restrict__ bucket, size_t bucket_stride,
                 uint32_t* __restrict__ heavy /* [0] = count, [1..] = bucket ids */,
                 const uint32_t* __restrict__ fold_start /* fold mode: partials of bucket g = segsum[fold_start[g] .. fold_start[g+1]) */) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    uint32_t s0, s1, np;
    if (fold_start) { s0 = fold_start[g]; s1 = fold_start[g + 1]; np = s1 - s0; }
    else { s0 = (uint32_t)(offs[g] >> 32); s1 = (uint32_t)(offs[g + 1] >> 32); np = partial_count(s0, s1); }
    if (np > (fold_start ? FOLD_SERIAL_MAX : FIN_SERIAL_MAX)) {
        heavy[1 + atomicAdd(&heavy[0], 1u)] = g;
        return;
    }
    Xyzz acc;
    xyzz_set_inf(acc);
    for (uint32_t k = 0; k < np; ++k) {
        Xyzz v, t;
        xyzz_load(v, segsum, seg_stride, fold_start ? s0 + k : partial_sid(s0, k));
        xyzz_add<true>(t, acc, v);
        acc = t;
    }
    xyzz_store(bucket, bucket_stride, m ? bucket_pos(g, m, n_chunks) : (size_t)g, acc);
}
// one wave per queued bucket (grid-stride over the queue): lanes take partials round-robin, then a shuffle tree
__global__ void __launch_bounds__(256)
k_msm_bucket_fin_heavy(const unsigned long long* __restrict__ offs, uint32_t m, uint32_t n_chunks,
                       const int32_t* __restrict__ segsum, size_t seg_stride, int32_t* __restrict__ bucket, size_t bucket_stride,
                       const uint32_t* __restrict__ heavy, const uint32_t* __restrict__ fold_start) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t count = heavy[0];
    for (uint32_t h = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; h < count; h += n_waves) {
        const uint32_t g = heavy[1 + h];
        uint32_t s0, s1, np;
        if (fold_start) { s0 = fold_start[g]; s1 = fold_start[g + 1]; np = s1 - s0; }
        else { s0 = (uint32_t)(offs[g] >> 32); s1 = (uint32_t)(offs[g + 1] >> 32); np = partial_count(s0, s1); }
        Xyzz acc;
        xyzz_set_inf(acc);
        for (uint32_t k = lane; k < np; k += 64) {
            Xyzz v, t;
            xyzz_load(v, segsum, seg_stride, fold_start ? s0 + k : partial_sid(s0, k));
            xyzz_add<true>(t, acc, v);
            acc = t;
        }
#pragma unroll 1
        for (int d = 32; d >= 1; d >>= 1) {
            Xyzz u, r;
            xyzz_shfl_down(u, acc, d);
            xyzz_add<true>(r, acc, u);
            acc = r;
        }
        if (lane == 0) xyzz_store(bucket, bucket_stride, m ? bucket_pos(g, m, n_chunks) : (size_t)g, acc);
    }
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
// 6b. low-latency bucket reduction for ONE bucket set (precomputed-table mode), B a multiple of 64
// -------------------------------------------------------------------------------------------------
// sum_b (b+1) V_b = T + sum_j 2^j S_j,  T = sum_b V_b,  S_j = sum_{b : bit j of b} V_b.
// A dependent EC addition costs ~13 us on a lone wave, so the classic running sum (2 serial adds per bucket of
// a chunk) is latency bound; here every partial sum is a 6-step wave tree over ds_bpermute shuffles:
// one wave per (group of 64 values, role); role j < 6 sums the lanes whose index has bit j set, role 6 sums all.
// Applied again to the group totals it yields bits 6..11, and so on; the few remaining values go to the host.
// Superset-sum ("zeta") transform over the 64 lanes of a wave: after the 6 steps lane x holds the sum of the values
// of all lanes l with (l & x) == x.  Lane 0 = total T; lane 2^k = S_k (sum over the lanes whose index has bit k set):
// all seven sums the bucket reduction needs from a group of 64 buckets come out of 6 wave-wide additions.
__device__ __forceinline__ void wave_zeta(Xyzz& v, uint32_t lane) {
#pragma unroll 1
    for (int k = 0; k < 6; ++k) {
        Xyzz u;
        xyzz_shfl_down(u, v, 1 << k);
        if (((lane >> k) & 1u) == 0) {
            Xyzz r;
            xyzz_add<true>(r, v, u);
            v = r;
        }
    }
}
__device__ __forceinline__ int zeta_role(uint32_t lane) {       // lane 0 -> role 6 (total), lane 2^k -> role k, else -1
    if (lane == 0) return 6;
    if ((lane & (lane - 1)) != 0) return -1;
    return __ffs((int)lane) - 1;
}
// level 1: X1[role * G1 + g] for the bucket group g (64 buckets): role k < 6 = S_k, role 6 = T.   One wave per group.
__global__ void __launch_bounds__(256)
k_red_bits1(const int32_t* __restrict__ bucket, size_t bucket_stride, uint32_t B, uint32_t G1,
            int32_t* __restrict__ x1, size_t x_stride) {
    const uint32_t g = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (g >= G1) return;
    Xyzz v;
    if (g * 64 + lane < B) xyzz_load(v, bucket, bucket_stride, (size_t)g * 64 + lane);
    else xyzz_set_inf(v);
    wave_zeta(v, lane);
    const int role = zeta_role(lane);
    if (role >= 0) xyzz_store(x1, x_stride, (size_t)role * G1 + g, v);
}
// level 2 (one launch, two kinds of job): with G1p = ceil(G1 / 64)
//   wave <  6 G1p : Y[a * G1p + g'] = sum of X1[a][g' * 64 .. +64)                      (a < 6: finishes bits 0..5)
//   wave >= 6 G1p : X2[role * G1p + g2], role 0..6, zeta transform of the totals X1[6][g2 * 64 .. +64)  (bits 6..11, totals)
__global__ void __launch_bounds__(256)
k_red_bits2(const int32_t* __restrict__ x1, size_t x_stride, uint32_t G1, uint32_t G1p,
            int32_t* __restrict__ y, int32_t* __restrict__ x2, size_t out_stride) {
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= 7u * G1p) return;
    const bool sum_job = wave < 6u * G1p;
    const uint32_t a = sum_job ? wave / G1p : 6u;
    const uint32_t g = sum_job ? wave % G1p : wave - 6u * G1p;
    const uint32_t cnt = G1 - g * 64 < 64 ? G1 - g * 64 : 64;
    Xyzz v;
    if (lane < cnt) xyzz_load(v, x1, x_stride, (size_t)a * G1 + (size_t)g * 64 + lane);
    else xyzz_set_inf(v);
    wave_zeta(v, lane);
    if (sum_job) {
        if (lane == 0) xyzz_store(y, out_stride, wave, v);
    } else {
        const int role = zeta_role(lane);
        if (role >= 0) xyzz_store(x2, out_stride, (size_t)role * G1p + g, v);
    }
}
// stored-form XYZZ planes -> wire words (32 u32 per element)
__global__ void __launch_bounds__(256)
k_xyzz_to_wire(const int32_t* __restrict__ in, size_t stride, uint32_t n, uint32_t* __restrict__ out_wire) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Xyzz v;
    xyzz_load(v, in, stride, i);
    uint32_t w[32];
    xyzz_to_wire(w, v);
#pragma unroll
    for (int j = 0; j < 32; ++j) out_wire[(size_t)i * 32 + j] = w[j];
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
