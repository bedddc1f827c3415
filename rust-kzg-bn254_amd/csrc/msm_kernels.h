// msm_kernels.h — hand-written gfx950 kernels for the G1 multi-scalar multiplication
//     sum_i s_i * P_i
// that backs `G1Projective::msm` at prover/src/kzg.rs:100,121 and primitives/src/helpers.rs:332.
//
// Pipeline (Pippenger, signed c-bit windows, sort-by-bucket):
//   k_msm_digits      scalars (wire) -> canonical integer -> W signed digits     (generic mode and small MSMs)
//   k_scan_*          exclusive scan of the bucket counts                        (one single-workgroup kernel)
//   k_sort*           entries (point index | sign) grouped by bucket             (counting sort; the two-level k_sort2_* path of
//                     table mode works straight from the scalars and produces the bucket offsets itself)
//   k_msm_accumulate  equal split of the sorted entries over the lanes: XYZZ mixed adds of 64-byte affine points
//   k_msm_bucket_*    lane partials -> bucket sums (fused with the first reduction level in table mode)
//   k_red_*           sum_k (k+1) * B_k: zeta transform over shuffles (table mode) / chunked running sums (generic mode)
// The W window sums leave the device as wire-format XYZZ; the O(W*c) Horner doublings and the single
// field inversion of `into_affine()` run on the host (host_curve.h) next to the D2H copy.
//
// All kernels are integer/VALU bound (no MFMA: there is no dense contraction here).
#pragma once
#include <hip/hip_runtime.h>
#include "curve_pair.h"
#include "curve_quad.h"
#include "naf.h"

namespace kzg {

// The latency-bound kernels of one MSM (sort, bucket sums, reductions) usually run BESIDE the accumulate kernel of the other MSM
// in flight: raised wave priority lets their dependent chains issue ahead of the accumulate waves they share a SIMD with (the
// accumulate kernel is throughput bound and loses nothing but those slots).  -DKZG_NO_SETPRIO: A/B switch.
__device__ __forceinline__ void latency_bound_kernel() {
#if !defined(KZG_NO_SETPRIO)
    __builtin_amdgcn_s_setprio(3);
#endif
}

constexpr uint32_t DIGIT_NONE = 0xFFFFFFFFu;
constexpr int RED_T = 512;          // chunks (= threads of the per-window scan block) per window

// -------------------------------------------------------------------------------------------------
// 1. scalars -> signed digits + histogram
// -------------------------------------------------------------------------------------------------
// digits[w * n + i] = (|d| - 1) | (d < 0) << 31, or DIGIT_NONE for d == 0.
// `n_total` scalars = batch * n; scalar j belongs to MSM j / n, whose digit rows are (j / n) * W + w.
__global__ void __launch_bounds__(256)
k_msm_digits(const uint4* __restrict__ scalars, uint32_t n_total, uint32_t n, int c, int W, uint32_t* __restrict__ digits,
             uint32_t* __restrict__ count = nullptr /* one bucket set (table mode, one MSM): the histogram of k_sort_small_hist in the same pass */) {
    latency_bound_kernel();
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
        if (mag != 0) {
            v = (mag - 1) | (neg << 31);
            if (count) atomicAdd(&count[mag - 1], 1u);
        }
        digits[((size_t)msm * W + w) * n + i] = v;
    }
}

// -------------------------------------------------------------------------------------------------
// 2. exclusive scan of the bucket counts: offs[g] = first sorted entry of bucket g, offs[G] = E (entries)
// -------------------------------------------------------------------------------------------------
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

// block-wide exclusive scan of one u32 per thread; returns the exclusive prefix, *total = block sum
template <int T>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* total, uint32_t* lds /* T */) {
    const int t = threadIdx.x;
    lds[t] = v;
    __syncthreads();
    for (int d = 1; d < T; d <<= 1) {
        uint32_t x = (t >= d) ? lds[t - d] : 0u;
        __syncthreads();
        lds[t] += x;
        __syncthreads();
    }
    uint32_t incl = lds[t];
    *total = lds[T - 1];
    __syncthreads();
    return incl - v;
}
// ONE workgroup does the whole scan (G <= SCAN1_MAX: every table-mode MSM and most generic ones): a dependent chain of three
// tiny launches (block sums, top, final) cost ~3 boundaries of 1.5 us plus their own latency for a few 10^4 counters.
// Every wave owns one contiguous range and walks it 256 counters (one 16-byte load per lane, coalesced) at a time, four such
// loads in flight; a first walk gives the wave totals, the second one the offsets (the counters come from L2 then).  The
// thread-per-chunk form before it read 256 B per thread with a stride of 256 B between lanes: 33 us for 2^16 counters.
constexpr int SCAN1_THREADS = 1024;
constexpr uint32_t SCAN1_MAX = 1u << 17;
__device__ __forceinline__ uint4 scan1_load(const uint32_t* __restrict__ count, uint32_t g, uint32_t hi) {
    if (g + 4 <= hi) return *reinterpret_cast<const uint4*>(count + g);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (g < hi) v.x = count[g];
    if (g + 1 < hi) v.y = count[g + 1];
    if (g + 2 < hi) v.z = count[g + 2];
    return v;
}
// LEAN (small table-mode sort): also leaves cursor[g] = offs[g] for the scatter's atomics and CLEARS the counters it has read, so the
// next MSM on this workspace needs neither memset (msm.hip keeps track: MsmWorkspace::count_zero_g).
template <bool LEAN>
__global__ void __launch_bounds__(SCAN1_THREADS)
k_scan_counts_1wg(uint32_t* __restrict__ count, uint32_t G, uint32_t* __restrict__ offs /* G + 1 */, uint32_t* __restrict__ cursor /* LEAN: G */) {
    latency_bound_kernel();
    __shared__ uint32_t wave_total[SCAN1_THREADS / 64];
    const uint32_t t = threadIdx.x, w = t >> 6, lane = t & 63;
    const uint32_t per_wave = (((G + SCAN1_THREADS / 64 - 1) / (SCAN1_THREADS / 64)) + 255u) & ~255u;
    const uint32_t lo = min(G, w * per_wave), hi = min(G, lo + per_wave);
    uint32_t s = 0;
    for (uint32_t g0 = lo; g0 < hi; g0 += 1024) {
        const uint32_t g = g0 + lane * 4;
        const uint4 a = scan1_load(count, g, hi), b = scan1_load(count, g + 256, hi), c = scan1_load(count, g + 512, hi), d = scan1_load(count, g + 768, hi);
        s += a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w + c.x + c.y + c.z + c.w + d.x + d.y + d.z + d.w;
    }
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if (lane == 0) wave_total[w] = s;
    __syncthreads();
    uint32_t run = 0, total = 0;
    for (uint32_t q = 0; q < SCAN1_THREADS / 64; ++q) { const uint32_t x = wave_total[q]; if (q < w) run += x; total += x; }
    for (uint32_t g0 = lo; g0 < hi; g0 += 1024) {
        uint4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = scan1_load(count, g0 + k * 256 + lane * 4, hi);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t g = g0 + k * 256 + lane * 4;
            const uint32_t sum4 = v[k].x + v[k].y + v[k].z + v[k].w;
            uint32_t incl = sum4;
            for (int d = 1; d < 64; d <<= 1) { const uint32_t x = __shfl_up(incl, d, 64); if ((int)lane >= d) incl += x; }
            uint4 o;
            o.x = run + incl - sum4; o.y = o.x + v[k].x; o.z = o.y + v[k].y; o.w = o.z + v[k].z;
            if (g + 4 <= hi) {
                *reinterpret_cast<uint4*>(offs + g) = o;
                if (LEAN) { *reinterpret_cast<uint4*>(cursor + g) = o; *reinterpret_cast<uint4*>(count + g) = make_uint4(0, 0, 0, 0); }
            } else {
                if (g < hi) { offs[g] = o.x; if (LEAN) { cursor[g] = o.x; count[g] = 0; } }
                if (g + 1 < hi) { offs[g + 1] = o.y; if (LEAN) { cursor[g + 1] = o.y; count[g + 1] = 0; } }
                if (g + 2 < hi) { offs[g + 2] = o.z; if (LEAN) { cursor[g + 2] = o.z; count[g + 2] = 0; } }
            }
            run += __shfl(incl, 63, 64);
        }
    }
    if (t == 0) offs[G] = total;
}
// multi-block form for larger G (generic mode with many windows x buckets)
__global__ void __launch_bounds__(SCAN_THREADS)
k_scan_block_sums(const uint32_t* __restrict__ count, uint32_t G, uint32_t* __restrict__ block_sums) {
    __shared__ uint32_t lds[SCAN_THREADS];
    size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) if (base + j < G) s += count[base + j];
    uint32_t total;
    block_excl_scan<SCAN_THREADS>(s, &total, lds);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}
// single block: exclusive scan of nb block sums in place (nb <= SCAN_TILE)
__global__ void __launch_bounds__(SCAN_THREADS)
k_scan_top(uint32_t* __restrict__ block_sums, uint32_t nb) {
    __shared__ uint32_t lds[SCAN_THREADS];
    size_t base = (size_t)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS], s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) { v[j] = (base + j < nb) ? block_sums[base + j] : 0u; s += v[j]; }
    uint32_t total;
    uint32_t pre = block_excl_scan<SCAN_THREADS>(s, &total, lds);
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) { if (base + j < nb) block_sums[base + j] = pre; pre += v[j]; }
}
__global__ void __launch_bounds__(SCAN_THREADS)
k_scan_final(const uint32_t* __restrict__ count, uint32_t G, const uint32_t* __restrict__ block_sums,
             uint32_t* __restrict__ offs /* G + 1 */) {
    __shared__ uint32_t lds[SCAN_THREADS];
    size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS], s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) { v[j] = (base + j < G) ? count[base + j] : 0u; s += v[j]; }
    uint32_t total;
    uint32_t pre = block_excl_scan<SCAN_THREADS>(s, &total, lds) + block_sums[blockIdx.x];
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
    latency_bound_kernel();
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
               const uint32_t* __restrict__ offs, const uint32_t* __restrict__ blockbase, uint32_t table_stride,
               uint32_t windows_per_msm, uint32_t* __restrict__ sorted) {
    latency_bound_kernel();
    extern __shared__ uint32_t lds_u32[];
    const uint32_t set = blockIdx.x / tiles_per_set, tile = blockIdx.x % tiles_per_set;
    const uint32_t base_idx = table_stride ? 0u : (set / windows_per_msm) * n;      // batched MSMs: bases are concatenated
    const uint32_t lo = tile * tile_len;
    const uint32_t hi = (set_len - lo < tile_len) ? set_len : lo + tile_len;
    for (uint32_t b = threadIdx.x; b < B; b += blockDim.x)
        lds_u32[b] = offs[(size_t)set * B + b] + blockbase[(size_t)blockIdx.x * B + b];
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
    latency_bound_kernel();
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_entries) return;
    const uint32_t v = digits[e];
    if (v != DIGIT_NONE) atomicAdd(&count[(size_t)(e / set_len) * B + (v & 0x7FFFFFFFu)], 1u);
}
__global__ void __launch_bounds__(256)
k_sort_small_scatter(const uint32_t* __restrict__ digits, uint32_t n_entries, uint32_t n, uint32_t set_len, uint32_t B,
                     const uint32_t* __restrict__ offs, uint32_t* __restrict__ cursor /* G zeros, or a copy of offs (cursor_is_offs) */, uint32_t table_stride,
                     uint32_t windows_per_msm, uint32_t* __restrict__ sorted, int cursor_is_offs = 0) {
    latency_bound_kernel();
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_entries) return;
    const uint32_t v = digits[e];
    if (v == DIGIT_NONE) return;
    const uint32_t set = e / set_len, es = e - set * set_len;
    const size_t gb = (size_t)set * B + (v & 0x7FFFFFFFu);
    const uint32_t pos = (cursor_is_offs ? 0u : offs[gb]) + atomicAdd(&cursor[gb], 1u);
    uint32_t idx;
    if (table_stride) { const uint32_t w = es / n; idx = w * table_stride + (es - w * n); }
    else idx = (set / windows_per_msm) * n + es;
    sorted[pos] = idx | (v & 0x80000000u);
}

// -------------------------------------------------------------------------------------------------
// 3b. two-level counting sort (table mode): coarse bins first, then the low key bits inside every bin
// -------------------------------------------------------------------------------------------------
// The single-pass scatter above writes 4-byte entries to ~2^15 different bucket regions: rocprofv3 counted 486 MiB
// written for 64 MiB of payload (partial-line writes), and its LDS histogram holds at most 2^15 buckets.  Here pass 1
// groups entries by the high key bits (<= 512 bins: every tile writes runs of >= 256 B per bin), pass 2 sorts each
// bin's entries by the low 7 key bits (runs of ~512 B per bucket): up to 2^16 buckets (c = 17).  Between the passes an
// entry is one u32: sign << 31 | low key << 24 | point index (< 2^24).
//   k_sort2_scalars<false>  scalars -> digits -> coarse histogram per tile      k_sort2_scan     bin starts, large-bin tiles
//   k_sort2_scalars<true>   scalars -> digits -> entries grouped by bin (tmp1)  k_sort2_bin      per bin: fine sort + bucket offsets
//   k_sort2_hist2 / k_sort2_scatter2: the tiled pass 2 for LARGE bins only (their grids exit at once otherwise)
constexpr int SORT2_LO_BITS = 7;
constexpr uint32_t SORT2_LO = 1u << SORT2_LO_BITS;
constexpr int SORT2_IDX_BITS = 24;
constexpr uint32_t SORT2_IDX_MASK = (1u << SORT2_IDX_BITS) - 1u;
constexpr uint32_t SORT2_CHUNK = 4096;          // entries per pass-2 tile of a LARGE bin
constexpr uint32_t SORT2_MAX_BINS = 512;

// Pass 1 works straight from the scalars: a tile is `tile_s` scalars (W entries each), and both kernels recompute the signed
// digits instead of going through a W*n digit array (round 2 first half: digits kernel + hist1 + scatter1 moved 284 MB for the
// 2^20 commitment; these two move 127 MB and drop one launch).
//   SCATTER = false : coarse histogram of the tile -> ccount (global) and the tile's base inside every bin (blockbase1)
//   SCATTER = true  : entries -> tmp1, grouped by coarse bin
// NAF mode (round 3, second half): the SRS carries one table per BIT position (srs.hip: Bit_j[i] = 2^j P_i, 255 x 64 B per point: HBM
// capacity spent to remove additions), so a scalar is recoded in width-w non-adjacent form, w = c + 1: odd digits |d| < 2^(w-1) at
// arbitrary bit positions, at least w positions apart -> 254 / (w + 1) entries per scalar on average instead of 255 / c (13.4 instead
// of 15 at 2^16 buckets) into the SAME number of buckets (key = (|d| - 1) / 2 < 2^(c-1)); the point of an entry is Bit_pos[i].  An
// index pos * stride + i needs 28 bits at 2^20 points, so between the passes the low 7 key bits travel in a byte array of their own
// (tmpk) beside the u32 entries sign << 31 | index.  The consuming recoder: skip the zeros of k + carry (trailing ones of k when a
// negative digit left a carry), take w bits, shift.
template <bool SCATTER>
__global__ void __launch_bounds__(256)
k_sort2_scalars(const uint4* __restrict__ scalars, uint32_t n, int c, int W, uint32_t tile_s, uint32_t Hb, uint32_t* __restrict__ ccount,
                uint32_t* __restrict__ blockbase1, const uint32_t* __restrict__ cstart, uint32_t table_stride, uint32_t* __restrict__ tmp1) {
    latency_bound_kernel();
    extern __shared__ uint32_t lds_u32[];
    const uint32_t lo = blockIdx.x * tile_s;
    const uint32_t hi = (n - lo < tile_s) ? n : lo + tile_s;
    for (uint32_t b = threadIdx.x; b < Hb; b += blockDim.x) lds_u32[b] = SCATTER ? cstart[b] + blockbase1[(size_t)blockIdx.x * Hb + b] : 0u;
    __syncthreads();
    const uint32_t mask = (1u << c) - 1u, half = 1u << (c - 1);
    for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const uint4 s_lo = scalars[2 * (size_t)i], s_hi = scalars[2 * (size_t)i + 1];
        uint32_t w32[8] = {s_lo.x, s_lo.y, s_lo.z, s_lo.w, s_hi.x, s_hi.y, s_hi.z, s_hi.w};
        uint32_t k[8];
        fe_wire_to_canonical_words<FrParams>(k, w32);
        uint32_t carry = 0, idx = i;
        for (int w = 0; w < W; ++w, idx += table_stride) {
            const uint32_t raw = (k[0] & mask) + carry;
#pragma unroll
            for (int j = 0; j < 7; ++j) k[j] = (k[j] >> c) | (k[j + 1] << (32 - c));
            k[7] >>= c;
            const uint32_t neg = raw > half;
            const uint32_t mag = neg ? (1u << c) - raw : raw;
            carry = neg;
            if (mag == 0) continue;
            const uint32_t key = mag - 1;
            if (SCATTER) {
                const uint32_t pos = atomicAdd(&lds_u32[key >> SORT2_LO_BITS], 1u);
                tmp1[pos] = (neg << 31) | ((key & (SORT2_LO - 1)) << SORT2_IDX_BITS) | idx;
            } else {
                atomicAdd(&lds_u32[key >> SORT2_LO_BITS], 1u);
            }
        }
    }
    if (SCATTER) return;
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < Hb; b += blockDim.x) {
        const uint32_t h = lds_u32[b];
        blockbase1[(size_t)blockIdx.x * Hb + b] = h ? atomicAdd(&ccount[b], h) : 0u;
    }
}
// NAF mode, pass 0: scalars -> digit array (NAF_DIGITS words per scalar: sign << 31 | position << 16 | bucket, NAF_NO_DIGIT in the
// unused slots) + the coarse histogram per tile, as k_sort2_scalars<false> leaves it.  The recoding runs ONCE, here (in the ISA the
// register recoder is 100 instructions per digit, the LDS-window one 45: recoding in all three passes, as the fixed windows do
// with their 12-instruction digits, cost more than the additions the NAF saves).  The digits of 256 scalars are staged in LDS and
// leave as 16-byte stores of consecutive lanes: one store per digit and lane (64 lines per wave instruction) bound the kernel at
// 82 us per 2^20 scalars whatever the recoder.
// ND = words per scalar (16: widths >= 16; 32: the width-8 digits of the batched mode).  poly_len != 0: BATCHED table mode -- the
// scalars are `n / poly_len` polynomials of poly_len coefficients over the same bases and scalar s belongs to polynomial s / poly_len,
// whose 64 buckets (width-8 NAF: keys < 64) are the group s / poly_len of the bucket array: one group of the first reduction level per
// polynomial, whose seven sums k_batch_finish turns into the polynomial's commitment.
// The polynomials of a batched launch may live in separate buffers (a stream of resident scalar sets): p[k] = polynomial k, or all null
// = one contiguous array `scalars`.
constexpr int MSM_BATCH_PTRS = 16;
struct PolyPtrs { const uint4* p[MSM_BATCH_PTRS]; };
template <int ND>
__global__ void __launch_bounds__(256)
k_naf_digits(const uint4* __restrict__ scalars, uint32_t n, int c, uint32_t tile_s, uint32_t Hb, uint32_t* __restrict__ ccount,
             uint32_t* __restrict__ blockbase1, uint4* __restrict__ digs, uint32_t poly_len, PolyPtrs ptrs) {
    latency_bound_kernel();
    extern __shared__ uint32_t lds_u32[];
    uint32_t* hist = lds_u32;                             // Hb
    uint32_t* col = hist + Hb + threadIdx.x;              // 11 x 256: this thread's scalar words (naf_for_digits_lds), words 8..10 zero
    uint32_t* stage = hist + Hb + 11 * 256;               // ND x 256: digit m of thread t at [m * 256 + t]
    const uint32_t t = threadIdx.x;
    const uint32_t lo = blockIdx.x * tile_s;
    const uint32_t hi = (n - lo < tile_s) ? n : lo + tile_s;
    for (uint32_t b = t; b < Hb; b += 256) hist[b] = 0u;
    col[8 * 256] = 0; col[9 * 256] = 0; col[10 * 256] = 0;
    for (uint32_t base = lo; base < hi; base += 256) {
        __syncthreads();                                  // histogram zeroed / the stage of the previous round written out
#pragma unroll
        for (int m = 0; m < ND; ++m) stage[m * 256 + t] = NAF_NO_DIGIT;
        const uint32_t i = base + t;
        if (i < hi) {
            const uint4* sp = scalars + 2 * (size_t)i;
            if (poly_len && ptrs.p[0]) {                  // separate buffers: a select chain over the (few) pointers, no indexed parameter access
                const uint32_t pk = i / poly_len;
                const uint4* base = ptrs.p[0];
#pragma unroll
                for (int q = 1; q < MSM_BATCH_PTRS; ++q) base = pk == (uint32_t)q ? ptrs.p[q] : base;
                sp = base + 2 * (size_t)(i - pk * poly_len);
            }
            const uint4 s_lo = sp[0], s_hi = sp[1];
            uint32_t w32[8] = {s_lo.x, s_lo.y, s_lo.z, s_lo.w, s_hi.x, s_hi.y, s_hi.z, s_hi.w};
            uint32_t k[8];
            fe_wire_to_canonical_words<FrParams>(k, w32);
#pragma unroll
            for (int j = 0; j < 8; ++j) col[j * 256] = k[j];
            uint32_t m = 0;
            const uint32_t group = poly_len ? (i / poly_len) << (c - 1) : 0u;        // batched mode: the polynomial's 2^(c-1) buckets
            naf_for_digits_lds(col, 256, c + 1, [&](uint32_t pos, uint32_t key, uint32_t neg) {
                const uint32_t g = group | (c - 1 > 6 ? naf_bucket(key, c - 1) : key);
                atomicAdd(&hist[g >> SORT2_LO_BITS], 1u);
                if (m < (uint32_t)ND) stage[m * 256 + t] = (neg << 31) | (pos << 16) | g;
                ++m;
            });
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < ND / 4; ++r) {
            const uint32_t idx = t + 256 * r, sc = idx / (ND / 4), q = idx % (ND / 4);
            if (base + sc < hi)
                digs[(size_t)(base + sc) * (ND / 4) + q] = make_uint4(stage[(4 * q) * 256 + sc], stage[(4 * q + 1) * 256 + sc],
                                                                    stage[(4 * q + 2) * 256 + sc], stage[(4 * q + 3) * 256 + sc]);
        }
    }
    __syncthreads();
    for (uint32_t b = t; b < Hb; b += 256) {
        const uint32_t h = hist[b];
        blockbase1[(size_t)blockIdx.x * Hb + b] = h ? atomicAdd(&ccount[b], h) : 0u;
    }
}
// Pass-1 scatter through LDS: the tile is walked in chunks of SORT2_P1_THREADS scalars (one per thread); a chunk's entries are
// counted per bin, placed bin by bin in an LDS buffer and written out from there, so that the lanes of a wave store runs of
// consecutive words (~15 entries per bin and chunk) instead of 64 single words to 64 different lines.  The digits are computed
// twice (count, place) rather than kept: W values per thread would not stay in registers.
// LDS entry: bin << 22 | low key << 15 | sign << 14 | window << 9 | scalar index inside the chunk.
constexpr int SORT2_P1_THREADS = 512;
static_assert(SORT2_MAX_BINS <= 512 && SORT2_LO_BITS == 7, "LDS entry layout of k_sort2_scatter1_lds");
template <class F>
__device__ __forceinline__ void sort2_for_digits(uint32_t k[8], int c, int W, F&& f) {
    const uint32_t mask = (1u << c) - 1u, half = 1u << (c - 1);
    uint32_t carry = 0;
    for (int w = 0; w < W; ++w) {
        const uint32_t raw = (k[0] & mask) + carry;
#pragma unroll
        for (int j = 0; j < 7; ++j) k[j] = (k[j] >> c) | (k[j + 1] << (32 - c));
        k[7] >>= c;
        const uint32_t neg = raw > half;
        const uint32_t mag = neg ? (1u << c) - raw : raw;
        carry = neg;
        if (mag != 0) f((uint32_t)w, mag - 1, neg);
    }
}
// NAF = true: W = most entries per scalar; LDS entry = sign << 31 | position << 9 | scalar index inside the chunk, the 16-bit key in a
// second LDS array behind it; tmp1 gets sign << 31 | position * table_stride + i, tmpk the low 7 key bits.
template <bool NAF = false, int ND = NAF_DIGITS>
__global__ void __launch_bounds__(SORT2_P1_THREADS)
k_sort2_scatter1_lds(const uint4* __restrict__ scalars, uint32_t n, int c, int W, uint32_t tile_s, uint32_t Hb, const uint32_t* __restrict__ blockbase1,
                     const uint32_t* __restrict__ cstart, uint32_t table_stride, uint32_t* __restrict__ tmp1, uint8_t* __restrict__ tmpk = nullptr,
                     uint32_t poly_len = 0) {
    latency_bound_kernel();
    extern __shared__ uint32_t lds_u32[];
    uint32_t* gpos = lds_u32;                 // Hb: next global position of the tile in every bin
    uint32_t* hist = gpos + Hb;               // Hb: entries of the chunk per bin, then the placement cursor
    uint32_t* lstart = hist + Hb;             // Hb: first buffer slot of every bin
    uint32_t* buf = lstart + Hb;              // SORT2_P1_THREADS * W
    uint16_t* bufk = reinterpret_cast<uint16_t*>(buf + (size_t)SORT2_P1_THREADS * W);   // NAF: SORT2_P1_THREADS * W keys
    const uint32_t t = threadIdx.x, lane = t & 63;
    const uint32_t lo = blockIdx.x * tile_s;
    const uint32_t hi = (n - lo < tile_s) ? n : lo + tile_s;
    for (uint32_t b = t; b < Hb; b += SORT2_P1_THREADS) gpos[b] = cstart[b] + blockbase1[(size_t)blockIdx.x * Hb + b];
    const uint32_t per_lane = (Hb + 63) / 64;           // bins per lane of wave 0 in the scan (<= 8)
    for (uint32_t base = lo; base < hi; base += SORT2_P1_THREADS) {
        __syncthreads();                                 // gpos ready / updated, buf and hist free
        for (uint32_t b = t; b < Hb; b += SORT2_P1_THREADS) hist[b] = 0;
        __syncthreads();
        const uint32_t i = base + t;
        const bool have = i < hi;
        uint32_t kk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        uint32_t dg[NAF ? ND : 1];                         // NAF: the scalar's digits as k_naf_digits left them
        if (NAF) {
#pragma unroll
            for (int m = 0; m < (NAF ? ND : 1); ++m) dg[m] = NAF_NO_DIGIT;
            if (have) {
                const uint4* src = reinterpret_cast<const uint4*>(scalars) + (size_t)i * (ND / 4);
#pragma unroll
                for (int q = 0; q < (NAF ? ND / 4 : 0); ++q) {
                    const uint4 v = src[q];
                    dg[4 * q] = v.x; dg[4 * q + 1] = v.y; dg[4 * q + 2] = v.z; dg[4 * q + 3] = v.w;
                }
                bool live = true;                           // words behind the first NAF_NO_DIGIT are stale
#pragma unroll
                for (int m = 0; m < (NAF ? ND : 0); ++m) {
                    live = live && dg[m] != NAF_NO_DIGIT;
                    if (!live) dg[m] = NAF_NO_DIGIT;
                    else atomicAdd(&hist[(dg[m] & 0xFFFFu) >> SORT2_LO_BITS], 1u);
                }
            }
        } else
        if (have) {
            const uint4 s_lo = scalars[2 * (size_t)i], s_hi = scalars[2 * (size_t)i + 1];
            const uint32_t w32[8] = {s_lo.x, s_lo.y, s_lo.z, s_lo.w, s_hi.x, s_hi.y, s_hi.z, s_hi.w};
            fe_wire_to_canonical_words<FrParams>(kk, w32);
            uint32_t k[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) k[j] = kk[j];
            sort2_for_digits(k, c, W, [&](uint32_t, uint32_t key, uint32_t) { atomicAdd(&hist[key >> SORT2_LO_BITS], 1u); });
        }
        __syncthreads();
        if (t < 64) {                                    // exclusive scan of the Hb counts: per_lane consecutive bins per lane
            uint32_t sum = 0;
            for (uint32_t q = 0; q < per_lane; ++q) { const uint32_t b = lane * per_lane + q; if (b < Hb) sum += hist[b]; }
            uint32_t incl = sum;
            for (int d = 1; d < 64; d <<= 1) { const uint32_t x = __shfl_up(incl, d, 64); if ((int)lane >= d) incl += x; }
            uint32_t run = incl - sum;
            for (uint32_t q = 0; q < per_lane; ++q) {
                const uint32_t b = lane * per_lane + q;
                if (b < Hb) { lstart[b] = run; run += hist[b]; }
            }
        }
        __syncthreads();
        const uint32_t cn = lstart[Hb - 1] + hist[Hb - 1];   // entries of this chunk
        __syncthreads();                                 // (everyone has read hist before it becomes the cursor)
        for (uint32_t b = t; b < Hb; b += SORT2_P1_THREADS) hist[b] = 0;
        __syncthreads();
        if (have) {
            if (NAF) {
#pragma unroll
                for (int m = 0; m < (NAF ? ND : 0); ++m) {
                    if (dg[m] == NAF_NO_DIGIT) continue;
                    const uint32_t key = dg[m] & 0xFFFFu;
                    const uint32_t bin = key >> SORT2_LO_BITS;
                    const uint32_t slot = lstart[bin] + atomicAdd(&hist[bin], 1u);
                    buf[slot] = (dg[m] & 0x80000000u) | (((dg[m] >> 16) & 255u) << 9) | t;
                    bufk[slot] = (uint16_t)key;
                }
            } else {
                sort2_for_digits(kk, c, W, [&](uint32_t w, uint32_t key, uint32_t neg) {
                    const uint32_t bin = key >> SORT2_LO_BITS;
                    const uint32_t slot = lstart[bin] + atomicAdd(&hist[bin], 1u);
                    buf[slot] = (bin << 22) | ((key & (SORT2_LO - 1)) << 15) | (neg << 14) | (w << 9) | t;
                });
            }
        }
        __syncthreads();
        if (NAF) {
            for (uint32_t q = t; q < cn; q += SORT2_P1_THREADS) {
                const uint32_t x = buf[q], key = bufk[q], bin = key >> SORT2_LO_BITS;
                const uint32_t sidx = base + (x & 511u);                                // scalar index; batched mode: base point = index inside its polynomial
                const uint32_t idx = ((x >> 9) & 255u) * table_stride + (poly_len ? sidx % poly_len : sidx);
                const uint32_t dst = gpos[bin] + (q - lstart[bin]);
                tmp1[dst] = (x & 0x80000000u) | idx;
                tmpk[dst] = (uint8_t)(key & (SORT2_LO - 1));
            }
        } else
        for (uint32_t q = t; q < cn; q += SORT2_P1_THREADS) {
            const uint32_t x = buf[q], bin = x >> 22;
            const uint32_t idx = ((x >> 9) & 31u) * table_stride + base + (x & 511u);
            tmp1[gpos[bin] + (q - lstart[bin])] = ((x >> 14) & 1u) << 31 | ((x >> 15) & (SORT2_LO - 1)) << SORT2_IDX_BITS | idx;
        }
        __syncthreads();
        for (uint32_t b = t; b < Hb; b += SORT2_P1_THREADS) gpos[b] += hist[b];
    }
}
// single block: cstart[h] = exclusive scan of the coarse counts.  Bins of more than SORT2_BIN_CAP entries (skewed scalars, the
// short top window) are LARGE: they are cut into tiles of SORT2_CHUNK entries for the tiled pass-2 kernels (tstart = exclusive
// scan of their tile counts, tile_bin = tile -> bin), and their fine counters are zeroed here; all other bins are sorted by one
// workgroup each in k_sort2_bin.
// The cap follows the load: a bin is LARGE above 5/4 of the average bin + 2 048 entries (at most SORT2_BIN_CAP): k_sort2_bin gives
// every other bin to ONE workgroup, so a bin of twice the average is a tail of twice the kernel's time (NAF mode: the bins that hold
// a heavy small-key bucket).  *cap_out tells k_sort2_bin.
constexpr uint32_t SORT2_BIN_CAP = 65536;
__global__ void __launch_bounds__(512)
k_sort2_scan(const uint32_t* __restrict__ ccount, uint32_t Hb, uint32_t* __restrict__ cstart, uint32_t* __restrict__ tstart,
             uint32_t* __restrict__ tile_bin, uint32_t* __restrict__ count, uint32_t* __restrict__ cap_out) {
    latency_bound_kernel();
    __shared__ uint32_t a[512], b[512];
    __shared__ uint32_t total;
    const uint32_t t = threadIdx.x;
    uint32_t c = t < Hb ? ccount[t] : 0u;
    if (t == 0) total = 0;
    __syncthreads();
    if (c) atomicAdd(&total, c);
    __syncthreads();
    const uint32_t cap_dyn = total / Hb + total / Hb / 4 + 2048u;
    const uint32_t cap = cap_dyn < SORT2_BIN_CAP ? cap_dyn : SORT2_BIN_CAP;
    if (t == 0) *cap_out = cap;
    uint32_t k = c > cap ? (c + SORT2_CHUNK - 1) / SORT2_CHUNK : 0u;
    a[t] = c; b[t] = k;
    __syncthreads();
    for (uint32_t d = 1; d < 512; d <<= 1) {
        uint32_t x = t >= d ? a[t - d] : 0u, y = t >= d ? b[t - d] : 0u;
        __syncthreads();
        a[t] += x; b[t] += y;
        __syncthreads();
    }
    if (t < Hb) {
        cstart[t] = a[t] - c; tstart[t] = b[t] - k;
        for (uint32_t q = b[t] - k; q < b[t]; ++q) tile_bin[q] = t;
        if (k) for (uint32_t q = 0; q < SORT2_LO; ++q) count[(size_t)t * SORT2_LO + q] = 0u;
    }
    if (t == Hb - 1) { cstart[Hb] = a[t]; tstart[Hb] = b[t]; }
}
// low key bits of the entry at position e between the passes (NAF: a byte array of its own)
template <bool NAF>
__device__ __forceinline__ uint32_t sort2_low_key(const uint32_t* __restrict__ tmp1, const uint8_t* __restrict__ tmpk, uint32_t e) {
    return NAF ? (uint32_t)tmpk[e] : (tmp1[e] >> SORT2_IDX_BITS) & (SORT2_LO - 1);
}
template <bool NAF = false>
__global__ void __launch_bounds__(256)
k_sort2_hist2(const uint32_t* __restrict__ tmp1, const uint32_t* __restrict__ cstart, const uint32_t* __restrict__ tstart,
              const uint32_t* __restrict__ tile_bin, uint32_t Hb, uint32_t* __restrict__ count, uint32_t* __restrict__ blockbase2,
              const uint8_t* __restrict__ tmpk = nullptr) {
    latency_bound_kernel();
    __shared__ uint32_t hist[SORT2_LO];
    const uint32_t tile = blockIdx.x;
    if (tile >= tstart[Hb]) return;
    const uint32_t h = tile_bin[tile];
    const uint32_t lo = cstart[h] + (tile - tstart[h]) * SORT2_CHUNK;
    const uint32_t hi = (cstart[h + 1] - lo < SORT2_CHUNK) ? cstart[h + 1] : lo + SORT2_CHUNK;
    if (threadIdx.x < SORT2_LO) hist[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t e = lo + threadIdx.x; e < hi; e += blockDim.x) atomicAdd(&hist[sort2_low_key<NAF>(tmp1, tmpk, e)], 1u);
    __syncthreads();
    if (threadIdx.x < SORT2_LO) {
        uint32_t c = hist[threadIdx.x];
        blockbase2[(size_t)tile * SORT2_LO + threadIdx.x] = c ? atomicAdd(&count[(size_t)h * SORT2_LO + threadIdx.x], c) : 0u;
    }
}
template <bool NAF = false>
__global__ void __launch_bounds__(256)
k_sort2_scatter2(const uint32_t* __restrict__ tmp1, const uint32_t* __restrict__ cstart, const uint32_t* __restrict__ tstart,
                 const uint32_t* __restrict__ tile_bin, uint32_t Hb, const uint32_t* __restrict__ offs,
                 const uint32_t* __restrict__ blockbase2, uint32_t* __restrict__ sorted, const uint8_t* __restrict__ tmpk = nullptr) {
    latency_bound_kernel();
    __shared__ uint32_t cur[SORT2_LO];
    const uint32_t tile = blockIdx.x;
    if (tile >= tstart[Hb]) return;
    const uint32_t h = tile_bin[tile];
    const uint32_t lo = cstart[h] + (tile - tstart[h]) * SORT2_CHUNK;
    const uint32_t hi = (cstart[h + 1] - lo < SORT2_CHUNK) ? cstart[h + 1] : lo + SORT2_CHUNK;
    if (threadIdx.x < SORT2_LO)
        cur[threadIdx.x] = offs[(size_t)h * SORT2_LO + threadIdx.x] + blockbase2[(size_t)tile * SORT2_LO + threadIdx.x];
    __syncthreads();
    for (uint32_t e = lo + threadIdx.x; e < hi; e += blockDim.x) {
        uint32_t v = tmp1[e];
        uint32_t pos = atomicAdd(&cur[sort2_low_key<NAF>(tmp1, tmpk, e)], 1u);
        sorted[pos] = NAF ? v : v & (0x80000000u | SORT2_IDX_MASK);
    }
}

// Pass 2 for every bin of at most SORT2_BIN_CAP entries, ONE workgroup per bin: fine histogram (LDS), its scan, the bucket
// offsets offs[h * 128 + k] (bins are contiguous in tmp1, so no global scan is needed) and the scatter into `sorted`.  It replaced
// three launches (tiled histogram, global scan of all bucket counts, tiled scatter).  For a LARGE bin the workgroup only turns
// the fine counts that k_sort2_hist2 gathered into offsets; k_sort2_scatter2 then moves its entries.
// The scatter goes through LDS: a chunk of SORT2_BIN_CHUNK entries is counting-sorted inside the workgroup first, so that a wave
// writes runs of consecutive positions (lane-per-entry stores of 4 bytes to 64 different lines ran at 2 TB/s of requests).
constexpr int SORT2_BIN_THREADS = 1024;
constexpr int SORT2_BIN_PER = 8;                                        // entries per thread and chunk
constexpr uint32_t SORT2_BIN_CHUNK = SORT2_BIN_THREADS * SORT2_BIN_PER;
template <bool NAF = false>
__global__ void __launch_bounds__(SORT2_BIN_THREADS)
k_sort2_bin(const uint32_t* __restrict__ tmp1, const uint32_t* __restrict__ cstart, uint32_t Hb, const uint32_t* __restrict__ count,
            uint32_t* __restrict__ offs, uint32_t* __restrict__ sorted, const uint32_t* __restrict__ cap_ptr, const uint8_t* __restrict__ tmpk = nullptr) {
    latency_bound_kernel();
    __shared__ uint32_t hist[SORT2_LO], gpos[SORT2_LO], lstart[SORT2_LO], buf[SORT2_BIN_CHUNK];
    __shared__ uint8_t bufk[NAF ? SORT2_BIN_CHUNK : 4];
    const uint32_t h = blockIdx.x, t = threadIdx.x;
    const uint32_t lo = cstart[h], hi = cstart[h + 1];
    const bool large = hi - lo > *cap_ptr;
    if (t < SORT2_LO) hist[t] = large ? count[(size_t)h * SORT2_LO + t] : 0u;
    __syncthreads();
    if (!large) {
        uint32_t e = lo + t;
        for (; e + 3 * SORT2_BIN_THREADS < hi; e += 4 * SORT2_BIN_THREADS) {
            const uint32_t k0 = sort2_low_key<NAF>(tmp1, tmpk, e), k1 = sort2_low_key<NAF>(tmp1, tmpk, e + SORT2_BIN_THREADS),
                           k2 = sort2_low_key<NAF>(tmp1, tmpk, e + 2 * SORT2_BIN_THREADS), k3 = sort2_low_key<NAF>(tmp1, tmpk, e + 3 * SORT2_BIN_THREADS);
            atomicAdd(&hist[k0], 1u);
            atomicAdd(&hist[k1], 1u);
            atomicAdd(&hist[k2], 1u);
            atomicAdd(&hist[k3], 1u);
        }
        for (; e < hi; e += SORT2_BIN_THREADS) atomicAdd(&hist[sort2_low_key<NAF>(tmp1, tmpk, e)], 1u);
        __syncthreads();
    }
    // exclusive scan of the SORT2_LO fine counts by wave 0 (two counters per lane)
    static_assert(SORT2_LO == 128, "two counters per lane of one wave");
    if (t < 64) {
        const uint32_t c0 = hist[2 * t], c1 = hist[2 * t + 1];
        uint32_t incl = c0 + c1;
        for (int d = 1; d < 64; d <<= 1) { const uint32_t x = __shfl_up(incl, d, 64); if ((int)t >= d) incl += x; }
        const uint32_t start = lo + incl - (c0 + c1);
        gpos[2 * t] = start; gpos[2 * t + 1] = start + c0;
        offs[(size_t)h * SORT2_LO + 2 * t] = start;
        offs[(size_t)h * SORT2_LO + 2 * t + 1] = start + c0;
    }
    if (h == Hb - 1 && t == 0) offs[(size_t)Hb * SORT2_LO] = cstart[Hb];
    if (large) return;
    for (uint32_t base = lo; base < hi; base += SORT2_BIN_CHUNK) {
        const uint32_t cn = hi - base < SORT2_BIN_CHUNK ? hi - base : SORT2_BIN_CHUNK;
        __syncthreads();                                           // (gpos of the previous chunk updated, buf free)
        if (t < SORT2_LO) hist[t] = 0;
        __syncthreads();
        uint32_t v[SORT2_BIN_PER], r[SORT2_BIN_PER], kq[SORT2_BIN_PER];
#pragma unroll
        for (int j = 0; j < SORT2_BIN_PER; ++j) {
            const uint32_t i = j * SORT2_BIN_THREADS + t;
            v[j] = i < cn ? tmp1[base + i] : 0u;
            kq[j] = NAF ? (i < cn ? (uint32_t)tmpk[base + i] : 0u) : (v[j] >> SORT2_IDX_BITS) & (SORT2_LO - 1);
        }
#pragma unroll
        for (int j = 0; j < SORT2_BIN_PER; ++j)
            if (j * SORT2_BIN_THREADS + t < cn) r[j] = atomicAdd(&hist[kq[j]], 1u);
        __syncthreads();
        if (t < 64) {
            const uint32_t c0 = hist[2 * t], c1 = hist[2 * t + 1];
            uint32_t incl = c0 + c1;
            for (int d = 1; d < 64; d <<= 1) { const uint32_t x = __shfl_up(incl, d, 64); if ((int)t >= d) incl += x; }
            lstart[2 * t] = incl - (c0 + c1); lstart[2 * t + 1] = incl - c1;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < SORT2_BIN_PER; ++j)
            if (j * SORT2_BIN_THREADS + t < cn) {
                const uint32_t slot = lstart[kq[j]] + r[j];
                buf[slot] = v[j];
                if (NAF) bufk[slot] = (uint8_t)kq[j];
            }
        __syncthreads();
        for (uint32_t i = t; i < cn; i += SORT2_BIN_THREADS) {
            const uint32_t x = buf[i], k = NAF ? (uint32_t)bufk[i] : (x >> SORT2_IDX_BITS) & (SORT2_LO - 1);
            sorted[gpos[k] + (i - lstart[k])] = NAF ? x : x & (0x80000000u | SORT2_IDX_MASK);
        }
        __syncthreads();
        if (t < SORT2_LO) gpos[t] += hist[t];
    }
}

// -------------------------------------------------------------------------------------------------
// 4. bucket accumulation: EQUAL SPLIT of the sorted entries over the lanes
// -------------------------------------------------------------------------------------------------
// The sorted array holds E entries grouped by bucket (offs[]).  Lane t of the nl launched lanes adds the entries
// [t L, (t+1) L), L = ceil(E / nl), whatever buckets they belong to: every lane of the chip runs the same trip count, and the
// host sizes nl to the number of resident wave slots (3 per SIMD), so there is exactly one full round of waves.
// (Round 1 cut every bucket into round(count / L) segments: at 2^20 pairs all buckets hold ~510 entries, so the segment count
// jumped between 5 and 6 per bucket, i.e. 2 560 waves of 102 additions or 3 072 + a second round: 1.36 ms where the equal
// split needs 86 additions on 3 072 waves.)
// A lane's partial sums go to
//   head[g]  the part of bucket g that starts inside the lane's range (every non-empty bucket has exactly one)
//   cont[t]  the part of the bucket that was already open at the lane's first entry (at most one per lane)
// so bucket g = head[g] + sum of cont[t], t in (t1, t2], t1 = offs[g] / L, t2 = (offs[g+1] - 1) / L.
// Runs of more than RUN_SERIAL continuation lanes (heavy buckets: skewed or repeated scalars, the short top window) are folded
// inside each wave by a segmented suffix scan over ds_bpermute shuffles first, so that only the first lane of the run and the
// lanes 0 of the following waves hold a partial: bucket_partial() below enumerates them.
#ifndef KZG_RUN_SERIAL
#define KZG_RUN_SERIAL 8
#endif
constexpr uint32_t RUN_SERIAL = KZG_RUN_SERIAL;        // continuation lanes a bucket's owner adds one by one
constexpr uint32_t NP_SERIAL = 10;        // partials (head included) above which a wave sums a bucket cooperatively

__device__ __forceinline__ void xyzz_shfl_down(Xyzz& r, const Xyzz& v, int d) {
    const Fq* s[4] = {&v.x, &v.y, &v.zz, &v.zzz};
    Fq* t[4] = {&r.x, &r.y, &r.zz, &r.zzz};
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < NL; ++j) t[q]->l[j] = __shfl_down(s[q]->l[j], d, 64);
    r.inf = __shfl_down((int)v.inf, d, 64) != 0;
}
__device__ __forceinline__ uint32_t acc_seg_len(uint32_t E, uint32_t nl) { return (E + nl - 1) / nl; }

// experiment switch: -DKZG_ACC_NOGATHER makes every lane read the same few table points (no HBM gather) to time the arithmetic alone
#ifdef KZG_ACC_NOGATHER
#define KZG_ACC_IDX(v) ((v) & 0xFFu)
#else
#define KZG_ACC_IDX(v) ((v) & 0x7FFFFFFFu)
#endif
// A sorted entry holds window * 2^idx_log + i (COMPACT index: SRS of more than 2^20 points, whose tables are table_stride >
// 2^idx_log points apart and would not fit the 24 index bits of the two-level sort): the point is at entry + window * stride_adj.
// Plain indices come with idx_log = 31 (window part = 0).
__device__ __forceinline__ size_t acc_point_index(uint32_t ix, uint32_t idx_log, uint32_t stride_adj) {
    return (size_t)ix + (size_t)(ix >> idx_log) * stride_adj;
}
#ifndef KZG_ACC_WAVES
#define KZG_ACC_WAVES 3        // waves per SIMD (<= 168 VGPRs); 4 (128 VGPRs, more spills) measured equal in round 1
#endif
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(KZG_ACC_WAVES, KZG_ACC_WAVES)))
k_msm_accumulate(const uint4* __restrict__ points, const uint32_t* __restrict__ sorted, const uint32_t* __restrict__ offs, uint32_t G,
                 int32_t* __restrict__ head, size_t head_stride, int32_t* __restrict__ cont, size_t cont_stride,
                 uint32_t idx_log, uint32_t stride_adj) {
    const uint32_t nl = gridDim.x * blockDim.x;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63;
#ifdef KZG_ACC_STAMPS     // diagnostic build only (tools/acc_stamps.py): when and where does every wave run
    unsigned long long* stamps = reinterpret_cast<unsigned long long*>(cont + cont_stride * 36) + 8 * (size_t)(t >> 6);
    const unsigned long long stamp0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long flush_ticks = 0, flush_events = 0, stamp_loop = 0;
#endif
    const uint32_t E = offs[G];
    const uint32_t L = acc_seg_len(E, nl);
    const unsigned long long b64 = (unsigned long long)t * L;
    const uint32_t begin = b64 < E ? (uint32_t)b64 : E;
    const uint32_t end = (b64 + L < E) ? (uint32_t)(b64 + L) : E;
    const bool active = begin < end;
    uint32_t g = 0, next = 0, g0 = 0xFFFFFFFFu;
    bool is_cont = false, long_run = false;
    if (active) {
        uint32_t lo = 0, hi = G;                       // invariant: offs[lo] <= begin < offs[hi]
#ifdef KZG_ACC_BINARY_SEARCH                           // (A/B: one dependent load per step, 16 steps at 2^16 buckets)
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (offs[mid] <= begin) lo = mid; else hi = mid;
        }
#else
        // 16-ary: fifteen INDEPENDENT probes per round, four dependent rounds instead of sixteen at 2^16 buckets (offs is L2 resident and
        // neighbouring lanes probe the same words); the prologue of a lane is ~10 us of dependent loads otherwise -- nothing at 2^20 pairs
        // (73 entries per lane), a tenth of the kernel at 2^12 .. 2^16
        while (hi - lo > 1) {
            const uint32_t s = (hi - lo + 15) >> 4;
            uint32_t cnt = 0;
#pragma unroll
            for (uint32_t j = 1; j < 16; ++j) {
                const uint32_t p = lo + j * s;
                const uint32_t v = p < hi ? offs[p] : 0xFFFFFFFFu;
                cnt += v <= begin ? 1u : 0u;       // offs is non-decreasing: the probes that are <= begin form a prefix
            }
            const uint32_t nlo = lo + cnt * s, nhi = lo + (cnt + 1) * s;
            if (cnt < 15 && nhi < hi) hi = nhi;
            lo = nlo;
        }
#endif
        g = lo;
        const uint32_t o0 = offs[g];
        next = offs[g + 1];
        is_cont = o0 < begin;
        if (is_cont) { long_run = ((next - 1) / L - o0 / L) > RUN_SERIAL; g0 = g; }
    }
    // Partials completed INSIDE the loop (a bucket ends before the lane's range does) are parked in LDS, one slot per lane, and
    // written to global memory after the loop: 36 global stores in the loop made the next iteration's load wait (vmcnt counts
    // in order) wait for their write acknowledgements as well -- 8.3 us per boundary, 15 % of a wave's lifetime at 2^20 pairs
    // (tools/acc_stamps.py).  A lane that crosses a second boundary (buckets shorter than its trip count) stores directly.
    __shared__ int32_t park[4 * NL * 256];
    const uint32_t tl = threadIdx.x;
    bool parked = false, parked_cont = false;          // slot in use; it holds the continuation partial (else head[park_g])
    uint32_t park_g = 0;
    bool cont_in_regs = false;                         // long-run continuation partial still in `acc` after the loop (scan input)
    Xyzz acc;
    xyzz_set_inf(acc);
    uint32_t next2 = 0;                                // offs[g + 2], loaded ahead: no dependent load at a boundary
    if (active) next2 = offs[g + 2 <= G ? g + 2 : G];
    // the partial of bucket g is complete for this lane: `more` = the lane goes on into the next bucket
    auto flush = [&](bool more) {
        const bool as_cont = is_cont && g == g0;
        if (more && !parked) {
            const Fq* c4[4] = {&acc.x, &acc.y, &acc.zz, &acc.zzz};
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < NL; ++j) park[(q * NL + j) * 256 + tl] = acc.inf ? 0 : c4[q]->l[j];
            parked = true; parked_cont = as_cont; park_g = g;
        } else if (!as_cont) {
            xyzz_store(head, head_stride, g, acc);
        } else if (!long_run || more) {
            xyzz_store(cont, cont_stride, t, acc);
        } else {
            cont_in_regs = true;                       // kept in registers for the segmented scan below
        }
    };
#ifdef KZG_ACC_STAMPS
    stamp_loop = __builtin_amdgcn_s_memrealtime();         // end of the prologue (bucket search, first offsets)
#endif
    if (active) {
        // two-deep software pipeline: index e+2 and point e+1 are in flight while entry e is added, so neither the index
        // load (dependent address) nor the 64-byte gather from the 1 GiB table is waited for inside an iteration.  The
        // prefetches are UNCONDITIONAL (indices clamped to the lane's last entry): with `if (e + 1 < end)` around them the
        // register allocator copied the freshly loaded registers right behind the loads, i.e. an s_waitcnt vmcnt(1) directly
        // after issue.
        const uint32_t last = end - 1;
        uint32_t v = sorted[begin];
        uint32_t v1 = sorted[begin + 1 < end ? begin + 1 : last];
        const uint4* src = points + 4 * acc_point_index(KZG_ACC_IDX(v), idx_log, stride_adj);
        uint4 q0 = src[0], q1 = src[1], q2 = src[2], q3 = src[3];
        for (uint32_t e = begin; e < end; ++e) {
#ifdef KZG_ACC_STAMPS
            const bool any_cross = __any(e == next);
            unsigned long long f0 = 0;
            if (any_cross) f0 = __builtin_amdgcn_s_memrealtime();
#endif
            if (__builtin_expect(e == next, 0)) {      // bucket boundary inside the lane's range
                flush(true);
                xyzz_set_inf(acc);
                ++g; next = next2;
                while (next == e) { ++g; next = offs[g + 1]; }         // empty buckets; e < end <= E = offs[G]: terminates
                next2 = offs[g + 2 <= G ? g + 2 : G];
            }
#ifdef KZG_ACC_STAMPS
            if (any_cross) { flush_ticks += __builtin_amdgcn_s_memrealtime() - f0; flush_events += 1; }
#endif
            const uint32_t neg = v >> 31;
            uint32_t wx[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
            uint32_t wy[8] = {q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
            const uint32_t any = q0.x | q0.y | q0.z | q0.w | q1.x | q1.y | q1.z | q1.w | q2.x | q2.y | q2.z | q2.w | q3.x | q3.y | q3.z | q3.w;
            Affine p;
            fe_unpack(p.x, wx);
            fe_unpack(p.y, wy);
            v = v1;                                    // entry e + 1 (the last iteration re-reads entry `last`: unused)
            src = points + 4 * acc_point_index(KZG_ACC_IDX(v), idx_log, stride_adj);
            q0 = src[0]; q1 = src[1]; q2 = src[2]; q3 = src[3];
            v1 = sorted[e + 2 < end ? e + 2 : last];
            if (any == 0) continue;                                                       // identity base
            xyzz_madd<true>(acc, p, neg);
        }
        flush(false);
    }
    const bool in_scan = active && is_cont && long_run;
    Xyzz c;                                            // scan input
    xyzz_set_inf(c);
    if (cont_in_regs) c = acc;
    if (parked) {                                      // parked partial -> its place (or into the scan)
        Xyzz v;
        Fq* c4[4] = {&v.x, &v.y, &v.zz, &v.zzz};
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < NL; ++j) c4[q]->l[j] = park[(q * NL + j) * 256 + tl];
        v.inf = fe_is_literal_zero(v.zz);
        if (!parked_cont) xyzz_store(head, head_stride, park_g, v);
        else if (!long_run) xyzz_store(cont, cont_stride, t, v);
        else c = v;
    }
#ifdef KZG_ACC_STAMPS
    if (lane == 0) {
        unsigned int hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        stamps[0] = stamp0; stamps[1] = __builtin_amdgcn_s_memrealtime(); stamps[2] = hwid; stamps[3] = xcc;
        stamps[4] = flush_ticks; stamps[5] = flush_events; stamps[6] = stamp_loop;
    }
#endif
    // segmented suffix scan over the continuation partials of long runs: lane = sum of the partials of lanes lane .. end of its
    // run in this wave; the first lane of each run stores
    if (!__any(in_scan)) return;                       // wave-uniform: the normal case (no heavy bucket in this wave)
    const uint32_t key = in_scan ? g0 : 0xFFFFFFFFu;
#pragma unroll 1
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t kd = __shfl_down(key, d, 64);
        const bool join = in_scan && (lane + d < 64) && kd == key;
        if (!__any(join)) continue;                    // wave-uniform: nobody has a partner at this distance
        Xyzz u;
        xyzz_shfl_down(u, c, d);
        if (join) {
            Xyzz r;
            xyzz_add<true>(r, c, u);
            c = r;
        }
    }
    const uint32_t kprev = __shfl_up(key, 1, 64);
    if (in_scan && (lane == 0 || kprev != key)) xyzz_store(cont, cont_stride, t, c);
}

// Bucket g of the G = W * B buckets is stored at a transposed position so that the reduction kernels,
// where lane t walks chunk t (buckets t*m .. t*m+m-1), read consecutive addresses across lanes.
__device__ __forceinline__ size_t bucket_pos(uint32_t g, uint32_t m, uint32_t n_chunks) {
    return (size_t)(g % m) * n_chunks + (g / m);
}
// the partials of one bucket, as the accumulate kernel leaves them
struct BucketSpan {
    uint32_t g, t1, np;      // np = number of partials, head included (0 for an empty bucket)
    bool long_run;
};
__device__ __forceinline__ BucketSpan bucket_span(const uint32_t* __restrict__ offs, uint32_t g, uint32_t L) {
    BucketSpan s;
    s.g = g;
    const uint32_t o0 = offs[g], o1 = offs[g + 1];
    s.t1 = 0; s.np = 0; s.long_run = false;
    if (o1 == o0) return s;
    s.t1 = o0 / L;
    const uint32_t t2 = (o1 - 1) / L, r = t2 - s.t1;
    s.long_run = r > RUN_SERIAL;
    // long run: the first continuation lane t1 + 1 and the lanes 0 of the later waves of (t1 + 1, t2]
    s.np = 1 + (s.long_run ? 1 + (t2 / 64 - (s.t1 + 1) / 64) : r);
    return s;
}
__device__ __forceinline__ void bucket_partial(Xyzz& v, const BucketSpan& s, uint32_t k, const int32_t* __restrict__ head, size_t head_stride,
                                               const int32_t* __restrict__ cont, size_t cont_stride) {
    if (k == 0) { xyzz_load(v, head, head_stride, s.g); return; }
    const uint32_t first = s.t1 + 1;
    const uint32_t t = !s.long_run ? s.t1 + k : (k == 1 ? first : (first / 64 + (k - 1)) * 64);
    xyzz_load(v, cont, cont_stride, t);
}
// Sum of all partials of this lane's bucket.  Buckets with up to NP_SERIAL partials are summed by their lane; heavier ones
// (skewed scalars: few distinct digits) one after the other by the whole wave (lanes take partials round-robin, then a
// shuffle tree).  Every lane of the wave must call this.
__device__ __forceinline__ void bucket_sum_wave(Xyzz& acc, const BucketSpan& s, uint32_t lane, const int32_t* __restrict__ head, size_t head_stride,
                                                const int32_t* __restrict__ cont, size_t cont_stride) {
    xyzz_set_inf(acc);
    const bool heavy = s.np > NP_SERIAL;
    if (!heavy) {
#pragma unroll 1
        for (uint32_t k = 0; k < s.np; ++k) {
            Xyzz v, r;
            bucket_partial(v, s, k, head, head_stride, cont, cont_stride);
            xyzz_add<true>(r, acc, v);
            acc = r;
        }
    }
    unsigned long long todo = __ballot(heavy);
    while (todo) {                                     // wave-uniform
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        BucketSpan h;
        h.g = __shfl(s.g, src, 64);
        h.t1 = __shfl(s.t1, src, 64);
        h.np = __shfl(s.np, src, 64);
        h.long_run = __shfl((int)s.long_run, src, 64) != 0;
        Xyzz part;
        xyzz_set_inf(part);
#pragma unroll 1
        for (uint32_t k = lane; k < h.np; k += 64) {
            Xyzz v, r;
            bucket_partial(v, h, k, head, head_stride, cont, cont_stride);
            xyzz_add<true>(r, part, v);
            part = r;
        }
#pragma unroll 1
        for (int d = 32; d >= 1; d >>= 1) {
            Xyzz u, r;
            xyzz_shfl_down(u, part, d);
            xyzz_add<true>(r, part, u);
            part = r;
        }
        Xyzz tot;                                      // lane 0 holds the sum
        const Fq* sp[4] = {&part.x, &part.y, &part.zz, &part.zzz};
        Fq* tp[4] = {&tot.x, &tot.y, &tot.zz, &tot.zzz};
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < NL; ++j) tp[q]->l[j] = __shfl(sp[q]->l[j], 0, 64);
        tot.inf = __shfl((int)part.inf, 0, 64) != 0;
        if ((int)lane == src) acc = tot;
    }
}

// generic mode: one lane per bucket -> bucket[] (transposed for the chunked reduction when m > 0)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
k_msm_bucket_fin(const uint32_t* __restrict__ offs, uint32_t G, uint32_t nl, uint32_t m, uint32_t n_chunks,
                 const int32_t* __restrict__ head, size_t head_stride, const int32_t* __restrict__ cont, size_t cont_stride,
                 int32_t* __restrict__ bucket, size_t bucket_stride) {
    latency_bound_kernel();
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
    const uint32_t L = acc_seg_len(offs[G], nl);
    BucketSpan s;
    s.g = g; s.t1 = 0; s.np = 0; s.long_run = false;
    if (g < G && L) s = bucket_span(offs, g, L);
    Xyzz acc;
    bucket_sum_wave(acc, s, lane, head, head_stride, cont, cont_stride);
    if (g < G) xyzz_store(bucket, bucket_stride, m ? bucket_pos(g, m, n_chunks) : (size_t)g, acc);
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
__device__ __forceinline__ int zeta_role(uint32_t lane) {       // lane 0 -> role 6 (total), lane 2^k -> role k, else -1
    if (lane == 0) return 6;
    if ((lane & (lane - 1)) != 0) return -1;
    return __ffs((int)lane) - 1;
}
__device__ __forceinline__ void xyzz_store_wire(uint32_t* __restrict__ out_wire, size_t i, const Xyzz& v) {
    uint32_t w[32];
    xyzz_to_wire(w, v);
#pragma unroll
    for (int j = 0; j < 32; j += 4) *reinterpret_cast<uint4*>(out_wire + i * 32 + j) = make_uint4(w[j], w[j + 1], w[j + 2], w[j + 3]);
}
// (the one-lane kernels of this form -- one wave per group of 64 buckets, X1[role * G1 + g]: role k < 6 = S_k, role 6 = T; then one launch that finishes
// bits 0..5 and transforms the group totals -- were the round-1 reduction; the lane-pair and lane-quad kernels below replaced them: history section 4b)
// -------------------------------------------------------------------------------------------------
// 6c. the same two reduction levels on LANE PAIRS (curve_pair.h): one point per pair of lanes, 7 multiplications per lane and
//     addition instead of 14; 64 values = one workgroup of two waves (32 pairs each).  Same inputs, same X1 / out_wire layout
//     and the same group elements as the one-lane form of 6b.
// -------------------------------------------------------------------------------------------------
__device__ __forceinline__ void bucket_partial_half(HalfXyzz& v, const BucketSpan& s, uint32_t k, const int32_t* __restrict__ head, size_t head_stride,
                                                    const int32_t* __restrict__ cont, size_t cont_stride, bool odd) {
    if (k == 0) { half_load(v, head, head_stride, s.g, odd); return; }
    const uint32_t first = s.t1 + 1;
    const uint32_t t = !s.long_run ? s.t1 + k : (k == 1 ? first : (first / 64 + (k - 1)) * 64);
    half_load(v, cont, cont_stride, t, odd);
}
// A heavy bucket (more than NP_SERIAL partials: skewed scalars, few distinct digits), summed by the 32 pairs of the wave: pair p
// takes the partials p, p + 32, ..; then a 5-step tree; every lane returns the sum.  Out of line: rare, and its registers (a second
// running sum) stay out of the kernel's budget.  ONE addition site for both phases.
__device__ __noinline__ void bucket_sum_heavy_pairs(HalfXyzz& tot, uint32_t g, uint32_t t1, uint32_t np, uint32_t long_run, uint32_t lane,
                                                    const int32_t* __restrict__ head, size_t head_stride, const int32_t* __restrict__ cont, size_t cont_stride) {
    const uint32_t pair = lane >> 1;
    const bool odd = (lane & 1u) != 0;
    BucketSpan h;
    h.g = g; h.t1 = t1; h.np = np; h.long_run = long_run != 0;
    HalfXyzz part;
    half_set_inf(part);
    const uint32_t loads = (np + 31) / 32;
#pragma unroll 1
    for (uint32_t step = 0; step < loads + 5; ++step) {
        HalfXyzz u;
        bool on;
        if (step < loads) {
            const uint32_t k = pair + 32 * step;
            on = k < np;
            if (on) bucket_partial_half(u, h, k, head, head_stride, cont, cont_stride, odd);
        } else {
            const uint32_t d = 16u >> (step - loads);
            half_shfl_down(u, part, (int)(2 * d));
            on = pair < d;
        }
        if (on) {
            HalfXyzz r;
            pair_add(r, part, u, odd);
            part = r;
        }
    }
    half_shfl(tot, part, odd ? 1 : 0);                 // pair 0 holds the sum
}
// Sum of all partials of this PAIR's bucket (s is the same in both lanes of the pair).  Buckets with up to NP_SERIAL partials are
// summed by their pair; heavier ones one after the other by the whole wave.  Every lane of the wave must call this.
__device__ __forceinline__ void bucket_sum_pairs(HalfXyzz& acc, const BucketSpan& s, uint32_t lane, bool odd, const int32_t* __restrict__ head,
                                                 size_t head_stride, const int32_t* __restrict__ cont, size_t cont_stride) {
    half_set_inf(acc);
    const bool heavy = s.np > NP_SERIAL;
    if (!heavy) {
#pragma unroll 1
        for (uint32_t k = 0; k < s.np; ++k) {
            HalfXyzz v, r;
            bucket_partial_half(v, s, k, head, head_stride, cont, cont_stride, odd);
            pair_add(r, acc, v, odd);
            acc = r;
        }
    }
    unsigned long long todo = __ballot(heavy && !odd);  // one bit per heavy pair (its even lane)
    while (todo) {                                     // wave-uniform
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        HalfXyzz tot;
        bucket_sum_heavy_pairs(tot, __shfl(s.g, src, 64), __shfl(s.t1, src, 64), __shfl(s.np, src, 64), (uint32_t)__shfl((int)s.long_run, src, 64), lane,
                               head, head_stride, cont, cont_stride);
        if ((int)(lane & ~1u) == src) acc = tot;
    }
}
// Superset-sum transform over the 64 values of a two-wave group, value gp = 32 w + pair: afterwards gp = 0 holds the total and
// gp = 2^k the sum over the values whose index has bit k set.  Steps 0..4 inside the wave, step 5 through LDS (wave 1 publishes,
// wave 0 adds).  Called by every lane of both waves (one barrier inside).
__device__ __forceinline__ void group_zeta64(HalfXyzz& v, uint32_t lane, uint32_t w, bool odd, int32_t* __restrict__ lds /* 2 NL x 64 words */) {
    const uint32_t pair = lane >> 1;
#pragma unroll 1
    for (int k = 0; k < 6; ++k) {                      // one addition site for the six steps
        HalfXyzz u;
        bool on;
        if (k < 5) {
            half_shfl_down(u, v, 2 << k);
            on = ((pair >> k) & 1u) == 0;
        } else {
            if (w == 1) {
#pragma unroll
                for (int j = 0; j < NL; ++j) { lds[j * 64 + lane] = v.inf ? 0 : v.u.l[j]; lds[(NL + j) * 64 + lane] = v.inf ? 0 : v.v.l[j]; }
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < NL; ++j) { u.u.l[j] = lds[j * 64 + lane]; u.v.l[j] = lds[(NL + j) * 64 + lane]; }
            u.inf = fe_is_literal_zero(u.v);
            on = w == 0;
        }
        if (on) {
            HalfXyzz r;
            pair_add(r, v, u, odd);
            v = r;
        }
    }
}
#ifndef KZG_PAIR_WAVES
#define KZG_PAIR_WAVES 3       // waves per SIMD of the pair kernels (3: <= 168 VGPRs, a wave fits beside two accumulate waves of the other MSM in flight)
#endif
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(KZG_PAIR_WAVES, KZG_PAIR_WAVES)))
k_msm_bucket_bits1p(const uint32_t* __restrict__ offs, uint32_t B, uint32_t nl, const int32_t* __restrict__ head, size_t head_stride,
                    const int32_t* __restrict__ cont, size_t cont_stride, uint32_t G1, int32_t* __restrict__ x1, size_t x_stride,
                    uint32_t* __restrict__ out_wire /* G1 == 1 only */) {
    latency_bound_kernel();
    __shared__ int32_t lds[2 * NL * 64];
    const uint32_t g = blockIdx.x, w = threadIdx.x >> 6, lane = threadIdx.x & 63, gp = w * 32 + (lane >> 1);
    const bool odd = (lane & 1u) != 0;
    const uint32_t L = acc_seg_len(offs[B], nl);
    const uint32_t bkt = g * 64 + gp;
    BucketSpan s;
    s.g = bkt; s.t1 = 0; s.np = 0; s.long_run = false;
    if (bkt < B && L) s = bucket_span(offs, bkt, L);
    HalfXyzz v;
#ifdef KZG_PROBE_NOSUM          // diagnostic builds (tools/ab_reduce.sh): where does this kernel's time go
    if (s.np) half_load(v, head, head_stride, s.g, odd); else half_set_inf(v);
#else
    bucket_sum_pairs(v, s, lane, odd, head, head_stride, cont, cont_stride);
#endif
#ifndef KZG_PROBE_NOZETA
    group_zeta64(v, lane, w, odd, lds);
#endif
    const int role = zeta_role(gp);
    if (role < 0) return;
    if (G1 == 1) half_store_wire(out_wire, (size_t)role, v, odd);
    else half_store(x1, x_stride, (size_t)role * G1 + g, v, odd);
}
// level 1 FUSED with the accumulation, for sparse MSMs (a few entries per bucket: commitments of <= 2^12 coefficients on the c = 15
// tables): pair = bucket, it adds its own sorted entries with pair_madd (5 multiplications per lane and entry) -- no k_msm_accumulate,
// no head / continuation partials to gather and sum -- then the zeta transform as above.  With ~2 entries per bucket the equal-split
// accumulate kernel spent a binary search and 4 dependent one-lane mixed additions (57 us) to leave ~2 partials per bucket, which the
// first level then added again (another ~30 us).
// A HEAVY bucket of the fused level (more than FUSED_HEAVY entries: a short top window puts e.g. 256 of a 512-coefficient polynomial's
// entries into one bucket of a small SRS's tables) is summed by the whole wave: pair p adds the entries hb + p, hb + p + 32, ..., a
// five-level tree joins the 32 partial sums.  One pair walking it alone kept its wave -- and the launch -- waiting for 256 dependent
// additions (0.93 ms for a 512-coefficient commitment on a 512-point SRS; 0.07 ms for the level otherwise).  Every lane of the wave calls.
constexpr uint32_t FUSED_HEAVY = 12;        // (24 while the groups held neighbouring buckets: the ~100 heavy buckets of a blob's top window then sat in two workgroups)
constexpr uint32_t FUSED_HEAVY_QUADS = 8;   // lane-quad form: 16 quads share a heavy bucket
__device__ __noinline__ void fused_heavy_bucket(HalfXyzz& tot, uint32_t hb, uint32_t he, uint32_t lane, const uint4* __restrict__ points,
                                                const uint32_t* __restrict__ sorted, uint32_t idx_log, uint32_t stride_adj) {
    const uint32_t pair = lane >> 1;
    const bool odd = (lane & 1u) != 0;
    HalfXyzz part;
    half_set_inf(part);
#pragma unroll 1
    for (uint32_t e = hb + pair; e < he; e += 32) {     // pair-uniform trip count
        const uint32_t cur = sorted[e];
        const uint4* src = points + 4 * acc_point_index(cur & 0x7FFFFFFFu, idx_log, stride_adj) + (odd ? 2 : 0);
        const uint4 q0 = src[0], q1 = src[1];
        const uint32_t w32[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
        int any = (q0.x | q0.y | q0.z | q0.w | q1.x | q1.y | q1.z | q1.w) != 0 ? 1 : 0;
        any |= pair_swap(any);
        if (!any) continue;
        Fq c;
        fe_unpack(c, w32);
        HalfXyzz r;
        pair_madd(r, part, c, cur >> 31, odd);
        part = r;
    }
#pragma unroll 1
    for (int d = 16; d >= 1; d >>= 1) {
        HalfXyzz u;
        half_shfl_down(u, part, 2 * d);
        if (pair < (uint32_t)d) {
            HalfXyzz r;
            pair_add(r, part, u, odd);
            part = r;
        }
    }
    half_shfl(tot, part, odd ? 1 : 0);                  // pair 0 holds the sum
}
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(KZG_PAIR_WAVES, KZG_PAIR_WAVES)))
k_msm_bucket_bits1p_fused(const uint4* __restrict__ points, const uint32_t* __restrict__ sorted, const uint32_t* __restrict__ offs, uint32_t B,
                          uint32_t idx_log, uint32_t stride_adj, uint32_t G1, int32_t* __restrict__ x1, size_t x_stride,
                          uint32_t* __restrict__ out_wire /* G1 == 1 only */) {
    latency_bound_kernel();
    __shared__ int32_t lds[2 * NL * 64];
    const uint32_t g = blockIdx.x, w = threadIdx.x >> 6, lane = threadIdx.x & 63, gp = w * 32 + (lane >> 1);
    const bool odd = (lane & 1u) != 0;
    const uint32_t bkt = gp * G1 + g;                   // strided: neighbouring buckets (the few heavy ones of a short top window) go to different workgroups
    uint32_t e = 0, end = 0;
    if (bkt < B) { e = offs[bkt]; end = offs[bkt + 1]; }
    const uint32_t hb = e, he = end;
    const bool heavy = end - e > FUSED_HEAVY;
    if (heavy) end = e;                                 // summed by the whole wave below
    HalfXyzz v;
    half_set_inf(v);
    // two-deep software pipeline as in k_msm_accumulate: entry e + 2 and the point of entry e + 1 are in flight while entry e is added
    // (clamped, unconditional prefetches: the dependent index -> point gather would otherwise be exposed in every iteration)
    const uint32_t last = end ? end - 1 : 0;
    uint32_t ent = e < end ? sorted[e] : 0u;
    uint32_t ent1 = e < end ? sorted[e + 1 < end ? e + 1 : last] : 0u;
    const uint4* src = points + 4 * acc_point_index(ent & 0x7FFFFFFFu, idx_log, stride_adj) + (odd ? 2 : 0);
    uint4 q0 = make_uint4(0, 0, 0, 0), q1 = q0;
    if (e < end) { q0 = src[0]; q1 = src[1]; }          // even lane: x, odd lane: y
#pragma unroll 1
    for (; e < end; ++e) {                              // pair-uniform trip count
        const uint32_t cur = ent;
        const uint32_t w32[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
        int any = (q0.x | q0.y | q0.z | q0.w | q1.x | q1.y | q1.z | q1.w) != 0 ? 1 : 0;
        ent = ent1;
        src = points + 4 * acc_point_index(ent & 0x7FFFFFFFu, idx_log, stride_adj) + (odd ? 2 : 0);
        q0 = src[0]; q1 = src[1];
        ent1 = sorted[e + 2 < end ? e + 2 : last];
        any |= pair_swap(any);
        if (!any) continue;                             // identity base (pair-uniform)
        Fq c;
        fe_unpack(c, w32);
        HalfXyzz r;
        pair_madd(r, v, c, cur >> 31, odd);
        v = r;
    }
    unsigned long long todo = __ballot(heavy && !odd);   // one bit per heavy pair (its even lane)
    while (todo) {                                      // wave-uniform
        const int src_lane = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        HalfXyzz tot;
        fused_heavy_bucket(tot, __shfl(hb, src_lane, 64), __shfl(he, src_lane, 64), lane, points, sorted, idx_log, stride_adj);
        if ((int)(lane & ~1u) == src_lane) v = tot;
    }
    group_zeta64(v, lane, w, odd, lds);
    const int role = zeta_role(gp);
    if (role < 0) return;
    if (G1 == 1) half_store_wire(out_wire, (size_t)role, v, odd);
    else half_store(x1, x_stride, (size_t)role * G1 + g, v, odd);
}

// Superset-sum transforms of TWO groups of 64 values at once on a two-wave workgroup, pair gp = 32 w + (lane >> 1).
// (Used by the second level only.  On the first level it HALVES the waves of a kernel that runs two waves per SIMD at 2^16 buckets and
// doubles the serial bucket sums of each: k_msm_bucket_bits1p went 0.123 -> 0.207 ms at 2^20 pairs, 77 -> 131 us at 2^11 (same box).)
// A zeta step k adds, for the pairs whose index has bit k CLEAR, the value of the pair 2^k further up -- the other half of the pairs
// would idle while the wave pays the issue slots.  So the workgroup carries a second group B in complemented order (pair gp holds
// B's value 63 - gp): B's step k is due exactly on the pairs whose bit k is SET, and its partner is the same pair gp ^ 2^k.  Every
// pair does one addition per step (for A or for B), six steps transform both groups: half the issue slots per group.
//   afterwards  A: gp = 0 holds the total, gp = 2^k the sum over the values whose index has bit k set;
//               B: gp = 63 the total,     gp = 63 - 2^k that sum.
// Steps 0..4 exchange inside the wave (ds_bpermute, xor partner), step 5 through LDS (wave 0 needs wave 1's A values, wave 1 needs
// wave 0's B values).  Called by every lane of both waves (one barrier inside).
__device__ __forceinline__ void group_zeta64_dual(HalfXyzz& vA, HalfXyzz& vB, uint32_t lane, uint32_t w, bool odd, int32_t* __restrict__ lds /* 2 x 2 NL x 64 words */) {
    const uint32_t gp = w * 32 + (lane >> 1);
#pragma unroll 1
    for (int k = 0; k < 6; ++k) {
        const bool bit = ((gp >> k) & 1u) != 0;
        HalfXyzz mine, pub, u, r;
        half_select(mine, bit, vB, vA);
        half_select(pub, bit, vA, vB);                 // what the partner pair (opposite bit k) adds
        if (k < 5) {
            half_shfl_xor(u, pub, 2 << k);
        } else {
            int32_t* mine_slot = lds + w * (2 * NL * 64);
            const int32_t* other_slot = lds + (1 - w) * (2 * NL * 64);
#pragma unroll
            for (int j = 0; j < NL; ++j) { mine_slot[j * 64 + lane] = pub.inf ? 0 : pub.u.l[j]; mine_slot[(NL + j) * 64 + lane] = pub.inf ? 0 : pub.v.l[j]; }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < NL; ++j) { u.u.l[j] = other_slot[j * 64 + lane]; u.v.l[j] = other_slot[(NL + j) * 64 + lane]; }
            u.inf = fe_is_literal_zero(u.v);
        }
        pair_add(r, mine, u, odd);
        half_select(vA, bit, vA, r);
        half_select(vB, bit, r, vB);
    }
}
// level 2 (one launch, two kinds of job), results straight to wire words: with G1p = ceil(G1 / 64)
//   job <  6 G1p : out[a * G1p + g'] = sum of X1[a][g' * 64 .. +64)                                  (a < 6: finishes bits 0..5)
//   job >= 6 G1p : out[6 G1p + role * G1p + g2], role 0..6, zeta transform of the totals X1[6][g2 * 64 .. +64)  (bits 6..11, totals)
// one two-wave workgroup per TWO jobs (jobs 2 blk and 2 blk + 1)
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(KZG_PAIR_WAVES, KZG_PAIR_WAVES)))
k_red_bits2p(const int32_t* __restrict__ x1, size_t x_stride, uint32_t G1, uint32_t G1p, uint32_t* __restrict__ out_wire) {
    latency_bound_kernel();
    __shared__ int32_t lds[2 * 2 * NL * 64];
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63, gp = w * 32 + (lane >> 1);
    const bool odd = (lane & 1u) != 0;
    const uint32_t n_jobs = 7u * G1p;
    HalfXyzz v[2];
    bool sum_job[2];
    uint32_t job[2], grp[2];
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        job[which] = 2 * blockIdx.x + which;
        const bool valid = job[which] < n_jobs;
        sum_job[which] = job[which] < 6u * G1p;
        const uint32_t a = sum_job[which] ? job[which] / G1p : 6u;
        grp[which] = sum_job[which] ? job[which] % G1p : job[which] - 6u * G1p;
        const uint32_t cnt = valid ? (G1 - grp[which] * 64 < 64 ? G1 - grp[which] * 64 : 64) : 0;
        const uint32_t idx = which ? 63 - gp : gp;
        if (idx < cnt) half_load(v[which], x1, x_stride, (size_t)a * G1 + (size_t)grp[which] * 64 + idx, odd);
        else half_set_inf(v[which]);
    }
    group_zeta64_dual(v[0], v[1], lane, w, odd, lds);
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        if (job[which] >= n_jobs) continue;
        const uint32_t idx = which ? 63 - gp : gp;
        if (sum_job[which]) {
            if (idx == 0) half_store_wire(out_wire, job[which], v[which], odd);
        } else {
            const int role = zeta_role(idx);
            if (role >= 0) half_store_wire(out_wire, (size_t)6 * G1p + (size_t)role * G1p + grp[which], v[which], odd);
        }
    }
}

// -------------------------------------------------------------------------------------------------
// 6d. the same two reduction levels on LANE QUADS (curve_quad.h): one point per four lanes, four multiplications per lane and addition
//     instead of seven; 64 values = one workgroup of four waves (16 quads each).  ~0.7 of the pair form's dependent instructions at
//     twice its lanes: used when no other MSM is in flight (then latency is everything and the SIMDs are idle anyway).
//     Same inputs, same X1 / out_wire layout, the same group elements (KZG_QUAD_REDUCE=0: always the pair kernels).
// -------------------------------------------------------------------------------------------------
#ifndef KZG_QUAD_SMALL_WAVES
#define KZG_QUAD_SMALL_WAVES 4     // waves per SIMD the small quad launches are compiled for (128 VGPRs with a few spills; 2 = no spills measured the same or 5 % slower)
#endif
__device__ __forceinline__ void bucket_partial_quad(QuadXyzz& v, const BucketSpan& s, uint32_t k, const int32_t* __restrict__ head, size_t head_stride,
                                                    const int32_t* __restrict__ cont, size_t cont_stride, uint32_t q) {
    if (k == 0) { quad_load(v, head, head_stride, s.g, q); return; }
    const uint32_t first = s.t1 + 1;
    const uint32_t t = !s.long_run ? s.t1 + k : (k == 1 ? first : (first / 64 + (k - 1)) * 64);
    quad_load(v, cont, cont_stride, t, q);
}
// a heavy bucket, summed by the 16 quads of the wave: quad p takes the partials p, p + 16, ..; then a 4-step tree
__device__ __noinline__ void bucket_sum_heavy_quads(QuadXyzz& tot, uint32_t g, uint32_t t1, uint32_t np, uint32_t long_run, uint32_t lane,
                                                    const int32_t* __restrict__ head, size_t head_stride, const int32_t* __restrict__ cont, size_t cont_stride) {
    const uint32_t quad = lane >> 2, q = lane & 3u;
    BucketSpan h;
    h.g = g; h.t1 = t1; h.np = np; h.long_run = long_run != 0;
    QuadXyzz part;
    quad_set_inf(part);
    const uint32_t loads = (np + 15) / 16;
#pragma unroll 1
    for (uint32_t step = 0; step < loads + 4; ++step) {
        QuadXyzz u;
        bool on;
        if (step < loads) {
            const uint32_t k = quad + 16 * step;
            on = k < np;
            if (on) bucket_partial_quad(u, h, k, head, head_stride, cont, cont_stride, q);
        } else {
            const uint32_t d = 8u >> (step - loads);
            quad_shfl_down(u, part, (int)(4 * d));
            on = quad < d;
        }
        if (on) {
            QuadXyzz r;
            quad_add(r, part, u, q);
            part = r;
        }
    }
    quad_shfl(tot, part, (int)q);                      // quad 0 holds the sum
}
// Sum of all partials of this QUAD's bucket (s is the same in the four lanes).  Every lane of the wave must call this.
__device__ __forceinline__ void bucket_sum_quads(QuadXyzz& acc, const BucketSpan& s, uint32_t lane, uint32_t q, const int32_t* __restrict__ head,
                                                 size_t head_stride, const int32_t* __restrict__ cont, size_t cont_stride) {
    quad_set_inf(acc);
    const bool heavy = s.np > NP_SERIAL;
    if (!heavy) {
#pragma unroll 1
        for (uint32_t k = 0; k < s.np; ++k) {
            QuadXyzz v, r;
            bucket_partial_quad(v, s, k, head, head_stride, cont, cont_stride, q);
            quad_add(r, acc, v, q);
            acc = r;
        }
    }
    unsigned long long todo = __ballot(heavy && q == 0);   // one bit per heavy quad (its lane 0)
    while (todo) {                                     // wave-uniform
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        QuadXyzz tot;
        bucket_sum_heavy_quads(tot, __shfl(s.g, src, 64), __shfl(s.t1, src, 64), __shfl(s.np, src, 64), (uint32_t)__shfl((int)s.long_run, src, 64), lane,
                               head, head_stride, cont, cont_stride);
        if ((int)(lane & ~3u) == src) acc = tot;
    }
}
// Superset-sum transform over the 64 values of a four-wave group, value gp = 16 w + quad: afterwards gp = 0 holds the total and
// gp = 2^k the sum over the values whose index has bit k set.  Steps 0..3 inside the wave, steps 4 and 5 through LDS (the waves whose
// index has the bit set publish, the others add).  Called by every lane of the four waves (four barriers inside).
__device__ __forceinline__ void group_zeta64q(QuadXyzz& v, uint32_t lane, uint32_t w, uint32_t q, int32_t* __restrict__ lds /* (NL + 1) x 256 words */) {
    const uint32_t quad = lane >> 2, tid = w * 64 + lane;
#pragma unroll 1
    for (int k = 0; k < 6; ++k) {                      // one addition site for the six steps
        QuadXyzz u;
        bool on;
        if (k < 4) {
            quad_shfl_down(u, v, 4 << k);
            on = ((quad >> k) & 1u) == 0;
        } else {
            const uint32_t bit = 1u << (k - 4);
            if (k == 5) __syncthreads();               // step 4's readers are done
            if (w & bit) {
#pragma unroll
                for (int j = 0; j < NL; ++j) lds[j * 256 + tid] = v.c.l[j];
                lds[NL * 256 + tid] = v.inf ? 1 : 0;
            }
            __syncthreads();
            on = (w & bit) == 0;
            const uint32_t src = tid + 64 * bit;
            if (on) {
#pragma unroll
                for (int j = 0; j < NL; ++j) u.c.l[j] = lds[j * 256 + src];
                u.inf = lds[NL * 256 + src] != 0;
            }
        }
        if (on) {
            QuadXyzz r;
            quad_add(r, v, u, q);
            v = r;
        }
    }
}
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(KZG_QUAD_SMALL_WAVES, KZG_QUAD_SMALL_WAVES)))
k_msm_bucket_bits1q(const uint32_t* __restrict__ offs, uint32_t B, uint32_t nl, const int32_t* __restrict__ head, size_t head_stride,
                    const int32_t* __restrict__ cont, size_t cont_stride, uint32_t G1, int32_t* __restrict__ x1, size_t x_stride,
                    uint32_t* __restrict__ out_wire /* G1 == 1 only */) {
    latency_bound_kernel();
    __shared__ int32_t lds[(NL + 1) * 256];
    const uint32_t g = blockIdx.x, w = threadIdx.x >> 6, lane = threadIdx.x & 63, gp = w * 16 + (lane >> 2), q = lane & 3u;
    const uint32_t L = acc_seg_len(offs[B], nl);
    const uint32_t bkt = g * 64 + gp;
    BucketSpan s;
    s.g = bkt; s.t1 = 0; s.np = 0; s.long_run = false;
    if (bkt < B && L) s = bucket_span(offs, bkt, L);
    QuadXyzz v;
    bucket_sum_quads(v, s, lane, q, head, head_stride, cont, cont_stride);
    group_zeta64q(v, lane, w, q, lds);
    const int role = zeta_role(gp);
    if (role < 0) return;
    if (G1 == 1) quad_store_wire(out_wire, (size_t)role, v, q);
    else quad_store(x1, x_stride, (size_t)role * G1 + g, v, q);
}
// level 1 FUSED with the accumulation (sparse MSMs; see k_msm_bucket_bits1p_fused): quad = bucket, entries added with quad_madd
__device__ __noinline__ void fused_heavy_bucket_quads(QuadXyzz& tot, uint32_t hb, uint32_t he, uint32_t lane, const uint4* __restrict__ points,
                                                      const uint32_t* __restrict__ sorted, uint32_t idx_log, uint32_t stride_adj) {
    const uint32_t quad = lane >> 2, q = lane & 3u;
    QuadXyzz part;
    quad_set_inf(part);
#pragma unroll 1
    for (uint32_t e = hb + quad; e < he; e += 16) {     // quad-uniform trip count
        const uint32_t cur = sorted[e];
        const uint4* src = points + 4 * acc_point_index(cur & 0x7FFFFFFFu, idx_log, stride_adj) + ((q & 1u) ? 2 : 0);
        const uint4 q0 = src[0], q1 = src[1];
        const uint32_t w32[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
        int any = (q0.x | q0.y | q0.z | q0.w | q1.x | q1.y | q1.z | q1.w) != 0 ? 1 : 0;
        any |= pair_swap(any);
        if (!any) continue;
        Fq c;
        fe_unpack(c, w32);
        QuadXyzz r;
        quad_madd(r, part, c, cur >> 31, q);
        part = r;
    }
#pragma unroll 1
    for (int d = 8; d >= 1; d >>= 1) {
        QuadXyzz u;
        quad_shfl_down(u, part, 4 * d);
        if (quad < (uint32_t)d) {
            QuadXyzz r;
            quad_add(r, part, u, q);
            part = r;
        }
    }
    quad_shfl(tot, part, (int)q);                       // quad 0 holds the sum
}
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(KZG_QUAD_SMALL_WAVES, KZG_QUAD_SMALL_WAVES)))      // (sparse MSMs: at most 256 workgroups)
k_msm_bucket_bits1q_fused(const uint4* __restrict__ points, const uint32_t* __restrict__ sorted, const uint32_t* __restrict__ offs, uint32_t B,
                          uint32_t idx_log, uint32_t stride_adj, uint32_t G1, int32_t* __restrict__ x1, size_t x_stride,
                          uint32_t* __restrict__ out_wire /* G1 == 1 only */) {
    latency_bound_kernel();
    __shared__ int32_t lds[(NL + 1) * 256];
    const uint32_t g = blockIdx.x, w = threadIdx.x >> 6, lane = threadIdx.x & 63, gp = w * 16 + (lane >> 2), q = lane & 3u;
    const uint32_t bkt = gp * G1 + g;                   // strided, as in k_msm_bucket_bits1p_fused
    uint32_t e = 0, end = 0;
    if (bkt < B) { e = offs[bkt]; end = offs[bkt + 1]; }
    const uint32_t hb = e, he = end;
    const bool heavy = end - e > FUSED_HEAVY_QUADS;
    if (heavy) end = e;                                 // summed by the whole wave below
    QuadXyzz v;
    quad_set_inf(v);
    const uint32_t half = (q & 1u) ? 2 : 0;             // lanes 0, 2: x; lanes 1, 3: y
    const uint32_t last = end ? end - 1 : 0;
    uint32_t ent = e < end ? sorted[e] : 0u;
    uint32_t ent1 = e < end ? sorted[e + 1 < end ? e + 1 : last] : 0u;
    const uint4* src = points + 4 * acc_point_index(ent & 0x7FFFFFFFu, idx_log, stride_adj) + half;
    uint4 q0 = make_uint4(0, 0, 0, 0), q1 = q0;
    if (e < end) { q0 = src[0]; q1 = src[1]; }
#pragma unroll 1
    for (; e < end; ++e) {                              // quad-uniform trip count
        const uint32_t cur = ent;
        const uint32_t w32[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
        int any = (q0.x | q0.y | q0.z | q0.w | q1.x | q1.y | q1.z | q1.w) != 0 ? 1 : 0;
        ent = ent1;
        src = points + 4 * acc_point_index(ent & 0x7FFFFFFFu, idx_log, stride_adj) + half;
        q0 = src[0]; q1 = src[1];
        ent1 = sorted[e + 2 < end ? e + 2 : last];
        any |= pair_swap(any);
        if (!any) continue;                             // identity base (quad-uniform)
        Fq c;
        fe_unpack(c, w32);
        QuadXyzz r;
        quad_madd(r, v, c, cur >> 31, q);
        v = r;
    }
    unsigned long long todo = __ballot(heavy && q == 0);
    while (todo) {                                      // wave-uniform
        const int src_lane = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        QuadXyzz tot;
        fused_heavy_bucket_quads(tot, __shfl(hb, src_lane, 64), __shfl(he, src_lane, 64), lane, points, sorted, idx_log, stride_adj);
        if ((int)(lane & ~3u) == src_lane) v = tot;
    }
    group_zeta64q(v, lane, w, q, lds);
    const int role = zeta_role(gp);
    if (role < 0) return;
    if (G1 == 1) quad_store_wire(out_wire, (size_t)role, v, q);
    else quad_store(x1, x_stride, (size_t)role * G1 + g, v, q);
}
// level 2: one four-wave workgroup per job of k_red_bits2
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(KZG_QUAD_SMALL_WAVES, KZG_QUAD_SMALL_WAVES)))      // (at most 7 x 16 workgroups)
k_red_bits2q(const int32_t* __restrict__ x1, size_t x_stride, uint32_t G1, uint32_t G1p, uint32_t* __restrict__ out_wire) {
    latency_bound_kernel();
    __shared__ int32_t lds[(NL + 1) * 256];
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63, gp = w * 16 + (lane >> 2), q = lane & 3u;
    const uint32_t job = blockIdx.x;                   // < 7 G1p
    const bool sum_job = job < 6u * G1p;
    const uint32_t a = sum_job ? job / G1p : 6u;
    const uint32_t grp = sum_job ? job % G1p : job - 6u * G1p;
    const uint32_t cnt = G1 - grp * 64 < 64 ? G1 - grp * 64 : 64;
    QuadXyzz v;
    if (gp < cnt) quad_load(v, x1, x_stride, (size_t)a * G1 + (size_t)grp * 64 + gp, q);
    else quad_set_inf(v);
    group_zeta64q(v, lane, w, q, lds);
    if (sum_job) {
        if (gp == 0) quad_store_wire(out_wire, job, v, q);
    } else {
        const int role = zeta_role(gp);
        if (role >= 0) quad_store_wire(out_wire, (size_t)6 * G1p + (size_t)role * G1p + grp, v, q);
    }
}

// -------------------------------------------------------------------------------------------------
// 6e. TINY MSMs over the per-bit tables as plain SUMS (n <= BITSUM_MAX_N pairs): sum_i s_i P_i = sum_i sum_p naf_p(s_i) Bit_p[i] with
//     the plain non-adjacent form of s_i (digits 0, +-1: ~85 table points per scalar).  No buckets, so no sort and no bit-weighted
//     transform: a quad takes 8 / 16 positions of one scalar (at most 4 / 8 non-zero digits: that many dependent mixed additions),
//     the 64 quads of a workgroup are summed by a six-step tree, a second launch sums 64 partial sums per workgroup, the host adds the <= 8 results.  A 512-coefficient
//     commitment is then ~19 dependent point additions deep instead of ~35 (bucket walk + two six-step transforms + the host's 14-bit
//     double-and-add), and two launches instead of five.
// -------------------------------------------------------------------------------------------------
constexpr uint32_t BITSUM_MAX_N = 8192;     // what the kernels take; the default policy (engine.h srs_bases) uses them up to 4 096
// CHUNK = bit positions per quad (8 / 16 / 32: 32 / 16 / 8 quads per scalar, at most 4 / 8 / 16 dependent mixed additions): chosen by n (msm.hip)
template <int CHUNK>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(KZG_QUAD_SMALL_WAVES, KZG_QUAD_SMALL_WAVES)))
k_bitsum_level1(const uint4* __restrict__ bits /* Bit_p[i] at (p stride + i) x 64 B */, uint32_t stride, const uint4* __restrict__ scalars, uint32_t n,
                int32_t* __restrict__ part /* gridDim.x partial sums, limb planes */) {
    latency_bound_kernel();
    __shared__ int32_t lds[(NL + 1) * 256];
    constexpr uint32_t PER = 256 / CHUNK, MASK = CHUNK == 32 ? 0xFFFFFFFFu : (1u << (CHUNK & 31)) - 1u;
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63, gp = w * 16 + (lane >> 2), q = lane & 3u;
    const uint32_t job = blockIdx.x * 64 + gp;          // (scalar, chunk of positions)
    const uint32_t i = job / PER, chunk = job % PER;
    QuadXyzz v;
    quad_set_inf(v);
    uint32_t digits = 0, negs = 0;                      // bit b: position CHUNK chunk + b carries a digit / the digit is -1
    if (i < n) {
        const uint4 lo = scalars[2 * (size_t)i], hi = scalars[2 * (size_t)i + 1];
        const uint32_t w32[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        uint32_t k[8];
        fe_wire_to_canonical_words<FrParams>(k, w32);
        // plain NAF: digit_p = bit_(p+1)(3k) - bit_(p+1)(k).  Only the two words around this chunk's positions are needed.
        uint32_t h[9];
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) { c += (uint64_t)k[j] * 3u; h[j] = (uint32_t)c; c >>= 32; }
        h[8] = (uint32_t)c;
        const uint32_t bit0 = (uint32_t)CHUNK * chunk + 1u;             // first bit of (3k, k) this chunk looks at
        const uint32_t wd = bit0 >> 5, sh = bit0 & 31u;
        uint32_t hw = 0, hw1 = 0, kw = 0, kw1 = 0;
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            if ((uint32_t)j == wd) { hw = h[j]; kw = j < 8 ? k[j] : 0u; }
            if ((uint32_t)j == wd + 1) { hw1 = h[j]; kw1 = j < 8 ? k[j] : 0u; }
        }
        const uint32_t hs = (uint32_t)((((uint64_t)hw1 << 32) | hw) >> sh) & MASK;
        const uint32_t ks = (uint32_t)((((uint64_t)kw1 << 32) | kw) >> sh) & MASK;
        digits = hs ^ ks;
        negs = ks & ~hs;
    }
    // (the same for the four lanes of the quad: quad-uniform trip count)
    uint32_t todo = digits;
#pragma unroll 1
    while (todo) {
        const uint32_t b = (uint32_t)__ffs((int)todo) - 1u;
        todo &= todo - 1u;
        const uint32_t pos = (uint32_t)CHUNK * chunk + b;
        const uint4* src = bits + 4 * ((size_t)pos * stride + i) + ((q & 1u) ? 2 : 0);
        const uint4 q0 = src[0], q1 = src[1];
        const uint32_t w32[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
        int any = (q0.x | q0.y | q0.z | q0.w | q1.x | q1.y | q1.z | q1.w) != 0 ? 1 : 0;
        any |= pair_swap(any);
        if (!any) continue;                             // identity base
        Fq cxy;
        fe_unpack(cxy, w32);
        QuadXyzz r;
        quad_madd(r, v, cxy, (negs >> b) & 1u, q);
        v = r;
    }
    group_zeta64q(v, lane, w, q, lds);                  // value 0 of the transform = the sum of the 64 quads
    if (gp == 0) quad_store(part, gridDim.x, blockIdx.x, v, q);
}
// workgroup b: the sum of the partial sums [64 b, 64 b + 64) -> out_wire[b] (at most 8 points, added on the host)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(KZG_QUAD_SMALL_WAVES, KZG_QUAD_SMALL_WAVES)))
k_bitsum_level2(const int32_t* __restrict__ part, uint32_t n_part, uint32_t* __restrict__ out_wire) {
    latency_bound_kernel();
    __shared__ int32_t lds[(NL + 1) * 256];
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63, gp = w * 16 + (lane >> 2), q = lane & 3u;
    const uint32_t j = blockIdx.x * 64 + gp;
    QuadXyzz v;
    if (j < n_part) quad_load(v, part, n_part, j, q);
    else quad_set_inf(v);
    group_zeta64q(v, lane, w, q, lds);
    if (gp == 0) quad_store_wire(out_wire, blockIdx.x, v, q);
}

// Batched table mode (k_naf_digits with poly_len != 0): group g of the first reduction level IS polynomial g.  Its seven sums
// (x1[role * G1 + g]: S_0 .. S_5 over the key bits, T the total) give the commitment sum_key (2 key + 1) V_key = 2 sum_j 2^j S_j + T:
// six doublings and six additions on one lane per polynomial, XYZZ wire words out.
__global__ void __launch_bounds__(64)
k_batch_finish(const int32_t* __restrict__ x1, size_t x_stride, uint32_t G1, uint32_t* __restrict__ out_wire) {
    latency_bound_kernel();
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G1) return;
    Xyzz acc, s, t;
    xyzz_load(acc, x1, x_stride, (size_t)5 * G1 + g);
#pragma unroll 1
    for (int j = 4; j >= -1; --j) {
        xyzz_dbl_impl(t, acc);
        xyzz_load(s, x1, x_stride, (size_t)(j >= 0 ? j : 6) * G1 + g);      // j = -1: the total
        xyzz_add<true>(acc, t, s);
    }
    xyzz_store_wire(out_wire, g, acc);
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

// the same conversion with the curve check y^2 == x^3 + 3 of every point that is not the identity (batch verification validates its
// 2n inputs, verifier/src/batch.rs:203-210: done here on the points the MSM uploads anyway); *flag |= 1 for a point off the curve
__global__ void __launch_bounds__(256)
k_points_wire_to_device_checked(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n, uint32_t* __restrict__ flag) {
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
    uint32_t any = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) any |= o[j];
    if (any == 0) return;                                     // identity
    Fq x, y, t, y2, b3;
    fe_unpack(x, o);
    fe_unpack(y, o + 8);
#pragma unroll
    for (int j = 0; j < NL; ++j) b3.l[j] = (int32_t)FqParams::B3[j];
    fe_sqr(t, x);
    fe_mul(t, t, x);
    fe_add(t, t, b3);
    fe_norm(t);                                               // x^3 + 3, |.| < 4m
    fe_sqr(y2, y);
    fe_sub(t, y2, t);
    fe_reduce(t);
    if (!fe_is_zero_mod(t)) atomicOr(flag, 1u);
}

}  // namespace kzg
