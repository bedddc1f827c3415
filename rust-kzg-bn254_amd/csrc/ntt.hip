// ntt.hip — radix-2^K Stockham NTT / inverse NTT over BN254 Fr, LDS-tiled (round 2: two passes at 2^20).
//
// Replaces `GeneralEvaluationDomain::<Fr>::new(n).fft(..)` / `.ifft(..)` at
// primitives/src/polynomial.rs:131-135 and :242-246: natural order in and out on the domain
// {w^i}, w = 5^((r-1)/n) (= PRIMITIVE_ROOTS_OF_UNITY[log2 n], primitives/src/consts.rs:22-52); the
// inverse uses w^-1 and scales by n^-1.
//
// Algorithm (decimation in time, Stockham autosort, P = ceil(log n / 10) passes: 10 + 10 at n = 2^20):
//   pass with radix R = 2^K brings the sub-transform length from n_cur/R to n_cur (stride s = N / n_cur):
//     y[u + j * N/R] = sum_j'  w_{n_cur}^(p j') x[q + s (R p + j')] * w_R^(j j'),   u = q + s p
//   A workgroup (512 threads, one per CU, tiles of 2 048 elements: 2^19, 2^20, >= 2^25 -- or 256 threads, two per CU, tiles of 1 024:
//   the other sizes, round 4) owns tiles of C = TILE / R consecutive units u.  Per tile: the R x C
//   elements come from global memory as canonical 256-bit words (runs of C x 32 B), go through the K radix-2 butterfly stages in
//   LDS (9 limb planes, padded rows; radix-4 steps in registers, their multiplies in independent PAIRS: fe_mul2), and leave as
//   256-bit words again.  The inter-pass twiddle w_{n_cur'}^(p' j') of the NEXT pass is applied at the STORE of this pass (the
//   multiply that brings the lazy value back into (-m, 2m) anyway), so later passes load, unpack and go.  Data stay in the wire
//   residue class (a * 2^256) throughout: the transform is linear and the twiddles are in internal Montgomery form.
//   Round 1 ran three passes (7 + 6 + 7) with 36-byte limb planes in between: SQ counters showed its VALU work fully issued
//   (27 of 45 us per pass) and the rest spent in the un-overlapped load and store phases of a single round of tiles.  Here a
//   workgroup walks several tiles and fetches the NEXT tile's words into registers while it computes the current one.
//   Algorithmic traffic: 64 B per element; this implementation moves P x 64 B.
#include "engine.h"
#include "field29.h"

#include <algorithm>
#include <map>
#include <mutex>
#include <tuple>

namespace kzg {

constexpr int NTT_KMAX = 10;
constexpr int NTT_EPT = 4;                         // elements per thread in the load / store phases = one radix-4 butterfly per thread and step
// Two tile sizes (round 4).  2 048 elements / 512 threads, one workgroup per CU: transforms of >= 2^19 elements (>= 256 tiles per pass).
// 1 024 elements / 256 threads, two workgroups per CU: smaller transforms -- twice the workgroups (2^18: 256 instead of 128 on 256
// CUs; one tile of a <= 1 024-point transform in half the threads) and two independent barrier domains per CU:
// 2^12 / 2^16 / 2^18 0.052 / 0.062 / 0.073 -> 0.040 / 0.046 / 0.057 ms per call; at 2^20 the big tile stays ahead (0.140 against 0.147).
template <int TILE_LOG> struct NttTile {
    static constexpr int LOG = TILE_LOG;
    static constexpr int TILE = 1 << TILE_LOG;     // elements per workgroup tile
    static constexpr int THREADS = TILE / NTT_EPT;
    // plane length: rows x (C + 1) words in the row-major layout (C >= 16, so at most TILE / 16 rows), C x (R + 32 / C) in the
    // column-major one (ntt_lds_pos) -> TILE + max(TILE / 16, 32) words
    static constexpr int PL = TILE + (TILE / 16 > 32 ? TILE / 16 : 32);
};
constexpr int NTT_TILE_LOG_BIG = 11, NTT_TILE_LOG_SMALL = 10;
// measured per call, small / big tile (tools/archive/time_ntt_variants.py, KZG_NTT_TILE_LOG=10 / 11): 2^10 0.034 / 0.043, 2^14 0.045 / 0.058,
// 2^17 0.055 / 0.067, 2^18 0.059 / 0.071, 2^19 0.086 / 0.082, 2^20 0.155 / 0.136, 2^21 0.280 / 0.297, 2^22 0.537 / 0.562, 2^23 1.195 /
// 1.246, 2^24 2.46 / 2.52, 2^25 5.31 / 5.34, 2^26 12.1 / 11.1 ms
// round 6: 2^25 moved to the small tile (its passes are 9 + 8 + 8 bits: the slim instantiation below, three workgroups per CU: 5.24 -> 4.95 ms; 2^26 stays: 10.9 against 11.5)
inline bool ntt_small_tile_pays(int log_n) { return log_n <= 18 || (log_n >= 21 && log_n <= 25); }
constexpr int NTT_LO_BITS = 10;

// ---- twiddle tables: planes[9][len] of w^(t * step), internal Montgomery form -----------------------
__device__ __forceinline__ void fr_pow_root(Fr& out, int log_n, bool inverse, uint32_t e) {
    // w_{2^log_n}^e by square-and-multiply from the generated root constants
    Fr base;
    const uint32_t* tab = inverse ? FrParams::ROOT_INV : FrParams::ROOT;
#pragma unroll
    for (int j = 0; j < NL; ++j) base.l[j] = (int32_t)tab[log_n * NL + j];
    Fr acc;
    fe_set_one(acc);
    while (e) {
        if (e & 1u) fe_mul(acc, acc, base);
        fe_sqr(base, base);
        e >>= 1;
    }
    out = acc;
}
__global__ void __launch_bounds__(256)
k_ntt_build_table(int32_t* __restrict__ planes, uint32_t len, int log_n, int inverse, uint32_t step) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= len) return;
    Fr w;
    fr_pow_root(w, log_n, inverse != 0, t * step);
#pragma unroll
    for (int j = 0; j < NL; ++j) planes[(size_t)j * len + t] = w.l[j];
}

__device__ __forceinline__ void load_planes(Fr& v, const int32_t* __restrict__ planes, size_t stride, size_t i) {
#pragma unroll
    for (int j = 0; j < NL; ++j) v.l[j] = planes[(size_t)j * stride + i];
}
// w_N^E from the two-level table (one multiply when E >= lo_len)
__device__ __forceinline__ void twiddle(Fr& w, const int32_t* __restrict__ lo, uint32_t lo_len, int lo_bits,
                                        const int32_t* __restrict__ hi, uint32_t hi_len, uint32_t E) {
    load_planes(w, lo, lo_len, E & (lo_len - 1));
    uint32_t eh = E >> lo_bits;
    if (eh != 0) {
        Fr h;
        load_planes(h, hi, hi_len, eh);
        fe_mul(w, w, h);
    }
}

// LDS position of (row i, column uu) of a tile.  C >= 16 columns: row-major with one pad word per row (consecutive lanes =
// consecutive columns = consecutive banks).  C <= 8 (K >= 8, e.g. the 1024 x 2 tiles of a 2^20 transform): a half-wave touches
// 32 / C different rows, so the layout is column-major with a column stride = R + 32 / C (column uu starts 32 / C banks further) and
// the row index is XOR-swizzled, pos = i ^ (((i >> 2) ^ (i >> 6)) & (32 / C - 1)): the rows of one access differ in bits {2..5}
// (first radix-4 step), {0,1,4,5} (second), {0..3} (later steps, final store) or the top bits (bit-reversed fill), and this map is
// injective on each of these sets -- without it the 1024 x 2 tile ran at 70-77 % LDS bank-conflict cycles (SQ_LDS_BANK_CONFLICT).
__device__ __forceinline__ uint32_t ntt_lds_pos(uint32_t i, uint32_t uu, int log_c, int K) {
    if (log_c >= 4) return i * ((1u << log_c) + 1u) + uu;
    const uint32_t spread = 32u >> log_c;                       // rows per half-wave
    const uint32_t cs = (1u << K) + spread;
    return uu * cs + (i ^ (((i >> 2) ^ (i >> 6)) & (spread - 1u)));
}

// inter-pass twiddles of one pass boundary for every element index: tw[idx] = w_N^((p' j') << log_s') as canonical words
__global__ void __launch_bounds__(256)
k_ntt_build_pass_twiddles(uint4* __restrict__ tw, int log_n, int next_K, int next_log_s,
                          const int32_t* __restrict__ tlo, uint32_t lo_len, int lo_bits, const int32_t* __restrict__ thi, uint32_t hi_len);

struct NttPassArgs {
    int log_n, K, log_s;          // this pass
    int next_K, next_log_s;       // the pass after it (next_K = 0: this is the last pass)
    int scale_log_n;              // last pass: >= 0 multiplies by (2^scale_log_n)^-1; -1: no factor left (forward transform, or the inverse's 1 / n
                                  // folded into the twiddle array of the previous pass boundary): the outputs are only reduced (fe_reduce_small)
    uint32_t n_tiles;
};

// global word index of element (uu, j) of tile `tile` on the INPUT side of the pass
__device__ __forceinline__ size_t ntt_in_index(const NttPassArgs& a, int tile_log, uint32_t tile, uint32_t t, uint32_t& uu, uint32_t& j, bool& valid) {
    const int log_c = tile_log - a.K;
    const uint32_t C = 1u << log_c, R = 1u << a.K, s = 1u << a.log_s;
    if (s >= C) { uu = t & (C - 1); j = t >> log_c; }
    else { const uint32_t q = t & (s - 1); j = (t >> a.log_s) & (R - 1); uu = ((t >> (a.log_s + a.K)) << a.log_s) | q; }
    const uint32_t u = (tile << log_c) + uu;
    valid = u < ((1u << a.log_n) >> a.K);
    const uint32_t q = u & (s - 1), p = u >> a.log_s;
    return (size_t)q + ((size_t)(R * p + j) << a.log_s);
}

// KMAX_T: largest radix 2^K this instantiation takes -- its per-workgroup twiddle table holds 2^(KMAX_T - 1) entries (18 KB at 10, 2.3 KB at 7: the
// three-pass plan of round 6, whose workgroups then fit three per CU)
template <int TILE_LOG, int KMAX_T = NTT_KMAX>
__global__ void __launch_bounds__(NttTile<TILE_LOG>::THREADS)
k_ntt_pass(const uint4* __restrict__ in_words, uint4* __restrict__ out_words, NttPassArgs a,
           const int32_t* __restrict__ tlo, uint32_t lo_len, int lo_bits, const int32_t* __restrict__ thi, uint32_t hi_len,
           const uint4* __restrict__ next_tw /* or nullptr: w_N^(E(idx)) of the next pass for every output index, canonical words of the internal form */
#ifdef KZG_NTT_STAMPS
           , unsigned long long* __restrict__ stamps /* diagnostic build (tools/ntt_stamps.py): 8 phase sums per workgroup, 100 MHz ticks */
#endif
           ) {
    constexpr int NTT_TILE_LOG = TILE_LOG, NTT_TILE = NttTile<TILE_LOG>::TILE, NTT_PL = NttTile<TILE_LOG>::PL, NTT_THREADS = NttTile<TILE_LOG>::THREADS;
    __shared__ int32_t lds[NL * NTT_PL];
    constexpr int NTT_TW = 1 << (KMAX_T - 1);
    __shared__ int32_t twl[NL * NTT_TW];
    const int K = a.K, log_n = a.log_n;
    const uint32_t N = 1u << log_n, R = 1u << K;
    const int log_c = NTT_TILE_LOG - K;
    const uint32_t C = 1u << log_c;
    const uint32_t n_units = N >> K;
    const uint32_t tid = threadIdx.x;
    const bool last = a.next_K == 0;

    // per-workgroup local twiddles w_R^t = w_N^(t * N/R), t < R/2 (the same for every tile of the pass)
    for (uint32_t t = tid; t < (R >> 1); t += NTT_THREADS) {
        Fr w;
        twiddle(w, tlo, lo_len, lo_bits, thi, hi_len, t << (log_n - K));
#pragma unroll
        for (int j = 0; j < NL; ++j) twl[j * NTT_TW + t] = w.l[j];
    }
    if (R == 1 && tid == 0) {
        Fr w; fe_set_one(w);
#pragma unroll
        for (int j = 0; j < NL; ++j) twl[j * NTT_TW] = w.l[j];
    }

    // words of the first tile
    uint4 pre[NTT_EPT][2];
    {
        const uint32_t tile = blockIdx.x;
#pragma unroll
        for (int k = 0; k < NTT_EPT; ++k) {
            uint32_t uu, j; bool valid;
            const size_t idx = ntt_in_index(a, NTT_TILE_LOG, tile, tid + k * NTT_THREADS, uu, j, valid);
            if (valid) { pre[k][0] = in_words[2 * idx]; pre[k][1] = in_words[2 * idx + 1]; }
            else { pre[k][0] = make_uint4(0, 0, 0, 0); pre[k][1] = make_uint4(0, 0, 0, 0); }
        }
    }
#ifdef KZG_NTT_STAMPS
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = __builtin_amdgcn_s_memrealtime();
#define KZG_NTT_STAMP(i) do { const unsigned long long t_now = __builtin_amdgcn_s_memrealtime(); ph[i] += t_now - t_prev; t_prev = t_now; } while (0)
#else
#define KZG_NTT_STAMP(i) do { } while (0)
#endif
    for (uint32_t tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
        const uint32_t tile_u0 = tile << log_c;
        KZG_NTT_STAMP(0);                                  // (set-up, or the previous tile's trailing barrier)
        // ---- registers -> LDS (bit-reversed rows) ----------------------------------------------
#pragma unroll
        for (int k = 0; k < NTT_EPT; ++k) {
            uint32_t uu, j; bool valid;
            (void)ntt_in_index(a, NTT_TILE_LOG, tile, tid + k * NTT_THREADS, uu, j, valid);
            const uint32_t w32[8] = {pre[k][0].x, pre[k][0].y, pre[k][0].z, pre[k][0].w, pre[k][1].x, pre[k][1].y, pre[k][1].z, pre[k][1].w};
            Fr v;
            fe_unpack(v, w32);                          // canonical, or < 3 m from the previous pass: both fine for the lazy stages
            uint32_t jr = K ? (__brev(j) >> (32 - K)) : 0;
            const uint32_t e = ntt_lds_pos(jr, uu, log_c, K);
#pragma unroll
            for (int l = 0; l < NL; ++l) lds[l * NTT_PL + e] = v.l[l];
        }
        __syncthreads();
        KZG_NTT_STAMP(1);                                  // wait for the tile's words + unpack + LDS fill
        // ---- this tile's inter-pass twiddle words (the STORE phase multiplies by them): with the 1 024-element tile they are issued now and are in
        // flight during the butterfly stages; fetched at the store they are a dependent global load in front of every output product.  Same-box A/B
        // (round 6, tools/time_ntt.py, ms per transform early / at the store): 2^12 0.0364 / 0.0386, 2^16 0.0425 / 0.0453, 2^18 0.0528 / 0.0551,
        // 2^21 0.2646 / 0.2788, 2^22 0.5085 / 0.5323 -- but the 2 048-element tile LOSES (2^19 0.0784 / 0.0768, 2^20 0.1390 / 0.1333), so it keeps the late read.
        // Reading them in front of radix-4 step 1 .. 4 of the 2 048-element tile instead: 0.1319-0.1336 ms at 2^20 against 0.1327 at the store -- noise.
        constexpr bool TW_EARLY = TILE_LOG == NTT_TILE_LOG_SMALL && KMAX_T == NTT_KMAX;    // (the three-pass instantiation keeps 145 VGPRs: three waves per SIMD)
        uint4 twpre[NTT_EPT][2];
        auto fetch_tw = [&]() {
#pragma unroll
            for (int k = 0; k < NTT_EPT; ++k) {
                const uint32_t t = tid + k * NTT_THREADS;
                const uint32_t uu = t & (C - 1), j = t >> log_c;
                const uint32_t u = tile_u0 + uu;
                if (u < n_units) {
                    const size_t idx = (size_t)u + ((size_t)j << (log_n - K));
                    twpre[k][0] = next_tw[2 * idx]; twpre[k][1] = next_tw[2 * idx + 1];
                } else { twpre[k][0] = make_uint4(0, 0, 0, 0); twpre[k][1] = make_uint4(0, 0, 0, 0); }
            }
        };
        if (TW_EARLY && !last && next_tw) fetch_tw();
        // ---- prefetch the next tile's words: in flight during the butterfly stages ---------------
        {
            const uint32_t nt = tile + gridDim.x;
            if (nt < a.n_tiles) {
#pragma unroll
                for (int k = 0; k < NTT_EPT; ++k) {
                    uint32_t uu, j; bool valid;
                    const size_t idx = ntt_in_index(a, NTT_TILE_LOG, nt, tid + k * NTT_THREADS, uu, j, valid);
                    if (valid) { pre[k][0] = in_words[2 * idx]; pre[k][1] = in_words[2 * idx + 1]; }
                    else { pre[k][0] = make_uint4(0, 0, 0, 0); pre[k][1] = make_uint4(0, 0, 0, 0); }
                }
            }
        }
        // ---- K DIT stages in LDS: pairs of stages as radix-4 steps in registers, a last radix-2 stage when K is odd ------
        // Radix-4 step over half-sizes h and 2h on rows i0, i0+h, i0+2h, i0+3h (same butterflies and twiddles as two radix-2
        // stages, so the results are identical): one LDS round trip, one barrier and four limb normalisations per four elements.
        uint32_t log_h = 0;
        KZG_NTT_STAMP(2);                                  // issue of the prefetch
        for (; log_h + 1 < (uint32_t)K; log_h += 2) {
            const uint32_t h = 1u << log_h;
            for (uint32_t gt = tid; gt < (uint32_t)(NTT_TILE / 4); gt += NTT_THREADS) {
                const uint32_t g = gt >> log_c, uu = gt & (C - 1);
                const uint32_t lowb = g & (h - 1);
                const uint32_t i0 = ((g >> log_h) << (log_h + 2)) | lowb;
                const uint32_t e0 = ntt_lds_pos(i0, uu, log_c, K), e1 = ntt_lds_pos(i0 + h, uu, log_c, K),
                               e2 = ntt_lds_pos(i0 + 2 * h, uu, log_c, K), e3 = ntt_lds_pos(i0 + 3 * h, uu, log_c, K);
                const uint32_t t1 = lowb << (K - 1 - log_h);                 // stage h:  w_R^(lowb R / 2h)
                const uint32_t t2 = lowb << (K - 2 - log_h);                 // stage 2h: w_R^(lowb R / 4h), rows i0 / i0+2h
                const uint32_t t3 = (lowb + h) << (K - 2 - log_h);           //           w_R^((lowb+h) R / 4h), rows i0+h / i0+3h
                Fr a0, a1, a2, a3, w, w3;
#pragma unroll
                for (int l = 0; l < NL; ++l) {
                    a0.l[l] = lds[l * NTT_PL + e0];
                    a1.l[l] = lds[l * NTT_PL + e1];
                    a2.l[l] = lds[l * NTT_PL + e2];
                    a3.l[l] = lds[l * NTT_PL + e3];
                }
                Fr p, q;
                if (h == 1) { p = a1; q = a3; }                              // first stage: every twiddle is 1
                else {
#pragma unroll
                    for (int l = 0; l < NL; ++l) w.l[l] = twl[l * NTT_TW + t1];
                    fe_mul2(p, a1, w, q, a3, w);
                }
                Fr b0, b1, b2, b3;
                fe_add(b0, a0, p); fe_sub(b1, a0, p);
                fe_add(b2, a2, q); fe_sub(b3, a2, q);
                Fr u, v;
#pragma unroll
                for (int l = 0; l < NL; ++l) { w.l[l] = twl[l * NTT_TW + t2]; w3.l[l] = twl[l * NTT_TW + t3]; }
                fe_mul2(u, b2, w, v, b3, w3);
                Fr c0, c1, c2, c3;
                fe_add(c0, b0, u); fe_norm(c0);
                fe_sub(c2, b0, u); fe_norm(c2);
                fe_add(c1, b1, v); fe_norm(c1);
                fe_sub(c3, b1, v); fe_norm(c3);
#pragma unroll
                for (int l = 0; l < NL; ++l) {
                    lds[l * NTT_PL + e0] = c0.l[l];
                    lds[l * NTT_PL + e1] = c1.l[l];
                    lds[l * NTT_PL + e2] = c2.l[l];
                    lds[l * NTT_PL + e3] = c3.l[l];
                }
            }
#ifdef KZG_NTT_PROBE_NOBARRIER   // timing probe only (wrong results): what do the barriers of the first four radix-4 steps cost
            if (log_h >= 6) __syncthreads();
#else
            __syncthreads();
#endif
            KZG_NTT_STAMP(3);                              // the radix-4 steps
        }
        if (log_h < (uint32_t)K) {
            const uint32_t h = 1u << log_h;
            for (uint32_t bt = tid; bt < (uint32_t)(NTT_TILE / 2); bt += NTT_THREADS) {
                uint32_t b = bt >> log_c, uu = bt & (C - 1);
                uint32_t lowb = b & (h - 1);
                uint32_t i0 = ((b >> log_h) << (log_h + 1)) | lowb;
                uint32_t e0 = ntt_lds_pos(i0, uu, log_c, K), e1 = ntt_lds_pos(i0 + h, uu, log_c, K);
                uint32_t tw_idx = lowb << (K - 1 - log_h);                 // (b mod h) * R / (2h)
                Fr x0, x, w, t;
#pragma unroll
                for (int l = 0; l < NL; ++l) {
                    x0.l[l] = lds[l * NTT_PL + e0];
                    x.l[l] = lds[l * NTT_PL + e1];
                    w.l[l] = twl[l * NTT_TW + tw_idx];
                }
                if (h == 1) t = x;                         // stage 1: every twiddle is w_R^0 = 1
                else fe_mul(t, x, w);
                Fr y0, y1;
                fe_add(y0, x0, t); fe_norm(y0);
                fe_sub(y1, x0, t); fe_norm(y1);
#pragma unroll
                for (int l = 0; l < NL; ++l) {
                    lds[l * NTT_PL + e0] = y0.l[l];
                    lds[l * NTT_PL + e1] = y1.l[l];
                }
            }
            __syncthreads();
        }
        KZG_NTT_STAMP(4);                                  // the odd radix-2 stage
        // ---- LDS -> global: rows j, C consecutive units each -------------------------------------------------
#pragma unroll
        for (int k = 0; k < NTT_EPT; ++k) {
            const uint32_t t = tid + k * NTT_THREADS;
            const uint32_t uu = t & (C - 1), j = t >> log_c;
            const uint32_t u = tile_u0 + uu;
            if (u >= n_units) continue;
            const size_t idx = (size_t)u + ((size_t)j << (log_n - K));
            const uint32_t e = ntt_lds_pos(j, uu, log_c, K);
            Fr v;
#pragma unroll
            for (int l = 0; l < NL; ++l) v.l[l] = lds[l * NTT_PL + e];
            uint32_t w32[8];
            if (last) {
                if (a.scale_log_n >= 0) {  // a single-pass inverse transform (or no twiddle array): scale by n^-1, which also reduces to (-m, 2m)
                    Fr kk;
#pragma unroll
                    for (int l = 0; l < NL; ++l) kk.l[l] = (int32_t)FrParams::NINV[a.scale_log_n * NL + l];
                    fe_mul(v, v, kk);
                } else {
                    fe_reduce_small(v);    // no factor: ~70 instructions instead of a 206-instruction product by one
                }
                fe_canon(v);
                fe_pack(w32, v);
            } else {
                // the next pass reads x[q' + s' (R' p' + j')] and needs it times w_{n_cur'}^(p' j') = w_N^(p' j' s'): applied here
                const uint32_t tt = (uint32_t)(idx >> a.next_log_s);
                const uint32_t jn = tt & ((1u << a.next_K) - 1), pn = tt >> a.next_K;
                const uint32_t E = (uint32_t)(((unsigned long long)pn * jn) << a.next_log_s) & (N - 1);
                Fr w;
                if (next_tw) {             // one coalesced 32-byte read (same access pattern as the data) instead of a two-level lookup + multiply
                    const uint4 ta = TW_EARLY ? twpre[k][0] : next_tw[2 * idx], tb2 = TW_EARLY ? twpre[k][1] : next_tw[2 * idx + 1];
                    const uint32_t tw32[8] = {ta.x, ta.y, ta.z, ta.w, tb2.x, tb2.y, tb2.z, tb2.w};
                    fe_unpack(w, tw32);
                } else if (E != 0) twiddle(w, tlo, lo_len, lo_bits, thi, hi_len, E);
                else fe_set_one(w);
                fe_mul(v, v, w);           // (-m, 2m)
#pragma unroll
                for (int l = 0; l < NL; ++l) v.l[l] += (int32_t)FrParams::P[l];      // (0, 3m) < 2^256: a non-negative 256-bit word
                fe_norm(v);
                fe_pack(w32, v);
            }
            out_words[2 * idx] = make_uint4(w32[0], w32[1], w32[2], w32[3]);
            out_words[2 * idx + 1] = make_uint4(w32[4], w32[5], w32[6], w32[7]);
        }
        KZG_NTT_STAMP(5);                                  // LDS read + the store-side multiply + pack + global stores (issue)
        __syncthreads();                   // the tile's rows are read before the next tile overwrites them
    }
#ifdef KZG_NTT_STAMPS
    KZG_NTT_STAMP(6);
    if (tid == 0 && stamps) for (int i = 0; i < 8; ++i) stamps[(size_t)blockIdx.x * 8 + i] = ph[i];
#endif
}

__global__ void __launch_bounds__(256)
k_ntt_build_pass_twiddles(uint4* __restrict__ tw, int log_n, int next_K, int next_log_s,
                          const int32_t* __restrict__ tlo, uint32_t lo_len, int lo_bits, const int32_t* __restrict__ thi, uint32_t hi_len,
                          int scale_log_n /* >= 0: every entry times (2^scale_log_n)^-1 (the inverse transform's scaling, folded in) */) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t N = 1u << log_n;
    if (idx >= N) return;
    const uint32_t tt = (uint32_t)(idx >> next_log_s);
    const uint32_t jn = tt & ((1u << next_K) - 1), pn = tt >> next_K;
    const uint32_t E = (uint32_t)(((unsigned long long)pn * jn) << next_log_s) & (N - 1);
    Fr w;
    if (E != 0) twiddle(w, tlo, lo_len, lo_bits, thi, hi_len, E);
    else fe_set_one(w);
    if (scale_log_n >= 0) {
        Fr kk;
#pragma unroll
        for (int l = 0; l < NL; ++l) kk.l[l] = (int32_t)FrParams::NINV[scale_log_n * NL + l];
        fe_norm(w);
        fe_mul(w, w, kk);
    }
    fe_canon(w);
    uint32_t o[8];
    fe_pack(o, w);
    tw[2 * idx] = make_uint4(o[0], o[1], o[2], o[3]);
    tw[2 * idx + 1] = make_uint4(o[4], o[5], o[6], o[7]);
}

// ---- host side ------------------------------------------------------------------------------------
struct TableKey { int dev; int log_n; int inverse; bool operator<(const TableKey& o) const { return dev != o.dev ? dev < o.dev : (log_n != o.log_n ? log_n < o.log_n : inverse < o.inverse); } };
static std::map<TableKey, NttTables> g_tables;
static std::mutex g_tables_mu;

int32_t ntt_get_tables(kzg_ctx* ctx, int log_n, bool inverse, NttTables* out) {
    std::lock_guard<std::mutex> lk(g_tables_mu);
    TableKey key{ctx->device, log_n, inverse ? 1 : 0};
    auto it = g_tables.find(key);
    if (it != g_tables.end()) { *out = it->second; return KZG_OK; }
    NttTables t;
    int lo_bits = log_n < NTT_LO_BITS ? log_n : NTT_LO_BITS;
    t.lo_len = 1u << lo_bits;
    t.lo_bits = lo_bits;
    t.hi_len = 1u << (log_n - lo_bits);
    KZG_HIP_TRY(ctx, hipMalloc(&t.lo, (size_t)t.lo_len * NL * 4));
    KZG_HIP_TRY(ctx, hipMalloc(&t.hi, (size_t)t.hi_len * NL * 4));
    hipLaunchKernelGGL(k_ntt_build_table, dim3((t.lo_len + 255) / 256), dim3(256), 0, ctx->stream, t.lo, t.lo_len, log_n, inverse ? 1 : 0, 1u);
    hipLaunchKernelGGL(k_ntt_build_table, dim3((t.hi_len + 255) / 256), dim3(256), 0, ctx->stream, t.hi, t.hi_len, log_n, inverse ? 1 : 0, t.lo_len);
    KZG_HIP_TRY(ctx, hipGetLastError());
    KZG_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));     // built once; afterwards read from any stream of the device
    g_tables[key] = t;
    *out = t;
    return KZG_OK;
}

// HBM capacity spent to remove work (like the MSM window tables): the inter-pass twiddle of every element as one 32-byte word,
// 32 MiB per pass boundary at 2^20; kept per (device, log n, direction, boundary) for transforms of up to 2^22 elements.
constexpr int NTT_FULL_TW_MAX_LOG = 22;
struct PassTwKey { int dev, log_n, inverse, next_K, next_log_s, scaled; bool operator<(const PassTwKey& o) const { return std::tie(dev, log_n, inverse, next_K, next_log_s, scaled) < std::tie(o.dev, o.log_n, o.inverse, o.next_K, o.next_log_s, o.scaled); } };
static std::map<PassTwKey, uint4*> g_pass_tw;
static int32_t ntt_get_pass_twiddles(kzg_ctx* ctx, int log_n, bool inverse, int next_K, int next_log_s, const NttTables& tb, int lo_bits, const uint4** out,
                                     bool scaled) {
    std::lock_guard<std::mutex> lk(g_tables_mu);
    PassTwKey key{ctx->device, log_n, inverse ? 1 : 0, next_K, next_log_s, scaled ? 1 : 0};
    auto it = g_pass_tw.find(key);
    if (it != g_pass_tw.end()) { *out = it->second; return KZG_OK; }
    const size_t n = (size_t)1 << log_n;
    uint4* p = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), n * 32);
    if (e != hipSuccess) { (void)hipGetLastError(); *out = nullptr; return KZG_OK; }     // no memory: the kernel looks the twiddles up itself
    hipLaunchKernelGGL(k_ntt_build_pass_twiddles, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, p, log_n, next_K, next_log_s,
                       tb.lo, tb.lo_len, lo_bits, tb.hi, tb.hi_len, scaled ? log_n : -1);
    KZG_HIP_TRY(ctx, hipGetLastError());
    KZG_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));     // built once; afterwards read from any stream of the device
    g_pass_tw[key] = p;
    *out = p;
    return KZG_OK;
}

// the last context of device `dev` is gone: free what the transforms cached for it (twiddle tables, per-element twiddle arrays)
void ntt_release_device_caches(int dev) {
    std::lock_guard<std::mutex> lk(g_tables_mu);
    for (auto it = g_tables.begin(); it != g_tables.end();) {
        if (it->first.dev == dev) { (void)hipFree(it->second.lo); (void)hipFree(it->second.hi); it = g_tables.erase(it); } else ++it;
    }
    for (auto it = g_pass_tw.begin(); it != g_pass_tw.end();) {
        if (it->first.dev == dev) { (void)hipFree(it->second); it = g_pass_tw.erase(it); } else ++it;
    }
}

#ifdef KZG_NTT_STAMPS
static unsigned long long* g_ntt_stamps = nullptr;
}  // namespace kzg
extern "C" int32_t kzg_debug_ntt_stamps(unsigned long long* out /* 4 x 1024 x 8 */) {
    if (!kzg::g_ntt_stamps) return -1;
    return hipMemcpy(out, kzg::g_ntt_stamps, 4 * 1024 * 8 * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -3;
}
namespace kzg {
#endif
int32_t ntt_run(kzg_ctx* ctx, void* d_data, size_t n, bool inverse, hipStream_t st, NttWorkspace* ws) {
    if (!st) st = ctx->stream;
    if (!ws) ws = &ctx->ntt;
    if (n == 0 || (n & (n - 1)) != 0) return KZG_ERR_NOT_POWER_OF_TWO;
    if (n > ((size_t)1 << 28)) return KZG_ERR_DOMAIN;
    if (n == 1) return KZG_OK;
    RoctxRange range(inverse ? "kzg:fr_intt" : "kzg:fr_ntt");
    int log_n = 0;
    while (((size_t)1 << log_n) < n) ++log_n;
    NttTables tb;
    int32_t rc = ntt_get_tables(ctx, log_n, inverse, &tb);
    if (rc != KZG_OK) return rc;
    int lo_bits = log_n < NTT_LO_BITS ? log_n : NTT_LO_BITS;

    int P = (log_n + NTT_KMAX - 1) / NTT_KMAX;
    // (2^20 as THREE passes of 7 + 7 + 6 bits on slim workgroups: 0.1305 / 0.1322 ms against 0.1333 for 10 + 10 -- 1.5 % for 60 % more HBM traffic: not taken;
    //  2^15 .. 2^19 lose 2 .. 13 % that way)
    int Ks[4];
    for (int pi = 0; pi < P; ++pi) Ks[pi] = log_n / P + (pi < log_n % P ? 1 : 0);
    // buffers: data -> A -> (B ->) data; a single pass works in place (one tile holds the whole transform)
    if (P > 1) KZG_HIP_TRY(ctx, ws->data.reserve(n * 32));
    if (P > 2) KZG_HIP_TRY(ctx, ws->tmp.reserve(n * 32));
    uint4* bufs[2] = {ws->data.as<uint4>(), ws->tmp.as<uint4>()};
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
    const int tile_env = opts().ntt_tile_log;                       // KZG_NTT_TILE_LOG = 10 / 11: one tile size at every transform size (tests cover both kernels everywhere)
    int kmax = 0;
    for (int pi = 0; pi < P; ++pi) kmax = std::max(kmax, log_n / P + (pi < log_n % P ? 1 : 0));
    // Round 6: the radix bits of the widest pass decide the per-workgroup twiddle table.  With K <= 9 a 1 024-element workgroup needs 41.5 / 43.8 / 48.4 KB of LDS
    // instead of 57.6, THREE fit a CU and at 145 VGPRs run three waves per SIMD -- the instantiations k_ntt_pass<10, 7 / 8 / 9>; fill and drain of one tile then
    // overlap the butterfly stages of two others.  Same-box A/B against the KMAX = 10 instantiation (two per CU, early twiddle read), ms per transform
    // (tools/time_ntt.py): 2^8 0.0267 -> 0.0254, 2^12 0.0385 -> 0.0372, 2^16 0.0450 -> 0.0430, 2^18 0.0552 -> 0.0531, 2^21 0.2723 -> 0.2423, 2^22 0.5213 -> 0.4880,
    // 2^23 1.168 -> 1.082, 2^24 2.430 -> 2.268, 2^25 5.24 (2 048-element tile) -> 4.95.  2^10, 2^19, 2^20 and 2^26 .. 2^30 have a 10-bit pass and stay as they were.
    const bool small_tile = tile_env == NTT_TILE_LOG_SMALL || (tile_env != NTT_TILE_LOG_BIG && ntt_small_tile_pays(log_n));
    const int slim = small_tile && kmax <= 9 ? std::max(kmax, 7) : 0;
    const int tile_log = small_tile ? NTT_TILE_LOG_SMALL : NTT_TILE_LOG_BIG;
    int log_ncur = 0;
    bool scale_folded = false;      // the inverse transform's 1 / n went into the twiddle array of the last pass boundary
    for (int pi = 0; pi < P; ++pi) {
        NttPassArgs a;
        a.log_n = log_n;
        a.K = Ks[pi];
        log_ncur += a.K;
        a.log_s = log_n - log_ncur;
        const bool last = pi == P - 1;
        a.next_K = last ? 0 : Ks[pi + 1];
        a.next_log_s = last ? 0 : log_n - (log_ncur + Ks[pi + 1]);
        a.scale_log_n = (last && inverse && !scale_folded) ? log_n : -1;
        const uint32_t n_units = (uint32_t)(n >> a.K);
        const uint32_t C = 1u << (tile_log - a.K);
        a.n_tiles = (n_units + C - 1) / C;
        const uint4* src = pi == 0 ? reinterpret_cast<const uint4*>(d_data) : bufs[(pi - 1) & 1];
        uint4* dst = last ? reinterpret_cast<uint4*>(d_data) : bufs[pi & 1];
        // big tile: one workgroup per CU (97 KB of LDS); small tile: two (58 KB each); a workgroup walks its tiles with the next one's words prefetched
        const uint32_t grid = std::min<uint32_t>(a.n_tiles, (uint32_t)cus * (slim ? 3u : small_tile ? 2u : 1u));
        const uint4* next_tw = nullptr;
        if (!last && log_n <= NTT_FULL_TW_MAX_LOG) {
            const bool fold = inverse && pi == P - 2;          // the boundary in front of the last pass carries the scaling
            rc = ntt_get_pass_twiddles(ctx, log_n, inverse, a.next_K, a.next_log_s, tb, lo_bits, &next_tw, fold);
            if (rc != KZG_OK) return rc;
            scale_folded = fold && next_tw != nullptr;
        }
#ifdef KZG_NTT_STAMPS
        static unsigned long long* d_stamps = nullptr;     // [pass][workgroup][8]; read back by kzg_debug_ntt_stamps
        if (!d_stamps) KZG_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&d_stamps), 4 * 1024 * 8 * 8));
        g_ntt_stamps = d_stamps;
        if (small_tile)
            hipLaunchKernelGGL(k_ntt_pass<NTT_TILE_LOG_SMALL>, dim3(grid), dim3(NttTile<NTT_TILE_LOG_SMALL>::THREADS), 0, st, src, dst, a, tb.lo, tb.lo_len, lo_bits, tb.hi, tb.hi_len, next_tw, d_stamps + (size_t)pi * 1024 * 8);
        else
            hipLaunchKernelGGL(k_ntt_pass<NTT_TILE_LOG_BIG>, dim3(grid), dim3(NttTile<NTT_TILE_LOG_BIG>::THREADS), 0, st, src, dst, a, tb.lo, tb.lo_len, lo_bits, tb.hi, tb.hi_len, next_tw, d_stamps + (size_t)pi * 1024 * 8);
#else
        if (slim == 7)
            hipLaunchKernelGGL((k_ntt_pass<NTT_TILE_LOG_SMALL, 7>), dim3(grid), dim3(NttTile<NTT_TILE_LOG_SMALL>::THREADS), 0, st, src, dst, a, tb.lo, tb.lo_len, lo_bits, tb.hi, tb.hi_len, next_tw);
        else if (slim == 8)
            hipLaunchKernelGGL((k_ntt_pass<NTT_TILE_LOG_SMALL, 8>), dim3(grid), dim3(NttTile<NTT_TILE_LOG_SMALL>::THREADS), 0, st, src, dst, a, tb.lo, tb.lo_len, lo_bits, tb.hi, tb.hi_len, next_tw);
        else if (slim == 9)
            hipLaunchKernelGGL((k_ntt_pass<NTT_TILE_LOG_SMALL, 9>), dim3(grid), dim3(NttTile<NTT_TILE_LOG_SMALL>::THREADS), 0, st, src, dst, a, tb.lo, tb.lo_len, lo_bits, tb.hi, tb.hi_len, next_tw);
        else if (small_tile)
            hipLaunchKernelGGL(k_ntt_pass<NTT_TILE_LOG_SMALL>, dim3(grid), dim3(NttTile<NTT_TILE_LOG_SMALL>::THREADS), 0, st, src, dst, a, tb.lo, tb.lo_len, lo_bits, tb.hi, tb.hi_len, next_tw);
        else
            hipLaunchKernelGGL(k_ntt_pass<NTT_TILE_LOG_BIG>, dim3(grid), dim3(NttTile<NTT_TILE_LOG_BIG>::THREADS), 0, st, src, dst, a, tb.lo, tb.lo_len, lo_bits, tb.hi, tb.hi_len, next_tw);
#endif
    }
    KZG_HIP_TRY(ctx, hipGetLastError());
    return KZG_OK;
}

}  // namespace kzg
