// ntt.hip — radix-2^K Stockham NTT / inverse NTT over BN254 Fr, LDS-tiled.
//
// Replaces `GeneralEvaluationDomain::<Fr>::new(n).fft(..)` / `.ifft(..)` at
// primitives/src/polynomial.rs:131-135 and :242-246: natural order in and out on the domain
// {w^i}, w = 5^((r-1)/n) (= PRIMITIVE_ROOTS_OF_UNITY[log2 n], primitives/src/consts.rs:22-52); the
// inverse uses w^-1 and scales by n^-1.
//
// Algorithm (decimation in time, Stockham autosort, P = ceil(log n / 7) passes):
//   pass with radix R = 2^K brings the sub-transform length from n_cur/R to n_cur (stride s = N / n_cur):
//     y[u + j * N/R] = sum_j'  w_{n_cur}^(p j') x[q + s (R p + j')] * w_R^(j j'),   u = q + s p
//   Each workgroup owns a tile of C = 2048 / R consecutive units u: it gathers the R x C elements
//   (contiguous runs in global memory), applies the inter-pass twiddles, runs the K radix-2 butterfly
//   stages in LDS (9 limb planes, padded rows; per-tile twiddle table w_R^t in LDS), and scatters rows
//   back.  Data stay in the wire residue class (a * 2^256) throughout: the transform is linear and the
//   twiddles are in internal Montgomery form, so no domain conversion is needed.  Intermediate passes
//   keep elements as 9 signed 29-bit limb planes (lazy, |v| < 16 m); the last pass reduces, scales and
//   packs canonical 256-bit words.
//   Algorithmic traffic: 64 B per element (32 B read + 32 B written); this implementation moves
//   P x (read + write) with 36 B planes in between.
#include "engine.h"
#include "field29.h"

#include <map>
#include <mutex>
#include <tuple>

namespace kzg {

constexpr int NTT_TILE_LOG = 11;
constexpr int NTT_TILE = 1 << NTT_TILE_LOG;       // elements per workgroup tile
constexpr int NTT_KMAX = 7;
constexpr int NTT_PL = NTT_TILE + (1 << NTT_KMAX); // plane length with one pad element per row
#ifndef KZG_NTT_THREADS
#define KZG_NTT_THREADS 512
#endif
constexpr int NTT_THREADS = KZG_NTT_THREADS;
constexpr int NTT_LO_BITS = 10;
constexpr int NTT_TW = 96;                         // LDS twiddle entries: R/2 local + R row entries must fit (2 tiles per CU)

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

// last-pass separable twiddles: sep[uu * R + j] = w_N^(uu * j), uu < C = 2048 / R, j < R = 2^K (tile-independent)
__global__ void __launch_bounds__(256)
k_ntt_build_sep(int32_t* __restrict__ planes, int log_n, int inverse, int K) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (uint32_t)NTT_TILE) return;
    Fr w;
    fr_pow_root(w, log_n, inverse != 0, (t >> K) * (t & ((1u << K) - 1)));
#pragma unroll
    for (int j = 0; j < NL; ++j) planes[(size_t)j * NTT_TILE + t] = w.l[j];
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

// One Stockham pass.  in_wire / out_wire (canonical 256-bit words) are used by the first / last pass,
// the planes otherwise.
__global__ void __launch_bounds__(NTT_THREADS)
k_ntt_pass(const uint4* __restrict__ in_wire, const int32_t* __restrict__ in_planes,
           uint4* __restrict__ out_wire, int32_t* __restrict__ out_planes,
           int log_n, int K, int log_s,
           const int32_t* __restrict__ tlo, uint32_t lo_len, int lo_bits, const int32_t* __restrict__ thi, uint32_t hi_len,
           int first, int last, int scale_log_n /* >= 0: multiply by (2^scale_log_n)^-1 at the end */,
           const int32_t* __restrict__ sep /* last pass (s = 1): w_N^(uu j) table, or nullptr */) {
    __shared__ int32_t lds[NL * NTT_PL];
    // twiddle scratch: entries [0, R/2) = per-tile w_R^t; entries [64, 64 + R) = inter-pass row (only when R <= 32...64 fits)
    __shared__ int32_t twl[NL * NTT_TW];

    const uint32_t N = 1u << log_n;
    const uint32_t R = 1u << K;
    const int log_c = NTT_TILE_LOG - K;
    const uint32_t C = 1u << log_c, Cp = C + 1;
    const uint32_t s = 1u << log_s;
    const uint32_t n_units = N >> K;
    const uint32_t tid = threadIdx.x;
    const uint32_t tile_u0 = blockIdx.x << log_c;

    // per-tile local twiddles w_R^t = w_N^(t * N/R), t < R/2
    if (tid < (R >> 1)) {
        Fr w;
        twiddle(w, tlo, lo_len, lo_bits, thi, hi_len, tid << (log_n - K));
        if (R == 1) fe_set_one(w);
#pragma unroll
        for (int j = 0; j < NL; ++j) twl[j * NTT_TW + tid] = w.l[j];
    }

    // every unit of the tile has the same p = tile_u0 >> log_s, and the row fits behind the local twiddles
    const bool row_tw = !first && s >= C && (R >> 1) + R <= (uint32_t)NTT_TW;
    if (row_tw && tid < R) {
        const uint32_t p = tile_u0 >> log_s;
        Fr w;
        twiddle(w, tlo, lo_len, lo_bits, thi, hi_len, (p * tid) << log_s);
#pragma unroll
        for (int j = 0; j < NL; ++j) twl[j * NTT_TW + (R >> 1) + tid] = w.l[j];
    }
    if (row_tw) __syncthreads();

    // Last pass (s = 1): every unit of the tile has its own p = tile_u0 + uu, so the inter-pass twiddle w_N^(p j) differs per
    // element.  A two-level table lookup per element was 18 scattered 4-byte loads (the slowest part of the pass: +29 us at
    // 2^20).  w_N^(p j) = w_N^(tile_u0 j) * w_N^(uu j): the first factor depends only on j, which is FIXED per thread here
    // (j = t mod R, NTT_THREADS a multiple of R) -> one lookup per thread, kept in registers; the second is a tile-independent
    // 2048-entry table read with consecutive lanes on consecutive entries.
    const bool sep_tw = !first && s == 1 && sep != nullptr;
    Fr tw_a;
    fe_set_one(tw_a);
    if (sep_tw && tile_u0 != 0) twiddle(tw_a, tlo, lo_len, lo_bits, thi, hi_len, tile_u0 * (tid & (R - 1)));

    // ---- gather -------------------------------------------------------------------------------
    for (uint32_t t = tid; t < (uint32_t)NTT_TILE; t += NTT_THREADS) {
        uint32_t uu, j;
        if (s >= C) { uu = t & (C - 1); j = t >> log_c; }
        else { uint32_t q = t & (s - 1); j = (t >> log_s) & (R - 1); uu = ((t >> (log_s + K)) << log_s) | q; }
        uint32_t u = tile_u0 + uu;
        Fr v;
        fe_set_zero(v);
        if (u < n_units) {
            uint32_t q = u & (s - 1), p = u >> log_s;
            size_t idx = (size_t)q + ((size_t)(R * p + j) << log_s);
            if (first) {
                uint4 a = in_wire[2 * idx], b = in_wire[2 * idx + 1];
                uint32_t w32[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
                fe_unpack(v, w32);
            } else {
                load_planes(v, in_planes, N, idx);
                if (sep_tw) {
                    if (p != 0 && j != 0) {
                        Fr w;
                        load_planes(w, sep, NTT_TILE, (size_t)uu * R + j);
                        if (tile_u0 != 0) fe_mul(w, w, tw_a);
                        fe_mul(v, v, w);
                    }
                } else if (p != 0 && j != 0) {
                    Fr w;
                    if (row_tw) {
#pragma unroll
                        for (int l = 0; l < NL; ++l) w.l[l] = twl[l * NTT_TW + (R >> 1) + j];
                    } else {
                        twiddle(w, tlo, lo_len, lo_bits, thi, hi_len, (p * j) << log_s);   // w_{n_cur}^(p j) = w_N^(p j s)
                    }
                    fe_mul(v, v, w);                                                  // |v| < 16 m, |w| < 2 m
                }
            }
        }
        uint32_t jr = __brev(j) >> (32 - K);
        if (K == 0) jr = 0;
        uint32_t e = jr * Cp + uu;
#pragma unroll
        for (int l = 0; l < NL; ++l) lds[l * NTT_PL + e] = v.l[l];
    }
    __syncthreads();

    // ---- K DIT stages in LDS: pairs of stages as radix-4 steps in registers, a last radix-2 stage when K is odd ---------
    // Radix-4 step over half-sizes h and 2h on rows i0, i0+h, i0+2h, i0+3h (same butterflies and twiddles as two radix-2 stages,
    // so the results are identical): one LDS round trip, one barrier and four limb normalisations per four elements instead of
    // two, eight and eight.  The intermediate values are multiplied un-normalised (limbs < 2^30 against twiddle limbs < 2^29).
    uint32_t log_h = 0;
    for (; log_h + 1 < (uint32_t)K; log_h += 2) {
        const uint32_t h = 1u << log_h;
        for (uint32_t gt = tid; gt < (uint32_t)(NTT_TILE / 4); gt += NTT_THREADS) {
            const uint32_t g = gt >> log_c, uu = gt & (C - 1);
            const uint32_t lowb = g & (h - 1);
            const uint32_t i0 = ((g >> log_h) << (log_h + 2)) | lowb;
            const uint32_t e0 = i0 * Cp + uu, e1 = e0 + h * Cp, e2 = e1 + h * Cp, e3 = e2 + h * Cp;
            const uint32_t t1 = lowb << (K - 1 - log_h);                 // stage h:  w_R^(lowb R / 2h)
            const uint32_t t2 = lowb << (K - 2 - log_h);                 // stage 2h: w_R^(lowb R / 4h), rows i0 / i0+2h
            const uint32_t t3 = (lowb + h) << (K - 2 - log_h);           //           w_R^((lowb+h) R / 4h), rows i0+h / i0+3h
            Fr a0, a1, a2, a3, w;
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
                fe_mul(p, a1, w);
                fe_mul(q, a3, w);
            }
            Fr b0, b1, b2, b3;
            fe_add(b0, a0, p); fe_sub(b1, a0, p);
            fe_add(b2, a2, q); fe_sub(b3, a2, q);
            Fr u, v;
#pragma unroll
            for (int l = 0; l < NL; ++l) w.l[l] = twl[l * NTT_TW + t2];
            fe_mul(u, b2, w);
#pragma unroll
            for (int l = 0; l < NL; ++l) w.l[l] = twl[l * NTT_TW + t3];
            fe_mul(v, b3, w);
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
        __syncthreads();
    }
    if (log_h < (uint32_t)K) {
        const uint32_t h = 1u << log_h;
        for (uint32_t bt = tid; bt < (uint32_t)(NTT_TILE / 2); bt += NTT_THREADS) {
            uint32_t b = bt >> log_c, uu = bt & (C - 1);
            uint32_t lowb = b & (h - 1);
            uint32_t i0 = ((b >> log_h) << (log_h + 1)) | lowb;
            uint32_t e0 = i0 * Cp + uu, e1 = (i0 + h) * Cp + uu;
            uint32_t tw_idx = lowb << (K - 1 - log_h);                 // (b mod h) * R / (2h)
            Fr a, x, w, t;
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                a.l[l] = lds[l * NTT_PL + e0];
                x.l[l] = lds[l * NTT_PL + e1];
                w.l[l] = twl[l * NTT_TW + tw_idx];
            }
            if (h == 1) t = x;                         // stage 1: every twiddle is w_R^0 = 1
            else fe_mul(t, x, w);
            Fr y0, y1;
            fe_add(y0, a, t); fe_norm(y0);
            fe_sub(y1, a, t); fe_norm(y1);
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                lds[l * NTT_PL + e0] = y0.l[l];
                lds[l * NTT_PL + e1] = y1.l[l];
            }
        }
        __syncthreads();
    }

    // ---- scatter ------------------------------------------------------------------------------------
    for (uint32_t t = tid; t < (uint32_t)NTT_TILE; t += NTT_THREADS) {
        uint32_t uu = t & (C - 1), j = t >> log_c;
        uint32_t u = tile_u0 + uu;
        if (u >= n_units) continue;
        size_t idx = (size_t)u + ((size_t)j << (log_n - K));
        uint32_t e = j * Cp + uu;
        Fr v;
#pragma unroll
        for (int l = 0; l < NL; ++l) v.l[l] = lds[l * NTT_PL + e];
        if (last) {
            Fr k;
            if (scale_log_n >= 0) {
#pragma unroll
                for (int l = 0; l < NL; ++l) k.l[l] = (int32_t)FrParams::NINV[scale_log_n * NL + l];
            } else {
                fe_set_one(k);
            }
            fe_mul(v, v, k);          // reduce to (-m, 2m) (and scale by n^-1 for the inverse transform)
            fe_canon(v);
            uint32_t w32[8];
            fe_pack(w32, v);
            out_wire[2 * idx] = make_uint4(w32[0], w32[1], w32[2], w32[3]);
            out_wire[2 * idx + 1] = make_uint4(w32[4], w32[5], w32[6], w32[7]);
        } else {
#pragma unroll
            for (int l = 0; l < NL; ++l) out_planes[(size_t)l * N + idx] = v.l[l];
        }
    }
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

struct SepKey { int dev, log_n, inverse, K; bool operator<(const SepKey& o) const { return std::tie(dev, log_n, inverse, K) < std::tie(o.dev, o.log_n, o.inverse, o.K); } };
static std::map<SepKey, int32_t*> g_sep;
static int32_t ntt_get_sep(kzg_ctx* ctx, int log_n, bool inverse, int K, const int32_t** out) {
    std::lock_guard<std::mutex> lk(g_tables_mu);
    SepKey key{ctx->device, log_n, inverse ? 1 : 0, K};
    auto it = g_sep.find(key);
    if (it != g_sep.end()) { *out = it->second; return KZG_OK; }
    int32_t* p = nullptr;
    KZG_HIP_TRY(ctx, hipMalloc(&p, (size_t)NTT_TILE * NL * 4));
    hipLaunchKernelGGL(k_ntt_build_sep, dim3(NTT_TILE / 256), dim3(256), 0, ctx->stream, p, log_n, inverse ? 1 : 0, K);
    KZG_HIP_TRY(ctx, hipGetLastError());
    KZG_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));     // built once; afterwards read from any stream of the device
    g_sep[key] = p;
    *out = p;
    return KZG_OK;
}

int32_t ntt_run(kzg_ctx* ctx, void* d_data, size_t n, bool inverse, hipStream_t st, NttWorkspace* ws) {
    if (!st) st = ctx->stream;
    if (!ws) ws = &ctx->ntt;
    if (n == 0 || (n & (n - 1)) != 0) return KZG_ERR_NOT_POWER_OF_TWO;
    if (n > ((size_t)1 << 28)) return KZG_ERR_DOMAIN;
    if (n == 1) return KZG_OK;
    int log_n = 0;
    while (((size_t)1 << log_n) < n) ++log_n;
    NttTables tb;
    int32_t rc = ntt_get_tables(ctx, log_n, inverse, &tb);
    if (rc != KZG_OK) return rc;
    int lo_bits = log_n < NTT_LO_BITS ? log_n : NTT_LO_BITS;

    int P = (log_n + NTT_KMAX - 1) / NTT_KMAX;
    int base = log_n / P, extra = log_n % P;
    if (P > 1) {
        KZG_HIP_TRY(ctx, ws->data.reserve(n * NL * 4));
        if (P > 2) KZG_HIP_TRY(ctx, ws->tmp.reserve(n * NL * 4));
    }
    int32_t* bufs[2] = {ws->data.as<int32_t>(), ws->tmp.as<int32_t>()};
    int Ks[8];
    for (int pi = 0; pi < P; ++pi) Ks[pi] = base;
    {
        int order[8], no = 0;
        order[no++] = 0;
        if (P > 1) order[no++] = P - 1;
        for (int pi = 1; pi + 1 < P; ++pi) order[no++] = pi;
        for (int e = 0; e < extra; ++e) Ks[order[e]] += 1;
    }
    const int32_t* sep = nullptr;
    if (P > 1) { rc = ntt_get_sep(ctx, log_n, inverse, Ks[P - 1], &sep); if (rc != KZG_OK) return rc; }
    int log_ncur = 0;
    for (int pi = 0; pi < P; ++pi) {
        int K = Ks[pi];
        log_ncur += K;
        int log_s = log_n - log_ncur;
        bool first = pi == 0, last = pi == P - 1;
        const int32_t* in_planes = first ? nullptr : bufs[(pi - 1) & 1];
        int32_t* out_planes = last ? nullptr : bufs[pi & 1];
        uint32_t n_units = (uint32_t)(n >> K);
        uint32_t C = 1u << (NTT_TILE_LOG - K);
        uint32_t tiles = (n_units + C - 1) / C;
        hipLaunchKernelGGL(k_ntt_pass, dim3(tiles), dim3(NTT_THREADS), 0, st,
                           reinterpret_cast<const uint4*>(d_data), in_planes, reinterpret_cast<uint4*>(d_data), out_planes,
                           log_n, K, log_s, tb.lo, tb.lo_len, lo_bits, tb.hi, tb.hi_len,
                           first ? 1 : 0, last ? 1 : 0, (last && inverse) ? log_n : -1, (last && !first) ? sep : nullptr);
    }
    KZG_HIP_TRY(ctx, hipGetLastError());
    return KZG_OK;
}

}  // namespace kzg
