// g1fft.hip — `KZG::g1_ifft` (prover/src/kzg.rs:263-285): the inverse FFT whose DATA are G1 points and whose
// twiddles are Fr scalars,  L_i = n^-1 * sum_j w^(-ij) P_j  (natural order) — the Lagrange-basis SRS.
// Reference: `GeneralEvaluationDomain::<Fr>::new(n).ifft(&[G1Projective])` + n `into_affine()` inversions, recomputed on
// every `commit_eval_form` (kzg.rs:96-98).  The commit / proof path of this library never needs it
// (commit_eval_form == MSM(srs, IFFT(evals))); it is public API with its own bench
// (prover/benches/bench_g1_ifft.rs) and golden vector (prover/tests/test-files/lagrangeG1SRS.txt), and the library can keep
// its result on the device as an SRS of its own (kzg_srs_lagrange) so that eval-form commitments become one MSM.
//
// What bounds it on a GPU is DEPTH: every FFT stage multiplies points by full-width scalars, i.e. a chain of ~254 dependent
// doublings (~4 us each on a lone wave), and radix 2 has log2 n such stages.  Work is spent to cut the depth:
//   * small n (n * R <= 65536 lanes): Stockham stages of radix R = 2^k, k <= 5, each output as a DIRECT sum of its R inputs,
//       y[u + j N/R] = sum_j' [w^-(s p j' + (N/R) j j')] x[q + s (R p + j')],   u = q + s p,
//     one lane per (output, term): one scalar multiplication deep per stage plus a k-step shuffle tree, ceil(log2 n / 5) stages
//     instead of log2 n (n = 2048: 3 instead of 11), for (2^k - 1)/k times the multiplications;
//   * large n: radix-2 butterflies (A, B) -> (A + [w]B, A - [w]B), one multiplication per two outputs (work bound);
//   * the scaling by n^-1 is folded into the scalars of the last stage (no chain of its own);
//   * Jacobian -> affine by Montgomery's trick (one inversion per lane for 16 points) when there are many points.
#include "engine.h"
#include "curve_pair.h"
#include "curve_quad.h"
#include "fe_invert.h"
#include "naf.h"
#include "host_curve.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <tuple>
#include <mutex>
#include <vector>

namespace kzg {

// the ONE inversion behind every batched affine conversion: Bernstein-Yang division steps (fe_invert.h: ~20 us on a lone lane) instead of
// a^(m-2) (381 dependent products, ~175 us).  -DKZG_INVERT_FERMAT restores the exponentiation (A/B).
template <class F>
__device__ __forceinline__ void fe_inverse_fermat(Fe<F>& out, const Fe<F>& a) {
#if !defined(KZG_INVERT_FERMAT)
    fe_inverse_safegcd(out, a);
#else
    Fe<F> acc, base = a;
    fe_set_one(acc);
    uint32_t e[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = F::P32[j];
    e[0] -= 2u;
    for (int w = 0; w < 8; ++w) {
        uint32_t bits = e[w];
        for (int b = 0; b < 32; ++b) {
            if (w == 7 && b >= 30) break;
            if (bits & 1u) fe_mul(acc, acc, base);
            fe_sqr(base, base);
            bits >>= 1;
        }
    }
    out = acc;
#endif
}

// ---- GLV: k P = k1 P + k2 phi(P), phi(x, y) = (beta x, y) = [lambda] P on BN254 G1, |k1|, |k2| < 2^127 ---------------------------
// beta, lambda: the cube roots of unity of Fq / Fr with phi(G) = [lambda] G (checked with big integers: tools note in DESIGN.md 9);
// lattice basis of {(x, y): x + y lambda = 0 mod r} from the extended Euclid on (r, lambda):
//   (a1, b1) = (9931322734385697763, -147946756881789319000765030803803410728), (a2, b2) = (147946756881789319010696353538189108491, a1),
// a1 b2 - a2 b1 = r.  c1 = floor(k g1 / 2^256), c2 = floor(k g2 / 2^256) with g1 = round(2^256 b2 / r), g2 = round(2^256 (-b1) / r);
// k1 = k - c1 a1 - c2 a2, k2 = c1 |b1| - c2 b2  (2 10^5 random k and the edge values: both below 2^127 in magnitude).
// Little-endian 32-bit words.
__device__ const uint32_t GLV_G1[3] = {0xc7e0b3d7u, 0xd91d232eu, 0x00000002u};
__device__ const uint32_t GLV_G2[5] = {0x391eb18eu, 0x7a7bd9d4u, 0xa773d2cfu, 0x4ccef014u, 0x00000002u};
__device__ const uint32_t GLV_A1[2] = {0x94d213e3u, 0x89d32568u};                                   // = b2
__device__ const uint32_t GLV_B1M[4] = {0x7d4f1128u, 0x8211bbebu, 0xeeb859fcu, 0x6f4d8248u};        // -b1
__device__ const uint32_t GLV_A2[4] = {0x1221250bu, 0x0be4e154u, 0xeeb859fdu, 0x6f4d8248u};
__device__ const uint32_t GLV_BETA[8] = {0x77fffffeu, 0x57634731u, 0xacdb5c4fu, 0xd4f263f1u, 0xa0d48bacu, 0x59e26bceu, 0u, 0u};   // plain integer

// out[0 .. na+nb) = a * b (schoolbook, 32-bit words)
template <int NA, int NB>
__device__ __forceinline__ void glv_mul_words(uint32_t* out, const uint32_t* a, const uint32_t* b) {
#pragma unroll
    for (int i = 0; i < NA + NB; ++i) out[i] = 0;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        uint64_t carry = 0;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const uint64_t t = (uint64_t)a[i] * b[j] + out[i + j] + carry;
            out[i + j] = (uint32_t)t;
            carry = t >> 32;
        }
        out[i + NB] = (uint32_t)carry;
    }
}
// 256-bit two's complement helpers
__device__ __forceinline__ void glv_sub8(uint32_t* r, const uint32_t* a, const uint32_t* b, int nb) {   // r = a - b (b has nb <= 8 words)
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint64_t bi = i < nb ? b[i] : 0u;
        const uint64_t t = (uint64_t)a[i] - bi - borrow;
        r[i] = (uint32_t)t;
        borrow = (t >> 32) & 1u;
    }
}
__device__ __forceinline__ uint32_t glv_abs8(uint32_t* v) {                                             // v = |v|, returns the sign
    const uint32_t neg = v[7] >> 31;
    if (neg) {
        uint64_t carry = 1;
#pragma unroll
        for (int i = 0; i < 8; ++i) { const uint64_t t = (uint64_t)(~v[i]) + carry; v[i] = (uint32_t)t; carry = t >> 32; }
    }
    return neg;
}
// k (canonical, < r) -> kk[0..3] = |k1|, kk[4..7] = |k2|, their signs in bit 31 of kk[3] / kk[7]
__device__ __forceinline__ void glv_decompose(uint32_t kk[8], const uint32_t k[8]) {
    uint32_t p1[11], p2[13];
    glv_mul_words<8, 3>(p1, k, GLV_G1);
    glv_mul_words<8, 5>(p2, k, GLV_G2);
    const uint32_t c1[2] = {p1[8], p1[9]};                          // < 2^64  (p1[10] = 0: k g1 < 2^320)
    const uint32_t c2[4] = {p2[8], p2[9], p2[10], p2[11]};          // < 2^128 (p2[12] = 0)
    uint32_t t1[4], t2[8], t3[6], t4[6];
    glv_mul_words<2, 2>(t1, c1, GLV_A1);
    glv_mul_words<4, 4>(t2, c2, GLV_A2);
    glv_mul_words<2, 4>(t3, c1, GLV_B1M);
    glv_mul_words<4, 2>(t4, c2, GLV_A1);
    uint32_t k1[8], k2[8], z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    glv_sub8(k1, k, t1, 4);
    glv_sub8(k1, k1, t2, 8);
    uint32_t t3w[8] = {t3[0], t3[1], t3[2], t3[3], t3[4], t3[5], 0, 0};
    (void)z;
    glv_sub8(k2, t3w, t4, 6);
    const uint32_t s1 = glv_abs8(k1), s2 = glv_abs8(k2);
#pragma unroll
    for (int i = 0; i < 4; ++i) { kk[i] = k1[i]; kk[4 + i] = k2[i]; }
    kk[3] |= s1 << 31;
    kk[7] |= s2 << 31;
}

// r = [k] * p with k given as its GLV halves (glv_decompose): 127 doublings, each followed by ONE addition of +-P, +-phi(P) or their
// sum selected per lane (a wave executes the addition whenever any of its lanes has a bit set, i.e. always: the plain
// double-and-add paid 254 doublings AND 254 additions per wave)
__device__ __forceinline__ void xyzz_scalar_mul(Xyzz& r, const Xyzz& p, const uint32_t kk[8]) {
    if (p.inf) { xyzz_set_inf(r); return; }
    const uint32_t s1 = kk[3] >> 31, s2 = kk[7] >> 31;
    Fq beta, kin, bx, y1, y2;
    {
        uint32_t bw[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) bw[j] = GLV_BETA[j];
        fe_unpack(beta, bw);
#pragma unroll
        for (int j = 0; j < NL; ++j) kin.l[j] = (int32_t)FqParams::K_PLAIN_IN[j];
        fe_mul(beta, beta, kin);                                   // plain integer -> internal form
    }
    fe_mul(bx, p.x, beta);
    fe_cneg(y1, p.y, s1);
    fe_cneg(y2, p.y, s2);
    Xyzz P1 = p, P2 = p, S;
    P1.y = y1;
    P2.x = bx; P2.y = y2;
    xyzz_add<true>(S, P1, P2);
    Xyzz acc;
    xyzz_set_inf(acc);
#pragma unroll 1
    for (int i = 126; i >= 0; --i) {
        Xyzz t;
        xyzz_dbl_impl(t, acc);
        acc = t;
        const uint32_t b1 = (kk[i >> 5] >> (i & 31)) & 1u, b2 = (kk[4 + (i >> 5)] >> (i & 31)) & 1u;
        Xyzz op;
        const bool both = b1 & b2;
        fe_select(op.x, both, S.x, b1 ? P1.x : P2.x);
        fe_select(op.y, both, S.y, b1 ? P1.y : P2.y);
        fe_select(op.zz, both, S.zz, p.zz);
        fe_select(op.zzz, both, S.zzz, p.zzz);
        op.inf = both ? S.inf : !(b1 | b2);
        xyzz_add<true>(t, acc, op);
        acc = t;
    }
    r = acc;
}

__device__ __forceinline__ void tw_load(Fr& w, const NttTables& tb, uint32_t E) {
#pragma unroll
    for (int j = 0; j < NL; ++j) w.l[j] = tb.lo[(size_t)j * tb.lo_len + (E & (tb.lo_len - 1))];
    uint32_t eh = E >> tb.lo_bits;
    if (eh != 0) {
        Fr h;
#pragma unroll
        for (int j = 0; j < NL; ++j) h.l[j] = tb.hi[(size_t)j * tb.hi_len + eh];
        fe_mul(w, w, h);
    }
}

// scal[e] = the GLV halves (glv_decompose) of the canonical integer of w^-e, or of w^-e / n (scaled != 0), e < n: the scalars of every stage
// (canon != 0: the canonical 256-bit integer itself, for the window-table stage)
__global__ void __launch_bounds__(256)
k_g1fft_scalars(uint4* __restrict__ scal, uint32_t n, int log_n, NttTables tb_inv, int scaled, int canon) {
    uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    Fr w;
    if (log_n > 0) tw_load(w, tb_inv, e); else fe_set_one(w);
    if (scaled) {
        Fr ninv;
#pragma unroll
        for (int j = 0; j < NL; ++j) ninv.l[j] = (int32_t)FrParams::NINV[log_n * NL + j];
        fe_mul(w, w, ninv);
    }
    Fr one_plain, c;
    fe_set_zero(one_plain);
    one_plain.l[0] = 1;
    fe_mul(c, w, one_plain);                       // internal Montgomery form -> plain integer
    fe_canon(c);
    uint32_t kc[8], k[8];
    fe_pack(kc, c);
    if (canon == 2) {                                // wire words (arkworks Montgomery form): the scalars of an MSM
        fe_to_wire(k, w);
    } else if (canon) {
#pragma unroll
        for (int j = 0; j < 8; ++j) k[j] = kc[j];
    } else {
        glv_decompose(k, kc);
    }
    scal[2 * (size_t)e] = make_uint4(k[0], k[1], k[2], k[3]);
    scal[2 * (size_t)e + 1] = make_uint4(k[4], k[5], k[6], k[7]);
}

// planes[i] = P[i] as XYZZ (natural order: the Stockham stages sort on the way)
__global__ void __launch_bounds__(256)
k_g1fft_load(const uint4* __restrict__ points, uint32_t n, int32_t* __restrict__ planes, int bitrev_log) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t j = bitrev_log > 0 ? (__brev(i) >> (32 - bitrev_log)) : i;
    Affine p;
    Xyzz v;
    if (affine_load(p, points + 4 * (size_t)j)) xyzz_from_affine(v, p, 0);
    else xyzz_set_inf(v);
    xyzz_store(planes, n, i, v);
}

// ---- small n: one Stockham stage of radix R = 2^K as direct sums, one lane per (output, term) ---------------------------
__global__ void __launch_bounds__(256)
k_g1fft_direct(const int32_t* __restrict__ x, int32_t* __restrict__ y, uint32_t n, int log_n, int K, int log_s,
               const uint4* __restrict__ scal /* n canonical scalars: w^-e, or w^-e / n in the last stage */, int last) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
    const uint32_t R = 1u << K;
    const uint32_t o = t >> K, jp = t & (R - 1);                 // output element, term j'
    const bool active = o < n;
    Xyzz term;
    xyzz_set_inf(term);
    if (active) {
        const uint32_t nr = n >> K;                               // N / R
        const uint32_t u = o & (nr - 1), j = o >> (log_n - K);
        const uint32_t s = 1u << log_s;
        const uint32_t q = u & (s - 1), p = u >> log_s;
        const uint32_t e = (uint32_t)(((unsigned long long)p * jp << log_s) + (unsigned long long)nr * j * jp) & (n - 1);
        Xyzz v;
        xyzz_load(v, x, n, (size_t)q + ((size_t)(R * p + jp) << log_s));
        if (e == 0 && !last) {
            term = v;
        } else {
            const uint4 lo = scal[2 * (size_t)e], hi = scal[2 * (size_t)e + 1];
            const uint32_t k[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            xyzz_scalar_mul(term, v, k);
        }
    }
    // sum of the R terms of an output: R consecutive lanes (R <= 32 < 64: a wave holds whole outputs)
#pragma unroll 1
    for (int d = 1; d < (int)R; d <<= 1) {
        Xyzz other, r;
        const Fq* sp[4] = {&term.x, &term.y, &term.zz, &term.zzz};
        Fq* tp[4] = {&other.x, &other.y, &other.zz, &other.zzz};
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int l = 0; l < NL; ++l) tp[c]->l[l] = __shfl_down(sp[c]->l[l], d, 64);
        other.inf = __shfl_down((int)term.inf, d, 64) != 0;
        if ((lane & (2 * d - 1)) == 0) {
            xyzz_add<true>(r, term, other);
            term = r;
        }
    }
    if (active && jp == 0) xyzz_store(y, n, o, term);
}

// ---- large n: radix-2 decimation in time on bit-reversed input, one butterfly per thread, in place ------------------------
__global__ void __launch_bounds__(256)
k_g1fft_stage(int32_t* __restrict__ planes, uint32_t n, int log_n, int s, const uint4* __restrict__ scal, int last) {
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n / 2) return;
    const uint32_t half = 1u << (s - 1);
    const uint32_t j = b & (half - 1);
    const uint32_t i0 = ((b >> (s - 1)) << s) | j, i1 = i0 + half;
    Xyzz A, B, t;
    xyzz_load(A, planes, n, i0);
    xyzz_load(B, planes, n, i1);
    const uint32_t E = j << (log_n - s);                 // w_n^(-j n/m)
    if (last) {                                          // (A +- [w]B) / n = [1/n]A +- [w/n]B: both products in this stage
        const uint4 l0 = scal[0], h0 = scal[1];
        const uint32_t k0[8] = {l0.x, l0.y, l0.z, l0.w, h0.x, h0.y, h0.z, h0.w};
        Xyzz a2;
        xyzz_scalar_mul(a2, A, k0);
        A = a2;
    }
    if (E == 0 && !last) {
        t = B;
    } else {
        const uint4 lo = scal[2 * (size_t)E], hi = scal[2 * (size_t)E + 1];
        const uint32_t k[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        xyzz_scalar_mul(t, B, k);
    }
    Xyzz r0, r1, tn = t;
    if (!tn.inf) { fe_neg(tn.y, t.y); fe_norm(tn.y); }
    xyzz_add<true>(r0, A, t);
    xyzz_add<true>(r1, A, tn);
    xyzz_store(planes, n, i0, r0);
    xyzz_store(planes, n, i1, r1);
}

// ---- the same two stage kernels on LANE PAIRS (curve_pair.h): one point per two lanes ----------------------------------------
// A stage is one scalar multiplication deep: 127 doublings + 127 additions one after the other.  On a pair of lanes a doubling costs
// 5 multiplications per lane instead of 9 and an addition 7 instead of 14, and the doubled number of waves fills the issue slots a
// lone wave per SIMD leaves empty: 1.45 -> ~0.8 ms per stage.  Same group elements (the affine results are bit-identical).
__device__ __forceinline__ void pair_dbl_any(HalfXyzz& r, const HalfXyzz& a, bool odd) {      // a may be the identity
    if (__all(a.inf)) { r = a; return; }
    HalfXyzz d;
    pair_dbl(d, a, odd);
    if (a.inf) r = a; else r = d;
}
// r = [k] p, k as its GLV halves (glv_decompose); every lane of a pair holds the same k
__device__ __forceinline__ void pair_scalar_mul(HalfXyzz& r, const HalfXyzz& p, const uint32_t kk[8], bool odd) {
    const uint32_t s1 = kk[3] >> 31, s2 = kk[7] >> 31;
    Fq beta, kin, bx, y1, y2;
    {
        uint32_t bw[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) bw[j] = GLV_BETA[j];
        fe_unpack(beta, bw);
#pragma unroll
        for (int j = 0; j < NL; ++j) kin.l[j] = (int32_t)FqParams::K_PLAIN_IN[j];
        fe_mul(beta, beta, kin);                                   // plain integer -> internal form
    }
    fe_mul(bx, p.u, beta);                                         // even lane: beta X
    fe_cneg(y1, p.u, s1); fe_norm(y1);                             // odd lane: +-Y
    fe_cneg(y2, p.u, s2); fe_norm(y2);
    HalfXyzz P1 = p, P2 = p, S;
    fe_select(P1.u, odd, y1, p.u);
    fe_select(P2.u, odd, y2, bx);
    pair_add(S, P1, P2, odd);
    HalfXyzz acc;
    half_set_inf(acc);
#pragma unroll 1
    for (int i = 126; i >= 0; --i) {
        HalfXyzz t;
        pair_dbl_any(t, acc, odd);
        acc = t;
        const uint32_t b1 = (kk[i >> 5] >> (i & 31)) & 1u, b2 = (kk[4 + (i >> 5)] >> (i & 31)) & 1u;
        HalfXyzz op;
        const bool both = b1 & b2;
        fe_select(op.u, both, S.u, b1 ? P1.u : P2.u);
        fe_select(op.v, both, S.v, p.v);
        op.inf = p.inf || (both ? S.inf : !(b1 | b2));
        pair_add(t, acc, op, odd);
        acc = t;
    }
    r = acc;
}

__global__ void __launch_bounds__(256)
k_g1fft_direct_pairs(const int32_t* __restrict__ x, int32_t* __restrict__ y, uint32_t n, int log_n, int K, int log_s,
                     const uint4* __restrict__ scal, int last) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63, pair = t >> 1;
    const bool odd = (t & 1u) != 0;
    const uint32_t R = 1u << K;
    const uint32_t o = pair >> K, jp = pair & (R - 1);           // output element, term j'
    const bool active = o < n;
    HalfXyzz term;
    half_set_inf(term);
    uint32_t k[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool plain = true;                                            // the term is the input itself (scalar 1)
    if (active) {
        const uint32_t nr = n >> K;
        const uint32_t u = o & (nr - 1), j = o >> (log_n - K);
        const uint32_t q = u & ((1u << log_s) - 1), p = u >> log_s;
        const uint32_t e = (uint32_t)(((unsigned long long)p * jp << log_s) + (unsigned long long)nr * j * jp) & (n - 1);
        half_load(term, x, n, (size_t)q + ((size_t)(R * p + jp) << log_s), odd);
        if (!(e == 0 && !last)) {
            const uint4 lo = scal[2 * (size_t)e], hi = scal[2 * (size_t)e + 1];
            k[0] = lo.x; k[1] = lo.y; k[2] = lo.z; k[3] = lo.w; k[4] = hi.x; k[5] = hi.y; k[6] = hi.z; k[7] = hi.w;
            plain = false;
        }
    }
    if (!__all(plain)) {                                          // wave-uniform: the multiplication runs for the whole wave or not at all
        HalfXyzz m;
        pair_scalar_mul(m, term, k, odd);
        if (!plain) term = m;
    }
    // sum of the R terms of an output: R consecutive pairs (R <= 32: a wave holds whole outputs)
#pragma unroll 1
    for (int d = 1; d < (int)R; d <<= 1) {
        HalfXyzz other;
        half_shfl_down(other, term, 2 * d);
        if (((lane >> 1) & (2 * d - 1)) == 0) {
            HalfXyzz r;
            pair_add(r, term, other, odd);
            term = r;
        }
    }
    if (active && jp == 0) half_store(y, n, o, term, odd);
}

// ---- the direct stage on LANE QUADS (curve_quad.h; round 4) -------------------------------------------------------------------------
// One step of the GLV chain is a doubling and an addition one after the other: 5 + 7 products deep on a lane pair, 3 + 4 on a quad.  A stage
// that fits one wave per SIMD is pure latency, so the quad form takes 7 / 12 of the pair form's time (measured: 0.83 -> 0.60 ms; with two bits
// per step 3 + 3 + 4 per two bits); the products
// [k] x of all (output, term) slots go to a partial array and k_g1fft_sum_partials adds the R = 2^K terms of an output (a wave holds 16 quads,
// so the tree no longer fits the multiplying wave for R = 32).  r = [k] p, k as its GLV halves; every lane of a quad holds the same k.
__device__ __forceinline__ void quad_scalar_mul(QuadXyzz& r, const QuadXyzz& p, const uint32_t kk[8], uint32_t q, int32_t* __restrict__ tab /* this lane's LDS column: word (entry 9 + limb) 256 */) {
    const uint32_t s1 = kk[3] >> 31, s2 = kk[7] >> 31;
    Fq beta, kin, bx, y1, y2;
    {
        uint32_t bw[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) bw[j] = GLV_BETA[j];
        fe_unpack(beta, bw);
#pragma unroll
        for (int j = 0; j < NL; ++j) kin.l[j] = (int32_t)FqParams::K_PLAIN_IN[j];
        fe_mul(beta, beta, kin);                                   // plain integer -> internal form
    }
    fe_mul(bx, p.c, beta);                                         // lane 0: beta X
    fe_cneg(y1, p.c, s1); fe_norm(y1);                             // lane 1: +-Y
    fe_cneg(y2, p.c, s2); fe_norm(y2);
    QuadXyzz P1 = p, P2 = p;
    fe_select(P1.c, q == 1u, y1, p.c);
    fe_select(P2.c, q == 1u, y2, p.c);
    fe_select(P2.c, q == 0u, bx, P2.c);
#if defined(KZG_G1FFT_QUAD_W1)
    // one bit of each half per step: a doubling and an addition of P1, P2 or P1 + P2 (7 products deep per bit)
    QuadXyzz S;
    quad_add(S, P1, P2, q);
    QuadXyzz acc;
    quad_set_inf(acc);
#pragma unroll 1
    for (int i = 126; i >= 0; --i) {
        QuadXyzz t;
        quad_dbl_any(t, acc, q);
        acc = t;
        const uint32_t b1 = (kk[i >> 5] >> (i & 31)) & 1u, b2 = (kk[4 + (i >> 5)] >> (i & 31)) & 1u;
        QuadXyzz op;
        const bool both = b1 & b2;
        fe_select(op.c, both, S.c, b1 ? P1.c : P2.c);
        op.inf = p.inf || (both ? S.inf : !(b1 | b2));
        quad_add(t, acc, op, q);
        acc = t;
    }
    r = acc;
#else
    // TWO bits of each half per step: two doublings and ONE addition of T[a + 4 b] = a P1 + b P2, a, b < 4 (3 + 3 + 4 = 10 products deep per two
    // bits instead of 14).  On a quad a point is nine words per lane, so the fifteen table points of a lane stay in registers (135 VGPRs)
    // and the entry is picked with a v_cndmask tree.  -DKZG_G1FFT_QUAD_LDS keeps them in a 540-byte LDS column per lane instead (144 KiB per
    // workgroup, nine ds_read_b32 per step issued ahead of the doublings): 195 instead of 288 VGPRs, the SAME time per stage (measured:
    // 0.728 / 0.975 / 1.428 ms against 0.729 / 0.978 / 1.441 ms at 512 / 1 024 / 2 048 points) -- the selects are not what a stage waits for.
    // No entry is the identity unless p is (a + b lambda != 0 mod r for these a, b): one flag for all.
    QuadXyzz t, u;
#if !defined(KZG_G1FFT_QUAD_LDS)
    Fq T[16];                                                      // T[0] unused
#define QTAB_PUT(e, v) T[e] = (v)
#define QTAB_GET(dst, e) dst = T[e]
#else
    auto tab_put = [&](int e, const Fq& v) {
#pragma unroll
        for (int j = 0; j < NL; ++j) tab[(e * NL + j) * 256] = v.l[j];
    };
    auto tab_get = [&](Fq& v, uint32_t e) {
#pragma unroll
        for (int j = 0; j < NL; ++j) v.l[j] = tab[(e * NL + j) * 256];
    };
#define QTAB_PUT(e, v) tab_put(e, v)
#define QTAB_GET(dst, e) tab_get(dst, e)
#endif
    QTAB_PUT(1, P1.c); QTAB_PUT(4, P2.c);
    Fq p1x2, p1x3, p2x2, p2x3;
    quad_dbl_any(t, P1, q); p1x2 = t.c; QTAB_PUT(2, t.c);
    quad_add(u, t, P1, q); p1x3 = u.c; QTAB_PUT(3, u.c);
    quad_dbl_any(t, P2, q); p2x2 = t.c; QTAB_PUT(8, t.c);
    quad_add(u, t, P2, q); p2x3 = u.c; QTAB_PUT(12, u.c);
#pragma unroll
    for (int bb = 1; bb < 4; ++bb)
#pragma unroll
        for (int aa = 1; aa < 4; ++aa) {
            QuadXyzz x, y;
            x.c = aa == 1 ? P1.c : aa == 2 ? p1x2 : p1x3; x.inf = p.inf;
            y.c = bb == 1 ? P2.c : bb == 2 ? p2x2 : p2x3; y.inf = p.inf;
            quad_add(t, x, y, q);
            QTAB_PUT(aa + 4 * bb, t.c);
        }
    QuadXyzz acc;
    quad_set_inf(acc);
    // the two 127-bit halves as shift registers (a dynamically indexed kk[] would live in scratch memory): the window is the top two bits
    uint32_t h1[4] = {kk[0], kk[1], kk[2], kk[3] & 0x7FFFFFFFu}, h2[4] = {kk[4], kk[5], kk[6], kk[7] & 0x7FFFFFFFu};   // bit 127 is the sign
#pragma unroll 1
    for (int i = 63; i >= 0; --i) {                                // bits 2 i + 1, 2 i of both halves
        const uint32_t a = h1[3] >> 30, b = h2[3] >> 30;
#pragma unroll
        for (int j = 3; j > 0; --j) { h1[j] = (h1[j] << 2) | (h1[j - 1] >> 30); h2[j] = (h2[j] << 2) | (h2[j - 1] >> 30); }
        h1[0] <<= 2; h2[0] <<= 2;
        const uint32_t idx = a | (b << 2);
        QuadXyzz op;
#if !defined(KZG_G1FFT_QUAD_LDS)
        // 16-way select as a binary tree over the index bits (entry 0 never used as a point: op.inf covers it)
        Fq s8[8], s4[4], s2[2];
#pragma unroll
        for (int m = 0; m < 8; ++m) fe_select(s8[m], (idx & 1u) != 0, T[2 * m + 1], T[m == 0 ? 1 : 2 * m]);
#pragma unroll
        for (int m = 0; m < 4; ++m) fe_select(s4[m], (idx & 2u) != 0, s8[2 * m + 1], s8[2 * m]);
#pragma unroll
        for (int m = 0; m < 2; ++m) fe_select(s2[m], (idx & 4u) != 0, s4[2 * m + 1], s4[2 * m]);
        fe_select(op.c, (idx & 8u) != 0, s2[1], s2[0]);
#else
        QTAB_GET(op.c, idx ? idx : 1u);                            // issued before the doublings: the reads are back when the addition needs them
#endif
        op.inf = p.inf || idx == 0u;
        quad_dbl_any(t, acc, q);
        quad_dbl_any(acc, t, q);
        quad_add(t, acc, op, q);
        acc = t;
    }
#undef QTAB_PUT
#undef QTAB_GET
    r = acc;
#endif
}

// slot (o, j') of a direct stage of radix R = 2^K (the index rule of k_g1fft_direct_pairs): partial[o R + j'] = [w^-e (/ n)] x[input]
__global__ void __launch_bounds__(256)
k_g1fft_mul_quads(const int32_t* __restrict__ x, int32_t* __restrict__ partial, uint32_t n, int log_n, int K, int log_s,
                  const uint4* __restrict__ scal, int last) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, q = t & 3u, quad = t >> 2;
    const uint32_t R = 1u << K;
    const uint32_t o = quad >> K, jp = quad & (R - 1);
    const bool active = o < n;
    QuadXyzz term;
    quad_set_inf(term);
    uint32_t k[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool plain = true;                                            // the term is the input itself (scalar 1)
    if (active) {
        const uint32_t nr = n >> K;
        const uint32_t u = o & (nr - 1), j = o >> (log_n - K);
        const uint32_t qq = u & ((1u << log_s) - 1), p = u >> log_s;
        const uint32_t e = (uint32_t)(((unsigned long long)p * jp << log_s) + (unsigned long long)nr * j * jp) & (n - 1);
        quad_load(term, x, n, (size_t)qq + ((size_t)(R * p + jp) << log_s), q);
        if (!(e == 0 && !last)) {
            const uint4 lo = scal[2 * (size_t)e], hi = scal[2 * (size_t)e + 1];
            k[0] = lo.x; k[1] = lo.y; k[2] = lo.z; k[3] = lo.w; k[4] = hi.x; k[5] = hi.y; k[6] = hi.z; k[7] = hi.w;
            plain = false;
        }
    }
    extern __shared__ int32_t qtab[];                             // -DKZG_G1FFT_QUAD_LDS: 16 entries x 9 words x 256 lanes; unused (size 0) otherwise
    if (!__all(plain)) {                                          // wave-uniform: the multiplication runs for the whole wave or not at all
        QuadXyzz m;
        quad_scalar_mul(m, term, k, q, qtab + threadIdx.x);
        if (!plain) term = m;
    }
    if (active) quad_store(partial, (size_t)n * R, (size_t)o * R + jp, term, q);
}

__global__ void __launch_bounds__(256)
k_g1fft_stage_pairs(int32_t* __restrict__ planes, uint32_t n, int log_n, int s, const uint4* __restrict__ scal, int last) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, b = t >> 1;
    const bool odd = (t & 1u) != 0;
    const bool active = b < n / 2;                               // n / 2 >= 32 pairs here: whole waves are active or not, except the last one
    const uint32_t half = 1u << (s - 1);
    const uint32_t j = b & (half - 1);
    const uint32_t i0 = active ? (((b >> (s - 1)) << s) | j) : 0, i1 = i0 + half;
    HalfXyzz A, B, tB;
    half_load(A, planes, n, i0, odd);
    half_load(B, planes, n, i1, odd);
    if (!active) { half_set_inf(A); half_set_inf(B); }
    const uint32_t E = j << (log_n - s);
    if (last) {                                                   // (A +- [w]B) / n = [1/n]A +- [w/n]B
        const uint4 l0 = scal[0], h0 = scal[1];
        const uint32_t k0[8] = {l0.x, l0.y, l0.z, l0.w, h0.x, h0.y, h0.z, h0.w};
        HalfXyzz a2;
        pair_scalar_mul(a2, A, k0, odd);
        A = a2;
    }
    const bool plain = E == 0 && !last;
    tB = B;
    if (!__all(plain)) {
        const uint4 lo = scal[2 * (size_t)E], hi = scal[2 * (size_t)E + 1];
        const uint32_t k[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        HalfXyzz m;
        pair_scalar_mul(m, B, k, odd);
        if (!plain) tB = m;
    }
    HalfXyzz r0, r1, tn = tB;
    {
        Fq ny;
        fe_neg(ny, tB.u); fe_norm(ny);
        fe_select(tn.u, odd && !tB.inf, ny, tB.u);                // -[w]B: the odd lane's Y changes sign
    }
    pair_add(r0, A, tB, odd);
    pair_add(r1, A, tn, odd);
    if (active) {
        half_store(planes, n, i0, r0, odd);
        half_store(planes, n, i1, r1, odd);
    }
}

// ---- the FIRST stage through the SRS window tables --------------------------------------------------------------------------------
// The inputs of the first stage are SRS points, and the SRS carries window tables T_w[i] = 2^(c w) P_i for its MSMs.  So a term
// [k] P_i = sum_w d_w(k) T_w[i] with the signed c-bit digits of k: W small multiplications of c double-and-add steps (the addend is an
// AFFINE table point: pair_madd) instead of one 127-step GLV chain -- c x (5 + 5) multiplications deep instead of 127 x (5 + 7) -- and
// a tree over the R W terms of an output.  One PAIR of lanes per (output, term, window); a wave holds 32 (term, window) slots of one
// output and leaves their sum; k_g1fft_sum_partials adds the waves of an output.  Same group elements.
__global__ void __launch_bounds__(256)
k_g1fft_first_tables(const uint4* __restrict__ tables, uint32_t table_stride, int c, int W, uint32_t n, int log_n, int K,
                     const uint4* __restrict__ scal_canon, uint32_t waves_per_out, int32_t* __restrict__ partial) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63, pair = lane >> 1;
    const bool odd = (t & 1u) != 0;
    const uint32_t gw = t >> 6;
    const uint32_t o = gw / waves_per_out, wv = gw - o * waves_per_out;
    if (o >= n) return;                                           // wave-uniform
    const uint32_t R = 1u << K, nr = n >> K;
    const uint32_t tt = wv * 32 + pair;
    const bool valid = tt < R * (uint32_t)W;
    const uint32_t jp = valid ? tt / (uint32_t)W : 0, w = valid ? tt - jp * (uint32_t)W : 0;
    const uint32_t u = o & (nr - 1), j = o >> (log_n - K);
    const uint32_t e = (uint32_t)((unsigned long long)nr * j * jp) & (n - 1);
    const uint32_t i = u + nr * jp;                               // first stage: stride s = N / R, p = 0
    // signed c-bit digit number w of the canonical scalar (the MSM's digit rule: k_msm_digits)
    uint32_t mag = 0, neg = 0;
    {
        const uint4 lo = scal_canon[2 * (size_t)e], hi = scal_canon[2 * (size_t)e + 1];
        uint32_t k[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        const uint32_t mask = (1u << c) - 1u, half = 1u << (c - 1);
        uint32_t carry = 0;
        for (uint32_t ww = 0; ww <= w; ++ww) {
            const uint32_t raw = (k[0] & mask) + carry;
#pragma unroll
            for (int q = 0; q < 7; ++q) k[q] = (k[q] >> c) | (k[q + 1] << (32 - c));
            k[7] >>= c;
            neg = raw > half;
            mag = neg ? (1u << c) - raw : raw;
            carry = neg;
        }
        if (!valid) mag = 0;
    }
    const uint4* src = tables + 4 * ((size_t)w * table_stride + i) + (odd ? 2 : 0);
    const uint4 q0 = src[0], q1 = src[1];                         // even lane: x, odd lane: y
    const uint32_t w32[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
    int any = (q0.x | q0.y | q0.z | q0.w | q1.x | q1.y | q1.z | q1.w) != 0 ? 1 : 0;
    any |= pair_swap(any);
    if (!any) mag = 0;                                            // identity table point
    Fq cpt;
    fe_unpack(cpt, w32);
    HalfXyzz acc;
    half_set_inf(acc);
#pragma unroll 1
    for (int b = c - 1; b >= 0; --b) {
        HalfXyzz r;
        pair_dbl_any(r, acc, odd);
        acc = r;
        if ((mag >> b) & 1u) {                                    // pair-uniform
            pair_madd(r, acc, cpt, neg, odd);
            acc = r;
        }
    }
#pragma unroll 1
    for (int d = 1; d < 32; d <<= 1) {                            // sum of the wave's 32 slots
        HalfXyzz other;
        half_shfl_down(other, acc, 2 * d);
        if ((pair & (2 * d - 1)) == 0) {
            HalfXyzz r;
            pair_add(r, acc, other, odd);
            acc = r;
        }
    }
    if (pair == 0) half_store(partial, (size_t)n * waves_per_out, (size_t)o * waves_per_out + wv, acc, odd);
}
// ---- the WHOLE transform of 64 .. 256 points through the per-bit SRS tables ----------------------------------------------------------
// An SRS of >= 2^15 points carries Bit_p[j] = 2^p P_j for every bit position p (srs.hip srs_build_bit_tables, the NAF mode of its
// MSMs).  With the scalars in plain non-adjacent form (digits +-1, ~85 per scalar) a term [k] P_j is a SUM OF TABLE POINTS,
// sum_t +-Bit_(p_t)[j], and an output L_o = sum_j [w^(-o j) / n] P_j is n x 85 signed table points: no doubling, no scalar
// multiplication, no stages -- mixed additions at the chip's throughput (n^2 x 85: 5.6 M for n = 256) and one tree.  The n distinct
// scalars are recoded once per size (k_g1fft_naf2); one PAIR of lanes per (output, term, slice of the digit list) adds its digits
// (pair_madd, 5 multiplications per lane); a wave holds 32 slots of one output and leaves their sum; k_g1fft_sum_partials adds the
// <= 32 waves of an output.  Same group elements as the staged transform.
constexpr uint32_t NAF2_MAX = 128;                       // digit slots per scalar (width-4 NAF of a scalar < 2^254: at most 254 / 4 + 1 = 64)
constexpr uint32_t G1FFT_T3_MAX = 2048;                  // the x3 tables cover the first min(SRS length, 2048) points: every point the table paths transform
// The odd multiples 3, 5, 7 of Bit_p[j] as XYZZ planes: k_g1fft_to_affine turns them into the tables.  3 x 255 x t3_points points (100 MB at
// 2 048), once per SRS (kzg_srs::d_t3, ::t3_n).
__global__ void __launch_bounds__(256)
k_g1fft_t3_planes(const uint4* __restrict__ bits, uint32_t stride, uint32_t t3_points, uint32_t total, int32_t* __restrict__ planes) {
    // slot t = ((key - 1) 255 + p) t3_points + j, key = 1, 2, 3: (2 key + 1) Bit_p[j] = Bit_p + Bit_(p+1) | Bit_p + Bit_(p+2) | Bit_(p+3) - Bit_p
    // (round 4: width-4 digits +-1, +-3, +-5, +-7).  A slot whose second plane would be beyond bit 254 stays the identity: no scalar
    // below the group order has such a digit (5 2^252 and 7 2^251 exceed it even with every lower digit negative).
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const uint32_t row = t / t3_points, j = t - row * t3_points;          // t3_points <= stride: every read stays inside its bit plane
    const uint32_t key = row / 255u + 1u, p = row - (key - 1u) * 255u;
    Xyzz v;
    xyzz_set_inf(v);
    if (p + key <= 254u) {
        Affine a, b;
        const bool ha = affine_load(a, bits + 4 * ((size_t)p * stride + j));
        const bool hb = affine_load(b, bits + 4 * ((size_t)(p + key) * stride + j));
        if (ha && hb) {
            if (key < 3u) { xyzz_from_affine(v, a, 0); xyzz_madd<true>(v, b, 0); }
            else { xyzz_from_affine(v, b, 0); xyzz_madd<true>(v, a, 1); }
        }
    }
    xyzz_store(planes, total, t, v);
}
__global__ void __launch_bounds__(64)
k_g1fft_naf2(const uint4* __restrict__ scal_canon, uint32_t n, uint16_t* __restrict__ list, uint32_t* __restrict__ cnt) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const uint4 lo = scal_canon[2 * (size_t)e], hi = scal_canon[2 * (size_t)e + 1];
    uint32_t k[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    uint32_t m = 0;
    // width-4 NAF (round 4; width 3 before): digits +-1, +-3, +-5, +-7 (key = 0 .. 3), ~51 per scalar instead of ~64; a digit with key > 0
    // reads the table of that odd multiple.  Entry: position (8 bits) | key << 8 | sign << 15
    naf_for_digits(k, 4, [&](uint32_t pos, uint32_t key, uint32_t neg) { if (m < NAF2_MAX) list[(size_t)e * NAF2_MAX + m] = (uint16_t)(pos | (key << 8) | (neg << 15)); ++m; });
    cnt[e] = m < NAF2_MAX ? m : NAF2_MAX;
}
// K < log n: the same kernel as the FIRST STAGE of a staged transform of radix R = 2^K (the index rule of k_g1fft_first_tables: output o
// sums the R inputs u + (n / R) j' with the scalars w^-(n / R . j . j'), u = o mod n / R, j = o / (n / R)); K = log n is the whole transform.
__global__ void __launch_bounds__(256)
k_g1fft_bits(const uint4* __restrict__ bits, uint32_t stride, const uint4* __restrict__ bits3 /* 3 Bit_p[i], i < t3_points, that many points apart */,
             uint32_t t3_points, uint32_t n, int log_n, int K, const uint16_t* __restrict__ list, const uint32_t* __restrict__ cnt,
             uint32_t Q, uint32_t waves_per_out, int32_t* __restrict__ partial) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63, pair = lane >> 1;
    const bool odd = (t & 1u) != 0;
    const uint32_t gw = t >> 6;
    const uint32_t o = gw / waves_per_out, wv = gw - o * waves_per_out;
    if (o >= n) return;                                           // wave-uniform
    const uint32_t slot = wv * 32 + pair;                         // < R Q (the host makes R Q a multiple of 32)
    const uint32_t jp = slot / Q, q = slot - jp * Q;
    const uint32_t nr = n >> K, u = o & (nr - 1), jo = o >> (log_n - K);
    const uint32_t j = u + nr * jp;                               // the input point
    const uint32_t e = (uint32_t)((unsigned long long)nr * jo * jp) & (n - 1);
    const uint16_t* L = list + (size_t)e * NAF2_MAX;
    const uint32_t c = cnt[e];
    HalfXyzz acc;
    half_set_inf(acc);
    // digit m of the list: the table point Bit_pos[j], even lane x, odd lane y; one point in flight ahead of the addition
    auto fetch = [&](uint32_t m, uint4& a, uint4& b, uint32_t& neg) {
        const uint32_t d = L[m < c ? m : (c ? c - 1 : 0)];
        neg = d >> 15;
        const uint32_t pos = d & 0xFFu, key = (d >> 8) & 3u;
        const uint4* src = (key ? bits3 + 4 * ((size_t)((key - 1u) * 255u + pos) * t3_points + j) : bits + 4 * ((size_t)pos * stride + j)) + (odd ? 2 : 0);
        a = src[0]; b = src[1];
    };
    uint4 a0, b0; uint32_t neg0 = 0;
    if (c) fetch(q, a0, b0, neg0);
#pragma unroll 1
    for (uint32_t m = q; m < c; m += Q) {                         // pair-uniform trip count
        const uint32_t w32[8] = {a0.x, a0.y, a0.z, a0.w, b0.x, b0.y, b0.z, b0.w};
        const uint32_t neg = neg0;
        int any = (a0.x | a0.y | a0.z | a0.w | b0.x | b0.y | b0.z | b0.w) != 0 ? 1 : 0;
        any |= pair_swap(any);
        fetch(m + Q, a0, b0, neg0);
        if (!any) continue;                                       // identity SRS point
        Fq cpt;
        fe_unpack(cpt, w32);
        HalfXyzz r;
        pair_madd(r, acc, cpt, neg, odd);
        acc = r;
    }
#pragma unroll 1
    for (int d = 1; d < 32; d <<= 1) {                            // sum of the wave's 32 slots
        HalfXyzz other;
        half_shfl_down(other, acc, 2 * d);
        if ((pair & (2 * d - 1)) == 0) {
            HalfXyzz r;
            pair_add(r, acc, other, odd);
            acc = r;
        }
    }
    if (pair == 0) half_store(partial, (size_t)n * waves_per_out, (size_t)o * waves_per_out + wv, acc, odd);
}
// y[o] = sum of the waves_per_out (<= 32) partial sums of output o: one wave per output
__global__ void __launch_bounds__(256)
k_g1fft_sum_partials(const int32_t* __restrict__ partial, uint32_t waves_per_out, uint32_t n, int32_t* __restrict__ y) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63, pair = lane >> 1;
    const bool odd = (t & 1u) != 0;
    const uint32_t o = t >> 6;
    if (o >= n) return;
    HalfXyzz acc;
    if (pair < waves_per_out) half_load(acc, partial, (size_t)n * waves_per_out, (size_t)o * waves_per_out + pair, odd);
    else half_set_inf(acc);
#pragma unroll 1
    for (int d = 1; d < 32; d <<= 1) {
        HalfXyzz other;
        half_shfl_down(other, acc, 2 * d);
        if ((pair & (2 * d - 1)) == 0) {
            HalfXyzz r;
            pair_add(r, acc, other, odd);
            acc = r;
        }
    }
    if (pair == 0) half_store(y, n, o, acc, odd);
}

// ---- XYZZ -> affine ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void affine_emit(uint4* __restrict__ out, size_t i, const Xyzz& r, const Fq& inv_zz_zzz /* 1 / (ZZ ZZZ) */, bool wire) {
    uint32_t o[16];
    if (r.inf) {
#pragma unroll
        for (int j = 0; j < 16; ++j) o[j] = 0;
    } else {
        Fq t, x, y;
        fe_mul(t, inv_zz_zzz, r.zzz);              // 1 / ZZ
        fe_mul(x, r.x, t);
        fe_mul(t, inv_zz_zzz, r.zz);               // 1 / ZZZ
        fe_mul(y, r.y, t);
        if (wire) {
            fe_norm(x); fe_norm(y);
            fe_to_wire(o, x);
            fe_to_wire(o + 8, y);
        } else {                                   // device affine format (curve.h): canonical residues of the internal form
            fe_canon(x); fe_canon(y);
            fe_pack(o, x);
            fe_pack(o + 8, y);
        }
    }
    out[4 * i] = make_uint4(o[0], o[1], o[2], o[3]);
    out[4 * i + 1] = make_uint4(o[4], o[5], o[6], o[7]);
    out[4 * i + 2] = make_uint4(o[8], o[9], o[10], o[11]);
    out[4 * i + 3] = make_uint4(o[12], o[13], o[14], o[15]);
}
// lane t converts the points i = t, t + T, t + 2T, ... (T = number of lanes) with ONE inversion (Montgomery's trick): prefix
// products of the denominators ZZ ZZZ go through `scratch` (9 limb planes, stride n).  per lane <= AFF_PER points.
constexpr uint32_t AFF_PER = 4;        // 16 while the lane's inversion was a 381-product chain; with division steps (fe_invert.h) shorter chains on more lanes win: 84 -> ~45 us at 1 024 points
__global__ void __launch_bounds__(256)
k_g1fft_to_affine(const int32_t* __restrict__ planes, uint32_t n, uint4* __restrict__ out, int wire, int32_t* __restrict__ scratch) {
    const uint32_t T = gridDim.x * blockDim.x, t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    Fq run;
    fe_set_one(run);
    for (uint32_t i = t; i < n; i += T) {                 // forward: prefix products (identity points contribute 1)
        Xyzz v;
        xyzz_load(v, planes, n, i);
        Fq d;
        if (v.inf) fe_set_one(d); else fe_mul(d, v.zz, v.zzz);
#pragma unroll
        for (int j = 0; j < NL; ++j) scratch[(size_t)j * n + i] = run.l[j];
        fe_mul(run, run, d);
    }
    Fq rinv;
    fe_inverse_fermat(rinv, run);
    const uint32_t cnt = (n - 1 - t) / T + 1;
    for (uint32_t k = cnt; k-- > 0;) {                    // backward: inv_i = rinv * prefix_i; rinv *= d_i
        const uint32_t i = t + k * T;
        Xyzz v;
        xyzz_load(v, planes, n, i);
        Fq pre, d, iv;
#pragma unroll
        for (int j = 0; j < NL; ++j) pre.l[j] = scratch[(size_t)j * n + i];
        if (v.inf) fe_set_one(d); else fe_mul(d, v.zz, v.zzz);
        fe_mul(iv, rinv, pre);
        fe_mul(rinv, rinv, d);
        affine_emit(out, i, v, iv, wire != 0);
    }
}

// ---- host -----------------------------------------------------------------------------------------------------------------
struct ScalKey { int dev, log_n, scaled; bool operator<(const ScalKey& o) const { return dev != o.dev ? dev < o.dev : (log_n != o.log_n ? log_n < o.log_n : scaled < o.scaled); } };   // scaled: bit 0 = times 1/n, bit 1 = canonical (not GLV-split)
static std::map<ScalKey, uint4*> g_scal;
static std::mutex g_scal_mu;
static int32_t get_scalars(kzg_ctx* ctx, int log_n, bool scaled, const uint4** out, int canon = 0) {     // canon: 0 GLV halves, 1 canonical integers, 2 wire words
    std::lock_guard<std::mutex> lk(g_scal_mu);
    ScalKey key{ctx->device, log_n, (scaled ? 1 : 0) | (canon << 1)};
    auto it = g_scal.find(key);
    if (it != g_scal.end()) { *out = it->second; return KZG_OK; }
    NttTables tb{};
    if (log_n > 0) { int32_t rc = ntt_get_tables(ctx, log_n, true, &tb); if (rc != KZG_OK) return rc; }
    const size_t n = (size_t)1 << log_n;
    uint4* p = nullptr;
    KZG_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&p), n * 32));
    hipLaunchKernelGGL(k_g1fft_scalars, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, p, (uint32_t)n, log_n, tb, scaled ? 1 : 0, canon);
    KZG_HIP_TRY(ctx, hipGetLastError());
    KZG_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    g_scal[key] = p;
    *out = p;
    return KZG_OK;
}

// width-3 NAF digit lists of the n scalars w^-e / n (scaled: the whole transform) or w^-e (a first stage), k_g1fft_naf2; cached per (device, log n, scaled)
struct Naf2Lists { uint16_t* list = nullptr; uint32_t* cnt = nullptr; };
static std::map<std::tuple<int, int, int>, Naf2Lists> g_naf2;
static int32_t get_naf2(kzg_ctx* ctx, int log_n, bool scaled, Naf2Lists* out) {
    const uint4* sc = nullptr;
    int32_t rc = get_scalars(ctx, log_n, scaled, &sc, 1);            // canonical integers
    if (rc != KZG_OK) return rc;
    std::lock_guard<std::mutex> lk(g_scal_mu);
    auto key = std::make_tuple(ctx->device, log_n, scaled ? 1 : 0);
    auto it = g_naf2.find(key);
    if (it != g_naf2.end()) { *out = it->second; return KZG_OK; }
    const size_t n = (size_t)1 << log_n;
    Naf2Lists l;
    KZG_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&l.list), n * NAF2_MAX * 2));
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&l.cnt), n * 4);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_g1fft_naf2, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, sc, (uint32_t)n, l.list, l.cnt);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);      // shared by every context of the device: complete before it is published
    if (e != hipSuccess) { (void)hipFree(l.list); if (l.cnt) (void)hipFree(l.cnt); return set_error(ctx, e, "recoding the scalars of g1_ifft"); }
    g_naf2[key] = l;
    *out = l;
    return KZG_OK;
}

// XYZZ planes -> wire XYZZ words (32 u32 per point) for the host-side affine conversion of small transforms
__global__ void __launch_bounds__(256)
k_g1fft_planes_to_wire(const int32_t* __restrict__ planes, uint32_t n, uint32_t* __restrict__ out_wire) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Xyzz v;
    xyzz_load(v, planes, n, i);
    uint32_t w[32];
    xyzz_to_wire(w, v);
#pragma unroll
    for (int j = 0; j < 32; j += 4) *reinterpret_cast<uint4*>(out_wire + (size_t)i * 32 + j) = make_uint4(w[j], w[j + 1], w[j + 2], w[j + 3]);
}

// The stages of the transform: *result_out = XYZZ planes (stride n) of the Lagrange basis of the first n SRS points, natural order
static int32_t g1_ifft_stages(kzg_ctx* ctx, const kzg_srs* srs, size_t n, const int32_t** result_out) {
    int log_n = 0;
    while (((size_t)1 << log_n) < n) ++log_n;
    hipStream_t st = ctx->stream;
    const unsigned gn = (unsigned)((n + 255) / 256);
    KZG_HIP_TRY(ctx, ctx->poly[0].b.reserve(n * 36 * 4 * 2));                 // two XYZZ plane sets (ping-pong)
    KZG_HIP_TRY(ctx, ctx->poly[0].a.reserve(n * NL * 4));                     // prefix products of the affine conversion
    int32_t* bufA = ctx->poly[0].b.as<int32_t>();
    int32_t* bufB = bufA + n * 36;
    const uint4 *scal = nullptr, *scal_n = nullptr;
    int32_t rc = get_scalars(ctx, log_n, false, &scal);
    if (rc == KZG_OK) rc = get_scalars(ctx, log_n, true, &scal_n);
    if (rc != KZG_OK) return rc;
    const int32_t* result = bufA;
    // 64 .. 256 points of an SRS with per-bit tables: the whole transform as sums of table points (k_g1fft_bits);
    // 512 .. 2048 points: the FIRST STAGE that way (radix 2^K0 over the SRS points: n 2^K0 x 64 mixed additions at the chip's throughput
    // instead of a 127-step scalar-multiplication chain), the rest as one or two direct stages on lane quads
    uint4* const d_bits = srs_bits(srs);
    const bool whole_by_bits = d_bits && srs->lagrange_of == 0 && n >= 64 && n <= 256;
    const bool first_by_bits = d_bits && srs->lagrange_of == 0 && n >= 512 && n <= G1FFT_T3_MAX;
    uint4* d_t3 = nullptr;
    uint32_t t3_points = 0;
    if (whole_by_bits || first_by_bits) {
        std::unique_lock<std::mutex> lazy(srs->lazy_mu);
        if (!srs->d_t3) {                                                  // x3 tables of the first min(SRS length, 2048) points, once per SRS (<= 33 MB)
            const uint32_t pts = (uint32_t)std::min<size_t>(srs->n, G1FFT_T3_MAX);
            const uint32_t total = 3 * 255 * pts;                              // the x3, x5, x7 tables (100 MB at 2 048 points)
            KZG_HIP_TRY(ctx, ctx->poly[0].c.reserve((size_t)total * 36 * 4));
            KZG_HIP_TRY(ctx, ctx->poly[0].a.reserve((size_t)total * NL * 4));
            uint4* t3 = nullptr;
            KZG_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&t3), (size_t)total * 64));
            hipError_t e = hipMemsetAsync(t3, 0, (size_t)total * 64, st);
            if (e == hipSuccess) {
                hipLaunchKernelGGL(k_g1fft_t3_planes, dim3((total + 255) / 256), dim3(256), 0, st, d_bits, (uint32_t)srs->n, pts, total, ctx->poly[0].c.as<int32_t>());
                const size_t lanes = (total + AFF_PER - 1) / AFF_PER;
                hipLaunchKernelGGL(k_g1fft_to_affine, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, st, ctx->poly[0].c.as<int32_t>(), total, t3, 0, ctx->poly[0].a.as<int32_t>());
                e = hipGetLastError();
            }
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e != hipSuccess) { (void)hipFree(t3); return set_error(ctx, e, "building the x3 tables of g1_ifft"); }
            srs->d_t3 = t3;
            srs->t3_n = pts;
        }
        d_t3 = srs->d_t3;
        t3_points = srs->t3_n;
    }
    if (whole_by_bits) {
        Naf2Lists nl;
        rc = get_naf2(ctx, log_n, true, &nl);
        if (rc != KZG_OK) return rc;
        // slices per term: two waves per SIMD in all (n^2 Q / 32 = 2048 waves) -- more waves only add tree additions (every wave ends in a
        // 5-level tree: a third of the work at 21 digits per pair), fewer leave lone waves at half the issue rate.  Measured: 256 points
        // 0.71 ms with Q = 4, 0.66 with Q = 1; 128 points 0.28 -> 0.26 (tools/time_g1ifft.py).  The floor is the additions themselves:
        // n^2 x 85 (5.6 M at 256 points = 0.36 ms of the chip).
        const uint32_t Q = (uint32_t)std::max<size_t>(1, 65536 / (n * n)), wpo = (uint32_t)(n * Q / 32);
        KZG_HIP_TRY(ctx, ctx->poly[0].c.reserve((size_t)n * wpo * 36 * 4));
        int32_t* partial = ctx->poly[0].c.as<int32_t>();
        hipLaunchKernelGGL(k_g1fft_bits, dim3((unsigned)((n * (size_t)wpo * 64 + 255) / 256)), dim3(256), 0, st, d_bits, (uint32_t)srs->n, d_t3, t3_points, (uint32_t)n,
                           log_n, log_n, nl.list, nl.cnt, Q, wpo, partial);
        hipLaunchKernelGGL(k_g1fft_sum_partials, dim3((unsigned)((n * 64 + 255) / 256)), dim3(256), 0, st, partial, wpo, (uint32_t)n, bufA);
        KZG_HIP_TRY(ctx, hipGetLastError());
        *result_out = bufA;
        return KZG_OK;
    }
    if (first_by_bits) {
        // Plan (bits of the stages, first one through the per-bit tables): the later stages are one 127-step GLV chain deep each, pure latency
        // while they fit one wave per SIMD (65 536 lanes = 16 384 quads = n 2^K <= 16 384).  512 = 2^4 . 2^5, 1024 = 2^6 . 2^4, 2048 = 2^5 . 2^3 . 2^3
        // (measured against 5,4 / 3,3,3 / 5,5 / 4,3,3 / 6,5 / 7,4 and against the later stages on lane pairs: tools/time_g1ifft.py, profiles/r03d_quad.md).
        int plan[4] = {0, 0, 0, 0}, np = 0;
        if (log_n == 9) { plan[0] = 4; plan[1] = 5; np = 2; }
        else if (log_n == 10) { plan[0] = 6; plan[1] = 4; np = 2; }
        else { plan[0] = 5; plan[1] = 3; plan[2] = 3; np = 3; }
        const int K0 = plan[0];
#if !defined(KZG_G1FFT_QUAD_LDS) || defined(KZG_G1FFT_QUAD_W1)
        constexpr size_t QUAD_TAB_LDS = 0;
#else
        constexpr size_t QUAD_TAB_LDS = (size_t)16 * NL * 256 * 4;     // entries 1 .. 15 of every lane (entry 0 unused): 144 KiB of the CU's 160
        static bool quad_attr_set = false;
        if (!quad_attr_set) {
            KZG_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_g1fft_mul_quads), hipFuncAttributeMaxDynamicSharedMemorySize, (int)QUAD_TAB_LDS));
            quad_attr_set = true;
        }
#endif
        Naf2Lists nl;
        rc = get_naf2(ctx, log_n, false, &nl);
        if (rc != KZG_OK) return rc;
        const size_t R0 = (size_t)1 << K0;
        const uint32_t Q = (uint32_t)std::max<size_t>(std::max<size_t>(1, 32 / R0), 65536 / (n * R0)), wpo = (uint32_t)(R0 * Q / 32);
        size_t part_points = (size_t)n * wpo;
        for (int i = 1; i < np; ++i) part_points = std::max(part_points, n << plan[i]);
        KZG_HIP_TRY(ctx, ctx->poly[0].c.reserve(part_points * 36 * 4));
        int32_t* partial = ctx->poly[0].c.as<int32_t>();
        int32_t* src = bufB;
        int32_t* dst = bufA;
        hipLaunchKernelGGL(k_g1fft_bits, dim3((unsigned)((n * (size_t)wpo * 64 + 255) / 256)), dim3(256), 0, st, d_bits, (uint32_t)srs->n, d_t3, t3_points, (uint32_t)n,
                           log_n, K0, nl.list, nl.cnt, Q, wpo, partial);
        hipLaunchKernelGGL(k_g1fft_sum_partials, dim3((unsigned)((n * 64 + 255) / 256)), dim3(256), 0, st, partial, wpo, (uint32_t)n, src);
        int done = K0;
        for (int i = 1; i < np; ++i) {
            const int K = plan[i];
            done += K;
            const int log_s = log_n - done;
            const bool last = i == np - 1;
            const size_t slots = n << K;
            hipLaunchKernelGGL(k_g1fft_mul_quads, dim3((unsigned)((4 * slots + 255) / 256)), dim3(256), QUAD_TAB_LDS, st, src, partial, (uint32_t)n, log_n, K, log_s,
                               last ? scal_n : scal, last ? 1 : 0);
            hipLaunchKernelGGL(k_g1fft_sum_partials, dim3((unsigned)((n * 64 + 255) / 256)), dim3(256), 0, st, partial, (uint32_t)1 << K, (uint32_t)n, dst);
            std::swap(src, dst);
        }
        KZG_HIP_TRY(ctx, hipGetLastError());
        *result_out = src;
        return KZG_OK;
    }
    // Stage plan.  A stage is one scalar multiplication deep whatever it computes, so the plan minimises (number of stages) x (time of a
    // stage).  Measured stage times on MI355X (tools/time_g1ifft.py, round 3) while the stage fits ONE wave per SIMD (65536 lanes):
    // 1.25 ms with one lane per point, 0.83 ms on lane pairs; beyond that a stage is throughput bound and scales with its lanes (a lone
    // wave already issues most of what its SIMD can: two pair waves per SIMD took 1.44 ms).  Candidates:
    //   direct stages of radix 2^K (one lane or pair per (output, term): n 2^K lanes or pairs), K <= 5
    //   radix-2 butterflies (n / 2 lanes or pairs, work bound: one multiplication per two outputs)
    int kmax = 0;
    bool pairs = false;
    {
        const double t_lane = 1.25, t_pair = 0.83, cap = 65536.0;
        double best = 1e300;
        for (int mode = 0; mode < 2; ++mode) {                        // 0: one lane per point, 1: lane pairs
            const double t1 = mode ? t_pair : t_lane, width = mode ? 2.0 : 1.0;
            for (int K = 2; K <= 5 && K <= std::max(log_n, 2); ++K) {  // direct stages
                const double lanes = (double)n * (double)(1u << K) * width;
                const double cost = (double)((log_n + K - 1) / K) * t1 * std::max(1.0, lanes / cap);
                if (cost < best) { best = cost; kmax = K; pairs = mode != 0; }
            }
            const double lanes2 = (double)n / 2.0 * width;             // radix-2 butterflies (+ the scaling multiplication of the last stage)
            const double cost2 = (double)(log_n + 1) * t1 * std::max(1.0, lanes2 / cap);
            if (cost2 < best) { best = cost2; kmax = 0; pairs = mode != 0; }
        }
    }
    if (log_n == 0) {
        hipLaunchKernelGGL(k_g1fft_load, dim3(gn), dim3(256), 0, st, srs->d_points, (uint32_t)n, bufA, 0);
    } else if (kmax >= 2) {
        const int stages = (log_n + kmax - 1) / kmax;
        // the first stage through the SRS window tables when the SRS has them and the stage fits four waves per SIMD
        // Prefers the narrow (c = 15) set: fewer double-and-add steps per digit.
        const uint4* tab = nullptr; int tab_c = 0, tab_W = 0;
        if (srs->lagrange_of == 0) {
            if (srs->d_small) { tab = srs->d_small; tab_c = srs->small_c; tab_W = srs->small_W; }
            else if (srs->pre_W > 0) { tab = srs->d_points; tab_c = srs->pre_c; tab_W = srs->pre_W; }
        }
        const int K0 = (log_n + stages - 1) / stages;                       // radix bits of the first stage (balanced split)
        const uint32_t wpo = tab ? (uint32_t)((((size_t)1 << K0) * tab_W + 31) / 32) : 0;
        const bool first_tables = tab && wpo <= 32 && n * (size_t)wpo <= 4096;
        if (!first_tables) hipLaunchKernelGGL(k_g1fft_load, dim3(gn), dim3(256), 0, st, srs->d_points, (uint32_t)n, bufA, 0);
        int done = 0;
        int32_t* src = bufA;
        int32_t* dst = bufB;
        for (int i = 0; i < stages; ++i) {
            const int K = (log_n - done + (stages - i) - 1) / (stages - i);   // balanced split of the remaining bits
            done += K;
            const int log_s = log_n - done;
            const bool last = i == stages - 1;
            const size_t lanes = n << K;
            if (i == 0 && first_tables) {
                const uint4* sc = nullptr;
                rc = get_scalars(ctx, log_n, last, &sc, 1);                // canonical scalars (scaled by 1/n when this is also the last stage)
                if (rc != KZG_OK) return rc;
                KZG_HIP_TRY(ctx, ctx->poly[0].c.reserve((size_t)n * wpo * 36 * 4));
                int32_t* partial = ctx->poly[0].c.as<int32_t>();
                hipLaunchKernelGGL(k_g1fft_first_tables, dim3((unsigned)((n * (size_t)wpo * 64 + 255) / 256)), dim3(256), 0, st, tab, (uint32_t)srs->n, tab_c, tab_W,
                                   (uint32_t)n, log_n, K, sc, wpo, partial);
                hipLaunchKernelGGL(k_g1fft_sum_partials, dim3((unsigned)((n * 64 + 255) / 256)), dim3(256), 0, st, partial, wpo, (uint32_t)n, dst);
            } else if (pairs)
                hipLaunchKernelGGL(k_g1fft_direct_pairs, dim3((unsigned)((2 * lanes + 255) / 256)), dim3(256), 0, st, src, dst, (uint32_t)n, log_n, K, log_s,
                                   last ? scal_n : scal, last ? 1 : 0);
            else
                hipLaunchKernelGGL(k_g1fft_direct, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, st, src, dst, (uint32_t)n, log_n, K, log_s,
                                   last ? scal_n : scal, last ? 1 : 0);
            std::swap(src, dst);
        }
        result = src;
    } else {
        hipLaunchKernelGGL(k_g1fft_load, dim3(gn), dim3(256), 0, st, srs->d_points, (uint32_t)n, bufA, log_n);
        for (int s = 1; s <= log_n; ++s) {
            const bool last = s == log_n;
            if (pairs)
                hipLaunchKernelGGL(k_g1fft_stage_pairs, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, bufA, (uint32_t)n, log_n, s,
                                   last ? scal_n : scal, last ? 1 : 0);
            else
                hipLaunchKernelGGL(k_g1fft_stage, dim3((unsigned)((n / 2 + 255) / 256)), dim3(256), 0, st, bufA, (uint32_t)n, log_n, s,
                                   last ? scal_n : scal, last ? 1 : 0);
        }
    }
    KZG_HIP_TRY(ctx, hipGetLastError());
    *result_out = result;
    return KZG_OK;
}

// Lagrange basis of the first n SRS points -> d_out (n affine points: wire format, or the device format of curve.h)
int32_t g1_ifft_device(kzg_ctx* ctx, const kzg_srs* srs, size_t n, uint4* d_out, bool wire) {
    const int32_t* result = nullptr;
    int32_t rc = g1_ifft_stages(ctx, srs, n, &result);
    if (rc != KZG_OK) return rc;
    const size_t lanes = std::max<size_t>(1, (n + AFF_PER - 1) / AFF_PER);
    const unsigned blocks = (unsigned)std::min<size_t>((lanes + 255) / 256, 4096);
    hipLaunchKernelGGL(k_g1fft_to_affine, dim3(blocks), dim3(256), 0, ctx->stream, result, (uint32_t)n, d_out, wire ? 1 : 0, ctx->poly[0].a.as<int32_t>());
    KZG_HIP_TRY(ctx, hipGetLastError());
    return KZG_OK;
}

// Up to this many points the one inversion of the affine conversion runs on the HOST (Montgomery's trick over the n points, ~20 us):
// on the device it is a 380-multiplication chain on lone lanes, 0.2 ms whatever n -- two thirds of a g1_ifft of 2..32 points.
constexpr size_t G1FFT_HOST_AFFINE_MAX = 256;

// the last context of device `dev` is gone: free the scalar tables and digit lists g1_ifft cached for it
void g1fft_release_device_caches(int dev) {
    std::lock_guard<std::mutex> lk(g_scal_mu);
    for (auto it = g_scal.begin(); it != g_scal.end();) {
        if (it->first.dev == dev) { (void)hipFree(it->second); it = g_scal.erase(it); } else ++it;
    }
    for (auto it = g_naf2.begin(); it != g_naf2.end();) {
        if (std::get<0>(it->first) == dev) { (void)hipFree(it->second.list); (void)hipFree(it->second.cnt); it = g_naf2.erase(it); } else ++it;
    }
}

int32_t g1_ifft_run(kzg_ctx* ctx, const kzg_srs* srs, size_t n, uint64_t* out_xy) {
    // (the transform as n batched MSMs of n pairs over the per-bit tables -- round 3's form at 512 points, 1.08 ms -- lost to the table first stage + one quad
    // stage, 0.68 ms, and was removed in round 6: docs/history, profiles/r03_g1ifft.txt)
    if (n <= G1FFT_HOST_AFFINE_MAX) {
        const int32_t* result = nullptr;
        int32_t rc = g1_ifft_stages(ctx, srs, n, &result);
        if (rc != KZG_OK) return rc;
        KZG_HIP_TRY(ctx, ctx->msm.bases_wire.reserve(n * 128));
        hipLaunchKernelGGL(k_g1fft_planes_to_wire, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, result, (uint32_t)n, ctx->msm.bases_wire.as<uint32_t>());
        KZG_HIP_TRY(ctx, hipGetLastError());
        static thread_local std::vector<kzg_host::Xyzz> host_pts;
        host_pts.resize(n);
        KZG_HIP_TRY(ctx, hipMemcpyAsync(host_pts.data(), ctx->msm.bases_wire.p, n * 128, hipMemcpyDeviceToHost, ctx->stream));
        KZG_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        kzg_host::xyzz_batch_to_affine(host_pts.data(), n, out_xy);
        return KZG_OK;
    }
    KZG_HIP_TRY(ctx, ctx->msm.bases_wire.reserve(n * 64));
    int32_t rc = g1_ifft_device(ctx, srs, n, ctx->msm.bases_wire.as<uint4>(), true);
    if (rc != KZG_OK) return rc;
    KZG_HIP_TRY(ctx, hipMemcpyAsync(out_xy, ctx->msm.bases_wire.p, n * 64, hipMemcpyDeviceToHost, ctx->stream));
    KZG_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return KZG_OK;
}

}  // namespace kzg
