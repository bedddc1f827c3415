// field29.h — BN254 prime-field arithmetic for gfx950 on SIGNED 9 x 29-bit limbs.
//
// Why this representation (measured, tools/ubench/valu_rates.hip on MI355X):
// v_mad_u64_u32 / v_mad_i64_i32 issue at the same rate as v_add_co_u32 / v_addc_co_u32 (~4.5 cycles
// per wave-instruction), so on this chip the cost of a 254-bit modular multiply is its INSTRUCTION
// COUNT, and carry-flag chains cost as much as the multiplies they serve.  With 29-bit limbs the
// 58-bit partial products of a whole product-scanning column (9 from a*b, 9 from m*p) fit one 64-bit
// accumulator, so a Montgomery multiply is a pure chain of 162 mads + 18 shifts + 9 (mul,and) with
// no carry flag at all, and add / sub are 9 independent 32-bit ops (the fast VALU class) with
// normalisation deferred.  Signed limbs make subtraction free of "add k*p" corrections.
//
// Value of an element: sum l[j] * 2^(29 j); Montgomery radix R' = 2^261.
//   normalised:  l[0..7] in [0, 2^29), l[8] signed (carries the sign of the value)
//   mul/sqr:     operands need |l_a| * |l_b| < 2^59.35 per limb pair and |a * b| < 2^261 * m (~169 m^2);
//                result is normalised and lies in (-m, 2m)
// The bounds each formula relies on are written next to it and are CHECKED on the host by the
// KZG_BOUND_CHECK build (tests/test_field29_host.py), which compiles this same header with g++.
//
// Replaces: the ark-ff 0.5 `Fp256<MontBackend>` arithmetic the reference calls through
// `G1Projective::msm` (prover/src/kzg.rs:100,121) and `domain.fft/ifft` (primitives/src/polynomial.rs:135,246).
#pragma once
#include <cstdint>
#include "field_constants.h"
#include "fe_asm.h"        // generated: the products below as single inline-asm statements (device pass)

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define KZG_HD __host__ __device__ __forceinline__
#define KZG_HD_NOINLINE __host__ __device__ __noinline__ inline
#else
#define KZG_HD inline
#define KZG_HD_NOINLINE inline
#endif

#if defined(KZG_BOUND_CHECK)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#endif

namespace kzg {

constexpr int NL = 9;
constexpr int LB = 29;
constexpr uint32_t LMASK = (1u << LB) - 1u;

template <class F>
struct Fe {
    int32_t l[NL];
};

#if defined(KZG_BOUND_CHECK)
template <class F>
inline long double fe_approx(const Fe<F>& a) {
    long double v = 0;
    for (int j = NL - 1; j >= 0; --j) v = v * 536870912.0L + (long double)a.l[j];
    return v;
}
template <class F>
inline long double fe_modulus_approx() {
    long double v = 0;
    for (int j = NL - 1; j >= 0; --j) v = v * 536870912.0L + (long double)F::P[j];
    return v;
}
template <class F>
inline void fe_check_mul_operands(const Fe<F>& a, const Fe<F>& b, const char* what) {
    long double ma = 0, mb = 0;
    for (int j = 0; j < NL; ++j) {
        ma = fmaxl(ma, fabsl((long double)a.l[j]));
        mb = fmaxl(mb, fabsl((long double)b.l[j]));
    }
    const long double lim = 7.3e17L;              // (2^63 - 9*2^58)/9 = 2^59.35 = 7.366e17
    if (ma * mb >= lim) { fprintf(stderr, "KZG_BOUND_CHECK: %s limb bound violated: %Lg * %Lg\n", what, ma, mb); abort(); }
    long double m = fe_modulus_approx<F>();
    long double prod = fabsl(fe_approx(a)) * fabsl(fe_approx(b));
    if (prod >= ldexpl(1.0L, 261) * m) { fprintf(stderr, "KZG_BOUND_CHECK: %s value bound violated: |a*b| / (R m) = %Lg\n", what, prod / (ldexpl(1.0L, 261) * m)); abort(); }
}
#define KZG_CHECK_MUL(a, b, what) fe_check_mul_operands(a, b, what)
#else
#define KZG_CHECK_MUL(a, b, what) ((void)0)
#endif

// The compiler tracks known-non-negative limbs (anything masked with LMASK) and, for a product of such a limb with a
// signed one, gives up v_mad_i64_i32 for a v_mad_u64_u32 + sign-correction pair (2 mads + 2 moves per product: +7 % VALU
// instructions in the mixed add).  Passing operand limbs through an empty asm hides the range without emitting code.
KZG_HD int32_t fe_opaque(int32_t x) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(KZG_NO_OPAQUE)      // -DKZG_NO_OPAQUE: A/B switch for the measurement in DESIGN.md section 4
    asm("" : "+v"(x));
#endif
    return x;
}
template <class F>
KZG_HD void fe_opaque_limbs(int32_t (&o)[NL], const Fe<F>& a) {
#pragma unroll
    for (int j = 0; j < NL; ++j) o[j] = fe_opaque(a.l[j]);
}

// One multiply-accumulate step of a column sum.  LLVM's reassociation sorts the operands of a long integer sum by rank, which
// moves the carry of the previous column (the latest value) to the END of the chain -- where it can no longer be the addend of a
// v_mad_i64_i32 and costs one 64-bit add per column (146 v_lshl_add_u64 per mixed addition, 6.5 % of its instructions).  An empty
// asm on the accumulator (-DKZG_CHAIN) keeps the additions in source order and removes those adds -- but hipcc pads every asm
// statement with an s_nop (1 215 per mixed addition), and the same-box A/B says the nops cost more than the adds:
// k_msm_accumulate 1.17-1.19 ms with the barrier, 1.10-1.11 ms without.  So the barrier is OFF by default.
KZG_HD void fe_mac(int64_t& acc, int32_t a, int32_t b) {
    acc += (int64_t)a * (int64_t)b;
#if defined(__HIP_DEVICE_COMPILE__) && defined(KZG_CHAIN)
    asm("" : "+v"(acc));
#endif
}

// ---------------------------------------------------------------------------------------------
// Montgomery product a * b * 2^-261 mod m.  Finely-integrated product scanning: column k sums
// a_j b_(k-j) and m_j p_(k-j); its low 29 bits are cancelled by m_k p_0; the rest carries on.
// ---------------------------------------------------------------------------------------------
template <class F>
KZG_HD void fe_mul(Fe<F>& r, const Fe<F>& a, const Fe<F>& b) {
    KZG_CHECK_MUL(a, b, "fe_mul");
#if defined(__HIP_DEVICE_COMPILE__) && !defined(KZG_NO_FE_ASM)
    {
        int32_t q[NL];
        fe_mul_asm<F>(q, a.l, b.l);
#pragma unroll
        for (int j = 0; j < NL; ++j) r.l[j] = q[j];
        return;
    }
#endif
    int64_t acc = 0;
    int32_t m[NL];
    int32_t out[NL];
    int32_t al[NL], bl[NL];
    fe_opaque_limbs(al, a);
    fe_opaque_limbs(bl, b);
#pragma unroll
    for (int k = 0; k < NL; ++k) {
#pragma unroll
        for (int j = 0; j <= k; ++j) fe_mac(acc, al[j], bl[k - j]);
#pragma unroll
        for (int j = 0; j < k; ++j) fe_mac(acc, m[j], (int32_t)F::P[k - j]);
        m[k] = (int32_t)(((uint32_t)acc * F::INV) & LMASK);
        fe_mac(acc, m[k], (int32_t)F::P[0]);
        acc >>= LB;
    }
#pragma unroll
    for (int k = NL; k < 2 * NL - 1; ++k) {
#pragma unroll
        for (int j = k - NL + 1; j < NL; ++j) fe_mac(acc, al[j], bl[k - j]);
#pragma unroll
        for (int j = k - NL + 1; j < NL; ++j) fe_mac(acc, m[j], (int32_t)F::P[k - j]);
        out[k - NL] = (int32_t)((uint32_t)acc & LMASK);
        acc >>= LB;
    }
    out[NL - 1] = (int32_t)acc;
#pragma unroll
    for (int j = 0; j < NL; ++j) r.l[j] = out[j];
}

// The same product for the LATENCY-bound kernels (a lone wave per SIMD: bucket reductions, G1 FFT, small polynomial kernels).  A lone
// wave issues a DEPENDENT instruction only every ~8 cycles, an independent one every 4: fe_mul above is one chain of ~206.  Here the 17
// column sums of a b are 17 independent chains (product scanning, no reduction inside), and the Montgomery reduction walks the columns
// afterwards: its own chain is ~half as long and the product chains fill the issue slots beside it.  Same value, same output range.
template <class F>
KZG_HD void fe_mul_ilp(Fe<F>& r, const Fe<F>& a, const Fe<F>& b) {
    KZG_CHECK_MUL(a, b, "fe_mul_ilp");
    int32_t al[NL], bl[NL];
    fe_opaque_limbs(al, a);
    fe_opaque_limbs(bl, b);
    int64_t T[2 * NL - 1];
#pragma unroll
    for (int k = 0; k < 2 * NL - 1; ++k) {
        int64_t t = 0;
#pragma unroll
        for (int j = (k < NL ? 0 : k - NL + 1); j <= (k < NL ? k : NL - 1); ++j) t += (int64_t)al[j] * (int64_t)bl[k - j];
        T[k] = t;
    }
    int64_t acc = 0;
    int32_t m[NL];
    int32_t out[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        int64_t mp = 0;                                  // the terms of the earlier m: off the critical path
#pragma unroll
        for (int j = 0; j + 1 < k; ++j) mp += (int64_t)m[j] * (int64_t)(int32_t)F::P[k - j];
        acc += T[k] + mp;
        if (k >= 1) acc += (int64_t)m[k - 1] * (int64_t)(int32_t)F::P[1];
        m[k] = (int32_t)(((uint32_t)acc * F::INV) & LMASK);
        acc += (int64_t)m[k] * (int64_t)(int32_t)F::P[0];
        acc >>= LB;
    }
#pragma unroll
    for (int k = NL; k < 2 * NL - 1; ++k) {
        int64_t mp = 0;                                  // all m known: independent of the carry chain
#pragma unroll
        for (int j = k - NL + 1; j < NL; ++j) mp += (int64_t)m[j] * (int64_t)(int32_t)F::P[k - j];
        acc += T[k] + mp;
        out[k - NL] = (int32_t)((uint32_t)acc & LMASK);
        acc >>= LB;
    }
    out[NL - 1] = (int32_t)acc;
#pragma unroll
    for (int j = 0; j < NL; ++j) r.l[j] = out[j];
}

// ---------------------------------------------------------------------------------------------
// TWO independent Montgomery products with their column sums interleaved statement by statement (used by xyzz_madd, curve.h):
// with the chain barrier of fe_mac on BOTH accumulators every column is one chain of mads that starts from the carry (no 64-bit
// join add), and the instruction after a barrier belongs to the OTHER product, so hipcc's one-state pad behind an asm statement
// is not needed.
// ---------------------------------------------------------------------------------------------
KZG_HD void fe_mac_b(int64_t& acc, int32_t a, int32_t b) {
    acc += (int64_t)a * (int64_t)b;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+v"(acc));
#endif
}
template <class F>
KZG_HD void fe_mul2(Fe<F>& r1, const Fe<F>& a1, const Fe<F>& b1, Fe<F>& r2, const Fe<F>& a2, const Fe<F>& b2) {
    KZG_CHECK_MUL(a1, b1, "fe_mul2");
    KZG_CHECK_MUL(a2, b2, "fe_mul2");
#if defined(__HIP_DEVICE_COMPILE__) && !defined(KZG_NO_FE_ASM)
    {
        int32_t q1[NL], q2[NL];
        fe_mul2_asm<F>(q1, q2, a1.l, b1.l, a2.l, b2.l);
#pragma unroll
        for (int j = 0; j < NL; ++j) { r1.l[j] = q1[j]; r2.l[j] = q2[j]; }
        return;
    }
#endif
    int64_t acc1 = 0, acc2 = 0;
    int32_t m1[NL], m2[NL], o1[NL], o2[NL];
    int32_t al1[NL], bl1[NL], al2[NL], bl2[NL];
    fe_opaque_limbs(al1, a1); fe_opaque_limbs(bl1, b1);
    fe_opaque_limbs(al2, a2); fe_opaque_limbs(bl2, b2);
#pragma unroll
    for (int k = 0; k < NL; ++k) {
#pragma unroll
        for (int j = 0; j <= k; ++j) { fe_mac_b(acc1, al1[j], bl1[k - j]); fe_mac_b(acc2, al2[j], bl2[k - j]); }
#pragma unroll
        for (int j = 0; j < k; ++j) { fe_mac_b(acc1, m1[j], (int32_t)F::P[k - j]); fe_mac_b(acc2, m2[j], (int32_t)F::P[k - j]); }
        m1[k] = (int32_t)(((uint32_t)acc1 * F::INV) & LMASK);
        m2[k] = (int32_t)(((uint32_t)acc2 * F::INV) & LMASK);
        fe_mac_b(acc1, m1[k], (int32_t)F::P[0]);
        fe_mac_b(acc2, m2[k], (int32_t)F::P[0]);
        acc1 >>= LB;
        acc2 >>= LB;
    }
#pragma unroll
    for (int k = NL; k < 2 * NL - 1; ++k) {
#pragma unroll
        for (int j = k - NL + 1; j < NL; ++j) { fe_mac_b(acc1, al1[j], bl1[k - j]); fe_mac_b(acc2, al2[j], bl2[k - j]); }
#pragma unroll
        for (int j = k - NL + 1; j < NL; ++j) { fe_mac_b(acc1, m1[j], (int32_t)F::P[k - j]); fe_mac_b(acc2, m2[j], (int32_t)F::P[k - j]); }
        o1[k - NL] = (int32_t)((uint32_t)acc1 & LMASK);
        o2[k - NL] = (int32_t)((uint32_t)acc2 & LMASK);
        acc1 >>= LB;
        acc2 >>= LB;
    }
    o1[NL - 1] = (int32_t)acc1;
    o2[NL - 1] = (int32_t)acc2;
#pragma unroll
    for (int j = 0; j < NL; ++j) { r1.l[j] = o1[j]; r2.l[j] = o2[j]; }
}

template <class F>
KZG_HD void fe_sqr2(Fe<F>& r1, const Fe<F>& a1, Fe<F>& r2, const Fe<F>& a2) {
    KZG_CHECK_MUL(a1, a1, "fe_sqr2");
    KZG_CHECK_MUL(a2, a2, "fe_sqr2");
#if defined(__HIP_DEVICE_COMPILE__) && !defined(KZG_NO_FE_ASM)
    {
        int32_t q1[NL], q2[NL], e1[NL], e2[NL];
#pragma unroll
        for (int j = 0; j < NL; ++j) { e1[j] = a1.l[j] * 2; e2[j] = a2.l[j] * 2; }
        fe_sqr2_asm<F>(q1, q2, a1.l, e1, a2.l, e2);
#pragma unroll
        for (int j = 0; j < NL; ++j) { r1.l[j] = q1[j]; r2.l[j] = q2[j]; }
        return;
    }
#endif
    int64_t acc1 = 0, acc2 = 0;
    int32_t m1[NL], m2[NL], o1[NL], o2[NL];
    int32_t d1[NL], d2[NL], al1[NL], al2[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) { d1[j] = fe_opaque(a1.l[j] * 2); d2[j] = fe_opaque(a2.l[j] * 2); }
    fe_opaque_limbs(al1, a1);
    fe_opaque_limbs(al2, a2);
#pragma unroll
    for (int k = 0; k < NL; ++k) {
#pragma unroll
        for (int j = 0; 2 * j < k; ++j) { fe_mac_b(acc1, d1[j], al1[k - j]); fe_mac_b(acc2, d2[j], al2[k - j]); }
        if ((k & 1) == 0) { fe_mac_b(acc1, al1[k / 2], al1[k / 2]); fe_mac_b(acc2, al2[k / 2], al2[k / 2]); }
#pragma unroll
        for (int j = 0; j < k; ++j) { fe_mac_b(acc1, m1[j], (int32_t)F::P[k - j]); fe_mac_b(acc2, m2[j], (int32_t)F::P[k - j]); }
        m1[k] = (int32_t)(((uint32_t)acc1 * F::INV) & LMASK);
        m2[k] = (int32_t)(((uint32_t)acc2 * F::INV) & LMASK);
        fe_mac_b(acc1, m1[k], (int32_t)F::P[0]);
        fe_mac_b(acc2, m2[k], (int32_t)F::P[0]);
        acc1 >>= LB;
        acc2 >>= LB;
    }
#pragma unroll
    for (int k = NL; k < 2 * NL - 1; ++k) {
#pragma unroll
        for (int j = k - NL + 1; 2 * j < k; ++j) { fe_mac_b(acc1, d1[j], al1[k - j]); fe_mac_b(acc2, d2[j], al2[k - j]); }
        if ((k & 1) == 0) { fe_mac_b(acc1, al1[k / 2], al1[k / 2]); fe_mac_b(acc2, al2[k / 2], al2[k / 2]); }
#pragma unroll
        for (int j = k - NL + 1; j < NL; ++j) { fe_mac_b(acc1, m1[j], (int32_t)F::P[k - j]); fe_mac_b(acc2, m2[j], (int32_t)F::P[k - j]); }
        o1[k - NL] = (int32_t)((uint32_t)acc1 & LMASK);
        o2[k - NL] = (int32_t)((uint32_t)acc2 & LMASK);
        acc1 >>= LB;
        acc2 >>= LB;
    }
    o1[NL - 1] = (int32_t)acc1;
    o2[NL - 1] = (int32_t)acc2;
#pragma unroll
    for (int j = 0; j < NL; ++j) { r1.l[j] = o1[j]; r2.l[j] = o2[j]; }
}

// Fused (a*b - c*d) * 2^-261 mod m with ONE Montgomery reduction (saves 81 mads + 9 mul_lo against two fe_mul and a
// subtraction).  Needs all four operands' limbs within +-2^29 (27 * 2^58 < 2^63 per column) and |a*b| + |c*d| < 2^261 m.
// Result normalised, in (-m, 2m).
template <class F>
KZG_HD void fe_mulsub(Fe<F>& r, const Fe<F>& a, const Fe<F>& b, const Fe<F>& c, const Fe<F>& d) {
#if defined(KZG_BOUND_CHECK)
    {
        long double mx = 0;
        for (int j = 0; j < NL; ++j) {
            mx = fmaxl(mx, fmaxl(fabsl((long double)a.l[j]) * fabsl((long double)b.l[j]), 0.0L));
        }
        long double ma = 0, mb = 0, mc = 0, md = 0;
        for (int j = 0; j < NL; ++j) {
            ma = fmaxl(ma, fabsl((long double)a.l[j])); mb = fmaxl(mb, fabsl((long double)b.l[j]));
            mc = fmaxl(mc, fabsl((long double)c.l[j])); md = fmaxl(md, fabsl((long double)d.l[j]));
        }
        if (9.0L * (ma * mb + mc * md) + 9.0L * 288230376151711744.0L >= 9223372036854775807.0L) {
            fprintf(stderr, "KZG_BOUND_CHECK: fe_mulsub limb bound violated: %Lg %Lg %Lg %Lg\n", ma, mb, mc, md); abort();
        }
        long double m = fe_modulus_approx<F>();
        long double prod = fabsl(fe_approx(a)) * fabsl(fe_approx(b)) + fabsl(fe_approx(c)) * fabsl(fe_approx(d));
        if (prod >= ldexpl(1.0L, 261) * m) { fprintf(stderr, "KZG_BOUND_CHECK: fe_mulsub value bound violated\n"); abort(); }
    }
#endif
#if defined(__HIP_DEVICE_COMPILE__) && !defined(KZG_NO_FE_ASM)
    {
        int32_t q[NL], nc[NL];
#pragma unroll
        for (int j = 0; j < NL; ++j) nc[j] = -c.l[j];
        fe_mulsub_asm<F>(q, a.l, b.l, nc, d.l);
#pragma unroll
        for (int j = 0; j < NL; ++j) r.l[j] = q[j];
        return;
    }
#endif
    int64_t acc = 0;
    int32_t m[NL];
    int32_t out[NL];
    int32_t al[NL], bl[NL], ncl[NL], dl[NL];
    fe_opaque_limbs(al, a);
    fe_opaque_limbs(bl, b);
    fe_opaque_limbs(dl, d);
#pragma unroll
    for (int j = 0; j < NL; ++j) ncl[j] = fe_opaque(-c.l[j]);
#pragma unroll
    for (int k = 0; k < NL; ++k) {
#pragma unroll
        for (int j = 0; j <= k; ++j) {
            fe_mac(acc, al[j], bl[k - j]);
            fe_mac(acc, ncl[j], dl[k - j]);
        }
#pragma unroll
        for (int j = 0; j < k; ++j) fe_mac(acc, m[j], (int32_t)F::P[k - j]);
        m[k] = (int32_t)(((uint32_t)acc * F::INV) & LMASK);
        fe_mac(acc, m[k], (int32_t)F::P[0]);
        acc >>= LB;
    }
#pragma unroll
    for (int k = NL; k < 2 * NL - 1; ++k) {
#pragma unroll
        for (int j = k - NL + 1; j < NL; ++j) {
            fe_mac(acc, al[j], bl[k - j]);
            fe_mac(acc, ncl[j], dl[k - j]);
        }
#pragma unroll
        for (int j = k - NL + 1; j < NL; ++j) fe_mac(acc, m[j], (int32_t)F::P[k - j]);
        out[k - NL] = (int32_t)((uint32_t)acc & LMASK);
        acc >>= LB;
    }
    out[NL - 1] = (int32_t)acc;
#pragma unroll
    for (int j = 0; j < NL; ++j) r.l[j] = out[j];
}

// Montgomery square: the symmetric products are taken once against the doubled limb.
template <class F>
KZG_HD void fe_sqr(Fe<F>& r, const Fe<F>& a) {
    KZG_CHECK_MUL(a, a, "fe_sqr");
#if defined(__HIP_DEVICE_COMPILE__) && !defined(KZG_NO_FE_ASM)
    {
        int32_t q[NL], e[NL];
#pragma unroll
        for (int j = 0; j < NL; ++j) e[j] = a.l[j] * 2;
        fe_sqr_asm<F>(q, a.l, e);
#pragma unroll
        for (int j = 0; j < NL; ++j) r.l[j] = q[j];
        return;
    }
#endif
    int64_t acc = 0;
    int32_t m[NL];
    int32_t out[NL];
    int32_t a2[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) a2[j] = fe_opaque(a.l[j] * 2);
    int32_t al[NL];
    fe_opaque_limbs(al, a);
#pragma unroll
    for (int k = 0; k < NL; ++k) {
#pragma unroll
        for (int j = 0; 2 * j < k; ++j) fe_mac(acc, a2[j], al[k - j]);
        if ((k & 1) == 0) fe_mac(acc, al[k / 2], al[k / 2]);
#pragma unroll
        for (int j = 0; j < k; ++j) fe_mac(acc, m[j], (int32_t)F::P[k - j]);
        m[k] = (int32_t)(((uint32_t)acc * F::INV) & LMASK);
        fe_mac(acc, m[k], (int32_t)F::P[0]);
        acc >>= LB;
    }
#pragma unroll
    for (int k = NL; k < 2 * NL - 1; ++k) {
#pragma unroll
        for (int j = k - NL + 1; 2 * j < k; ++j) fe_mac(acc, a2[j], al[k - j]);
        if ((k & 1) == 0) fe_mac(acc, al[k / 2], al[k / 2]);
#pragma unroll
        for (int j = k - NL + 1; j < NL; ++j) fe_mac(acc, m[j], (int32_t)F::P[k - j]);
        out[k - NL] = (int32_t)((uint32_t)acc & LMASK);
        acc >>= LB;
    }
    out[NL - 1] = (int32_t)acc;
#pragma unroll
    for (int j = 0; j < NL; ++j) r.l[j] = out[j];
}

// ---------------------------------------------------------------------------------------------
// Lazy limb-wise add / sub / neg / small multiples (no carry, no reduction), normalisation.
// ---------------------------------------------------------------------------------------------
template <class F>
KZG_HD void fe_add(Fe<F>& r, const Fe<F>& a, const Fe<F>& b) {
#pragma unroll
    for (int j = 0; j < NL; ++j) r.l[j] = a.l[j] + b.l[j];
}
template <class F>
KZG_HD void fe_sub(Fe<F>& r, const Fe<F>& a, const Fe<F>& b) {
#pragma unroll
    for (int j = 0; j < NL; ++j) r.l[j] = a.l[j] - b.l[j];
}
template <class F>
KZG_HD void fe_neg(Fe<F>& r, const Fe<F>& a) {
#pragma unroll
    for (int j = 0; j < NL; ++j) r.l[j] = -a.l[j];
}
// r = neg ? -a : a   (neg is 0 / 1)
template <class F>
KZG_HD void fe_cneg(Fe<F>& r, const Fe<F>& a, uint32_t neg) {
    int32_t s = -(int32_t)neg;                 // 0 or -1
#pragma unroll
    for (int j = 0; j < NL; ++j) r.l[j] = (a.l[j] ^ s) - s;
}
template <class F>
KZG_HD void fe_dbl(Fe<F>& r, const Fe<F>& a) {
#pragma unroll
    for (int j = 0; j < NL; ++j) r.l[j] = a.l[j] * 2;
}
template <class F>
KZG_HD void fe_set_zero(Fe<F>& r) {
#pragma unroll
    for (int j = 0; j < NL; ++j) r.l[j] = 0;
}
template <class F>
KZG_HD void fe_set_one(Fe<F>& r) {
#pragma unroll
    for (int j = 0; j < NL; ++j) r.l[j] = (int32_t)F::ONE[j];
}
template <class F>
KZG_HD void fe_select(Fe<F>& r, bool c, const Fe<F>& a, const Fe<F>& b) {   // r = c ? a : b
#pragma unroll
    for (int j = 0; j < NL; ++j) r.l[j] = c ? a.l[j] : b.l[j];
}
// Signed carry propagation: limbs 0..7 -> [0, 2^29), limb 8 keeps the sign.  Value unchanged.
template <class F>
KZG_HD void fe_norm(Fe<F>& a) {
#pragma unroll
    for (int j = 0; j < NL - 1; ++j) {
        int32_t c = a.l[j] >> LB;
        a.l[j] &= (int32_t)LMASK;
        a.l[j + 1] += c;
    }
}

// a is a NORMALISED value in (-m, 2m) (any fe_mul / fe_sqr result): a == 0 mod m  <=>  a in {0, m}
template <class F>
KZG_HD bool fe_is_zero_mod(const Fe<F>& a) {
    uint32_t z0 = 0, zp = 0;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        z0 |= (uint32_t)a.l[j];
        zp |= (uint32_t)a.l[j] ^ F::P[j];
    }
    return (z0 == 0) | (zp == 0);
}
template <class F>
KZG_HD bool fe_is_literal_zero(const Fe<F>& a) {
    uint32_t z0 = 0;
#pragma unroll
    for (int j = 0; j < NL; ++j) z0 |= (uint32_t)a.l[j];
    return z0 == 0;
}

// normalised a in (-m, 2m)  ->  canonical [0, m)
template <class F>
KZG_HD void fe_canon(Fe<F>& a) {
    Fe<F> t;
    bool neg = a.l[NL - 1] < 0;
#pragma unroll
    for (int j = 0; j < NL; ++j) t.l[j] = a.l[j] + (neg ? (int32_t)F::P[j] : 0);
    fe_norm(t);
    Fe<F> u;
#pragma unroll
    for (int j = 0; j < NL; ++j) u.l[j] = t.l[j] - (int32_t)F::P[j];
    fe_norm(u);
    bool ge = u.l[NL - 1] >= 0;
#pragma unroll
    for (int j = 0; j < NL; ++j) a.l[j] = ge ? u.l[j] : t.l[j];
}

// any lazy value with |a| < 169 m  ->  normalised representative in (-m, 2m)  (mont_mul by R' mod m)
template <class F>
KZG_HD void fe_reduce(Fe<F>& a) {
    Fe<F> one;
    fe_set_one(one);
    fe_norm(a);
    fe_mul(a, a, one);
}

// The same reduction WITHOUT a Montgomery product (no factor involved): subtract q m with q = floor(top limb / (P[8] + 1)), an
// under-estimate of a / m by less than 1.0001, so the result lies in [0, 1.0001 m): ~70 instructions instead of the 206 of fe_mul
// (the last pass of the Fr NTT reduces every output once).  |a| < 169 m as for fe_reduce (the top limb stays inside int32).
template <class F>
KZG_HD void fe_reduce_small(Fe<F>& a) {
    fe_norm(a);
    const int32_t D = (int32_t)F::P[NL - 1] + 1;
    const int32_t top = a.l[NL - 1];
    const int32_t q = (top >= 0 ? top : top - (D - 1)) / D;            // floor(top / D)
#pragma unroll
    for (int j = 0; j < NL - 1; ++j) {
        const int64_t prod = (int64_t)q * (int64_t)(int32_t)F::P[j];   // |.| < 2^37
        a.l[j] -= (int32_t)(prod & (int64_t)LMASK);
        a.l[j + 1] -= (int32_t)(prod >> LB);
    }
    a.l[NL - 1] -= q * (int32_t)F::P[NL - 1];
    fe_norm(a);
}

// ---------------------------------------------------------------------------------------------
// 256-bit words <-> limbs.  w[8] little-endian u32 words of a non-negative integer < 2^256.
// ---------------------------------------------------------------------------------------------
template <class F>
KZG_HD void fe_unpack(Fe<F>& r, const uint32_t w[8]) {
    r.l[0] = (int32_t)(w[0] & LMASK);
#pragma unroll
    for (int j = 1; j < 8; ++j) r.l[j] = (int32_t)(((w[j - 1] >> (32 - 3 * j)) | (w[j] << (3 * j))) & LMASK);
    r.l[8] = (int32_t)(w[7] >> 8);
}
// a canonical (limbs in [0, 2^29), value < 2^256)
template <class F>
KZG_HD void fe_pack(uint32_t w[8], const Fe<F>& a) {
#pragma unroll
    for (int k = 0; k < 8; ++k) w[k] = ((uint32_t)a.l[k] >> (3 * k)) | ((uint32_t)a.l[k + 1] << (29 - 3 * k));
}

// wire (arkworks Montgomery, a * 2^256 mod m, canonical) -> internal (a * 2^261), result in (-m, 2m)
template <class F>
KZG_HD void fe_from_wire(Fe<F>& r, const uint32_t w[8]) {
    Fe<F> t, k;
    fe_unpack(t, w);
#pragma unroll
    for (int j = 0; j < NL; ++j) k.l[j] = (int32_t)F::K_IN[j];
    fe_mul(r, t, k);
}
// internal (normalised, |a| < 169 m) -> wire words (canonical)
template <class F>
KZG_HD void fe_to_wire(uint32_t w[8], const Fe<F>& a) {
    Fe<F> t, k;
#pragma unroll
    for (int j = 0; j < NL; ++j) k.l[j] = (int32_t)F::K_OUT[j];
    fe_mul(t, a, k);
    fe_canon(t);
    fe_pack(w, t);
}
// wire -> canonical integer words (the `into_bigint()` of the reference's scalars)
template <class F>
KZG_HD void fe_wire_to_canonical_words(uint32_t out[8], const uint32_t w[8]) {
    Fe<F> t, k;
    fe_unpack(t, w);
    fe_set_zero(k);
    k.l[0] = 32;                                  // a*2^256 * 2^5 * 2^-261 = a
    fe_mul(t, t, k);
    fe_canon(t);
    fe_pack(out, t);
}

using Fq = Fe<FqParams>;
using Fr = Fe<FrParams>;

}  // namespace kzg
