// fe_invert.h — modular inversion by Bernstein-Yang "safegcd" division steps on the signed 9 x 29-bit limbs of field29.h.
//
// Why: a^(m-2) is 381 DEPENDENT field products (254 squarings + 127 multiplications); on a lone lane that is ~175 us whatever the
// batch (a lone wave issues one dependent product per ~0.46 us, tools/ubench/mul_latency.hip) — the 0.19-0.21 ms of every
// k_g1fft_to_affine launch, and the one-inversion-deep part of every kernel that batches its inversions with Montgomery's trick.  The
// division-step recurrence needs no field product at all: 21 batches of 29 steps on the low limb (cheap 32-bit ops), each followed by a
// 2 x 2 integer matrix applied to (f, g) and, modulo m, to (d, e) — ~4 x 9 multiply-adds per update.  ~13 000 cheap instructions instead
// of 381 x 206, and a dependent chain an order of magnitude shorter.
//
// Algorithm and bounds: Bernstein & Yang, "Fast constant-time gcd computation and modular inversion" (2019), in the half-delta form
// (zeta = -(delta + 1/2), starting at -1) for which 590 division steps are proven enough for inputs below 2^256 (libsecp256k1's modinv32
// uses the same recurrence on 9 x 30-bit limbs with 20 x 30 = 600 steps; here 21 x 29 = 609).  The transition matrix of a batch has
// entries in [-2^29, 2^29], so every column sum below is < 2 x 2^29 x 2^29 + 2^29 x 2^29 + carry < 2^61: it fits the 64-bit accumulator.
// Control flow is the same for every lane (fixed trip counts, masks instead of branches).
//
// Replaces: nothing in the reference (arkworks inverts with its own binary-gcd variant inside `into_affine()` / `inverse()`); results
// are field elements, identical to the Fermat form bit for bit after canonicalisation — tests/test_field29_host.py::test_safegcd_inverse
// compares both on the host build of this header, tools/ubench (device) through every g1_ifft / batch-verify parity test.
#pragma once
#include "field29.h"

namespace kzg {

struct InvTrans { int32_t u, v, q, r; };       // (f, g) <- (u f + v g, q f + r g) / 2^29

// 29 division steps on the low limbs of f and g; returns the new zeta
KZG_HD int32_t inv_divsteps_29(int32_t zeta, uint32_t f0, uint32_t g0, InvTrans& t) {
    uint32_t u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int i = 0; i < LB; ++i) {
        const uint32_t mask1 = (uint32_t)(zeta >> 31);            // zeta < 0
        const uint32_t mask2 = 0u - (g & 1u);                     // g odd
        const uint32_t x = (f ^ mask1) - mask1, y = (u ^ mask1) - mask1, z = (v ^ mask1) - mask1;   // conditionally negated f, u, v
        g += x & mask2; q += y & mask2; r += z & mask2;
        const uint32_t m = mask1 & mask2;
        zeta = (int32_t)(((uint32_t)zeta ^ m) - 1u);              // zeta -> -zeta - 2 or zeta - 1
        f += g & m; u += q & m; v += r & m;
        g >>= 1; u <<= 1; v <<= 1;
    }
    t.u = (int32_t)u; t.v = (int32_t)v; t.q = (int32_t)q; t.r = (int32_t)r;
    return zeta;
}

// (f, g) <- t (f, g) / 2^29, exactly (the low 29 bits of both combinations are zero by construction); limbs stay normalised
template <class F>
KZG_HD void inv_update_fg(Fe<F>& f, Fe<F>& g, const InvTrans& t) {
    int64_t cf = (int64_t)t.u * f.l[0] + (int64_t)t.v * g.l[0];
    int64_t cg = (int64_t)t.q * f.l[0] + (int64_t)t.r * g.l[0];
    cf >>= LB; cg >>= LB;
#pragma unroll
    for (int i = 1; i < NL; ++i) {
        cf += (int64_t)t.u * f.l[i] + (int64_t)t.v * g.l[i];
        cg += (int64_t)t.q * f.l[i] + (int64_t)t.r * g.l[i];
        f.l[i - 1] = (int32_t)((uint32_t)cf & LMASK); cf >>= LB;
        g.l[i - 1] = (int32_t)((uint32_t)cg & LMASK); cg >>= LB;
    }
    f.l[NL - 1] = (int32_t)cf; g.l[NL - 1] = (int32_t)cg;
}

// (d, e) <- t (d, e) / 2^29 mod m, d and e in (-2m, m): a multiple of m is added to each combination so that it becomes divisible by 2^29
template <class F>
KZG_HD void inv_update_de(Fe<F>& d, Fe<F>& e, const InvTrans& t) {
    const int32_t sd = d.l[NL - 1] >> 31, se = e.l[NL - 1] >> 31;
    int32_t md = (t.u & sd) + (t.v & se), me = (t.q & sd) + (t.r & se);
    int64_t cd = (int64_t)t.u * d.l[0] + (int64_t)t.v * e.l[0];
    int64_t ce = (int64_t)t.q * d.l[0] + (int64_t)t.r * e.l[0];
    const uint32_t minv = (0u - F::INV) & LMASK;                  // m^-1 mod 2^29 (F::INV is -m^-1)
    md -= (int32_t)((minv * (uint32_t)cd + (uint32_t)md) & LMASK);
    me -= (int32_t)((minv * (uint32_t)ce + (uint32_t)me) & LMASK);
    cd += (int64_t)(int32_t)F::P[0] * md;
    ce += (int64_t)(int32_t)F::P[0] * me;
    cd >>= LB; ce >>= LB;
#pragma unroll
    for (int i = 1; i < NL; ++i) {
        cd += (int64_t)t.u * d.l[i] + (int64_t)t.v * e.l[i] + (int64_t)(int32_t)F::P[i] * md;
        ce += (int64_t)t.q * d.l[i] + (int64_t)t.r * e.l[i] + (int64_t)(int32_t)F::P[i] * me;
        d.l[i - 1] = (int32_t)((uint32_t)cd & LMASK); cd >>= LB;
        e.l[i - 1] = (int32_t)((uint32_t)ce & LMASK); ce >>= LB;
    }
    d.l[NL - 1] = (int32_t)cd; e.l[NL - 1] = (int32_t)ce;
}

// out = a^-1 in the internal (Montgomery, radix 2^261) form, out in (-m, 2m) normalised; a any normalised value in (-m, 2m); 0 -> 0
template <class F>
KZG_HD void fe_inverse_safegcd(Fe<F>& out, const Fe<F>& a) {
    Fe<F> f, g, d, e;
    g = a;
    fe_canon(g);                                                  // the integer a R' mod m in [0, m)
#pragma unroll
    for (int j = 0; j < NL; ++j) { f.l[j] = (int32_t)F::P[j]; d.l[j] = 0; e.l[j] = 0; }
    e.l[0] = 1;
    int32_t zeta = -1;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int b = 0; b < 21; ++b) {                                // 21 x 29 = 609 >= 590 division steps
        InvTrans t;
        zeta = inv_divsteps_29(zeta, (uint32_t)f.l[0], (uint32_t)g.l[0], t);
        inv_update_de(d, e, t);
        inv_update_fg(f, g, t);
    }
    // g == 0 and f == +-gcd = +-1 (or f == +-m for a == 0, where d == 0): the inverse of the INTEGER is sign(f) d mod m, d in (-2m, m)
    const int32_t sf = f.l[NL - 1] >> 31;                          // -1 when f is negative
    Fe<F> r;
#pragma unroll
    for (int j = 0; j < NL; ++j) r.l[j] = (d.l[j] ^ sf) - sf;      // limb-wise negation: (-m, 2m) either way
    fe_norm(r);
    // r = (a R')^-1 as a plain integer mod m; a^-1 R' = r R'^2: two products by K_PLAIN_IN = R'^2 (plain -> internal is one of them)
    Fe<F> k;
#pragma unroll
    for (int j = 0; j < NL; ++j) k.l[j] = (int32_t)F::K_PLAIN_IN[j];
    // |r| < 2m and the limbs are normalised: within fe_mul's bounds
    fe_mul(r, r, k);
    fe_mul(out, r, k);
}

}  // namespace kzg
