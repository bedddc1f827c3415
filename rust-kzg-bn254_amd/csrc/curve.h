// curve.h — BN254 G1 (y^2 = x^3 + 3) group law in extended Jacobian ("XYZZ") coordinates on the
// signed 9 x 29-bit field of field29.h.   x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2.
//
// This is the arithmetic behind the reference's `G1Projective::msm` call sites
// (prover/src/kzg.rs:100,121; primitives/src/helpers.rs:332).  The exceptional cases the reference's
// tests exercise are handled explicitly: identity operands (verifier/tests/tests.rs:271-311),
// P + P (duplicate points, tests.rs:343-346), P + (-P), zero scalars (zero-padded polynomials).
//
// Coordinate invariants for an XYZZ value held in registers or memory ("stored form"):
//   all four coordinates normalised;  X in (-7m, 5m),  Y in (-3m, 3m),  ZZ, ZZZ in (-m, 2m);
//   the identity is flagged by `inf` in registers and by ZZ == literal 0 in memory.
// Formula bounds are annotated inline and verified by the KZG_BOUND_CHECK host build.
#pragma once
#include "field29.h"

#if !defined(__HIPCC__)
struct uint4 { uint32_t x, y, z, w; };      // host-check build only
#endif

// The mixed addition multiplies in PAIRS of independent products with interleaved column sums (field29.h fe_mul2 / fe_sqr2):
// 128 of its 146 64-bit join adds disappear for 245 s_nop pads; same-box A/B on k_msm_accumulate: 1.211-1.220 -> 1.182-1.189 ms.
// -DKZG_NO_MADD_PAIRED restores the one-product-at-a-time form.
#if !defined(KZG_NO_MADD_PAIRED) && !defined(KZG_MADD_PAIRED)
#define KZG_MADD_PAIRED 1
#endif

namespace kzg {

struct Affine {        // unpacked affine point, coordinates canonical in [0, m) (internal Montgomery)
    Fq x, y;
};

struct Xyzz {
    Fq x, y, zz, zzz;
    bool inf;
};

KZG_HD void xyzz_set_inf(Xyzz& r) {
    fe_set_zero(r.x); fe_set_zero(r.y); fe_set_zero(r.zz); fe_set_zero(r.zzz);
    r.inf = true;
}

// r = (x, +-y) as XYZZ
KZG_HD void xyzz_from_affine(Xyzz& r, const Affine& p, uint32_t neg) {
    r.x = p.x;
    fe_cneg(r.y, p.y, neg);
    fe_norm(r.y);                              // stored form: limbs 0..7 non-negative
    fe_set_one(r.zz);
    fe_set_one(r.zzz);
    r.inf = false;
}

// Doubling of an affine point (mdbl-2008-s-1 with ZZ = ZZZ = 1).  y2 already carries the sign.
KZG_HD void xyzz_dbl_affine_impl(Xyzz& r, const Fq& x2, const Fq& y2) {
    Fq u, v, w, s, m, t, xx;
    fe_dbl(u, y2); fe_norm(u);                 // |u| < 2m
    fe_sqr(v, u);                              // V = U^2
    fe_mul(w, u, v);                           // W = U V
    fe_mul(s, x2, v);                          // S = X V
    fe_sqr(xx, x2);
    fe_add(m, xx, xx); fe_add(m, m, xx); fe_norm(m);     // M = 3 X^2, |M| < 6m
    fe_sqr(r.x, m);
    fe_sub(r.x, r.x, s); fe_sub(r.x, r.x, s); fe_norm(r.x);   // X3 = M^2 - 2S in (-4m, 4m)
    fe_sub(t, s, r.x);                         // (-5m, 6m), limbs within +-2^29
    fe_mul(t, m, t);                           // 6m * 6m = 36 m^2 < 169 m^2
    fe_mul(u, w, y2);
    fe_sub(r.y, t, u); fe_norm(r.y);           // Y3 in (-3m, 3m)
    r.zz = v;
    r.zzz = w;
    r.inf = false;
}

KZG_HD_NOINLINE void xyzz_dbl_affine(Xyzz& r, const Fq& x2, const Fq& y2) { xyzz_dbl_affine_impl(r, x2, y2); }

// Doubling of a stored XYZZ value (dbl-2008-s-1).
KZG_HD void xyzz_dbl_impl(Xyzz& r, const Xyzz& p) {
    if (p.inf) { r = p; return; }
    Fq u, v, w, s, m, t, xx;
    fe_dbl(u, p.y); fe_norm(u);                // |u| < 6m
    fe_sqr(v, u);                              // 36 m^2
    fe_mul(w, u, v);                           // 6m * 2m
    fe_mul(s, p.x, v);                         // 7m * 2m
    fe_sqr(xx, p.x);                           // 49 m^2
    fe_add(m, xx, xx); fe_add(m, m, xx); fe_norm(m);     // |M| < 6m
    Fq x3, y3;
    fe_sqr(x3, m);
    fe_sub(x3, x3, s); fe_sub(x3, x3, s); fe_norm(x3);   // (-4m, 4m)
    fe_sub(t, s, x3);
    fe_mul(t, m, t);
    fe_mul(u, w, p.y);                         // 2m * 3m
    fe_sub(y3, t, u); fe_norm(y3);
    fe_mul(r.zz, v, p.zz);
    fe_mul(r.zzz, w, p.zzz);
    r.x = x3; r.y = y3; r.inf = false;
}

KZG_HD_NOINLINE void xyzz_dbl(Xyzz& r, const Xyzz& p) { xyzz_dbl_impl(r, p); }

// Slow path of the mixed add when X1 == x2 (P == +-Q): double or cancel.
// INLINE_SLOW = true keeps the slow path inside the caller (no call: the kernel's register budget then also binds
// the slow path, which simply spills there); false calls an out-of-line copy (smaller code, callee picks its budget).
template <bool INLINE_SLOW>
KZG_HD void xyzz_madd_exceptional(Xyzz& acc, const Fq& x2, const Fq& y2s, const Fq& rr) {
    if (fe_is_zero_mod(rr)) {                                   // same point
        if (INLINE_SLOW) xyzz_dbl_affine_impl(acc, x2, y2s);
        else xyzz_dbl_affine(acc, x2, y2s);
    } else {
        xyzz_set_inf(acc);                                      // opposite points
    }
}

// acc += (neg ? -p : p), p affine and NOT the identity (callers skip identity bases).   madd-2008-s
template <bool INLINE_SLOW = false>
KZG_HD void xyzz_madd(Xyzz& acc, const Affine& p, uint32_t neg) {
    if (acc.inf) { xyzz_from_affine(acc, p, neg); return; }
    Fq y2s, u2, s2, pp_, rr_, P, R, ppp, q, t, v;
    fe_cneg(y2s, p.y, neg);
#if defined(KZG_MADD_PAIRED)
    fe_mul2(u2, p.x, acc.zz, s2, y2s, acc.zzz);
#else
    fe_mul(u2, p.x, acc.zz);                   // U2 = x2 ZZ1
    fe_mul(s2, y2s, acc.zzz);                  // S2 = y2 ZZZ1
#endif
    fe_sub(P, u2, acc.x);                      // P in (-6m, 9m), limbs within +-2^29
    fe_sub(R, s2, acc.y);                      // R in (-4m, 5m)
#if defined(KZG_MADD_PAIRED)
    fe_sqr2(pp_, P, rr_, R);
#else
    fe_sqr(pp_, P);                            // 81 m^2 < 169 m^2
    fe_sqr(rr_, R);
#endif
    if (__builtin_expect(fe_is_zero_mod(pp_), 0)) { xyzz_madd_exceptional<INLINE_SLOW>(acc, p.x, y2s, rr_); return; }
#if defined(KZG_MADD_PAIRED)
    fe_mul2(ppp, P, pp_, q, acc.x, pp_);
#else
    fe_mul(ppp, P, pp_);                       // 9m * 2m
    fe_mul(q, acc.x, pp_);                     // 7m * 2m
#endif
    fe_sub(v, rr_, ppp); fe_sub(v, v, q); fe_sub(v, v, q); fe_norm(v);   // X3 = RR - PPP - 2Q in (-7m, 5m)
    fe_sub(t, q, v);                           // (-6m, 9m), limbs within +-2^29
    fe_mulsub(acc.y, R, t, acc.y, ppp);        // Y3 = R (Q - X3) - Y1 PPP: 45 m^2 + 6 m^2, one reduction -> (-m, 2m)
    acc.x = v;
#if defined(KZG_MADD_PAIRED)
    { Fq z2, z3; fe_mul2(z2, acc.zz, pp_, z3, acc.zzz, ppp); acc.zz = z2; acc.zzz = z3; }
#else
    fe_mul(acc.zz, acc.zz, pp_);
    fe_mul(acc.zzz, acc.zzz, ppp);
#endif
}

template <bool INLINE_SLOW>
KZG_HD void xyzz_add_exceptional(Xyzz& r, const Xyzz& a, const Fq& rr) {
    if (fe_is_zero_mod(rr)) {
        if (INLINE_SLOW) xyzz_dbl_impl(r, a);
        else xyzz_dbl(r, a);
    } else {
        xyzz_set_inf(r);
    }
}

// r = a + b, both stored-form XYZZ.   add-2008-s
template <bool INLINE_SLOW = false>
KZG_HD void xyzz_add(Xyzz& r, const Xyzz& a, const Xyzz& b) {
    if (a.inf) { r = b; return; }
    if (b.inf) { r = a; return; }
    Fq u1, u2, s1, s2, P, R, pp_, rr_, ppp, q, t, v;
    fe_mul(u1, a.x, b.zz);                     // 7m * 2m
    fe_mul(u2, b.x, a.zz);
    fe_mul(s1, a.y, b.zzz);                    // 3m * 2m
    fe_mul(s2, b.y, a.zzz);
    fe_sub(P, u2, u1);                         // (-3m, 3m)
    fe_sub(R, s2, s1);
    fe_sqr(pp_, P);
    fe_sqr(rr_, R);
    if (__builtin_expect(fe_is_zero_mod(pp_), 0)) { xyzz_add_exceptional<INLINE_SLOW>(r, a, rr_); return; }
    fe_mul(ppp, P, pp_);
    fe_mul(q, u1, pp_);
    fe_sub(v, rr_, ppp); fe_sub(v, v, q); fe_sub(v, v, q); fe_norm(v);   // (-6m, 5m)
    fe_sub(t, q, v);                           // (-6m, 8m)
    Fq y3;
    fe_mulsub(y3, R, t, s1, ppp);              // Y3 = R (Q - X3) - S1 PPP, one reduction -> (-m, 2m)
    fe_mul(t, a.zz, b.zz);
    fe_mul(r.zz, t, pp_);
    fe_mul(t, a.zzz, b.zzz);
    fe_mul(r.zzz, t, ppp);
    r.x = v; r.y = y3; r.inf = false;
}

// ---------------------------------------------------------------------------------------------
// Memory formats
// ---------------------------------------------------------------------------------------------
// Device-resident affine point: 16 u32 = x[8] || y[8], canonical residues of the INTERNAL Montgomery
// form (a * 2^261 mod p), 64 B, read as four 128-bit loads.  Identity = all zero.
KZG_HD bool affine_load(Affine& p, const uint4* __restrict__ src) {
    uint4 a = src[0], b = src[1], c = src[2], d = src[3];
    uint32_t wx[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint32_t wy[8] = {c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
    fe_unpack(p.x, wx);
    fe_unpack(p.y, wy);
    uint32_t any = a.x | a.y | a.z | a.w | b.x | b.y | b.z | b.w | c.x | c.y | c.z | c.w | d.x | d.y | d.z | d.w;
    return any != 0;       // false = identity
}

// XYZZ in global memory: 36 int32 limbs, struct-of-arrays: limb k of element i at base[k * stride + i].
KZG_HD void xyzz_store(int32_t* __restrict__ base, size_t stride, size_t i, const Xyzz& v) {
    const Fq* c[4] = {&v.x, &v.y, &v.zz, &v.zzz};
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < NL; ++j) base[(size_t)(q * NL + j) * stride + i] = v.inf ? 0 : c[q]->l[j];
}
KZG_HD void xyzz_load(Xyzz& v, const int32_t* __restrict__ base, size_t stride, size_t i) {
    Fq* c[4] = {&v.x, &v.y, &v.zz, &v.zzz};
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < NL; ++j) c[q]->l[j] = base[(size_t)(q * NL + j) * stride + i];
    v.inf = fe_is_literal_zero(v.zz);
}

// wire affine (x || y, arkworks Montgomery radix 2^256, 16 u32) -> device affine format
KZG_HD void affine_wire_to_device(uint32_t out[16], const uint32_t in[16]) {
    Fq x, y;
    fe_from_wire(x, in);
    fe_from_wire(y, in + 8);
    fe_canon(x);
    fe_canon(y);
    fe_pack(out, x);
    fe_pack(out + 8, y);
}
// stored-form XYZZ -> 32 u32 wire words X || Y || ZZ || ZZZ (radix 2^256, canonical); identity = zeros
KZG_HD void xyzz_to_wire(uint32_t out[32], const Xyzz& v) {
    if (v.inf) {
#pragma unroll
        for (int j = 0; j < 32; ++j) out[j] = 0;
        return;
    }
    fe_to_wire(out, v.x);
    fe_to_wire(out + 8, v.y);
    fe_to_wire(out + 16, v.zz);
    fe_to_wire(out + 24, v.zzz);
}

}  // namespace kzg
