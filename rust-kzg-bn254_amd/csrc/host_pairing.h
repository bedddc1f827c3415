// host_pairing.h — host-side BN254 pairing check for the verifier surface:
// `helpers::pairings_verify` (primitives/src/helpers.rs:392-398), `verify_proof` (verifier/src/verify.rs:10-72) and the
// final check of `verify_kzg_proof_batch` (verifier/src/batch.rs:253-254).  O(1) work per verification; the data-parallel
// part of batch verification (three n-point MSMs) runs on the GPU (kzg_msm_g1_batch).
//
// The reference calls arkworks' optimal-ate `Bn254::multi_pairing` and only looks at `result.is_zero()` (is the product
// the identity of GT).  Any bilinear non-degenerate pairing decides that predicate identically, so this file implements
// the simplest one to verify: the reduced Tate pairing  t(P, Q) = f_{r,P}(psi(Q))^((p^12 - 1)/r)  with the G1 point as the
// Miller-loop point (affine arithmetic in Fq), psi the untwist G2 -> E(Fq12), and Fq12 = Fq[w]/(w^12 - 18 w^6 + 82)
// (w^6 = 9 + u).  Two constructions: a literal one (affine Miller loop, generic square-and-multiply final exponentiation)
// and the one the library uses (shared projective Miller loop, (p^6-1)(p^2+1) easy part, base-p Straus hard part); the
// self-check build compares them.
#pragma once
#include <system_error>
#include <thread>
#include "host_curve.h"
#include "pairing_constants.h"

namespace kzg_host {

// ---- Fq helpers -------------------------------------------------------------------------------------------------
inline Fq fq_zero() { Fq z; memset(&z, 0, sizeof z); return z; }
inline Fq neg(const Fq& a) { return is_zero(a) ? a : sub(fq_zero(), a); }

// ---- Fr (scalar field) in wire form: Montgomery product, sum, and conversion to canonical integer words --------------------------------
static const uint64_t FR_INV = 0xc2e1f593efffffffULL;           // -r^-1 mod 2^64
inline bool fr_geq_r(const uint64_t t[4]) {
    for (int i = 3; i >= 0; --i) { if (t[i] != FR_MODULUS_WORDS[i]) return t[i] > FR_MODULUS_WORDS[i]; }
    return true;
}
inline void fr_sub_r(uint64_t t[4]) {
    uint64_t br = 0;
    for (int i = 0; i < 4; ++i) { u128 d = (u128)t[i] - FR_MODULUS_WORDS[i] - br; t[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
}
inline void fr_mul(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]) {
    uint64_t t[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        u128 c = 0;
        for (int j = 0; j < 4; ++j) { c += (u128)a[j] * b[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        u128 top = (u128)t[4] + (uint64_t)c;
        uint64_t m = t[0] * FR_INV;
        c = ((u128)m * FR_MODULUS_WORDS[0] + t[0]) >> 64;
        for (int j = 1; j < 4; ++j) { c += (u128)m * FR_MODULUS_WORDS[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        top += (uint64_t)c;
        t[3] = (uint64_t)top; t[4] = (uint64_t)(top >> 64);
    }
    if (t[4] || fr_geq_r(t)) fr_sub_r(t);
    memcpy(out, t, 32);
}
inline void fr_add(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]) {
    uint64_t t[4]; u128 c = 0;
    for (int i = 0; i < 4; ++i) { c += (u128)a[i] + b[i]; t[i] = (uint64_t)c; c >>= 64; }
    if (c || fr_geq_r(t)) fr_sub_r(t);
    memcpy(out, t, 32);
}
inline void fr_wire_to_canonical(const uint64_t in[4], uint64_t out[4]) {    // x 2^-256: Montgomery product with the integer 1
    const uint64_t one[4] = {1, 0, 0, 0};
    fr_mul(in, one, out);
}

// ---- G1 affine (host) -----------------------------------------------------------------------------------------------------
struct G1 { Fq x, y; bool inf; };
inline G1 g1_from_wire(const uint64_t xy[8]) {
    G1 p; memcpy(p.x.l, xy, 32); memcpy(p.y.l, xy + 4, 32); p.inf = is_zero(p.x) && is_zero(p.y); return p;
}
inline bool g1_on_curve(const G1& p) {
    if (p.inf) return true;
    return eq(sqr(p.y), add(mul(sqr(p.x), p.x), FQ_THREE));
}
inline G1 g1_neg(const G1& p) { G1 r = p; if (!p.inf) r.y = neg(p.y); return r; }
inline G1 g1_add(const G1& a, const G1& b) {
    if (a.inf) return b;
    if (b.inf) return a;
    Fq lam;
    if (eq(a.x, b.x)) {
        if (!eq(a.y, b.y) || is_zero(a.y)) { G1 r; r.x = fq_zero(); r.y = fq_zero(); r.inf = true; return r; }
        Fq xx = sqr(a.x);
        lam = mul(add(dbl(xx), xx), inv(dbl(a.y)));
    } else {
        lam = mul(sub(b.y, a.y), inv(sub(b.x, a.x)));
    }
    G1 r; r.inf = false;
    r.x = sub(sub(sqr(lam), a.x), b.x);
    r.y = sub(mul(lam, sub(a.x, r.x)), a.y);
    return r;
}
inline G1 g1_mul(const G1& p, const uint64_t k[4]) {             // XYZZ double-and-add, one inversion at the end
    G1 out; out.x = fq_zero(); out.y = fq_zero(); out.inf = true;
    if (p.inf) return out;
    Xyzz base; base.x = p.x; base.y = p.y; base.zz = FQ_ONE; base.zzz = FQ_ONE;
    Xyzz acc = xyzz_inf();
    for (int i = 255; i >= 0; --i) {
        acc = xyzz_dbl(acc);
        if ((k[i >> 6] >> (i & 63)) & 1) acc = xyzz_add(acc, base);
    }
    if (is_inf(acc)) return out;
    Fq iz = inv(mul(acc.zz, acc.zzz));
    out.x = mul(acc.x, mul(iz, acc.zzz));
    out.y = mul(acc.y, mul(iz, acc.zz));
    out.inf = false;
    return out;
}
inline void g1_to_wire(const G1& p, uint64_t xy[8]) {
    if (p.inf) { memset(xy, 0, 64); return; }
    memcpy(xy, p.x.l, 32); memcpy(xy + 4, p.y.l, 32);
}

// ---- Fq2 = Fq[u]/(u^2 + 1), G2 affine over Fq2 on y^2 = x^3 + 3/(9+u) --------------------------------------------------------
struct Fq2 { Fq c0, c1; };
inline Fq2 add(const Fq2& a, const Fq2& b) { return {add(a.c0, b.c0), add(a.c1, b.c1)}; }
inline Fq2 sub(const Fq2& a, const Fq2& b) { return {sub(a.c0, b.c0), sub(a.c1, b.c1)}; }
inline Fq2 neg(const Fq2& a) { return {neg(a.c0), neg(a.c1)}; }
inline Fq2 mul(const Fq2& a, const Fq2& b) {
    Fq t0 = mul(a.c0, b.c0), t1 = mul(a.c1, b.c1);
    return {sub(t0, t1), sub(sub(mul(add(a.c0, a.c1), add(b.c0, b.c1)), t0), t1)};
}
inline Fq2 sqr(const Fq2& a) {                    // (c0 + c1)(c0 - c1) + 2 c0 c1 u
    Fq t = mul(a.c0, a.c1);
    return {mul(add(a.c0, a.c1), sub(a.c0, a.c1)), add(t, t)};
}
inline Fq2 mul_fq(const Fq2& a, const Fq& k) { return {mul(a.c0, k), mul(a.c1, k)}; }
inline bool is_zero(const Fq2& a) { return is_zero(a.c0) && is_zero(a.c1); }
inline bool eq(const Fq2& a, const Fq2& b) { return eq(a.c0, b.c0) && eq(a.c1, b.c1); }
inline Fq2 inv(const Fq2& a) {                    // (c0 - c1 u) / (c0^2 + c1^2)
    Fq n = inv(add(sqr(a.c0), sqr(a.c1)));
    return {mul(a.c0, n), neg(mul(a.c1, n))};
}
struct G2 { Fq2 x, y; bool inf; };
inline G2 g2_inf() { G2 r; memset(&r, 0, sizeof r); r.inf = true; return r; }
inline G2 g2_from_wire(const uint64_t w[16]) {
    G2 p; memcpy(p.x.c0.l, w, 32); memcpy(p.x.c1.l, w + 4, 32); memcpy(p.y.c0.l, w + 8, 32); memcpy(p.y.c1.l, w + 12, 32);
    p.inf = is_zero(p.x) && is_zero(p.y);
    return p;
}
inline void g2_to_wire(const G2& p, uint64_t w[16]) {
    if (p.inf) { memset(w, 0, 128); return; }
    memcpy(w, p.x.c0.l, 32); memcpy(w + 4, p.x.c1.l, 32); memcpy(w + 8, p.y.c0.l, 32); memcpy(w + 12, p.y.c1.l, 32);
}
inline G2 g2_generator() { G2 g; g.x = {G2_GEN_X0, G2_GEN_X1}; g.y = {G2_GEN_Y0, G2_GEN_Y1}; g.inf = false; return g; }
inline G2 g2_tau_mainnet() { G2 g; g.x = {G2_TAU_X0, G2_TAU_X1}; g.y = {G2_TAU_Y0, G2_TAU_Y1}; g.inf = false; return g; }
inline bool g2_on_curve(const G2& p) {             // helpers.rs:264-285 is_on_curve_g2
    if (p.inf) return true;
    Fq2 b = {TWIST_B0, TWIST_B1};
    return eq(sqr(p.y), add(mul(sqr(p.x), p.x), b));
}
inline G2 g2_neg(const G2& p) { G2 r = p; if (!p.inf) r.y = neg(p.y); return r; }
inline G2 g2_add(const G2& a, const G2& b) {
    if (a.inf) return b;
    if (b.inf) return a;
    Fq2 lam;
    if (eq(a.x, b.x)) {
        if (!eq(a.y, b.y) || is_zero(a.y)) return g2_inf();
        Fq2 xx = sqr(a.x);
        lam = mul(add(add(xx, xx), xx), inv(add(a.y, a.y)));
    } else {
        lam = mul(sub(b.y, a.y), inv(sub(b.x, a.x)));
    }
    G2 r; r.inf = false;
    r.x = sub(sub(sqr(lam), a.x), b.x);
    r.y = sub(mul(lam, sub(a.x, r.x)), a.y);
    return r;
}
// Jacobian double-and-add over Fq2 (a = 0), one inversion at the end; the affine g2_add above stays for single additions
struct G2Jac { Fq2 X, Y, Z; bool inf; };
inline Fq2 dbl2(const Fq2& a) { return add(a, a); }
inline G2Jac g2j_dbl(const G2Jac& t) {                             // dbl-2009-l
    if (t.inf || is_zero(t.Y)) { G2Jac r = t; r.inf = true; return r; }
    Fq2 A = sqr(t.X), B = sqr(t.Y), C = sqr(B);
    Fq2 D = dbl2(sub(sub(sqr(add(t.X, B)), A), C));
    Fq2 E = add(dbl2(A), A);
    G2Jac r; r.inf = false;
    r.X = sub(sqr(E), dbl2(D));
    r.Y = sub(mul(E, sub(D, r.X)), dbl2(dbl2(dbl2(C))));
    r.Z = dbl2(mul(t.Y, t.Z));
    return r;
}
inline G2Jac g2j_madd(const G2Jac& t, const G2& p) {               // mixed addition with an affine point
    if (p.inf) return t;
    if (t.inf) { G2Jac r; r.X = p.x; r.Y = p.y; r.Z = {FQ_ONE, fq_zero()}; r.inf = false; return r; }
    Fq2 Z2 = sqr(t.Z);
    Fq2 H = sub(mul(p.x, Z2), t.X);
    Fq2 rr = sub(mul(p.y, mul(Z2, t.Z)), t.Y);
    if (is_zero(H)) {
        if (is_zero(rr)) return g2j_dbl(t);
        G2Jac r = t; r.inf = true; return r;
    }
    Fq2 HH = sqr(H), HHH = mul(H, HH), V = mul(t.X, HH);
    G2Jac r; r.inf = false;
    r.X = sub(sub(sqr(rr), HHH), dbl2(V));
    r.Y = sub(mul(rr, sub(V, r.X)), mul(t.Y, HHH));
    r.Z = mul(t.Z, H);
    return r;
}
inline G2 g2_mul(const G2& p, const uint64_t k[4]) {
    G2Jac acc; acc.inf = true; acc.X = {fq_zero(), fq_zero()}; acc.Y = acc.X; acc.Z = acc.X;
    for (int i = 255; i >= 0; --i) {
        acc = g2j_dbl(acc);
        if ((k[i >> 6] >> (i & 63)) & 1) acc = g2j_madd(acc, p);
    }
    if (acc.inf) return g2_inf();
    Fq2 zi = inv(acc.Z), zi2 = sqr(zi);
    G2 r; r.inf = false;
    r.x = mul(acc.X, zi2);
    r.y = mul(acc.Y, mul(zi2, zi));
    return r;
}

// ---- fixed-base multiplication of the two generators (verify.rs:37-51: [y] G1 and [z] G2 in every verification) --------------------
// 64 windows of 4 bits: k G = sum_w T_w[d_w], T_w[d] = d 2^(4w) G as affine points, so a multiplication is at most 64 mixed additions
// and no doubling (a 254-step double-and-add was 0.25 ms for G2, a fifth of verify_proof).  Tables are built on first use:
// Jacobian / XYZZ sums, normalised with ONE inversion each (Montgomery's trick).
struct FixedBaseTables {
    G1 g1[64][15];
    G2 g2[64][15];
    FixedBaseTables() {
        // G1
        {
            static Xyzz pts[64 * 15];
            Xyzz base; base.x = FQ_ONE; base.y = FQ_TWO; base.zz = FQ_ONE; base.zzz = FQ_ONE;
            for (int w = 0; w < 64; ++w) {
                Xyzz acc = base;
                for (int d = 0; d < 15; ++d) { pts[w * 15 + d] = acc; acc = xyzz_add(acc, base); }
                base = acc;                                          // 16 * base
            }
            // batch inversion of zz * zzz: 1/zz = zzz * inv, 1/zzz = zz * inv  (all points are finite: d 2^(4w) < r)
            static Fq pre[64 * 15];
            Fq run = FQ_ONE;
            for (int i = 0; i < 64 * 15; ++i) { pre[i] = run; run = mul(run, mul(pts[i].zz, pts[i].zzz)); }
            Fq iv = inv(run);
            for (int i = 64 * 15 - 1; i >= 0; --i) {
                const Fq zi = mul(iv, pre[i]);                       // 1 / (zz zzz)
                iv = mul(iv, mul(pts[i].zz, pts[i].zzz));
                G1& o = g1[i / 15][i % 15];
                o.x = mul(pts[i].x, mul(zi, pts[i].zzz));
                o.y = mul(pts[i].y, mul(zi, pts[i].zz));
                o.inf = false;
            }
        }
        // G2
        {
            static G2Jac pts[64 * 15];
            const G2 gen = g2_generator();
            G2Jac base; base.X = gen.x; base.Y = gen.y; base.Z = {FQ_ONE, fq_zero()}; base.inf = false;
            for (int w = 0; w < 64; ++w) {
                // affine copy of the window base for the mixed additions below
                const Fq2 zi = inv(base.Z), zi2 = sqr(zi);
                G2 b; b.inf = false; b.x = mul(base.X, zi2); b.y = mul(base.Y, mul(zi2, zi));
                G2Jac acc; acc.X = b.x; acc.Y = b.y; acc.Z = {FQ_ONE, fq_zero()}; acc.inf = false;
                for (int d = 0; d < 15; ++d) { pts[w * 15 + d] = acc; acc = g2j_madd(acc, b); }
                base = acc;
            }
            static Fq2 pre[64 * 15];
            Fq2 run = {FQ_ONE, fq_zero()};
            for (int i = 0; i < 64 * 15; ++i) { pre[i] = run; run = mul(run, pts[i].Z); }
            Fq2 iv = inv(run);
            for (int i = 64 * 15 - 1; i >= 0; --i) {
                const Fq2 zi = mul(iv, pre[i]), zi2 = sqr(zi);
                iv = mul(iv, pts[i].Z);
                G2& o = g2[i / 15][i % 15];
                o.x = mul(pts[i].X, zi2);
                o.y = mul(pts[i].Y, mul(zi2, zi));
                o.inf = false;
            }
        }
    }
};
inline const FixedBaseTables& fixed_base_tables() { static const FixedBaseTables t; return t; }
// k in canonical integer words, any value below 2^256 (the windows cover 256 bits; multiples of r come out as the identity through the group law)
inline G1 g1_mul_generator(const uint64_t k[4]) {
    const FixedBaseTables& t = fixed_base_tables();
    Xyzz acc = xyzz_inf();
    for (int w = 0; w < 64; ++w) {
        const int d = (int)((k[w >> 4] >> ((w & 15) * 4)) & 15);
        if (!d) continue;
        const G1& p = t.g1[w][d - 1];
        Xyzz q; q.x = p.x; q.y = p.y; q.zz = FQ_ONE; q.zzz = FQ_ONE;
        acc = xyzz_add(acc, q);
    }
    G1 out; out.x = fq_zero(); out.y = fq_zero(); out.inf = true;
    if (is_inf(acc)) return out;
    const Fq iz = inv(mul(acc.zz, acc.zzz));
    out.x = mul(acc.x, mul(iz, acc.zzz));
    out.y = mul(acc.y, mul(iz, acc.zz));
    out.inf = false;
    return out;
}
inline G2 g2_mul_generator(const uint64_t k[4]) {
    const FixedBaseTables& t = fixed_base_tables();
    G2Jac acc; acc.inf = true; acc.X = {fq_zero(), fq_zero()}; acc.Y = acc.X; acc.Z = acc.X;
    for (int w = 0; w < 64; ++w) {
        const int d = (int)((k[w >> 4] >> ((w & 15) * 4)) & 15);
        if (d) acc = g2j_madd(acc, t.g2[w][d - 1]);
    }
    if (acc.inf) return g2_inf();
    const Fq2 zi = inv(acc.Z), zi2 = sqr(zi);
    G2 r; r.inf = false;
    r.x = mul(acc.X, zi2);
    r.y = mul(acc.Y, mul(zi2, zi));
    return r;
}

// ---- Fq12 = Fq[w] / (w^12 - 18 w^6 + 82) ----------------------------------------------------------------------------------------
struct Fq12 { Fq c[12]; };
inline Fq12 fq12_one() { Fq12 r; memset(&r, 0, sizeof r); r.c[0] = FQ_ONE; return r; }
inline bool fq12_is_one(const Fq12& a) {
    if (!eq(a.c[0], FQ_ONE)) return false;
    for (int i = 1; i < 12; ++i) if (!is_zero(a.c[i])) return false;
    return true;
}
inline void fq12_reduce(Fq t[23], Fq12& r) {          // w^12 = 18 w^6 - 82
    for (int k = 22; k >= 12; --k) {
        if (is_zero(t[k])) continue;
        t[k - 6] = add(t[k - 6], mul(t[k], FQ_EIGHTEEN));
        t[k - 12] = sub(t[k - 12], mul(t[k], FQ_EIGHTYTWO));
    }
    for (int i = 0; i < 12; ++i) r.c[i] = t[i];
}
// 6 x 6 coefficient product (degree-5 polynomials), zero coefficients skipped (line functions are sparse)
inline void poly6_mul(const Fq* a, const Fq* b, Fq out[11]) {
    for (int i = 0; i < 11; ++i) out[i] = fq_zero();
    for (int i = 0; i < 6; ++i) {
        if (is_zero(a[i])) continue;
        for (int j = 0; j < 6; ++j) {
            if (is_zero(b[j])) continue;
            out[i + j] = add(out[i + j], mul(a[i], b[j]));
        }
    }
}
// One level of Karatsuba over the split f = f0 + f1 w^6: three 6 x 6 products (108 multiplications instead of 144)
inline Fq12 mul(const Fq12& a, const Fq12& b) {
    int zeros = 0;
    for (int i = 0; i < 12; ++i) zeros += is_zero(b.c[i]) + is_zero(a.c[i]);
    if (zeros >= 6) {                                      // sparse operand (Miller-loop lines): schoolbook with zero skipping is cheaper
        Fq t[23];
        for (int i = 0; i < 23; ++i) t[i] = fq_zero();
        for (int i = 0; i < 12; ++i) {
            if (is_zero(a.c[i])) continue;
            for (int j = 0; j < 12; ++j) {
                if (is_zero(b.c[j])) continue;
                t[i + j] = add(t[i + j], mul(a.c[i], b.c[j]));
            }
        }
        Fq12 r; fq12_reduce(t, r); return r;
    }
    Fq sa[6], sb[6], lo[11], hi[11], mid[11];
    for (int i = 0; i < 6; ++i) { sa[i] = add(a.c[i], a.c[i + 6]); sb[i] = add(b.c[i], b.c[i + 6]); }
    poly6_mul(a.c, b.c, lo);
    poly6_mul(a.c + 6, b.c + 6, hi);
    poly6_mul(sa, sb, mid);
    Fq t[23];
    for (int i = 0; i < 23; ++i) t[i] = fq_zero();
    for (int i = 0; i < 11; ++i) {
        t[i] = add(t[i], lo[i]);
        t[i + 12] = add(t[i + 12], hi[i]);
        t[i + 6] = add(t[i + 6], sub(sub(mid[i], lo[i]), hi[i]));
    }
    Fq12 r; fq12_reduce(t, r); return r;
}
// Squaring: the same Karatsuba split with three 6-coefficient SQUARINGS (21 multiplications each: 63 instead of 108)
inline void poly6_sqr(const Fq* a, Fq out[11]) {
    for (int i = 0; i < 11; ++i) out[i] = fq_zero();
    for (int i = 0; i < 6; ++i) {
        out[2 * i] = add(out[2 * i], sqr(a[i]));
        for (int j = i + 1; j < 6; ++j) out[i + j] = add(out[i + j], dbl(mul(a[i], a[j])));
    }
}
inline Fq12 sqr(const Fq12& a) {
    Fq sa[6], lo[11], hi[11], mid[11];
    for (int i = 0; i < 6; ++i) sa[i] = add(a.c[i], a.c[i + 6]);
    poly6_sqr(a.c, lo);
    poly6_sqr(a.c + 6, hi);
    poly6_sqr(sa, mid);
    Fq t[23];
    for (int i = 0; i < 23; ++i) t[i] = fq_zero();
    for (int i = 0; i < 11; ++i) {
        t[i] = add(t[i], lo[i]);
        t[i + 12] = add(t[i + 12], hi[i]);
        t[i + 6] = add(t[i + 6], sub(sub(mid[i], lo[i]), hi[i]));
    }
    Fq12 r; fq12_reduce(t, r); return r;
}
inline Fq12 fq12_pow(const Fq12& a, const uint64_t* e, int words) {
    Fq12 acc = fq12_one();
    bool started = false;
    for (int i = words * 64 - 1; i >= 0; --i) {
        if (started) acc = mul(acc, acc);
        if ((e[i >> 6] >> (i & 63)) & 1) { acc = started ? mul(acc, a) : a; started = true; }
    }
    return acc;
}

// untwisted coordinates of a G2 point inside Fq12:  x' = x w^2, y' = y w^3, where a + b u = (a - 9 b) + b w^6
struct UntwistedQ { Fq x2, x8, y3, y9; };             // x' = x2 w^2 + x8 w^8,  y' = y3 w^3 + y9 w^9
inline UntwistedQ untwist(const G2& q) {
    UntwistedQ r;
    r.x2 = sub(q.x.c0, mul(q.x.c1, FQ_NINE)); r.x8 = q.x.c1;
    r.y3 = sub(q.y.c0, mul(q.y.c1, FQ_NINE)); r.y9 = q.y.c1;
    return r;
}
// line through T1, T2 in G1 (slope lam) evaluated at the untwisted Q:  lam (x' - x1) - (y' - y1)
inline Fq12 line_eval(const Fq& lam, const G1& t1, const UntwistedQ& q) {
    Fq12 l; memset(&l, 0, sizeof l);
    l.c[0] = sub(t1.y, mul(lam, t1.x));
    l.c[2] = mul(lam, q.x2);
    l.c[8] = mul(lam, q.x8);
    l.c[3] = neg(q.y3);
    l.c[9] = neg(q.y9);
    return l;
}
// Miller function f_{r,P}(psi(Q)) of the Tate pairing (vertical lines dropped: they lie in Fq6 and die in the final exponentiation)
inline Fq12 miller_tate(const G1& p, const G2& q2) {
    Fq12 f = fq12_one();
    if (p.inf || q2.inf) return f;
    const UntwistedQ q = untwist(q2);
    G1 t = p;
    int top = 255;
    while (!((FR_MODULUS_WORDS[top >> 6] >> (top & 63)) & 1)) --top;
    for (int i = top - 1; i >= 0; --i) {
        // doubling step
        {
            Fq xx = sqr(t.x);
            Fq lam = mul(add(dbl(xx), xx), inv(dbl(t.y)));
            f = mul(mul(f, f), line_eval(lam, t, q));
            G1 n; n.inf = false;
            n.x = sub(sqr(lam), dbl(t.x));
            n.y = sub(mul(lam, sub(t.x, n.x)), t.y);
            t = n;
        }
        if ((FR_MODULUS_WORDS[i >> 6] >> (i & 63)) & 1) {
            if (eq(t.x, p.x)) {                       // T = -P (last step): vertical line, result is the identity
                t.inf = true;
                continue;
            }
            Fq lam = mul(sub(p.y, t.y), inv(sub(p.x, t.x)));
            f = mul(f, line_eval(lam, t, q));
            G1 n; n.inf = false;
            n.x = sub(sub(sqr(lam), t.x), p.x);
            n.y = sub(mul(lam, sub(t.x, n.x)), t.y);
            t = n;
        }
    }
    return f;
}
inline Fq12 final_exponentiation(const Fq12& f) { return fq12_pow(f, FINAL_EXP, FINAL_EXP_WORDS); }
inline Fq12 pairing(const G1& p, const G2& q) { return final_exponentiation(miller_tate(p, q)); }

// ---- fast path: shared projective Miller loop + decomposed final exponentiation ------------------------------------------------
// Product of Miller functions prod_k f_{r,P_k}(psi(Q_k)) with one squaring of f per bit for all pairs and the loop points in
// Jacobian coordinates: each line is scaled by an element of Fq (2YZ^3 for tangents, HZ for chords), which the final
// exponentiation kills, so no field inversion is needed.  Line through/at T=(X,Y,Z) evaluated at (x', y'):
//   tangent: (2Y^2 - 3X^3) + 3X^2 Z^2 x' - Z3 Z^2 y'        with Z3 = 2YZ
//   chord T,P: (Z3 yp - r xp) + r x' - Z3 y'                  with H = xp Z^2 - X, r = yp Z^3 - Y, Z3 = HZ
struct MillerPoint { Fq X, Y, Z; G1 p; UntwistedQ q; bool done; };
inline Fq12 line_from_coeffs(const Fq& c0, const Fq& cx, const Fq& cy, const UntwistedQ& q) {
    Fq12 l; memset(&l, 0, sizeof l);
    l.c[0] = c0;
    l.c[2] = mul(cx, q.x2); l.c[8] = mul(cx, q.x8);
    l.c[3] = mul(cy, q.y3); l.c[9] = mul(cy, q.y9);
    return l;
}
inline Fq12 miller_tate_product(const G1* ps, const G2* qs, int count) {
    Fq12 f = fq12_one();
    MillerPoint pts[4];
    int m = 0;
    for (int k = 0; k < count && m < 4; ++k) {
        if (ps[k].inf || qs[k].inf) continue;                   // e(O, Q) = e(P, O) = 1
        pts[m].X = ps[k].x; pts[m].Y = ps[k].y; pts[m].Z = FQ_ONE; pts[m].p = ps[k]; pts[m].q = untwist(qs[k]); pts[m].done = false;
        ++m;
    }
    if (m == 0) return f;
    int top = 255;
    while (!((FR_MODULUS_WORDS[top >> 6] >> (top & 63)) & 1)) --top;
    for (int i = top - 1; i >= 0; --i) {
        f = sqr(f);
        for (int k = 0; k < m; ++k) {
            MillerPoint& t = pts[k];
            if (t.done) continue;
            // doubling step (a = 0): dbl-2009-l
            Fq A = sqr(t.X), B = sqr(t.Y), C = sqr(B);
            Fq D = dbl(sub(sub(sqr(add(t.X, B)), A), C));
            Fq E = add(dbl(A), A);
            Fq Z2 = sqr(t.Z);
            Fq X3 = sub(sqr(E), dbl(D));
            Fq Y3 = sub(mul(E, sub(D, X3)), dbl(dbl(dbl(C))));
            Fq Z3 = dbl(mul(t.Y, t.Z));
            f = mul(f, line_from_coeffs(sub(dbl(B), mul(E, t.X)), mul(E, Z2), neg(mul(Z3, Z2)), t.q));
            t.X = X3; t.Y = Y3; t.Z = Z3;
        }
        if ((FR_MODULUS_WORDS[i >> 6] >> (i & 63)) & 1) {
            for (int k = 0; k < m; ++k) {
                MillerPoint& t = pts[k];
                if (t.done) continue;
                Fq Z2 = sqr(t.Z);
                Fq H = sub(mul(t.p.x, Z2), t.X);
                Fq r = sub(mul(t.p.y, mul(Z2, t.Z)), t.Y);
                if (is_zero(H)) { t.done = true; continue; }   // T = -P (last step, r P = O): vertical line, in Fq6
                Fq Z3 = mul(t.Z, H);
                f = mul(f, line_from_coeffs(sub(mul(Z3, t.p.y), mul(r, t.p.x)), r, neg(Z3), t.q));
                Fq HH = sqr(H), HHH = mul(H, HH), V = mul(t.X, HH);
                Fq X3 = sub(sub(sqr(r), HHH), dbl(V));
                t.Y = sub(mul(r, sub(V, X3)), mul(t.Y, HHH));
                t.X = X3; t.Z = Z3;
            }
        }
    }
    return f;
}

// f^-1 by solving (multiplication-by-f) x = 1 over Fq: 12 x 12 Gauss-Jordan elimination
inline bool fq12_inverse(const Fq12& f, Fq12& out) {
    Fq M[12][13];
    Fq12 col = f;                                               // column j = f * w^j
    for (int j = 0; j < 12; ++j) {
        for (int i = 0; i < 12; ++i) M[i][j] = col.c[i];
        Fq top = col.c[11];                                     // multiply by w: shift up, w^12 = 18 w^6 - 82
        for (int i = 11; i > 0; --i) col.c[i] = col.c[i - 1];
        col.c[0] = neg(mul(top, FQ_EIGHTYTWO));
        col.c[6] = add(col.c[6], mul(top, FQ_EIGHTEEN));
    }
    for (int i = 0; i < 12; ++i) M[i][12] = (i == 0) ? FQ_ONE : fq_zero();
    for (int c = 0; c < 12; ++c) {
        int piv = -1;
        for (int r = c; r < 12; ++r) if (!is_zero(M[r][c])) { piv = r; break; }
        if (piv < 0) return false;
        if (piv != c) for (int j = 0; j < 13; ++j) { Fq t = M[c][j]; M[c][j] = M[piv][j]; M[piv][j] = t; }
        Fq iv = inv(M[c][c]);
        for (int j = c; j < 13; ++j) M[c][j] = mul(M[c][j], iv);
        for (int r = 0; r < 12; ++r) {
            if (r == c || is_zero(M[r][c])) continue;
            Fq fac = M[r][c];
            for (int j = c; j < 13; ++j) M[r][j] = sub(M[r][j], mul(fac, M[c][j]));
        }
    }
    for (int i = 0; i < 12; ++i) out.c[i] = M[i][12];
    return true;
}

// Frobenius maps x -> x^(p^k), k = 1..3, in the polynomial basis: coefficients are fixed, (w^i)^(p^k) are precomputed
struct FrobeniusTables {
    Fq12 w[3][12];
    FrobeniusTables() {
        Fq12 wgen; memset(&wgen, 0, sizeof wgen); wgen.c[1] = FQ_ONE;
        Fq12 g = fq12_pow(wgen, FQ_MODULUS_WORDS, 4);           // w^p
        for (int k = 0; k < 3; ++k) {
            w[k][0] = fq12_one();
            for (int i = 1; i < 12; ++i) w[k][i] = mul(w[k][i - 1], g);
            if (k < 2) { Fq12 n; memset(&n, 0, sizeof n); for (int i = 0; i < 12; ++i) if (!is_zero(g.c[i])) for (int j = 0; j < 12; ++j) n.c[j] = add(n.c[j], mul(g.c[i], w[0][i].c[j])); g = n; }   // g <- g^p
        }
    }
};
inline const FrobeniusTables& frobenius_tables() { static const FrobeniusTables t; return t; }
inline Fq12 frobenius(const Fq12& a, int k /* 1..3 */) {
    const FrobeniusTables& t = frobenius_tables();
    Fq12 r; memset(&r, 0, sizeof r);
    for (int i = 0; i < 12; ++i) {
        if (is_zero(a.c[i])) continue;
        for (int j = 0; j < 12; ++j) r.c[j] = add(r.c[j], mul(a.c[i], t.w[k - 1][i].c[j]));
    }
    return r;
}
inline Fq12 conjugate_p6(const Fq12& a) {                       // x -> x^(p^6): w -> -w
    Fq12 r = a;
    for (int i = 1; i < 12; i += 2) r.c[i] = neg(r.c[i]);
    return r;
}
// f^((p^12-1)/r) = ((f^(p^6-1))^(p^2+1))^hard, hard = sum_i d_i p^i evaluated by interleaved (Straus) exponentiation of
// g, g^p, g^(p^2), g^(p^3): ~254 squarings + <= 254 multiplications by one of 15 precomputed products.
inline Fq12 final_exponentiation_fast(const Fq12& f) {
    Fq12 fi;
    if (!fq12_inverse(f, fi)) return f;                         // f = 0 cannot occur for valid inputs
    Fq12 g = mul(conjugate_p6(f), fi);                          // f^(p^6 - 1)
    g = mul(frobenius(g, 2), g);                                // ^(p^2 + 1)
    Fq12 tab[16];
    tab[0] = fq12_one();
    Fq12 base[4] = {g, frobenius(g, 1), frobenius(g, 2), frobenius(g, 3)};
    for (int msk = 1; msk < 16; ++msk) {
        int low = msk & -msk, idx = low == 1 ? 0 : low == 2 ? 1 : low == 4 ? 2 : 3;
        tab[msk] = (msk == low) ? base[idx] : mul(tab[msk ^ low], base[idx]);
    }
    Fq12 acc = fq12_one();
    bool started = false;
    for (int i = 255; i >= 0; --i) {
        if (started) acc = sqr(acc);
        int msk = 0;
        for (int d = 0; d < 4; ++d) msk |= (int)((HARD_EXP_DIGITS[d][i >> 6] >> (i & 63)) & 1) << d;
        if (msk) { acc = started ? mul(acc, tab[msk]) : tab[msk]; started = true; }
    }
    return acc;
}

// f^-1 through norms instead of a 12 x 12 elimination with twelve field inversions: g = f * f^(p^6) lies in Fq6, d = g * g^(p^2) * g^(p^4)
// in Fq2 (only the coefficients of 1 and w^6 = 9 + u are non-zero), so f^-1 = f^(p^6) * g^(p^2) * g^(p^4) * d^-1 with ONE inversion in Fq.
inline bool fq12_inverse_norm(const Fq12& f, Fq12& out) {
    const Fq12 fbar = conjugate_p6(f);
    const Fq12 g = mul(f, fbar);
    const Fq12 g2 = frobenius(g, 2), g4 = frobenius(g2, 2);
    const Fq12 t = mul(g2, g4);
    const Fq12 d = mul(g, t);                                   // = d.c[0] + d.c[6] w^6 = (d.c[0] + 9 d.c[6]) + d.c[6] u
    const Fq nine = add(dbl(dbl(dbl(d.c[6]))), d.c[6]);
    const Fq2 d2 = {add(d.c[0], nine), d.c[6]};
    if (is_zero(d2)) return false;
    const Fq2 di = inv(d2);
    Fq12 dinv; memset(&dinv, 0, sizeof dinv);
    const Fq nine_i = add(dbl(dbl(dbl(di.c1))), di.c1);
    dinv.c[0] = sub(di.c0, nine_i);
    dinv.c[6] = di.c1;
    out = mul(mul(fbar, t), dinv);
    return true;
}
// The same value with the hard part (p^4 - p^2 + 1)/r = l0 + l1 p + l2 p^2 + p^3 written in the BN parameter x = 4965661367192848881
// (l2 = 6x^2 + 1, l1 = -36x^3 - 18x^2 - 12x + 1, l0 = -36x^3 - 30x^2 - 18x - 2; Scott, Benger, Charlemagne, Dominguez Perez, Kachisa,
// "On the final exponentiation for calculating pairings on ordinary elliptic curves", 2009): three exponentiations by the 63-bit x
// (62 squarings + 27 multiplications each) and a 13-step vector addition chain, instead of 254 squarings + ~240 multiplications.
// After the easy part g lies in the cyclotomic subgroup, where the inverse is the conjugation w -> -w.
inline Fq2 mul_xi(const Fq2& a) {                               // (9 + u)(a0 + a1 u)
    Fq n0 = add(dbl(dbl(dbl(a.c0))), a.c0), n1 = add(dbl(dbl(dbl(a.c1))), a.c1);
    return {sub(n0, a.c1), add(n1, a.c0)};
}
// Squaring in the cyclotomic subgroup (where the easy part of the final exponentiation lands): Granger-Scott, "Faster squaring in
// the cyclotomic subgroup of sixth degree extensions" -- three squarings in Fq4 = Fq2[s]/(s^2 - xi), i.e. nine Fq2 squarings = 18
// multiplications in Fq instead of 63.  The element is read as six Fq2 coefficients g_i of w^i (as in mul_by_line); in the tower
// Fq12 = Fq6[w]/(w^2 - v), Fq6 = Fq2[v]/(v^3 - xi), v = w^2 they are c0 = (g0, g2, g4), c1 = (g1, g3, g5).
inline void fq4_sqr(const Fq2& a, const Fq2& b, Fq2& c0, Fq2& c1) {       // (a + b s)^2 = (a^2 + xi b^2) + 2ab s
    const Fq2 t0 = sqr(a), t1 = sqr(b);
    c0 = add(mul_xi(t1), t0);
    c1 = sub(sub(sqr(add(a, b)), t0), t1);
}
inline Fq12 cyclotomic_sqr(const Fq12& f) {
    Fq2 g[6];
    for (int i = 0; i < 6; ++i) {
        const Fq nine = add(dbl(dbl(dbl(f.c[i + 6]))), f.c[i + 6]);
        g[i] = {add(f.c[i], nine), f.c[i + 6]};
    }
    // z0 = c0.c0 = g0, z4 = c0.c1 = g2, z3 = c0.c2 = g4, z2 = c1.c0 = g1, z1 = c1.c1 = g3, z5 = c1.c2 = g5
    const Fq2 z0 = g[0], z4 = g[2], z3 = g[4], z2 = g[1], z1 = g[3], z5 = g[5];
    Fq2 t0, t1, t2, t3, t4, t5;
    fq4_sqr(z0, z1, t0, t1);
    fq4_sqr(z2, z3, t2, t3);
    fq4_sqr(z4, z5, t4, t5);
    auto three_minus_two = [](const Fq2& t, const Fq2& z) { return add(dbl2(sub(t, z)), t); };      // 3t - 2z
    auto three_plus_two = [](const Fq2& t, const Fq2& z) { return add(dbl2(add(t, z)), t); };        // 3t + 2z
    Fq2 h[6];
    h[0] = three_minus_two(t0, z0);                // z0
    h[3] = three_plus_two(t1, z1);                 // z1
    h[1] = three_plus_two(mul_xi(t5), z2);         // z2
    h[4] = three_minus_two(t4, z3);                // z3
    h[2] = three_minus_two(t2, z4);                // z4
    h[5] = three_plus_two(t3, z5);                 // z5
    Fq12 r;
    for (int i = 0; i < 6; ++i) {
        const Fq nine = add(dbl(dbl(dbl(h[i].c1))), h[i].c1);
        r.c[i] = sub(h[i].c0, nine);
        r.c[i + 6] = h[i].c1;
    }
    return r;
}
static const uint64_t BN_X = 0x44e992b44a6909f1ULL;
inline Fq12 fq12_pow_x(const Fq12& a) {
    Fq12 acc = a;
    for (int i = 61; i >= 0; --i) {                            // bit 62 is the top bit of x; `a` is in the cyclotomic subgroup
        acc = cyclotomic_sqr(acc);
        if ((BN_X >> i) & 1) acc = mul(acc, a);
    }
    return acc;
}
inline Fq12 final_exponentiation_x(const Fq12& f) {
    Fq12 fi;
    if (!fq12_inverse_norm(f, fi)) return f;                    // f = 0 cannot occur for valid inputs
    Fq12 g = mul(conjugate_p6(f), fi);                          // f^(p^6 - 1)
    g = mul(frobenius(g, 2), g);                                // ^(p^2 + 1)
    const Fq12 fx = fq12_pow_x(g), fx2 = fq12_pow_x(fx), fx3 = fq12_pow_x(fx2);
    const Fq12 y0 = mul(mul(frobenius(g, 1), frobenius(g, 2)), frobenius(g, 3));
    const Fq12 y1 = conjugate_p6(g);
    const Fq12 y2 = frobenius(fx2, 2);
    const Fq12 y3 = conjugate_p6(frobenius(fx, 1));
    const Fq12 y4 = conjugate_p6(mul(fx, frobenius(fx2, 1)));
    const Fq12 y5 = conjugate_p6(fx2);
    const Fq12 y6 = conjugate_p6(mul(fx3, frobenius(fx3, 1)));
    // y0 y1^2 y2^6 y3^12 y4^18 y5^30 y6^36
    Fq12 t0 = sqr(y6);
    t0 = mul(t0, y4);
    t0 = mul(t0, y5);
    Fq12 t1 = mul(y3, y5);
    t1 = mul(t1, t0);
    t0 = mul(t0, y2);
    t1 = sqr(t1);
    t1 = mul(t1, t0);
    t1 = sqr(t1);
    t0 = mul(t1, y1);
    t1 = mul(t1, y0);
    t0 = sqr(t0);
    return mul(t0, t1);
}

// ---- optimal ate pairing (what arkworks' Bn254::multi_pairing computes up to the representation of GT) --------------------------------
// e(P, Q) = ( f_{6x+2, Q}(P) * l_{[6x+2]Q, pi(Q)}(P) * l_{[6x+2]Q + pi(Q), -pi^2(Q)}(P) )^((p^12 - 1)/r)
// The loop runs over the 65-bit 6x + 2 (signed digits: 22 non-zero of 66) on the G2 point, in Jacobian coordinates on the twist
// E': y^2 = x^3 + 3/(9+u) over Fq2, four times shorter than the Tate loop over r above.  With the untwist (x', y') -> (x' w^2, y' w^3)
// the line through T with slope lambda, evaluated at P = (xP, yP) in G1, is  yP - lambda xP w + (lambda xT - yT) w^3  with Fq2
// coefficients; scaling a line by an element of Fq2 does not change the pairing (the final exponentiation kills Fq6), which removes
// every inversion:
//   tangent at T = (X, Y, Z):  (Z3 Z^2) yP - (3X^2 Z^2) xP w + (3X^3 - 2Y^2) w^3,          Z3 = 2YZ
//   chord through T and Q:     Z3 yP - r xP w + (r xQ - Z3 yQ) w^3,   H = xQ Z^2 - X, r = yQ Z^3 - Y, Z3 = HZ
// An Fq2 coefficient c0 + c1 u sits at w^k as (c0 - 9 c1) w^k + c1 w^(k+6)  (u = w^6 - 9).
// Formulas checked against a big-integer model (bilinearity, order r, pi(Q) = [p]Q) before they were written here, and by
// tests/hostcheck/pairingcheck.cpp against the Tate construction above.
struct AteLine { Fq2 a, b, c; };                                // a + b w + c w^3
inline void ate_dbl_step(G2Jac& T, const G1& P, AteLine& l) {
    const Fq2 A = sqr(T.X), B = sqr(T.Y), C = sqr(B);
    const Fq2 D = dbl2(sub(sub(sqr(add(T.X, B)), A), C));
    const Fq2 E = add(dbl2(A), A), Z2 = sqr(T.Z);
    const Fq2 X3 = sub(sqr(E), dbl2(D));
    const Fq2 Y3 = sub(mul(E, sub(D, X3)), dbl2(dbl2(dbl2(C))));
    const Fq2 Z3 = dbl2(mul(T.Y, T.Z));
    l.a = mul_fq(mul(Z3, Z2), P.y);
    l.b = neg(mul_fq(mul(E, Z2), P.x));
    l.c = sub(mul(E, T.X), dbl2(B));
    T.X = X3; T.Y = Y3; T.Z = Z3;
}
// Returns false in the degenerate case H == 0 (T == +-Q): it cannot happen for Q in the order-r subgroup of the twist (the loop scalar
// 6x + 2 and the Frobenius steps never bring a multiple of Q back onto +-Q), only for a caller-supplied point outside it -- which is
// merely checked to be on the curve, as in the reference.  The chord formula has no meaning there; the caller answers "not equal".
inline bool ate_add_step(G2Jac& T, const G2& Q, const G1& P, AteLine& l) {
    const Fq2 Z2 = sqr(T.Z);
    const Fq2 H = sub(mul(Q.x, Z2), T.X);
    if (is_zero(H)) return false;
    const Fq2 r = sub(mul(Q.y, mul(Z2, T.Z)), T.Y);
    const Fq2 Z3 = mul(T.Z, H);
    l.a = mul_fq(Z3, P.y);
    l.b = neg(mul_fq(r, P.x));
    l.c = sub(mul(r, Q.x), mul(Z3, Q.y));
    const Fq2 HH = sqr(H), HHH = mul(H, HH), V = mul(T.X, HH);
    const Fq2 X3 = sub(sub(sqr(r), HHH), dbl2(V));
    T.Y = sub(mul(r, sub(V, X3)), mul(T.Y, HHH));
    T.X = X3; T.Z = Z3;
    return true;
}
// f * (a + b w + c w^3): f is read as six Fq2 coefficients g_i of w^i (g_i = (c_i + 9 c_{i+6}) + c_{i+6} u), w^6 = xi = 9 + u
inline Fq12 mul_by_line(const Fq12& f, const AteLine& l) {
    Fq2 g[6], h[6];
    for (int i = 0; i < 6; ++i) {
        const Fq nine = add(dbl(dbl(dbl(f.c[i + 6]))), f.c[i + 6]);
        g[i] = {add(f.c[i], nine), f.c[i + 6]};
    }
    for (int k = 0; k < 6; ++k) {
        Fq2 t = mul(l.a, g[k]);
        const Fq2 gb = k >= 1 ? g[k - 1] : mul_xi(g[5]);          // g_{k-1} w^(k-1) * w, wrapping through w^6 = xi
        const Fq2 gc = k >= 3 ? g[k - 3] : mul_xi(g[k + 3]);
        t = add(t, mul(l.b, gb));
        t = add(t, mul(l.c, gc));
        h[k] = t;
    }
    Fq12 r;
    for (int i = 0; i < 6; ++i) {
        const Fq nine = add(dbl(dbl(dbl(h[i].c1))), h[i].c1);
        r.c[i] = sub(h[i].c0, nine);
        r.c[i + 6] = h[i].c1;
    }
    return r;
}
inline Fq2 conj(const Fq2& a) { return {a.c0, neg(a.c1)}; }
inline G2 g2_frobenius(const G2& q) {                           // pi on the twist
    G2 r; r.inf = q.inf;
    r.x = mul(conj(q.x), Fq2{TWIST_FROB_X0, TWIST_FROB_X1});
    r.y = mul(conj(q.y), Fq2{TWIST_FROB_Y0, TWIST_FROB_Y1});
    return r;
}
inline G2 g2_frobenius2_neg(const G2& q) {                      // -pi^2: (x GX2, y)   (pi^2 negates y)
    G2 r; r.inf = q.inf;
    r.x = mul_fq(q.x, TWIST_FROB2_X);
    r.y = q.y;
    return r;
}
struct AteNaf {
    int8_t d[66]; int len;
    AteNaf() {
        unsigned __int128 s = (unsigned __int128)BN_X * 6 + 2;
        len = 0;
        while (s) {
            int z = 0;
            if (s & 1) { z = 2 - (int)(s & 3); s -= (unsigned __int128)(__int128)z; }
            d[len++] = (int8_t)z;
            s >>= 1;
        }
    }
};
inline const AteNaf& ate_naf() { static const AteNaf n; return n; }
// product over the pairs of the optimal-ate Miller functions (pairs with an identity point contribute 1)
// *degenerate (optional) is set when an addition step met T == +-Q (see ate_add_step): the value returned is then meaningless
inline Fq12 miller_ate_product(const G1* ps, const G2* qs, int count, bool* degenerate = nullptr) {
    if (degenerate) *degenerate = false;
    bool bad = false;
    struct Pt { G2Jac T; G2 q, nq; G1 p; } pts[4];
    int m = 0;
    for (int k = 0; k < count && m < 4; ++k) {
        if (ps[k].inf || qs[k].inf) continue;
        pts[m].q = qs[k]; pts[m].nq = g2_neg(qs[k]); pts[m].p = ps[k];
        pts[m].T.X = qs[k].x; pts[m].T.Y = qs[k].y; pts[m].T.Z = {FQ_ONE, fq_zero()}; pts[m].T.inf = false;
        ++m;
    }
    Fq12 f = fq12_one();
    if (m == 0) return f;
    const AteNaf& naf = ate_naf();
    AteLine l;
    for (int i = naf.len - 2; i >= 0; --i) {
        f = sqr(f);
        for (int k = 0; k < m; ++k) { ate_dbl_step(pts[k].T, pts[k].p, l); f = mul_by_line(f, l); }
        if (naf.d[i]) for (int k = 0; k < m; ++k) { bad |= !ate_add_step(pts[k].T, naf.d[i] > 0 ? pts[k].q : pts[k].nq, pts[k].p, l); f = mul_by_line(f, l); }
    }
    for (int k = 0; k < m; ++k) {
        bad |= !ate_add_step(pts[k].T, g2_frobenius(pts[k].q), pts[k].p, l); f = mul_by_line(f, l);
        bad |= !ate_add_step(pts[k].T, g2_frobenius2_neg(pts[k].q), pts[k].p, l); f = mul_by_line(f, l);
    }
    if (degenerate) *degenerate = bad;
    return f;
}
inline Fq12 pairing_ate(const G1& p, const G2& q) { return final_exponentiation_x(miller_ate_product(&p, &q, 1)); }

// helpers::pairings_verify(a1, a2, b1, b2): e(a1, a2) * e(-b1, b2) == 1   (helpers.rs:392-398)
// The two Miller loops run on two host threads (the second one on a std::thread of this call) and their values are multiplied: 0.29 ms instead of
// 0.49 for the shared loop on the box's EPYC 9575F (tools/ubench/host_pairing_time.cpp; the same Fq12 value, bit for bit); the final
// exponentiation (0.38 ms) is one dependent chain.  Falls back to the shared loop when no thread can be started.
inline bool pairings_verify(const G1& a1, const G2& a2, const G1& b1, const G2& b2) {
    G1 ps[2] = {a1, g1_neg(b1)};
    G2 qs[2] = {a2, b2};
    Fq12 f;
    bool degenerate = false, threaded = false;
#if !defined(KZG_PAIRING_ONE_THREAD)
    {
        Fq12 f1, f2;
        bool d1 = false, d2 = false;
        try {
            std::thread second([&] { f2 = miller_ate_product(ps + 1, qs + 1, 1, &d2); });
            f1 = miller_ate_product(ps, qs, 1, &d1);
            second.join();
            threaded = true;
        } catch (const std::system_error&) {}          // (std::thread's constructor: nothing was started)
        if (threaded) { f = mul(f1, f2); degenerate = d1 || d2; }
    }
#endif
    if (!threaded) f = miller_ate_product(ps, qs, 2, &degenerate);
    if (degenerate) return false;                      // a G2 input outside the order-r subgroup: no pairing value to compare (header: kzg_verify_proof)
    return fq12_is_one(final_exponentiation_x(f));
}
// the same predicate through the Tate construction (self-check builds compare the two)
inline bool pairings_verify_tate(const G1& a1, const G2& a2, const G1& b1, const G2& b2) {
    G1 ps[2] = {a1, g1_neg(b1)};
    G2 qs[2] = {a2, b2};
    return fq12_is_one(final_exponentiation_x(miller_tate_product(ps, qs, 2)));
}
// the straightforward construction (affine Miller loops, generic final exponentiation), kept as the cross-check of the fast path
inline bool pairings_verify_reference(const G1& a1, const G2& a2, const G1& b1, const G2& b2) {
    Fq12 f = mul(miller_tate(a1, a2), miller_tate(g1_neg(b1), b2));
    return fq12_is_one(final_exponentiation(f));
}

}  // namespace kzg_host
