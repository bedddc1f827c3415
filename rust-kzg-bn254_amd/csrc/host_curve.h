// host_curve.h — host-side (CPU) epilogue arithmetic of the product library.
//
// The device returns O(1) data per MSM: W window sums as wire-format XYZZ (Montgomery radix 2^256, the
// reference's in-memory format).  What is left is inherently serial and tiny: the Horner combination
// sum_w 2^(c w) S_w (<= 255 doublings), folding partial sums of several GPUs, and the single field
// inversion of `G1Projective::into_affine()` (prover/src/kzg.rs:101, :122).  It runs here on 4 x 64-bit
// limbs, beside the D2H copy.  This is NOT a CPU fallback for the MSM/NTT: no scalar or SRS data ever
// reaches these functions.
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

namespace kzg_host {

typedef unsigned __int128 u128;

struct Fq { uint64_t l[4]; };

static const Fq FQ_P = {{0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL}};
static const uint64_t FQ_INV = 0x87d20782e4866389ULL;                                   // -p^-1 mod 2^64
static const Fq FQ_ONE = {{0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL}};   // 2^256 mod p

inline bool is_zero(const Fq& a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }
inline bool eq(const Fq& a, const Fq& b) { return memcmp(&a, &b, sizeof(Fq)) == 0; }
inline bool geq_p(const Fq& a) {
    for (int i = 3; i >= 0; --i) { if (a.l[i] > FQ_P.l[i]) return true; if (a.l[i] < FQ_P.l[i]) return false; }
    return true;
}
inline void sub_p(Fq& a) {
    uint64_t br = 0;
    for (int i = 0; i < 4; ++i) { u128 d = (u128)a.l[i] - FQ_P.l[i] - br; a.l[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
}
inline Fq add(const Fq& a, const Fq& b) {
    Fq r; u128 c = 0;
    for (int i = 0; i < 4; ++i) { c += (u128)a.l[i] + b.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
    if (c || geq_p(r)) sub_p(r);
    return r;
}
inline Fq sub(const Fq& a, const Fq& b) {
    Fq r; uint64_t br = 0;
    for (int i = 0; i < 4; ++i) { u128 d = (u128)a.l[i] - b.l[i] - br; r.l[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
    if (br) { u128 c = 0; for (int i = 0; i < 4; ++i) { c += (u128)r.l[i] + FQ_P.l[i]; r.l[i] = (uint64_t)c; c >>= 64; } }
    return r;
}
inline Fq dbl(const Fq& a) { return add(a, a); }
// Montgomery product, operand scanning with interleaved reduction
inline Fq mul(const Fq& a, const Fq& b) {
    uint64_t t[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        u128 c = 0;
        for (int j = 0; j < 4; ++j) { c += (u128)a.l[j] * b.l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        u128 top = (u128)t[4] + (uint64_t)c;
        uint64_t m = t[0] * FQ_INV;
        c = ((u128)m * FQ_P.l[0] + t[0]) >> 64;
        for (int j = 1; j < 4; ++j) { c += (u128)m * FQ_P.l[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        top += (uint64_t)c;
        t[3] = (uint64_t)top; t[4] = (uint64_t)(top >> 64);
    }
    Fq r = {{t[0], t[1], t[2], t[3]}};
    if (t[4] || geq_p(r)) sub_p(r);
    return r;
}
inline Fq sqr(const Fq& a) { return mul(a, a); }
inline Fq inv_fermat(const Fq& a) {     // a^(p-2): 381 Montgomery products, ~11 us (kept as the cross-check of inv below)
    Fq e = FQ_P; e.l[0] -= 2;
    Fq acc = FQ_ONE, base = a;
    for (int i = 0; i < 254; ++i) {
        if ((e.l[i >> 6] >> (i & 63)) & 1) acc = mul(acc, base);
        base = sqr(base);
    }
    return acc;
}
// plain 256-bit helpers of the binary inversion
inline bool geq(const Fq& a, const Fq& b) {
    for (int i = 3; i >= 0; --i) { if (a.l[i] > b.l[i]) return true; if (a.l[i] < b.l[i]) return false; }
    return true;
}
inline void sub_raw(Fq& a, const Fq& b) {
    uint64_t br = 0;
    for (int i = 0; i < 4; ++i) { u128 d = (u128)a.l[i] - b.l[i] - br; a.l[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
}
inline void shr1(Fq& a) {
    for (int i = 0; i < 3; ++i) a.l[i] = (a.l[i] >> 1) | (a.l[i + 1] << 63);
    a.l[3] >>= 1;
}
inline void half_mod_p(Fq& x) {         // x / 2 mod p, x < p < 2^254 (x + p does not overflow)
    if (x.l[0] & 1) { u128 c = 0; for (int i = 0; i < 4; ++i) { c += (u128)x.l[i] + FQ_P.l[i]; x.l[i] = (uint64_t)c; c >>= 64; } }
    shr1(x);
}
// Inverse of a Montgomery-form element (a R -> a^-1 R), a != 0: binary extended Euclid on the plain integers (at most 2 x 254 shift /
// subtract steps on four words, ~2-3 us) and two Montgomery products by R^2 to get back into the form.  `into_affine()` is on the
// critical path of every commitment: the Fermat form above was 11 of the ~70 us of a 512-coefficient commitment.
inline const Fq& fq_r2() {
    static const Fq r2 = []() { Fq v = FQ_ONE; for (int i = 0; i < 256; ++i) v = add(v, v); return v; }();     // R * 2^256 = R^2 mod p
    return r2;
}
inline Fq inv(const Fq& a_in) {
    // caller-supplied words (kzg_g1_fold_partials takes XYZZ partials from the wire) may be any 256-bit value: canonicalise first.  A
    // non-zero multiple of p would otherwise reach u == v == p, u = 0, and the halving loop below would never end
    Fq a = a_in;
    for (int i = 0; i < 6 && geq(a, FQ_P); ++i) sub_raw(a, FQ_P);     // 2^256 / p < 6
    if (is_zero(a)) return a;           // (as a^(p-2) would: 0)
    Fq u = a, v = FQ_P, x1 = {{1, 0, 0, 0}}, x2 = {{0, 0, 0, 0}};
    const Fq one = {{1, 0, 0, 0}};
    for (int guard = 0; guard < 1024 && !eq(u, one) && !eq(v, one); ++guard) {     // gcd(a, p) = 1: at most 2 x 254 rounds
        while ((u.l[0] & 1) == 0) { shr1(u); half_mod_p(x1); }
        while ((v.l[0] & 1) == 0) { shr1(v); half_mod_p(x2); }
        if (geq(u, v)) { sub_raw(u, v); x1 = sub(x1, x2); }
        else { sub_raw(v, u); x2 = sub(x2, x1); }
    }
    const Fq b = eq(u, one) ? x1 : x2;  // (a R)^-1 = a^-1 R^-1 as a plain integer < p
    return mul(mul(b, fq_r2()), fq_r2());       // . R^2 R^-1 . R^2 R^-1 = a^-1 R
}

struct Xyzz { Fq x, y, zz, zzz; };          // identity: zz == 0
inline Xyzz xyzz_inf() { Xyzz r; memset(&r, 0, sizeof r); return r; }
inline bool is_inf(const Xyzz& p) { return is_zero(p.zz); }

inline Xyzz xyzz_dbl(const Xyzz& p) {       // dbl-2008-s-1
    if (is_inf(p)) return p;
    Fq u = dbl(p.y), v = sqr(u), w = mul(u, v), s = mul(p.x, v), xx = sqr(p.x);
    Fq m = add(dbl(xx), xx);
    Xyzz r;
    r.x = sub(sqr(m), dbl(s));
    r.y = sub(mul(m, sub(s, r.x)), mul(w, p.y));
    r.zz = mul(v, p.zz);
    r.zzz = mul(w, p.zzz);
    return r;
}
inline Xyzz xyzz_add(const Xyzz& a, const Xyzz& b) {   // add-2008-s
    if (is_inf(a)) return b;
    if (is_inf(b)) return a;
    Fq u1 = mul(a.x, b.zz), u2 = mul(b.x, a.zz), s1 = mul(a.y, b.zzz), s2 = mul(b.y, a.zzz);
    Fq p = sub(u2, u1), r = sub(s2, s1);
    if (is_zero(p)) return is_zero(r) ? xyzz_dbl(a) : xyzz_inf();
    Fq pp = sqr(p), ppp = mul(p, pp), q = mul(u1, pp);
    Xyzz o;
    o.x = sub(sub(sqr(r), ppp), dbl(q));
    o.y = sub(mul(r, sub(q, o.x)), mul(s1, ppp));
    o.zz = mul(mul(a.zz, b.zz), pp);
    o.zzz = mul(mul(a.zzz, b.zzz), ppp);
    return o;
}
// `into_affine()`: out_xy = x || y (wire), identity -> zeros with *is_inf = 1
inline void xyzz_to_affine(const Xyzz& p, uint64_t out_xy[8], uint8_t* is_infinity) {
    if (is_inf(p)) { memset(out_xy, 0, 64); if (is_infinity) *is_infinity = 1; return; }
    Fq i = inv(mul(p.zz, p.zzz));
    Fq x = mul(p.x, mul(i, p.zzz)), y = mul(p.y, mul(i, p.zz));
    memcpy(out_xy, x.l, 32); memcpy(out_xy + 4, y.l, 32);
    if (is_infinity) *is_infinity = 0;
}
// n points at once with ONE field inversion (Montgomery's trick): out_xy = n x (x || y) wire words, identity -> zeros
inline void xyzz_batch_to_affine(const Xyzz* p, size_t n, uint64_t* out_xy) {
    std::vector<Fq> pre(n);
    Fq run = FQ_ONE;
    for (size_t i = 0; i < n; ++i) {                     // prefix products of the denominators ZZ ZZZ (identity points contribute 1)
        pre[i] = run;
        if (!is_inf(p[i])) run = mul(run, mul(p[i].zz, p[i].zzz));
    }
    Fq rinv = inv(run);
    for (size_t i = n; i-- > 0;) {
        if (is_inf(p[i])) { memset(out_xy + 8 * i, 0, 64); continue; }
        const Fq d = mul(p[i].zz, p[i].zzz);
        const Fq iv = mul(rinv, pre[i]);                 // 1 / (ZZ ZZZ)
        rinv = mul(rinv, d);
        const Fq x = mul(p[i].x, mul(iv, p[i].zzz)), y = mul(p[i].y, mul(iv, p[i].zz));
        memcpy(out_xy + 8 * i, x.l, 32); memcpy(out_xy + 8 * i + 4, y.l, 32);
    }
}
// sum_w 2^(c w) * S_w, windows given low to high
inline Xyzz horner_windows(const Xyzz* sums, int W, int c) {
    Xyzz acc = xyzz_inf();
    for (int w = W - 1; w >= 0; --w) {
        for (int k = 0; k < c; ++k) acc = xyzz_dbl(acc);
        acc = xyzz_add(acc, sums[w]);
    }
    return acc;
}
inline Xyzz xyzz_from_affine_wire(const uint64_t xy[8]) {
    Xyzz r;
    memcpy(r.x.l, xy, 32); memcpy(r.y.l, xy + 4, 32);
    if (is_zero(r.x) && is_zero(r.y)) return xyzz_inf();
    r.zz = FQ_ONE; r.zzz = FQ_ONE;
    return r;
}

}  // namespace kzg_host
