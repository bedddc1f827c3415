// tests/hostcheck/hostcheck.cpp — TEST-ONLY host build of the product's device math headers
// (field29.h, curve.h) compiled by g++ with -DKZG_BOUND_CHECK, so that the lazy-reduction bounds every
// formula relies on are asserted on the CPU.  Not part of libkzg_bn254_mi355x.so; loaded only by
// tests/test_field29_host.py.
#include <cstdint>
#include <cstring>
#include "field29.h"
#include "curve.h"
#include "naf.h"
#include "fe_invert.h"

using namespace kzg;

extern "C" {

// wire a, b (8 u32 each) -> wire a*b      which: 0 = Fq, 1 = Fr
void hc_mul(int which, const uint32_t* a, const uint32_t* b, uint32_t* out, int square) {
    if (which == 0) {
        Fq x, y, r; fe_from_wire(x, a); fe_from_wire(y, b);
        if (square) fe_sqr(r, x); else fe_mul(r, x, y);
        fe_to_wire(out, r);
    } else {
        Fr x, y, r; fe_from_wire(x, a); fe_from_wire(y, b);
        if (square) fe_sqr(r, x); else fe_mul(r, x, y);
        fe_to_wire(out, r);
    }
}
// lazy chain: ((a - b) + (a - b) - b) * (a + a - b), exercises signed limbs through mul
void hc_lazy(int which, const uint32_t* a, const uint32_t* b, uint32_t* out) {
    if (which == 0) {
        Fq x, y, t, u, r; fe_from_wire(x, a); fe_from_wire(y, b);
        fe_sub(t, x, y);
        fe_add(u, x, x); fe_sub(u, u, y); fe_norm(u);
        Fq t2; fe_add(t2, t, t); fe_sub(t2, t2, y); fe_norm(t2);
        fe_mul(r, t2, u); fe_to_wire(out, r);
    } else {
        Fr x, y, t, u, r; fe_from_wire(x, a); fe_from_wire(y, b);
        fe_sub(t, x, y);
        fe_add(u, x, x); fe_sub(u, u, y); fe_norm(u);
        Fr t2; fe_add(t2, t, t); fe_sub(t2, t2, y); fe_norm(t2);
        fe_mul(r, t2, u); fe_to_wire(out, r);
    }
}
// The PAIRED routines the mixed addition runs on the GPU (fe_mul2 / fe_sqr2 / fe_mulsub, field29.h) against the one-product forms,
// on raw SIGNED limb patterns (9 int32 each, no normalisation: the lazy operands of curve.h).  Returns 0 when all four pairs of
// results are limb-for-limb equal.  The KZG_BOUND_CHECK asserts of both forms run on the way.
int hc_paired_vs_single(const int32_t* a1, const int32_t* b1, const int32_t* a2, const int32_t* b2) {
    Fq x1, y1, x2, y2;
    for (int j = 0; j < NL; ++j) { x1.l[j] = a1[j]; y1.l[j] = b1[j]; x2.l[j] = a2[j]; y2.l[j] = b2[j]; }
    Fq p1, p2, q1, q2, s1, s2, t1, t2;
    fe_mul2(p1, x1, y1, p2, x2, y2);
    fe_mul(q1, x1, y1); fe_mul(q2, x2, y2);
    fe_sqr2(s1, x1, s2, y2);
    fe_sqr(t1, x1); fe_sqr(t2, y2);
    int bad = 0;
    for (int j = 0; j < NL; ++j) bad |= (p1.l[j] != q1.l[j]) | (p2.l[j] != q2.l[j]) << 1 | (s1.l[j] != t1.l[j]) << 2 | (s2.l[j] != t2.l[j]) << 3;
    return bad;
}
// wire a -> wire a^-1 by the safegcd division steps of fe_invert.h (0 -> 0); lazy != 0: the input is first made a lazy value in (-m, 2m)
// that is NOT canonical (a + m or a - m where that stays inside the range) to exercise the canonicalisation at the entry
void hc_inverse_safegcd(int which, const uint32_t* a, uint32_t* out, int lazy) {
    if (which == 0) {
        Fq x, r; fe_from_wire(x, a);
        if (lazy) { Fq one; fe_set_one(one); fe_mul(x, x, one); }       // another representative of the same residue, still in (-m, 2m)
        fe_inverse_safegcd(r, x); fe_to_wire(out, r);
    } else {
        Fr x, r; fe_from_wire(x, a);
        if (lazy) { Fr one; fe_set_one(one); fe_mul(x, x, one); }
        fe_inverse_safegcd(r, x); fe_to_wire(out, r);
    }
}
void hc_wire_to_canonical(int which, const uint32_t* a, uint32_t* out) {
    if (which == 0) fe_wire_to_canonical_words<FqParams>(out, a);
    else fe_wire_to_canonical_words<FrParams>(out, a);
}
void hc_affine_wire_to_device(const uint32_t* in, uint32_t* out) { affine_wire_to_device(out, in); }

// acc = sum_i (+-) P_i by mixed adds, in order; points in wire format (16 u32 each); sign[i] = 1 -> subtract.
// out = XYZZ wire (32 u32)
void hc_madd_chain(const uint32_t* pts_wire, const uint8_t* sign, size_t n, uint32_t* out) {
    Xyzz acc; xyzz_set_inf(acc);
    for (size_t i = 0; i < n; ++i) {
        uint32_t dev[16]; affine_wire_to_device(dev, pts_wire + 16 * i);
        uint4 v[4]; memcpy(v, dev, 64);
        Affine p;
        if (!affine_load(p, v)) continue;
        xyzz_madd(acc, p, sign[i]);
    }
    xyzz_to_wire(out, acc);
}
// tree-free pairwise test: out = (chain over first half) + (chain over second half) via xyzz_add, then doubled k times
void hc_add_halves(const uint32_t* pts_wire, const uint8_t* sign, size_t n, int doublings, uint32_t* out) {
    Xyzz a, b; xyzz_set_inf(a); xyzz_set_inf(b);
    for (size_t i = 0; i < n; ++i) {
        uint32_t dev[16]; affine_wire_to_device(dev, pts_wire + 16 * i);
        uint4 v[4]; memcpy(v, dev, 64);
        Affine p;
        if (!affine_load(p, v)) continue;
        xyzz_madd(i < n / 2 ? a : b, p, sign[i]);
    }
    Xyzz r; xyzz_add(r, a, b);
    for (int k = 0; k < doublings; ++k) { Xyzz t; xyzz_dbl(t, r); r = t; }
    // round trip through the memory format
    int32_t mem[36]; xyzz_store(mem, 1, 0, r); Xyzz s; xyzz_load(s, mem, 1, 0);
    xyzz_to_wire(out, s);
}
// running-sum shape used by the bucket reduction: out = sum_{k} (k+1) * B_k with B_k = P_k (k < n)
void hc_running_sum(const uint32_t* pts_wire, size_t n, uint32_t* out) {
    Xyzz run, acc; xyzz_set_inf(run); xyzz_set_inf(acc);
    for (size_t i = n; i-- > 0;) {
        uint32_t dev[16]; affine_wire_to_device(dev, pts_wire + 16 * i);
        uint4 v[4]; memcpy(v, dev, 64);
        Affine p; Xyzz b; xyzz_set_inf(b);
        if (affine_load(p, v)) xyzz_from_affine(b, p, 0);
        Xyzz t; xyzz_add(t, run, b); run = t;
        xyzz_add(t, acc, run); acc = t;
    }
    xyzz_to_wire(out, acc);
}
// width-w NAF digits of the canonical integer k (8 u32 words): pos / key = (|d| - 1) / 2 / sign of every digit; returns the count
int hc_naf(const uint32_t* k_words, int w, uint32_t* pos, uint32_t* key, uint32_t* neg, int cap) {
    uint32_t k[8];
    memcpy(k, k_words, 32);
    int n = 0;
    naf_for_digits(k, w, [&](uint32_t p, uint32_t kk, uint32_t s) { if (n < cap) { pos[n] = p; key[n] = kk; neg[n] = s; } ++n; });
    return n;
}
int hc_naf_max_digits(int w) { return naf_max_digits(w); }
// fe_reduce_small on a lazy value: k doublings of +-a with normalisations in between (|value| = 2^k |a| < 169 m for k <= 7), then the
// cheap reduction + fe_canon; out = wire words of the result (must equal +-2^k a mod m)
void hc_reduce_small(int which, const uint32_t* a, int k, int negate, uint32_t* out) {
    if (which == 0) {
        Fq v; fe_from_wire(v, a); if (negate) fe_neg(v, v);
        for (int i = 0; i < k; ++i) { fe_add(v, v, v); fe_norm(v); }
        fe_reduce_small(v); fe_canon(v); fe_to_wire(out, v);
    } else {
        Fr v; fe_from_wire(v, a); if (negate) fe_neg(v, v);
        for (int i = 0; i < k; ++i) { fe_add(v, v, v); fe_norm(v); }
        fe_reduce_small(v); fe_canon(v); fe_to_wire(out, v);
    }
}
}
