// tests/hostcheck/hostcheck.cpp — TEST-ONLY host build of the product's device math headers
// (field29.h, curve.h) compiled by g++ with -DKZG_BOUND_CHECK, so that the lazy-reduction bounds every
// formula relies on are asserted on the CPU.  Not part of libkzg_bn254_mi355x.so; loaded only by
// tests/test_field29_host.py.
#include <cstdint>
#include <cstring>
#include "field29.h"
#include "curve.h"

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
}
