// curve_pair.h — ONE XYZZ point spread over a PAIR of adjacent lanes, for the latency-bound kernels (bucket reduction,
// G1 FFT stages): the even lane holds (X, ZZ), the odd lane (Y, ZZZ).
//
// Why: a full XYZZ addition is 14 field multiplications in a row on one lane (8.7 us on a lone wave, which issues at half the
// rate of a shared SIMD); the reductions behind every MSM are ~15 such additions one after the other.  The formulas split
// almost symmetrically between the x-side and the y-side of the point (add-2008-s):
//     even lane                              odd lane
//     U1 = X1 ZZ2,  U2 = X2 ZZ1,  T = ZZ1 ZZ2     S1 = Y1 ZZZ2,  S2 = Y2 ZZZ1,  T' = ZZZ1 ZZZ2      (the same three products of (u, v))
//     P = U2 - U1,  PP = P^2                      R = S2 - S1,   RR = R^2
//            -- exchange (P, PP) <-> (R, RR) --
//     Q = U1 PP,    ZZ3 = T PP                    PPP = P PP,    ZZZ3 = T' PPP
//            -- PPP -> even --
//     X3 = RR - PPP - 2Q,  W1 = R (Q - X3)        W2 = S1 PPP
//            -- W1 -> odd --
//                                                 Y3 = W1 - W2
// so each lane multiplies 7 times instead of 14, both lanes run ONE instruction stream (the operands that differ are picked with
// v_cndmask), and the three exchanges are DPP quad_perm moves (lane ^ 1: VALU, no LDS round trip).  A wave then carries 32
// points and twice as many waves are resident per SIMD -- which also gives the SIMD back the issue slots a lone wave cannot use.
// The exceptional cases (P == 0: doubling or cancellation) are detected on the even lane and handled, wave-uniformly, by rebuilding
// the whole points in both lanes and running the one-lane formula (curve.h).
//
// Value ranges: as in curve.h ("stored form"): X in (-7m, 5m), Y in (-3m, 3m), ZZ / ZZZ in (-m, 2m), all normalised.
#pragma once
#include "curve.h"

#if defined(__HIPCC__)
namespace kzg {

struct HalfXyzz {      // even lane: u = X, v = ZZ;  odd lane: u = Y, v = ZZZ;  `inf` is the same in both lanes
    Fq u, v;
    bool inf;
};

__device__ __forceinline__ void half_set_inf(HalfXyzz& h) { fe_set_zero(h.u); fe_set_zero(h.v); h.inf = true; }

// value of the other lane of the pair (lane ^ 1): DPP quad_perm [1, 0, 3, 2]
__device__ __forceinline__ int32_t pair_swap(int32_t x) { return __builtin_amdgcn_mov_dpp(x, 0xB1, 0xF, 0xF, true); }
__device__ __forceinline__ void fe_pair_swap(Fq& r, const Fq& a) {
#pragma unroll
    for (int j = 0; j < NL; ++j) r.l[j] = pair_swap(a.l[j]);
}

// r = 2 a (dbl-2008-s-1), a not the identity:
//     even lane                                   odd lane
//     XX = X^2,  M = 3 XX,  MM = M^2              U = 2 Y,  V = U^2,  W = U V
//            -- V -> even --
//     S = X V,   ZZ3 = ZZ V                       WY = W Y,  ZZZ3 = ZZZ W        (u times c, v times c with c = V | W)
//     X3 = MM - 2 S,  D = M (S - X3)
//            -- D -> odd --
//                                                 Y3 = D - WY
// five multiplications per lane instead of nine.
__device__ __forceinline__ void pair_dbl(HalfXyzz& r, const HalfXyzz& a, bool odd) {
    Fq s1, p1, p2, q2, c, oc;
    fe_dbl(s1, a.u); fe_norm(s1);              // odd: U = 2 Y, |U| < 6m
    fe_select(s1, odd, s1, a.u);               // even: X in (-7m, 5m)
    fe_sqr(p1, s1);                            // XX (49 m^2) | V (36 m^2)
    Fq m;
    fe_add(m, p1, p1); fe_add(m, m, p1); fe_norm(m);   // even: M = 3 XX, |M| < 6m
    fe_select(p2, odd, s1, m);                 // M | U
    fe_select(q2, odd, p1, m);                 // M | V
    fe_mul(c, p2, q2);                         // MM | W = U V
    fe_pair_swap(oc, p1);                      // even: V
    Fq k;
    fe_select(k, odd, c, oc);                  // V | W
    Fq su, sv;
    fe_mul2(su, a.u, k, sv, a.v, k);           // S = X V | WY = W Y;   ZZ3 = ZZ V | ZZZ3 = ZZZ W
    Fq x3, t, d, od, y3;
    fe_sub(x3, c, su); fe_sub(x3, x3, su); fe_norm(x3);    // even: X3 = MM - 2 S in (-4m, 4m)
    fe_sub(t, su, x3);                         // even: S - X3 in (-5m, 6m)
    fe_mul(d, m, t);                           // even: M (S - X3)     (odd: unused)
    fe_pair_swap(od, d);
    fe_sub(y3, od, su); fe_norm(y3);           // odd: Y3 = D - WY in (-3m, 3m)
    fe_select(r.u, odd, y3, x3);
    r.v = sv;
    r.inf = false;
}

// r = a + b.  Every lane of the pair must be active together (callers branch on pair-uniform conditions only).
__device__ __forceinline__ void pair_add(HalfXyzz& r, const HalfXyzz& a, const HalfXyzz& b, bool odd) {
    if (__all(a.inf || b.inf)) {               // nothing to add in this wave (first partial of a bucket, empty groups): copy
        if (a.inf) r = b; else r = a;
        return;
    }
    // what survives of the operands: the one to return when the other is the identity, and the point to double when both are the
    // same point (then `a` itself)
    HalfXyzz keep;
    fe_select(keep.u, a.inf, b.u, a.u);
    fe_select(keep.v, a.inf, b.v, a.v);
    keep.inf = a.inf && b.inf;
    const bool any_inf = a.inf || b.inf;
    Fq m1, m2, m3, df, sq, odf, osq;
    fe_mul2(m1, a.u, b.v, m2, b.u, a.v);       // U1 | S1,  U2 | S2     (7m * 2m, 3m * 2m)
    fe_mul(m3, a.v, b.v);                      // ZZ1 ZZ2 | ZZZ1 ZZZ2
    fe_sub(df, m2, m1);                        // P | R in (-3m, 3m)
    fe_sqr(sq, df);                            // PP | RR
    const int zero_here = (!any_inf && fe_is_zero_mod(sq)) ? 1 : 0;
    const int zero_there = pair_swap(zero_here);
    const bool exc = odd ? zero_there != 0 : zero_here != 0;        // P == 0: X1 / ZZ1 == X2 / ZZ2, the same or opposite points
    const bool same = exc && (odd ? zero_here != 0 : zero_there != 0);   // and R == 0: the same point
    fe_pair_swap(odf, df);
    fe_pair_swap(osq, sq);
    Fq e1, e2, ta, f, tb, ota;
    fe_select(e1, odd, odf, m1);
    fe_select(e2, odd, osq, sq);
    fe_mul(ta, e1, e2);                        // Q = U1 PP | PPP = P PP
    fe_select(f, odd, ta, sq);
    fe_mul(tb, m3, f);                         // ZZ3 | ZZZ3
    fe_pair_swap(ota, ta);                     // even: PPP
    Fq x3, t, g1, g2, w, ow, y3;
    fe_sub(x3, osq, ota); fe_sub(x3, x3, ta); fe_sub(x3, x3, ta); fe_norm(x3);      // even: X3 = RR - PPP - 2Q in (-7m, 5m)
    fe_sub(t, ta, x3);                         // even: Q - X3 in (-6m, 9m), limbs within +-2^29
    fe_select(g1, odd, m1, odf);
    fe_select(g2, odd, ta, t);
    fe_mul(w, g1, g2);                         // W1 = R (Q - X3) | W2 = S1 PPP
    fe_pair_swap(ow, w);
    fe_sub(y3, ow, w); fe_norm(y3);            // odd: Y3 = W1 - W2 in (-3m, 3m)
    fe_select(r.u, odd, y3, x3);
    r.v = tb;
    r.inf = false;
    if (__builtin_expect(__any(exc), 0)) {     // some pair of this wave doubles or cancels
        HalfXyzz d;
        pair_dbl(d, keep, odd);
        if (exc) {
            if (same) r = d;
            else half_set_inf(r);
        }
    }
    if (any_inf) r = keep;
}

// acc += (neg ? -p : p) for an AFFINE point p that is not the identity: the even lane passes c = x, the odd lane c = y (canonical, as
// unpacked from the device affine format).  madd-2008-s split like pair_add: five multiplications per lane instead of ten.
//     even lane                              odd lane
//     U2 = x2 ZZ1,  P = U2 - X1,  PP = P^2       S2 = y2 ZZZ1,  R = S2 - Y1,  RR = R^2
//            -- exchange (P, PP) <-> (R, RR) --
//     Q = X1 PP,    ZZ3 = ZZ1 PP                 PPP = P PP,    ZZZ3 = ZZZ1 PPP
//            -- PPP -> even --
//     X3 = RR - PPP - 2Q,  W1 = R (Q - X3)       W2 = Y1 PPP
//            -- W1 -> odd --                     Y3 = W1 - W2
__device__ __forceinline__ void pair_madd(HalfXyzz& r, const HalfXyzz& acc, const Fq& c, uint32_t neg, bool odd) {
    HalfXyzz from;                             // the point itself as XYZZ (ZZ = ZZZ = 1)
    {
        Fq cs;
        fe_cneg(cs, c, odd ? neg : 0u);
        fe_norm(cs);
        from.u = cs;
        fe_set_one(from.v);
        from.inf = false;
    }
    if (__all(acc.inf)) { r = from; return; }
    Fq m, df, sq, odf, osq;
    fe_mul(m, from.u, acc.v);                  // U2 | S2
    fe_sub(df, m, acc.u);                      // P in (-6m, 9m) | R in (-4m, 5m): limbs within +-2^29
    fe_sqr(sq, df);                            // PP | RR (81 m^2)
    const int zero_here = (!acc.inf && fe_is_zero_mod(sq)) ? 1 : 0;
    const int zero_there = pair_swap(zero_here);
    const bool exc = odd ? zero_there != 0 : zero_here != 0;
    const bool same = exc && (odd ? zero_here != 0 : zero_there != 0);
    fe_pair_swap(odf, df);
    fe_pair_swap(osq, sq);
    Fq e1, e2, ta, f, tb, ota;
    fe_select(e1, odd, odf, acc.u);
    fe_select(e2, odd, osq, sq);
    fe_mul(ta, e1, e2);                        // Q = X1 PP (7m * 2m) | PPP = P PP (9m * 2m)
    fe_select(f, odd, ta, sq);
    fe_mul(tb, acc.v, f);                      // ZZ3 | ZZZ3
    fe_pair_swap(ota, ta);                     // even: PPP
    Fq x3, t, g1, g2, w, ow, y3;
    fe_sub(x3, osq, ota); fe_sub(x3, x3, ta); fe_sub(x3, x3, ta); fe_norm(x3);      // even: X3 in (-7m, 5m)
    fe_sub(t, ta, x3);                         // even: Q - X3 in (-6m, 9m)
    fe_select(g1, odd, acc.u, odf);            // R | Y1
    fe_select(g2, odd, ta, t);                 // Q - X3 | PPP
    fe_mul(w, g1, g2);                         // W1 = R (Q - X3) (5m * 9m) | W2 = Y1 PPP (3m * 2m)
    fe_pair_swap(ow, w);
    fe_sub(y3, ow, w); fe_norm(y3);            // odd: Y3 in (-3m, 3m)
    fe_select(r.u, odd, y3, x3);
    r.v = tb;
    r.inf = false;
    if (__builtin_expect(__any(exc), 0)) {     // the bucket already holds +-p
        HalfXyzz d;
        pair_dbl(d, from, odd);
        if (exc) {
            if (same) r = d;
            else half_set_inf(r);
        }
    }
    if (acc.inf) r = from;
}

// ---- memory: the struct-of-arrays XYZZ layout of curve.h (36 limb planes), each lane touching its half ----------------------
__device__ __forceinline__ void half_load(HalfXyzz& h, const int32_t* __restrict__ base, size_t stride, size_t i, bool odd) {
    const int q = odd ? 1 : 0;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        h.u.l[j] = base[(size_t)(q * NL + j) * stride + i];
        h.v.l[j] = base[(size_t)((2 + q) * NL + j) * stride + i];
    }
    h.inf = fe_is_literal_zero(h.v);            // the identity is stored as all zeros; a finite point has ZZ, ZZZ != 0
}
__device__ __forceinline__ void half_store(int32_t* __restrict__ base, size_t stride, size_t i, const HalfXyzz& h, bool odd) {
    const int q = odd ? 1 : 0;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        base[(size_t)(q * NL + j) * stride + i] = h.inf ? 0 : h.u.l[j];
        base[(size_t)((2 + q) * NL + j) * stride + i] = h.inf ? 0 : h.v.l[j];
    }
}
// 32 u32 wire words X || Y || ZZ || ZZZ of point i: the even lane converts and writes X and ZZ, the odd lane Y and ZZZ
__device__ __forceinline__ void half_store_wire(uint32_t* __restrict__ out_wire, size_t i, const HalfXyzz& h, bool odd) {
    uint32_t wu[8], wv[8];
    if (h.inf) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { wu[j] = 0; wv[j] = 0; }
    } else {
        fe_to_wire(wu, h.u);
        fe_to_wire(wv, h.v);
    }
    uint32_t* pu = out_wire + i * 32 + (odd ? 8 : 0);
    uint32_t* pv = out_wire + i * 32 + (odd ? 24 : 16);
    *reinterpret_cast<uint4*>(pu) = make_uint4(wu[0], wu[1], wu[2], wu[3]);
    *reinterpret_cast<uint4*>(pu + 4) = make_uint4(wu[4], wu[5], wu[6], wu[7]);
    *reinterpret_cast<uint4*>(pv) = make_uint4(wv[0], wv[1], wv[2], wv[3]);
    *reinterpret_cast<uint4*>(pv + 4) = make_uint4(wv[4], wv[5], wv[6], wv[7]);
}
// the half of the lane `d` lanes up (d even: the same role)
__device__ __forceinline__ void half_shfl_down(HalfXyzz& r, const HalfXyzz& h, int d) {
#pragma unroll
    for (int j = 0; j < NL; ++j) { r.u.l[j] = __shfl_down(h.u.l[j], d, 64); r.v.l[j] = __shfl_down(h.v.l[j], d, 64); }
    r.inf = __shfl_down((int)h.inf, d, 64) != 0;
}
// the half of lane ^ d (d even: the same role)
__device__ __forceinline__ void half_shfl_xor(HalfXyzz& r, const HalfXyzz& h, int d) {
#pragma unroll
    for (int j = 0; j < NL; ++j) { r.u.l[j] = __shfl_xor(h.u.l[j], d, 64); r.v.l[j] = __shfl_xor(h.v.l[j], d, 64); }
    r.inf = __shfl_xor((int)h.inf, d, 64) != 0;
}
__device__ __forceinline__ void half_select(HalfXyzz& r, bool c, const HalfXyzz& a, const HalfXyzz& b) {   // r = c ? a : b
    fe_select(r.u, c, a.u, b.u);
    fe_select(r.v, c, a.v, b.v);
    r.inf = c ? a.inf : b.inf;
}
// the half held by lane `src` (same parity as the caller)
__device__ __forceinline__ void half_shfl(HalfXyzz& r, const HalfXyzz& h, int src) {
#pragma unroll
    for (int j = 0; j < NL; ++j) { r.u.l[j] = __shfl(h.u.l[j], src, 64); r.v.l[j] = __shfl(h.v.l[j], src, 64); }
    r.inf = __shfl((int)h.inf, src, 64) != 0;
}

}  // namespace kzg
#endif
