// curve_quad.h — ONE XYZZ point spread over a QUAD of adjacent lanes (lane & 3 = 0: X, 1: Y, 2: ZZ, 3: ZZZ), for the reduction
// kernels when nothing else runs beside them (one MSM at a time: small commitments, the tail of a lone large MSM).
//
// The lane-pair form (curve_pair.h) runs 7 multiplications per lane and addition one after the other; the 14 products of add-2008-s
// have a dependency depth of FOUR, and four lanes reach it:
//     round   lane 0 (X)               lane 1 (Y)                lane 2 (ZZ)              lane 3 (ZZZ)
//       1     U1 = X1 ZZ2              S1 = Y1 ZZZ2              U2 = ZZ1 X2              S2 = ZZZ1 Y2        (own coordinate of a times the
//             P = U2 - U1              R = S2 - S1                                                              coordinate of b two lanes over)
//       2     PP = P^2                 RR = R^2                  T = ZZ1 ZZ2              T' = ZZZ1 ZZZ2
//       3     Q = U1 PP                --                        ZZ3 = T PP               PPP = P PP
//             X3 = RR - PPP - 2 Q
//       4     W1 = R (Q - X3)          W2 = S1 PPP               --                       ZZZ3 = T' PPP
//                                      Y3 = W1 - W2
// One instruction stream for the four lanes, operands picked with v_cndmask, every exchange a DPP quad_perm move (VALU, no LDS):
// ~0.7 of the pair form's instructions per addition, at twice its lanes -- worth it only where latency is all that counts.
// The exceptional cases (P == 0: doubling or cancellation) rebuild the whole point in every lane of the quad and run the one-lane
// doubling (curve.h), wave-uniformly.
//
// Value ranges as in curve.h / curve_pair.h: X in (-7m, 5m), Y in (-3m, 3m), ZZ / ZZZ in (-m, 2m), all normalised.
#pragma once
#include "curve_pair.h"

#if defined(__HIPCC__)
namespace kzg {

struct QuadXyzz {      // lane q = lane & 3 of the quad holds coordinate q of X, Y, ZZ, ZZZ; `inf` is the same in the four lanes
    Fq c;
    bool inf;
};

__device__ __forceinline__ void quad_set_inf(QuadXyzz& h) { fe_set_zero(h.c); h.inf = true; }

// DPP quad_perm moves: CTRL = p0 | p1 << 2 | p2 << 4 | p3 << 6, lane q of the quad reads lane p_q
constexpr int QUAD_BCAST0 = 0x00, QUAD_BCAST1 = 0x55, QUAD_BCAST2 = 0xAA, QUAD_BCAST3 = 0xFF, QUAD_SWAP2 = 0x4E /* [2, 3, 0, 1] */;
template <int CTRL>
__device__ __forceinline__ int32_t quad_mov(int32_t x) {
#ifdef KZG_QUAD_NO_DPP          // diagnostic: the same move through ds_bpermute
    const int lane = (int)(threadIdx.x & 63u);
    return __shfl(x, (lane & ~3) | ((CTRL >> (2 * (lane & 3))) & 3), 64);
#else
    return __builtin_amdgcn_mov_dpp(x, CTRL, 0xF, 0xF, true);
#endif
}
template <int CTRL>
__device__ __forceinline__ void fe_quad_mov(Fq& r, const Fq& a) {
#pragma unroll
    for (int j = 0; j < NL; ++j) r.l[j] = quad_mov<CTRL>(a.l[j]);
}
// the whole point of the quad, in every lane
__device__ __forceinline__ void quad_gather(Xyzz& p, const QuadXyzz& h) {
    fe_quad_mov<QUAD_BCAST0>(p.x, h.c);
    fe_quad_mov<QUAD_BCAST1>(p.y, h.c);
    fe_quad_mov<QUAD_BCAST2>(p.zz, h.c);
    fe_quad_mov<QUAD_BCAST3>(p.zzz, h.c);
    p.inf = h.inf;
}
__device__ __forceinline__ void quad_pick(QuadXyzz& h, const Xyzz& p, uint32_t q) {
    Fq lo, hi;
    fe_select(lo, (q & 1u) != 0, p.y, p.x);
    fe_select(hi, (q & 1u) != 0, p.zzz, p.zz);
    fe_select(h.c, q >= 2u, hi, lo);
    h.inf = p.inf;
}

// rounds 3 and 4 and the exceptional cases, shared by quad_add and quad_madd.
//   u1s1: lane 0: U1, lane 1: S1;  df: lane 0: P, lane 1: R;  sq: PP | RR | T | T';  keep: the point to double when P == R == 0
__device__ __forceinline__ void quad_add_tail(QuadXyzz& r, const Fq& u1s1, const Fq& df, const Fq& sq, bool may_be_exceptional, const QuadXyzz& keep, uint32_t q) {
    const int zero_here = (may_be_exceptional && q < 2u && fe_is_zero_mod(sq)) ? 1 : 0;
    const bool exc = quad_mov<QUAD_BCAST0>(zero_here) != 0;                 // P == 0: the same or opposite points
    const bool same = exc && quad_mov<QUAD_BCAST1>(zero_here) != 0;          // and R == 0: the same point
    Fq df0, sq0, e1, t3;
    fe_quad_mov<QUAD_BCAST0>(df0, df);
    fe_quad_mov<QUAD_BCAST0>(sq0, sq);
    fe_select(e1, q == 2u, sq, u1s1);
    fe_select(e1, q == 3u, df0, e1);
    fe_mul(t3, e1, sq0);                       // Q = U1 PP | (unused) | ZZ3 = T PP | PPP = P PP
    Fq rr, ppp, x3, tq;
    fe_quad_mov<QUAD_BCAST1>(rr, sq);
    fe_quad_mov<QUAD_BCAST3>(ppp, t3);
    fe_sub(x3, rr, ppp); fe_sub(x3, x3, t3); fe_sub(x3, x3, t3); fe_norm(x3);     // lane 0: X3 = RR - PPP - 2Q in (-7m, 5m)
    fe_sub(tq, t3, x3);                        // lane 0: Q - X3 in (-6m, 9m), limbs within +-2^29
    // (W1 on lane 0, W2 on lane 1: Y3 = dpp(t4) - t4.  The other way round hipcc folds the move into v_subrev_u32_dpp, and the kernels then
    //  returned a wrong Y in every generic case -- tools/ubench/quad_check.hip; v_sub_u32_dpp, as the pair form uses, is fine.)
    Fq df1, g1, g2, t4, w1, y3;
    fe_quad_mov<QUAD_BCAST1>(df1, df);
    fe_select(g1, q == 1u, u1s1, sq);
    fe_select(g1, q == 0u, df1, g1);
    fe_select(g2, q == 0u, tq, ppp);
    fe_mul(t4, g1, g2);                        // W1 = R (Q - X3) | W2 = S1 PPP | (unused) | ZZZ3 = T' PPP
    fe_quad_mov<QUAD_BCAST0>(w1, t4);
    fe_sub(y3, w1, t4); fe_norm(y3);           // lane 1: Y3 = W1 - W2 in (-3m, 3m)
    Fq lo, hi;
    fe_select(lo, q == 1u, y3, x3);
    fe_select(hi, q == 3u, t4, t3);
    fe_select(r.c, q >= 2u, hi, lo);
    r.inf = false;
    if (__builtin_expect(__any(exc), 0)) {     // some quad of this wave doubles or cancels
        Xyzz k, d;
        quad_gather(k, keep);
        xyzz_dbl(d, k);
        if (exc) {
            if (same) quad_pick(r, d, q);
            else quad_set_inf(r);
        }
    }
}

// r = a + b.  The four lanes of the quad must be active together (callers branch on quad-uniform conditions only).
__device__ __forceinline__ void quad_add(QuadXyzz& r, const QuadXyzz& a, const QuadXyzz& b, uint32_t q) {
    if (__all(a.inf || b.inf)) {               // nothing to add in this wave: copy
        if (a.inf) r = b; else r = a;
        return;
    }
    QuadXyzz keep;                             // what survives when one operand is the identity; `a` itself when both are finite
    fe_select(keep.c, a.inf, b.c, a.c);
    keep.inf = a.inf && b.inf;
    const bool any_inf = a.inf || b.inf;
    Fq bx, m1, o1, df, s1, s2, sq;
    fe_quad_mov<QUAD_SWAP2>(bx, b.c);
    fe_mul(m1, a.c, bx);                       // U1 = X1 ZZ2 | S1 = Y1 ZZZ2 | U2 = ZZ1 X2 | S2 = ZZZ1 Y2     (7m * 2m, 3m * 2m)
    fe_quad_mov<QUAD_SWAP2>(o1, m1);
    fe_sub(df, o1, m1);                        // P | R in (-3m, 3m)    (lanes 2, 3: unused)
    fe_select(s1, q >= 2u, a.c, df);
    fe_select(s2, q >= 2u, b.c, df);
    fe_mul(sq, s1, s2);                        // PP | RR | T = ZZ1 ZZ2 | T' = ZZZ1 ZZZ2
    quad_add_tail(r, m1, df, sq, !any_inf, keep, q);
    if (any_inf) r = keep;
}

// acc += (neg ? -p : p) for an AFFINE point p that is not the identity: lanes 0 and 2 pass c = x, lanes 1 and 3 c = y (canonical, as
// unpacked from the device affine format).  madd-2008-s: U1 = X1, S1 = Y1, T = ZZ1, T' = ZZZ1, so
//     round 1:  lane 2: U2 = ZZ1 x2, lane 3: S2 = ZZZ1 y2;   round 2:  lane 0: PP, lane 1: RR;   rounds 3 and 4 as in quad_add.
__device__ __forceinline__ void quad_madd(QuadXyzz& r, const QuadXyzz& acc, const Fq& c, uint32_t neg, uint32_t q) {
    QuadXyzz from;                             // the point itself as XYZZ (ZZ = ZZZ = 1)
    Fq cs, one;
    fe_cneg(cs, c, (q & 1u) ? neg : 0u);
    fe_norm(cs);
    fe_set_one(one);
    fe_select(from.c, q >= 2u, one, cs);
    from.inf = false;
    if (__all(acc.inf)) { r = from; return; }
    Fq m, om, df, sq, s;
    fe_mul(m, acc.c, cs);                      // (unused) | (unused) | U2 | S2
    fe_quad_mov<QUAD_SWAP2>(om, m);
    fe_sub(df, om, acc.c);                     // P in (-6m, 9m) | R in (-4m, 5m): limbs within +-2^29
    fe_sqr(s, df);                             // PP | RR (81 m^2)
    fe_select(sq, q >= 2u, acc.c, s);          // PP | RR | T = ZZ1 | T' = ZZZ1
    quad_add_tail(r, acc.c, df, sq, !acc.inf, from, q);
    if (acc.inf) r = from;
}

// r = 2 a (dbl-2008-s-1), a finite.  The nine products of the one-lane form have a dependency depth of THREE, and the quad reaches it:
//     round   lane 0 (X)                lane 1 (Y)              lane 2 (ZZ)           lane 3 (ZZZ)
//       1     XX = X^2                  V = U^2  (U = 2 Y)      V = U^2               V = U^2
//             M = 3 XX
//       2     S = X V                   W = U V                 ZZ3 = ZZ V            MM = M^2   (M from lane 0)
//             X3 = MM - 2 S
//       3     D = M (3 S - MM)          WY = W Y                --                    ZZZ3 = ZZZ W   (W from lane 1)
//                                       Y3 = D - WY                                                   (S - X3 = 3 S - MM)
// (the pair form is five products deep).  Ranges as curve.h: X3 in (-5m, 4m), Y3 in (-3m, 3m), ZZ3 / ZZZ3 in (-m, 2m).
__device__ __forceinline__ void quad_dbl(QuadXyzz& r, const QuadXyzz& a, uint32_t q) {
    Fq y, u, s1, t1;
    fe_quad_mov<QUAD_BCAST1>(y, a.c);
    fe_dbl(u, y); fe_norm(u);                  // U = 2 Y, |U| < 6m
    fe_select(s1, q == 0u, a.c, u);
    fe_sqr(t1, s1);                            // XX (49 m^2) | V | V | V (36 m^2)
    Fq m, m0, v1;
    fe_add(m, t1, t1); fe_add(m, m, t1); fe_norm(m);     // lane 0: M = 3 XX, |M| < 6m
    fe_quad_mov<QUAD_BCAST0>(m0, m);
    fe_quad_mov<QUAD_BCAST1>(v1, t1);          // V in every lane
    Fq a2, b2, t2;
    fe_select(a2, q == 1u, u, a.c);            // X | U | ZZ | (ZZZ)
    fe_select(a2, q == 3u, m0, a2);            // X | U | ZZ | M
    fe_select(b2, q == 3u, m0, v1);            // V | V | V | M
    fe_mul(t2, a2, b2);                        // S = X V (7m * 2m) | W = U V | ZZ3 = ZZ V | MM = M^2 (36 m^2)
    Fq mm, w1, x3, tq;
    fe_quad_mov<QUAD_BCAST3>(mm, t2);
    fe_quad_mov<QUAD_BCAST1>(w1, t2);
    fe_sub(x3, mm, t2); fe_sub(x3, x3, t2); fe_norm(x3);         // lane 0: X3 = MM - 2 S in (-5m, 4m)
    fe_add(tq, t2, t2); fe_add(tq, tq, t2); fe_sub(tq, tq, mm); fe_norm(tq);      // lane 0: 3 S - MM in (-5m, 7m)
    Fq a3, b3, t3;
    fe_select(a3, q == 0u, m, w1);             // M | W | W | W
    fe_select(b3, q == 0u, tq, a.c);           // 3 S - MM | Y | (ZZ) | ZZZ
    fe_mul(t3, a3, b3);                        // D = M (3 S - MM) (42 m^2) | WY = W Y | (unused) | ZZZ3 = W ZZZ
    Fq d0, y3;
    fe_quad_mov<QUAD_BCAST0>(d0, t3);
    fe_sub(y3, d0, t3); fe_norm(y3);           // lane 1: Y3 = D - WY in (-3m, 3m)      (dpp(t) - t: see quad_add_tail on v_subrev_u32_dpp)
    Fq lo, hi;
    fe_select(lo, q == 1u, y3, x3);
    fe_select(hi, q == 3u, t3, t2);
    fe_select(r.c, q >= 2u, hi, lo);
    r.inf = false;
}
// a may be the identity (quad-uniform flag)
__device__ __forceinline__ void quad_dbl_any(QuadXyzz& r, const QuadXyzz& a, uint32_t q) {
    if (__all(a.inf)) { r = a; return; }
    QuadXyzz d;
    quad_dbl(d, a, q);
    if (a.inf) r = a; else r = d;
}

// ---- memory: the struct-of-arrays XYZZ layout of curve.h (36 limb planes), each lane touching its coordinate ---------------------
__device__ __forceinline__ void quad_load(QuadXyzz& h, const int32_t* __restrict__ base, size_t stride, size_t i, uint32_t q) {
#pragma unroll
    for (int j = 0; j < NL; ++j) h.c.l[j] = base[(size_t)(q * NL + j) * stride + i];
    const int z = fe_is_literal_zero(h.c) ? 1 : 0;         // the identity is stored as all zeros; a finite point has ZZ != 0
    h.inf = quad_mov<QUAD_BCAST2>(z) != 0;
}
__device__ __forceinline__ void quad_store(int32_t* __restrict__ base, size_t stride, size_t i, const QuadXyzz& h, uint32_t q) {
#pragma unroll
    for (int j = 0; j < NL; ++j) base[(size_t)(q * NL + j) * stride + i] = h.inf ? 0 : h.c.l[j];
}
// 32 u32 wire words X || Y || ZZ || ZZZ of point i: lane q converts and writes its 8 words
__device__ __forceinline__ void quad_store_wire(uint32_t* __restrict__ out_wire, size_t i, const QuadXyzz& h, uint32_t q) {
    uint32_t w[8];
    if (h.inf) {
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = 0;
    } else {
        fe_to_wire(w, h.c);
    }
    uint32_t* p = out_wire + i * 32 + q * 8;
    *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    *reinterpret_cast<uint4*>(p + 4) = make_uint4(w[4], w[5], w[6], w[7]);
}
// the coordinate of the lane `d` lanes up (d a multiple of 4: the same role)
__device__ __forceinline__ void quad_shfl_down(QuadXyzz& r, const QuadXyzz& h, int d) {
#pragma unroll
    for (int j = 0; j < NL; ++j) r.c.l[j] = __shfl_down(h.c.l[j], d, 64);
    r.inf = __shfl_down((int)h.inf, d, 64) != 0;
}
// the coordinate held by lane `src` (same role as the caller)
__device__ __forceinline__ void quad_shfl(QuadXyzz& r, const QuadXyzz& h, int src) {
#pragma unroll
    for (int j = 0; j < NL; ++j) r.c.l[j] = __shfl(h.c.l[j], src, 64);
    r.inf = __shfl((int)h.inf, src, 64) != 0;
}

}  // namespace kzg
#endif
