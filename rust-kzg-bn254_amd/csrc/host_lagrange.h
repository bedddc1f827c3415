// host_lagrange.h -- the HOST folds of the Lagrange-sharded proofs (csrc/lagrange.hip; include/kzg_bn254_mi355x.h kzg_lagrange_fold_y /
// kzg_lagrange_fold_proof): y from the G gathered (S_g | f_m) rows, the proof point from the G gathered (XYZZ | T_g | L_m | owner) rows.
// Pure host code over host_fr.h / host_curve.h / host_pairing.h: O(G) field and group operations, no device, also built under
// ASAN / UBSAN by tests/test_sanitizers_host.py.  Replaces, for a sharded prover, the tail of helpers::evaluate_polynomial_in_evaluation_form
// (primitives/src/helpers.rs:497-504, :529-532) and of KZG::compute_quotient_eval_on_domain (prover/src/kzg.rs:237-260).
#pragma once
#include "host_fr.h"
#include "host_curve.h"
#include "host_pairing.h"

namespace kzg {

constexpr int32_t LAG_ERR_ROOT_NOT_FOUND = -12;      // = KZG_ERR_ROOT_NOT_FOUND of the C header

inline int ilog2_sz(size_t n) { int k = 0; while (((size_t)1 << k) < n) ++k; return k; }
inline void h_one(uint64_t out[4]) { const uint64_t one_int[4] = {1, 0, 0, 0}; h_fr_mul(H_FR_R2, one_int, out); }
inline void h_fr_add(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]) {
    uint64_t t[4]; hu128 c = 0;
    for (int i = 0; i < 4; ++i) { c += (hu128)a[i] + b[i]; t[i] = (uint64_t)c; c >>= 64; }
    if (c || h_geq_r(t)) h_sub_r(t);
    memcpy(out, t, 32);
}
// z^n (n = 2^log_n) in wire form
inline void h_pow2k(const uint64_t z[4], int log_n, uint64_t out[4]) {
    memcpy(out, z, 32);
    for (int a = 0; a < log_n; ++a) h_fr_mul(out, out, out);
}
inline bool h_is_one(const uint64_t a[4]) { uint64_t o[4]; h_one(o); return memcmp(a, o, 32) == 0; }


// y from the gathered partials (count x 8 words: S_g | f_m): helpers.rs:497-504 (z on the domain: y = f_m) / :507-532
inline int32_t lag_fold_y(const uint64_t* parts, size_t count, size_t n, const uint64_t z[4], uint64_t out_y[4]) {
    const int log_n = ilog2_sz(n);
    uint64_t zn[4], s[4] = {0, 0, 0, 0}, fm[4] = {0, 0, 0, 0};
    h_pow2k(z, log_n, zn);
    for (size_t g = 0; g < count; ++g) { h_fr_add(s, parts + 8 * g, s); h_fr_add(fm, parts + 8 * g + 4, fm); }
    if (h_is_one(zn)) { memcpy(out_y, fm, 32); return 0; }
    uint64_t one[4], num[4], n_inv_int[4], n_inv[4];
    h_one(one);
    h_fr_sub(zn, one, num);                                   // z^n - 1
    // 1 / n for n = 2^k, k <= 28, without an inversion: n divides r - 1, and n (r - (r - 1) / n) = n r - (r - 1) = 1 mod r
    {
        uint64_t q[4] = {H_FR[0] - 1, H_FR[1], H_FR[2], H_FR[3]};
        for (int i = 0; i < 4; ++i) q[i] = log_n ? ((q[i] >> log_n) | (i < 3 ? q[i + 1] << (64 - log_n) : 0)) : q[i];       // (r - 1) >> log_n
        uint64_t br = 0;
        for (int i = 0; i < 4; ++i) { hu128 d = (hu128)H_FR[i] - q[i] - br; n_inv_int[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
        if (log_n == 0) { n_inv_int[0] = 1; n_inv_int[1] = n_inv_int[2] = n_inv_int[3] = 0; }                               // n = 1: r - (r - 1) = 1
    }
    h_fr_mul(H_FR_R2, n_inv_int, n_inv);
    h_fr_mul(s, num, s);
    h_fr_mul(s, n_inv, out_y);
    return 0;
}

// proof from the gathered parts (count x 32 words, lag_end's layout): fold of the partial points, plus q_m L_m when z = w^m with
// q_m = -(1/z) sum_g T_g (kzg.rs:237-260)
inline int32_t lag_fold_proof(const uint64_t* parts, size_t count, size_t n, const uint64_t z[4], uint64_t out_xy[8], uint8_t* out_inf) {
    namespace H = kzg_host;              // (qualified: curve.h has device-math types of the same names in namespace kzg)
    H::Xyzz acc = H::xyzz_inf();
    for (size_t g = 0; g < count; ++g) { H::Xyzz p; memcpy(&p, parts + 32 * g, 128); acc = H::xyzz_add(acc, p); }
    uint64_t zn[4];
    h_pow2k(z, ilog2_sz(n), zn);
    if (h_is_one(zn)) {
        uint64_t t[4] = {0, 0, 0, 0}, zinv[4], qm[4], zero[4] = {0, 0, 0, 0}, qm_int[4];
        const uint64_t* lm = nullptr;
        for (size_t g = 0; g < count; ++g) { h_fr_add(t, parts + 32 * g + 16, t); if (parts[32 * g + 28] == 1 && !lm) lm = parts + 32 * g + 20; }
        if (!lm) return LAG_ERR_ROOT_NOT_FOUND;               // no slice owned w^m: the slices do not cover the domain
        h_fr_inv(z, zinv);
        h_fr_mul(t, zinv, qm);
        h_fr_sub(zero, qm, qm);
        H::fr_wire_to_canonical(qm, qm_int);
        H::G1 term = H::g1_mul(H::g1_from_wire(lm), qm_int);
        if (!term.inf) {
            uint64_t txy[8];
            H::g1_to_wire(term, txy);
            acc = H::xyzz_add(acc, H::xyzz_from_affine_wire(txy));
        }
    }
    H::xyzz_to_affine(acc, out_xy, out_inf);
    return 0;
}


}  // namespace kzg
