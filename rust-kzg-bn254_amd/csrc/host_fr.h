// host_fr.h -- Fr arithmetic on wire words (4 x u64 Montgomery limbs, R = 2^256) for the handful of scalars a proof needs on the HOST:
// the one inversion of a proof's coset chain, z^n, the domain index of an on-domain z, the folds of the sharded proofs.  Pure host code
// (no HIP): also compiled by g++ under ASAN / UBSAN (tests/test_sanitizers_host.py).
#pragma once
#include <cstdint>
#include <cstring>

namespace kzg {

// ---- host Fr arithmetic on wire words (Montgomery, R = 2^256) for the one inversion of a proof -------------------------
typedef unsigned __int128 hu128;
static const uint64_t H_FR[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
static const uint64_t H_FR_NINV = 0xc2e1f593efffffffULL;          // -r^-1 mod 2^64
static const uint64_t H_FR_R2[4] = {0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL};   // 2^512 mod r
inline bool h_geq_r(const uint64_t t[4]) {
    for (int i = 3; i >= 0; --i) if (t[i] != H_FR[i]) return t[i] > H_FR[i];
    return true;
}
inline void h_sub_r(uint64_t t[4]) {
    uint64_t br = 0;
    for (int i = 0; i < 4; ++i) { hu128 d = (hu128)t[i] - H_FR[i] - br; t[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
}
inline void h_fr_mul(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]) {
    uint64_t t[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        hu128 c = 0;
        for (int j = 0; j < 4; ++j) { c += (hu128)a[j] * b[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        hu128 top = (hu128)t[4] + (uint64_t)c;
        const uint64_t m = t[0] * H_FR_NINV;
        c = ((hu128)m * H_FR[0] + t[0]) >> 64;
        for (int j = 1; j < 4; ++j) { c += (hu128)m * H_FR[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        top += (uint64_t)c;
        t[3] = (uint64_t)top; t[4] = (uint64_t)(top >> 64);
    }
    if (t[4] || h_geq_r(t)) h_sub_r(t);
    memcpy(out, t, 32);
}
// out = a - b mod r (both < r)
inline void h_fr_sub(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]) {
    uint64_t t[4], br = 0;
    for (int i = 0; i < 4; ++i) { hu128 d = (hu128)a[i] - b[i] - br; t[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
    if (br) { hu128 c = 0; for (int i = 0; i < 4; ++i) { c += (hu128)t[i] + H_FR[i]; t[i] = (uint64_t)c; c >>= 64; } }
    memcpy(out, t, 32);
}
// a^(r-2), wire in / wire out: 381 dependent products (kept as the cross-check of h_fr_inv below: tests/hostcheck/sanitize_main.cpp)
inline void h_fr_inv_fermat(const uint64_t a[4], uint64_t out[4]) {
    uint64_t e[4] = {H_FR[0] - 2, H_FR[1], H_FR[2], H_FR[3]};
    uint64_t acc[4], base[4];
    const uint64_t one_int[4] = {1, 0, 0, 0};
    h_fr_mul(H_FR_R2, one_int, acc);                         // 1 in wire form
    memcpy(base, a, 32);
    for (int i = 0; i < 254; ++i) {
        if ((e[i >> 6] >> (i & 63)) & 1) h_fr_mul(acc, base, acc);
        h_fr_mul(base, base, base);
    }
    memcpy(out, acc, 32);
}
// Inverse, wire in / wire out (0 -> 0): binary extended Euclid on the plain integers -- at most 2 x 254 shift / subtract rounds on four words -- and two
// Montgomery products by R^2 back into the form: a third of the Fermat form's time, on the host's critical path of every proof (the one inversion of the
// coset chain, the folds of the sharded proofs).  Same construction as kzg_host::inv for Fq (host_curve.h).
inline void h_fr_inv(const uint64_t a_in[4], uint64_t out[4]) {
    uint64_t u[4], v[4], x1[4] = {1, 0, 0, 0}, x2[4] = {0, 0, 0, 0};
    memcpy(u, a_in, 32);
    for (int i = 0; i < 6 && h_geq_r(u); ++i) h_sub_r(u);                     // any 256-bit input: canonical first
    if ((u[0] | u[1] | u[2] | u[3]) == 0) { memset(out, 0, 32); return; }
    memcpy(v, H_FR, 32);
    auto is_one = [](const uint64_t t[4]) { return t[0] == 1 && (t[1] | t[2] | t[3]) == 0; };
    auto geq = [](const uint64_t a[4], const uint64_t b[4]) { for (int i = 3; i >= 0; --i) if (a[i] != b[i]) return a[i] > b[i]; return true; };
    auto shr1 = [](uint64_t t[4]) { for (int i = 0; i < 3; ++i) t[i] = (t[i] >> 1) | (t[i + 1] << 63); t[3] >>= 1; };
    auto sub_raw = [](uint64_t a[4], const uint64_t b[4]) { uint64_t br = 0; for (int i = 0; i < 4; ++i) { hu128 d = (hu128)a[i] - b[i] - br; a[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; } };
    auto half_mod_r = [&](uint64_t x[4]) {                                    // x / 2 mod r, x < r < 2^254
        if (x[0] & 1) { hu128 c = 0; for (int i = 0; i < 4; ++i) { c += (hu128)x[i] + H_FR[i]; x[i] = (uint64_t)c; c >>= 64; } }
        shr1(x);
    };
    for (int guard = 0; guard < 1024 && !is_one(u) && !is_one(v); ++guard) {   // gcd(a, r) = 1
        while ((u[0] & 1) == 0) { shr1(u); half_mod_r(x1); }
        while ((v[0] & 1) == 0) { shr1(v); half_mod_r(x2); }
        if (geq(u, v)) { sub_raw(u, v); h_fr_sub(x1, x2, x1); }
        else { sub_raw(v, u); h_fr_sub(x2, x1, x2); }
    }
    const uint64_t* b = is_one(u) ? x1 : x2;                                  // (a R)^-1 = a^-1 R^-1 as a plain integer < r
    uint64_t t[4];
    h_fr_mul(b, H_FR_R2, t);                                                  // . R^2 R^-1 = a^-1
    h_fr_mul(t, H_FR_R2, out);                                                // . R^2 R^-1 = a^-1 R
}
// 1/(i - 1), -1/2, 1/(-i - 1) in wire form, i = 5^((r-1)/4) (= w_n^(n/4) for every n >= 4: arkworks' roots are powers of 5^((r-1)/2^28))
inline const uint64_t* h_on_domain_constants() {
    static const struct Init {
        uint64_t c[12];
        Init() {
            const uint64_t one_int[4] = {1, 0, 0, 0}, five_int[4] = {5, 0, 0, 0}, zero[4] = {0, 0, 0, 0};
            uint64_t one_w[4], five_w[4], two_w[4], qi[4], acc[4], base[4], t[4];
            h_fr_mul(H_FR_R2, one_int, one_w);
            h_fr_mul(H_FR_R2, five_int, five_w);
            uint64_t e[4] = {H_FR[0] - 1, H_FR[1], H_FR[2], H_FR[3]};            // (r - 1) / 4
            for (int i = 0; i < 4; ++i) e[i] = (e[i] >> 2) | (i < 3 ? e[i + 1] << 62 : 0);
            memcpy(acc, one_w, 32); memcpy(base, five_w, 32);
            for (int i = 0; i < 254; ++i) {
                if ((e[i >> 6] >> (i & 63)) & 1) h_fr_mul(acc, base, acc);
                h_fr_mul(base, base, base);
            }
            memcpy(qi, acc, 32);
            h_fr_sub(qi, one_w, t); h_fr_inv(t, c);                             // 1 / (i - 1)
            h_fr_sub(zero, one_w, two_w); h_fr_sub(two_w, one_w, two_w);         // -2
            h_fr_inv(two_w, c + 4);                                             // -1/2
            h_fr_sub(zero, qi, t); h_fr_sub(t, one_w, t); h_fr_inv(t, c + 8);   // 1 / (-i - 1)
        }
    } init;
    return init.c;
}
// w_(2^k) and its inverse in wire form, k <= 28: arkworks' roots of unity are the powers of g = 5^((r-1)/2^28)
struct HRoots { uint64_t w[29][4], winv[29][4]; };
inline const HRoots& h_roots() {
    static const struct Init {
        HRoots r;
        Init() {
            const uint64_t one_int[4] = {1, 0, 0, 0}, five_int[4] = {5, 0, 0, 0};
            uint64_t acc[4], base[4];
            h_fr_mul(H_FR_R2, one_int, acc);
            h_fr_mul(H_FR_R2, five_int, base);
            uint64_t e[4] = {H_FR[0] - 1, H_FR[1], H_FR[2], H_FR[3]};            // (r - 1) >> 28
            for (int i = 0; i < 4; ++i) e[i] = (e[i] >> 28) | (i < 3 ? e[i + 1] << 36 : 0);
            for (int i = 0; i < 254; ++i) {
                if ((e[i >> 6] >> (i & 63)) & 1) h_fr_mul(acc, base, acc);
                h_fr_mul(base, base, base);
            }
            memcpy(r.w[28], acc, 32);
            for (int k = 27; k >= 0; --k) h_fr_mul(r.w[k + 1], r.w[k + 1], r.w[k]);
            h_fr_inv(r.w[28], r.winv[28]);
            for (int k = 27; k >= 0; --k) h_fr_mul(r.winv[k + 1], r.winv[k + 1], r.winv[k]);
        }
    } init;
    return init.r;
}
// m with w_n^m = z for a z of the n-point domain (n = 2^log_n, z^n = 1): one bit per step, lowest first (Pohlig-Hellman in a group of
// order 2^k: ~k^2 / 2 host multiplications, 3 us at k = 11).  Returns false if z turns out not to be a power of w_n.
inline bool h_domain_index(const uint64_t z[4], int log_n, uint32_t* m_out) {
    const HRoots& R = h_roots();
    uint64_t one_w[4];
    const uint64_t one_int[4] = {1, 0, 0, 0};
    h_fr_mul(H_FR_R2, one_int, one_w);
    uint64_t h[4];
    memcpy(h, z, 32);
    uint32_t m = 0;
    for (int b = 0; b < log_n; ++b) {
        uint64_t t[4];
        memcpy(t, h, 32);
        for (int q = 0; q < log_n - 1 - b; ++q) h_fr_mul(t, t, t);      // h^(2^(k-1-b)) = (-1)^(bit b of m)
        if (memcmp(t, one_w, 32) != 0) {
            m |= 1u << b;
            h_fr_mul(h, R.winv[log_n - b], h);                         // h *= w_n^-(2^b) = w_(n / 2^b)^-1
        }
    }
    if (memcmp(h, one_w, 32) != 0) return false;
    *m_out = m;
    return true;
}


}  // namespace kzg
