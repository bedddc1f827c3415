/*
 * oracle/field.h — TEST INFRASTRUCTURE ONLY (CPU oracle).  Never linked into, imported by or
 * executed from the product path; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use anything under oracle/.
 *
 * 4x64-bit-limb Montgomery arithmetic for the two BN254 prime fields, restating what the
 * reference delegates to the third-party crate ark-ff 0.5 (`Fp256<MontBackend<..,4>>`, reference
 * Cargo.toml:57-62; not vendored in /root/reference).  In-memory layout = arkworks': four
 * little-endian u64 limbs holding a*R mod m, R = 2^256, canonical (< m).  The reference itself
 * relies on that layout at primitives/src/helpers.rs:158 (`z.0 .0[i]`) and restates the
 * reduction at primitives/src/arith.rs:4-55, whose KATs (arith.rs:145-200) pin mont_reduce() here.
 *
 * Moduli: SURVEY.md Appendix A (Fq: curve base field, Fr: scalar field).  Every derived constant
 * (-m^-1 mod 2^64, R mod m, R^2 mod m) is COMPUTED at init from the modulus, then checked against
 * the published values in tests/test_oracle.py.
 */
#ifndef KZG_ORACLE_FIELD_H
#define KZG_ORACLE_FIELD_H

#include <stdint.h>
#include <string.h>

typedef unsigned __int128 u128;

typedef struct { uint64_t l[4]; } fe;

typedef struct {
    fe m;          /* modulus */
    uint64_t inv;  /* -m^-1 mod 2^64 */
    fe one;        /* R mod m */
    fe r2;         /* R^2 mod m */
} field_t;

extern field_t FQ, FR;
void oracle_init(void);

static inline int fe_is_zero(const fe *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static inline int fe_eq(const fe *a, const fe *b) { return memcmp(a, b, sizeof(fe)) == 0; }
static inline int fe_geq(const fe *a, const fe *b) {
    for (int i = 3; i >= 0; --i) {
        if (a->l[i] > b->l[i]) return 1;
        if (a->l[i] < b->l[i]) return 0;
    }
    return 1;
}
static inline uint64_t fe_add_raw(fe *r, const fe *a, const fe *b) {
    u128 c = 0;
    for (int i = 0; i < 4; ++i) { c += (u128)a->l[i] + b->l[i]; r->l[i] = (uint64_t)c; c >>= 64; }
    return (uint64_t)c;
}
static inline uint64_t fe_sub_raw(fe *r, const fe *a, const fe *b) {
    uint64_t borrow = 0;
    for (int i = 0; i < 4; ++i) {
        u128 d = (u128)a->l[i] - b->l[i] - borrow;
        r->l[i] = (uint64_t)d; borrow = (uint64_t)(d >> 64) & 1;
    }
    return borrow;
}
static inline void fe_add(const field_t *F, fe *r, const fe *a, const fe *b) {
    fe t; uint64_t c = fe_add_raw(&t, a, b);
    if (c || fe_geq(&t, &F->m)) fe_sub_raw(&t, &t, &F->m);
    *r = t;
}
static inline void fe_sub(const field_t *F, fe *r, const fe *a, const fe *b) {
    fe t; if (fe_sub_raw(&t, a, b)) fe_add_raw(&t, &t, &F->m);
    *r = t;
}
static inline void fe_neg(const field_t *F, fe *r, const fe *a) {
    if (fe_is_zero(a)) { *r = *a; return; }
    fe_sub_raw(r, &F->m, a);
}
static inline void fe_dbl(const field_t *F, fe *r, const fe *a) { fe_add(F, r, a, a); }

/* CIOS Montgomery product a*b*R^-1 mod m (ark-ff MontBackend::mul_assign, restated). */
static inline void fe_mul(const field_t *F, fe *r, const fe *a, const fe *b) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        u128 x; uint64_t carry = 0;
        for (int j = 0; j < 4; ++j) {
            x = (u128)a->l[j] * b->l[i] + t[j] + carry;
            t[j] = (uint64_t)x; carry = (uint64_t)(x >> 64);
        }
        x = (u128)t[4] + carry; t[4] = (uint64_t)x; t[5] = (uint64_t)(x >> 64);
        uint64_t m = t[0] * F->inv;
        x = (u128)m * F->m.l[0] + t[0]; carry = (uint64_t)(x >> 64);
        for (int j = 1; j < 4; ++j) {
            x = (u128)m * F->m.l[j] + t[j] + carry;
            t[j - 1] = (uint64_t)x; carry = (uint64_t)(x >> 64);
        }
        x = (u128)t[4] + carry; t[3] = (uint64_t)x; t[4] = t[5] + (uint64_t)(x >> 64);
    }
    fe out = {{t[0], t[1], t[2], t[3]}};
    if (t[4] || fe_geq(&out, &F->m)) fe_sub_raw(&out, &out, &F->m);
    *r = out;
}
static inline void fe_sqr(const field_t *F, fe *r, const fe *a) { fe_mul(F, r, a, a); }

/* Montgomery form <-> canonical integer (limbs LE). */
static inline void fe_from_canonical(const field_t *F, fe *r, const fe *a) { fe_mul(F, r, a, &F->r2); }
static inline void fe_to_canonical(const field_t *F, fe *r, const fe *a) {
    fe one = {{1, 0, 0, 0}}; fe_mul(F, r, a, &one);
}
static inline void fe_from_u64(const field_t *F, fe *r, uint64_t v) {
    fe t = {{v, 0, 0, 0}}; fe_from_canonical(F, r, &t);
}

void fe_pow(const field_t *F, fe *r, const fe *a, const fe *e_canonical);
void fe_pow_u64(const field_t *F, fe *r, const fe *a, uint64_t e);
int  fe_inv(const field_t *F, fe *r, const fe *a);            /* 0 if a == 0 */
void fe_from_be_bytes_mod_order(const field_t *F, fe *r, const uint8_t *bytes, size_t len);
void fe_to_be_bytes(const field_t *F, uint8_t out[32], const fe *a);
int  fq_sqrt(fe *r, const fe *a);                              /* 1 if a is a square */

#endif
