/*
 * oracle/field.c — TEST INFRASTRUCTURE ONLY (CPU oracle); see field.h.
 */
#include "field.h"

field_t FQ, FR;

/* -m^-1 mod 2^64 by Newton iteration on the low limb. */
static uint64_t neg_inv64(uint64_t m0) {
    uint64_t x = 1;
    for (int i = 0; i < 6; ++i) x *= 2 - m0 * x;   /* x = m0^-1 mod 2^64 */
    return (uint64_t)0 - x;
}

/* r = 2^k mod m by repeated modular doubling (only used at init). */
static void pow2_mod(const fe *m, fe *r, int k) {
    fe t = {{1, 0, 0, 0}};
    for (int i = 0; i < k; ++i) {
        fe d; uint64_t c = fe_add_raw(&d, &t, &t);
        if (c || fe_geq(&d, m)) fe_sub_raw(&d, &d, m);
        t = d;
    }
    *r = t;
}

static void field_setup(field_t *F, const uint64_t m[4]) {
    memcpy(F->m.l, m, 32);
    F->inv = neg_inv64(m[0]);
    pow2_mod(&F->m, &F->one, 256);
    pow2_mod(&F->m, &F->r2, 512);
}

static int g_init_done = 0;
void oracle_init(void) {
    if (g_init_done) return;
    /* SURVEY.md Appendix A; Fq = helpers.rs:201 `Fq::from_be_bytes_mod_order`, Fr = consts.rs:22 */
    static const uint64_t q[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
    static const uint64_t r[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
    field_setup(&FQ, q);
    field_setup(&FR, r);
    g_init_done = 1;
}

void fe_pow(const field_t *F, fe *r, const fe *a, const fe *e) {
    fe acc = F->one, base = *a;
    for (int i = 0; i < 256; ++i) {
        if ((e->l[i >> 6] >> (i & 63)) & 1) fe_mul(F, &acc, &acc, &base);
        fe_sqr(F, &base, &base);
    }
    *r = acc;
}
void fe_pow_u64(const field_t *F, fe *r, const fe *a, uint64_t e) {
    fe ee = {{e, 0, 0, 0}}; fe_pow(F, r, a, &ee);
}
/* a^(m-2) (Fermat) — ark-ff `Field::inverse` returns None on zero; so do we. */
int fe_inv(const field_t *F, fe *r, const fe *a) {
    if (fe_is_zero(a)) return 0;
    fe e = F->m, two = {{2, 0, 0, 0}};
    fe_sub_raw(&e, &e, &two);
    fe_pow(F, r, a, &e);
    return 1;
}

/* ark-ff `PrimeField::from_be_bytes_mod_order` (used at helpers.rs:32-34, :201, :387): the
 * big-endian byte string is interpreted as an integer and reduced mod m. Horner over bytes. */
void fe_from_be_bytes_mod_order(const field_t *F, fe *r, const uint8_t *bytes, size_t len) {
    fe acc = {{0, 0, 0, 0}}, c256; fe_from_u64(F, &c256, 256);
    for (size_t i = 0; i < len; ++i) {
        fe b; fe_from_u64(F, &b, bytes[i]);
        fe_mul(F, &acc, &acc, &c256);
        fe_add(F, &acc, &acc, &b);
    }
    *r = acc;
}
/* `into_bigint().to_bytes_be()` (helpers.rs:91) */
void fe_to_be_bytes(const field_t *F, uint8_t out[32], const fe *a) {
    fe c; fe_to_canonical(F, &c, a);
    for (int i = 0; i < 4; ++i)
        for (int b = 0; b < 8; ++b) out[31 - (8 * i + b)] = (uint8_t)(c.l[i] >> (8 * b));
}

/* p = 3 mod 4  =>  sqrt(a) = a^((p+1)/4) (helpers.rs:203 `y_squared.sqrt()`). */
int fq_sqrt(fe *r, const fe *a) {
    fe e = FQ.m, one = {{1, 0, 0, 0}};
    fe_add_raw(&e, &e, &one);
    /* e >>= 2 */
    for (int i = 0; i < 4; ++i) e.l[i] = (e.l[i] >> 2) | (i < 3 ? e.l[i + 1] << 62 : 0);
    fe s; fe_pow(&FQ, &s, a, &e);
    fe chk; fe_sqr(&FQ, &chk, &s);
    if (!fe_eq(&chk, a)) return 0;
    *r = s;
    return 1;
}
