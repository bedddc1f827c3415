// Host-only self-check of csrc/host_pairing.h (built by tests/test_pairing_host.py with g++; no GPU, no oracle).
// Prints one line per property: "<name> 0|1".
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include "host_pairing.h"
using namespace kzg_host;

static bool fq12_eq(const Fq12& a, const Fq12& b) { for (int i = 0; i < 12; ++i) if (!eq(a.c[i], b.c[i])) return false; return true; }

int main() {
    G1 g; g.x = FQ_ONE; g.y = FQ_TWO; g.inf = false;
    G2 h = g2_generator();
    {   // binary-Euclid inversion (host_curve.h inv) against a^(p-2) and against a * a^-1 = 1: edge values and 20 000 pseudo-random ones
        bool ok = true;
        Fq pm1 = FQ_P; pm1.l[0] -= 1;
        Fq edge[6] = {FQ_ONE, FQ_TWO, pm1, {{1, 0, 0, 0}}, {{2, 0, 0, 0}}, {{0, 0, 0, 1ULL << 61}}};
        for (const Fq& a : edge) ok = ok && eq(inv(a), inv_fermat(a)) && eq(mul(a, inv(a)), FQ_ONE);
        uint64_t s = 88172645463325252ULL;
        Fq a = FQ_TWO;
        for (int i = 0; i < 20000 && ok; ++i) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            Fq t = {{s, s * 0x9E3779B97F4A7C15ULL, ~s, s >> 3}};
            t.l[3] &= (1ULL << 61) - 1;                    // < 2^253 < p
            a = mul(add(a, t), FQ_TWO);
            if (is_zero(a)) continue;
            const Fq b = inv(a);
            ok = eq(mul(a, b), FQ_ONE) && (i % 64 != 0 || eq(b, inv_fermat(a)));
        }
        printf("binary_inverse %d\n", (int)(ok && is_zero(inv(fq_zero()))));
        // non-canonical input: p, 2p, 5p invert to 0 like the literal zero, a + p inverts like a -- and none of them hangs
        Fq k2 = FQ_P, k5 = FQ_P, ap = FQ_TWO;
        { unsigned __int128 c = 0; for (int i = 0; i < 4; ++i) { c += (unsigned __int128)k2.l[i] + FQ_P.l[i]; k2.l[i] = (uint64_t)c; c >>= 64; } }
        for (int r = 0; r < 4; ++r) { unsigned __int128 c = 0; for (int i = 0; i < 4; ++i) { c += (unsigned __int128)k5.l[i] + FQ_P.l[i]; k5.l[i] = (uint64_t)c; c >>= 64; } }
        { unsigned __int128 c = 0; for (int i = 0; i < 4; ++i) { c += (unsigned __int128)ap.l[i] + FQ_P.l[i]; ap.l[i] = (uint64_t)c; c >>= 64; } }
        printf("inverse_noncanonical %d\n", (int)(is_zero(inv(FQ_P)) && is_zero(inv(k2)) && is_zero(inv(k5)) && eq(inv(ap), inv(FQ_TWO))));
    }
    printf("g1_on_curve %d\n", (int)g1_on_curve(g));
    printf("g2_gen_on_curve %d\n", (int)g2_on_curve(h));
    printf("g2_tau_on_curve %d\n", (int)g2_on_curve(g2_tau_mainnet()));
    uint64_t r[4]; memcpy(r, FR_MODULUS_WORDS, 32);
    printf("g2_gen_order_r %d\n", (int)g2_mul(h, r).inf);
    printf("g1_gen_order_r %d\n", (int)g1_mul(g, r).inf);
    auto t0 = std::chrono::steady_clock::now();
    Fq12 e = pairing(g, h);
    auto t1 = std::chrono::steady_clock::now();
    printf("nondegenerate %d\n", (int)!fq12_is_one(e));
    printf("order_r %d\n", (int)fq12_is_one(fq12_pow(e, r, 4)));
    uint64_t a[4] = {0x1234567890abcdefULL, 0x0fedcba987654321ULL, 0x1111, 0}, b[4] = {0xdeadbeefcafef00dULL, 77, 0, 0};
    Fq12 eab = fq12_pow(fq12_pow(e, a, 4), b, 4);
    printf("bilinear %d\n", (int)fq12_eq(pairing(g1_mul(g, a), g2_mul(h, b)), eab));
    printf("bilinear_left %d\n", (int)fq12_eq(pairing(g1_mul(g, a), h), fq12_pow(e, a, 4)));
    printf("bilinear_right %d\n", (int)fq12_eq(pairing(g, g2_mul(h, b)), fq12_pow(e, b, 4)));
    printf("verify_true %d\n", (int)pairings_verify(g1_mul(g, a), g2_mul(h, b), g1_mul(g, b), g2_mul(h, a)));
    printf("verify_false %d\n", (int)!pairings_verify(g1_mul(g, a), g2_mul(h, b), g1_mul(g, b), h));
    printf("identity_pairs %d\n", (int)pairings_verify(g1_mul(g, r), h, g, g2_inf()));
    // fast path vs literal construction
    {
        G1 pa = g1_mul(g, a), pb = g1_mul(g, b);
        G2 qa = g2_mul(h, a), qb = g2_mul(h, b);
        G1 one_p[1] = {pa}; G2 one_q[1] = {qb};
        Fq12 fast = final_exponentiation_fast(miller_tate_product(one_p, one_q, 1));
        printf("fast_equals_literal %d\n", (int)fq12_eq(fast, pairing(pa, qb)));
        printf("final_exp_fast_equals_generic %d\n", (int)fq12_eq(final_exponentiation_fast(miller_tate(pa, qb)), final_exponentiation(miller_tate(pa, qb))));
        printf("final_exp_x_equals_straus %d\n", (int)fq12_eq(final_exponentiation_x(miller_tate(pa, qb)), final_exponentiation_fast(miller_tate(pa, qb))));
        printf("fq12_sqr_equals_mul %d\n", (int)(fq12_eq(sqr(e), mul(e, e)) && fq12_eq(sqr(miller_tate(pa, qb)), mul(miller_tate(pa, qb), miller_tate(pa, qb)))));
        {   // cyclotomic squaring == squaring on elements of the cyclotomic subgroup (pairing values are such elements)
            Fq12 c = fq12_pow(e, a, 4);
            printf("cyclotomic_sqr %d\n", (int)(fq12_eq(cyclotomic_sqr(e), sqr(e)) && fq12_eq(cyclotomic_sqr(c), sqr(c))));
        }
        {   // fixed-base tables against the generic double-and-add, incl. 0, 1, r - 1, r and a full-width value
            uint64_t zero[4] = {0, 0, 0, 0}, one[4] = {1, 0, 0, 0}, rm1[4], full[4] = {~0ULL, ~0ULL, ~0ULL, ~0ULL};
            memcpy(rm1, r, 32); rm1[0] -= 1;
            const uint64_t* ks[6] = {zero, one, rm1, r, a, full};
            bool ok1 = true, ok2 = true;
            for (int i = 0; i < 6; ++i) {
                G1 x1 = g1_mul_generator(ks[i]), y1 = g1_mul(g, ks[i]);
                ok1 = ok1 && x1.inf == y1.inf && (x1.inf || (eq(x1.x, y1.x) && eq(x1.y, y1.y)));
                G2 x2 = g2_mul_generator(ks[i]), y2 = g2_mul(h, ks[i]);
                ok2 = ok2 && x2.inf == y2.inf && (x2.inf || (eq(x2.x, y2.x) && eq(x2.y, y2.y)));
            }
            printf("fixed_base_g1 %d\n", (int)ok1);
            printf("fixed_base_g2 %d\n", (int)ok2);
        }
        Fq12 fi; bool okinv = fq12_inverse(e, fi);
        printf("fq12_inverse %d\n", (int)(okinv && fq12_is_one(mul(e, fi))));
        {
            Fq12 m = miller_tate(pa, qb), i1, i2;               // a generic element (not in any subgroup)
            const bool ok1 = fq12_inverse(m, i1), ok2 = fq12_inverse_norm(m, i2);
            printf("fq12_inverse_norm %d\n", (int)(ok1 && ok2 && fq12_eq(i1, i2) && fq12_is_one(mul(m, i2))));
        }
        printf("frobenius %d\n", (int)fq12_eq(frobenius(e, 1), fq12_pow(e, FQ_MODULUS_WORDS, 4)));
        printf("frobenius3 %d\n", (int)fq12_eq(frobenius(e, 3), frobenius(frobenius(frobenius(e, 1), 1), 1)));
        printf("reference_verify_true %d\n", (int)pairings_verify_reference(pa, qb, pb, qa));
        printf("reference_verify_false %d\n", (int)!pairings_verify_reference(pa, qb, pb, h));
        // optimal ate construction: a pairing of its own (different value from the Tate one), so its properties are checked directly
        {
            Fq12 ea = pairing_ate(g, h);
            printf("ate_nondegenerate %d\n", (int)!fq12_is_one(ea));
            printf("ate_order_r %d\n", (int)fq12_is_one(fq12_pow(ea, r, 4)));
            printf("ate_bilinear %d\n", (int)fq12_eq(pairing_ate(pa, qb), fq12_pow(fq12_pow(ea, a, 4), b, 4)));
            printf("ate_bilinear_left %d\n", (int)fq12_eq(pairing_ate(pa, h), fq12_pow(ea, a, 4)));
            printf("ate_bilinear_right %d\n", (int)fq12_eq(pairing_ate(g, qb), fq12_pow(ea, b, 4)));
            printf("ate_frobenius_is_p %d\n", (int)(eq(g2_frobenius(h).x, g2_mul(h, FQ_MODULUS_WORDS).x) && eq(g2_frobenius(h).y, g2_mul(h, FQ_MODULUS_WORDS).y)));
            printf("tate_verify_true %d\n", (int)pairings_verify_tate(pa, qb, pb, qa));
            printf("tate_verify_false %d\n", (int)!pairings_verify_tate(pa, qb, pb, h));
        }
        auto t2 = std::chrono::steady_clock::now();
        bool okv = pairings_verify(pa, qb, pb, qa);
        auto t3 = std::chrono::steady_clock::now();
        printf("fast_verify %d\n", (int)okv);
        fprintf(stderr, "pairings_verify (fast): %.1f ms\n", std::chrono::duration<double, std::milli>(t3 - t2).count());
    }
    fprintf(stderr, "one pairing: %.1f ms\n", std::chrono::duration<double, std::milli>(t1 - t0).count());
    return 0;
}
