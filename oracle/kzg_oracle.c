/*
 * oracle/kzg_oracle.c — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, gcc, no dependencies) of the reference's KZG-BN254 prover hot path, used
 * ONLY as the parity checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 * The product (rust-kzg-bn254_amd/, include/) never links, imports or calls this file.
 *
 * What it restates, and from where:
 *   - prover/src/kzg.rs:84-104 commit_eval_form, :107-125 commit_coeff_form, :128-178
 *     compute_proof_impl, :237-260 compute_quotient_eval_on_domain, :263-285 g1_ifft
 *   - primitives/src/polynomial.rs:130-140 (IFFT), :241-251 (FFT)
 *   - primitives/src/helpers.rs:40-57 to_fr_array, :175-226 read_g1_point_from_bytes_be,
 *     :328-337 g1_lincomb, :411-472 compute_challenge, :475-535
 *     evaluate_polynomial_in_evaluation_form, :553-610 roots of unity, :823-840 pad_payload
 *   - primitives/src/arith.rs:4-55 montgomery_reduce
 * The arithmetic itself lives in third-party crates that are NOT in /root/reference (arkworks 0.5:
 * ark-ff, ark-ec, ark-poly, ark-bn254, ark-serialize — Cargo.toml:57-62, no Cargo.lock, so the
 * patch version is unpinned; sha2 0.10.8).  Their published algorithms are restated here
 * (Montgomery CIOS; Jacobian short-Weierstrass group law; signed-window Pippenger
 * `VariableBaseMSM::msm`; radix-2 `EvaluationDomain::{fft,ifft}`); group and field results are
 * canonical, so any correct restatement is bit-identical to arkworks.
 *
 * PINNING: checked in tests/test_oracle.py against every golden vector the reference holds for this
 * path: g1.point <-> srs.g1.points.string (3000 decompressions),
 * lagrangeG1SRS.txt (g1_ifft(64)), kzg.proof.eq.input (40 proofs), blobs.txt <-> blobs-from-fr.txt,
 * PRIMITIVE_ROOTS_OF_UNITY decimals, arith.rs KATs, pad_payload byte vectors.  The reference is
 * Rust and no Rust toolchain exists in the image, so oracle/_ref cannot be built.
 *
 * Wire format everywhere: Fr/Fq = 4 LE u64 limbs, Montgomery R = 2^256, canonical; affine G1 =
 * x[4] || y[4]; the identity is encoded as all-zero (x = y = 0, which is not on the curve), the
 * same coordinates arkworks' `G1Affine::identity()` carries next to its `infinity` flag.
 */
#include "field.h"
#include <stdlib.h>
#include <pthread.h>

#define EXPORT __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------ */
/* G1: y^2 = x^3 + 3 over Fq (helpers.rs:202, :244), Jacobian coordinates like ark-ec's         */
/* short_weierstrass::Projective.                                                              */
/* ------------------------------------------------------------------------------------------ */
typedef struct { fe x, y; } g1a;            /* affine; identity = (0,0) */
typedef struct { fe x, y, z; } g1j;         /* Jacobian; identity = z == 0 */

static inline int g1a_is_inf(const g1a *p) { return fe_is_zero(&p->x) && fe_is_zero(&p->y); }
static inline int g1j_is_inf(const g1j *p) { return fe_is_zero(&p->z); }
static inline void g1j_set_inf(g1j *p) { p->x = FQ.one; p->y = FQ.one; memset(&p->z, 0, sizeof(fe)); }
static inline void g1j_from_affine(g1j *r, const g1a *p) {
    if (g1a_is_inf(p)) { g1j_set_inf(r); return; }
    r->x = p->x; r->y = p->y; r->z = FQ.one;
}

/* dbl-2009-l, a = 0 */
static void g1j_double(g1j *r, const g1j *p) {
    if (g1j_is_inf(p)) { *r = *p; return; }
    fe a, b, c, d, e, f, t, x3, y3, z3;
    fe_sqr(&FQ, &a, &p->x); fe_sqr(&FQ, &b, &p->y); fe_sqr(&FQ, &c, &b);
    fe_add(&FQ, &t, &p->x, &b); fe_sqr(&FQ, &t, &t); fe_sub(&FQ, &t, &t, &a); fe_sub(&FQ, &t, &t, &c);
    fe_dbl(&FQ, &d, &t);
    fe_dbl(&FQ, &e, &a); fe_add(&FQ, &e, &e, &a);
    fe_sqr(&FQ, &f, &e);
    fe_dbl(&FQ, &t, &d); fe_sub(&FQ, &x3, &f, &t);
    fe_mul(&FQ, &z3, &p->y, &p->z); fe_dbl(&FQ, &z3, &z3);
    fe_sub(&FQ, &t, &d, &x3); fe_mul(&FQ, &y3, &e, &t);
    fe_dbl(&FQ, &c, &c); fe_dbl(&FQ, &c, &c); fe_dbl(&FQ, &c, &c);
    fe_sub(&FQ, &y3, &y3, &c);
    r->x = x3; r->y = y3; r->z = z3;
}

/* add-2007-bl with the exceptional cases handled explicitly */
static void g1j_add(g1j *r, const g1j *p, const g1j *q) {
    if (g1j_is_inf(p)) { *r = *q; return; }
    if (g1j_is_inf(q)) { *r = *p; return; }
    fe z1z1, z2z2, u1, u2, s1, s2, h, i, j, rr, v, t, x3, y3, z3;
    fe_sqr(&FQ, &z1z1, &p->z); fe_sqr(&FQ, &z2z2, &q->z);
    fe_mul(&FQ, &u1, &p->x, &z2z2); fe_mul(&FQ, &u2, &q->x, &z1z1);
    fe_mul(&FQ, &s1, &p->y, &q->z); fe_mul(&FQ, &s1, &s1, &z2z2);
    fe_mul(&FQ, &s2, &q->y, &p->z); fe_mul(&FQ, &s2, &s2, &z1z1);
    if (fe_eq(&u1, &u2)) {
        if (fe_eq(&s1, &s2)) { g1j_double(r, p); return; }
        g1j_set_inf(r); return;
    }
    fe_sub(&FQ, &h, &u2, &u1);
    fe_dbl(&FQ, &i, &h); fe_sqr(&FQ, &i, &i);
    fe_mul(&FQ, &j, &h, &i);
    fe_sub(&FQ, &rr, &s2, &s1); fe_dbl(&FQ, &rr, &rr);
    fe_mul(&FQ, &v, &u1, &i);
    fe_sqr(&FQ, &x3, &rr); fe_sub(&FQ, &x3, &x3, &j); fe_dbl(&FQ, &t, &v); fe_sub(&FQ, &x3, &x3, &t);
    fe_sub(&FQ, &t, &v, &x3); fe_mul(&FQ, &y3, &rr, &t);
    fe_mul(&FQ, &t, &s1, &j); fe_dbl(&FQ, &t, &t); fe_sub(&FQ, &y3, &y3, &t);
    fe_add(&FQ, &z3, &p->z, &q->z); fe_sqr(&FQ, &z3, &z3); fe_sub(&FQ, &z3, &z3, &z1z1);
    fe_sub(&FQ, &z3, &z3, &z2z2); fe_mul(&FQ, &z3, &z3, &h);
    r->x = x3; r->y = y3; r->z = z3;
}

/* madd-2007-bl (q affine), exceptional cases handled; `neg` adds -q */
static void g1j_add_affine(g1j *r, const g1j *p, const g1a *q, int neg) {
    if (g1a_is_inf(q)) { *r = *p; return; }
    g1a qq = *q; if (neg) fe_neg(&FQ, &qq.y, &qq.y);
    if (g1j_is_inf(p)) { g1j_from_affine(r, &qq); return; }
    fe z1z1, u2, s2, h, hh, i, j, rr, v, t, x3, y3, z3;
    fe_sqr(&FQ, &z1z1, &p->z);
    fe_mul(&FQ, &u2, &qq.x, &z1z1);
    fe_mul(&FQ, &s2, &qq.y, &p->z); fe_mul(&FQ, &s2, &s2, &z1z1);
    if (fe_eq(&p->x, &u2)) {
        if (fe_eq(&p->y, &s2)) { g1j_double(r, p); return; }
        g1j_set_inf(r); return;
    }
    fe_sub(&FQ, &h, &u2, &p->x); fe_sqr(&FQ, &hh, &h);
    fe_dbl(&FQ, &i, &hh); fe_dbl(&FQ, &i, &i);
    fe_mul(&FQ, &j, &h, &i);
    fe_sub(&FQ, &rr, &s2, &p->y); fe_dbl(&FQ, &rr, &rr);
    fe_mul(&FQ, &v, &p->x, &i);
    fe_sqr(&FQ, &x3, &rr); fe_sub(&FQ, &x3, &x3, &j); fe_dbl(&FQ, &t, &v); fe_sub(&FQ, &x3, &x3, &t);
    fe_sub(&FQ, &t, &v, &x3); fe_mul(&FQ, &y3, &rr, &t);
    fe_mul(&FQ, &t, &p->y, &j); fe_dbl(&FQ, &t, &t); fe_sub(&FQ, &y3, &y3, &t);
    fe_add(&FQ, &z3, &p->z, &h); fe_sqr(&FQ, &z3, &z3); fe_sub(&FQ, &z3, &z3, &z1z1); fe_sub(&FQ, &z3, &z3, &hh);
    r->x = x3; r->y = y3; r->z = z3;
}

/* `into_affine()` (kzg.rs:101, :122): one Fq inversion; identity -> (0,0) */
static void g1j_to_affine(g1a *r, const g1j *p) {
    if (g1j_is_inf(p)) { memset(r, 0, sizeof(*r)); return; }
    fe zi, zi2, zi3; fe_inv(&FQ, &zi, &p->z);
    fe_sqr(&FQ, &zi2, &zi); fe_mul(&FQ, &zi3, &zi2, &zi);
    fe_mul(&FQ, &r->x, &p->x, &zi2); fe_mul(&FQ, &r->y, &p->y, &zi3);
}

/* double-and-add by a canonical 256-bit integer k (`point *= Fr`, `G1Affine * Fr`) */
static void g1j_mul_canonical(g1j *r, const g1j *p, const fe *k) {
    g1j acc; g1j_set_inf(&acc);
    for (int i = 255; i >= 0; --i) {
        g1j_double(&acc, &acc);
        if ((k->l[i >> 6] >> (i & 63)) & 1) g1j_add(&acc, &acc, p);
    }
    *r = acc;
}
static void g1j_mul_fr(g1j *r, const g1j *p, const fe *k_mont) {
    fe k; fe_to_canonical(&FR, &k, k_mont); g1j_mul_canonical(r, p, &k);
}

static int g1a_on_curve(const g1a *p) {
    if (g1a_is_inf(p)) return 1;
    fe l, rr, three; fe_sqr(&FQ, &l, &p->y);
    fe_sqr(&FQ, &rr, &p->x); fe_mul(&FQ, &rr, &rr, &p->x);
    fe_from_u64(&FQ, &three, 3); fe_add(&FQ, &rr, &rr, &three);
    return fe_eq(&l, &rr);
}

/* ------------------------------------------------------------------------------------------ */
/* MSM                                                                                         */
/* ------------------------------------------------------------------------------------------ */
/* Definition of the result: sum_i s_i * P_i by per-term double-and-add (no windowing). */
static void msm_naive(g1j *out, const g1a *bases, const fe *scalars, size_t n) {
    g1j acc; g1j_set_inf(&acc);
    for (size_t i = 0; i < n; ++i) {
        g1j b, t; g1j_from_affine(&b, &bases[i]);
        g1j_mul_fr(&t, &b, &scalars[i]);
        g1j_add(&acc, &acc, &t);
    }
    *out = acc;
}

/* ark-ec 0.5 `VariableBaseMSM::msm` -> msm_bigint_wnaf, restated (the routine behind kzg.rs:100,
 * :121 and helpers.rs:332): window c = 3 if n < 32 else ceil(log2 n)*69/100 + 2; signed radix-2^c
 * digits (make_digits); one task per window (rayon par_iter over windows -> pthreads here); 2^(c-1)
 * buckets filled with mixed adds; running-sum bucket reduction; Horner over windows with c
 * doublings each. */
static unsigned ceil_log2(size_t n) { unsigned k = 0; while (((size_t)1 << k) < n) ++k; return k; }
static unsigned ark_window(size_t n) { return n < 32 ? 3u : ceil_log2(n) * 69u / 100u + 2u; }

static void make_digits(const fe *k, unsigned w, unsigned num_bits, int64_t *out) {
    uint64_t radix = (uint64_t)1 << w, mask = radix - 1, carry = 0;
    unsigned count = (num_bits + w - 1) / w;
    for (unsigned i = 0; i < count; ++i) {
        unsigned off = i * w, idx = off / 64, bit = off % 64;
        uint64_t buf;
        if (bit < 64 - w || idx == 3) buf = k->l[idx] >> bit;
        else buf = (k->l[idx] >> bit) | (k->l[idx + 1] << (64 - bit));
        uint64_t coef = carry + (buf & mask);
        carry = (coef + radix / 2) >> w;
        int64_t digit = (int64_t)coef - (int64_t)(carry << w);
        if (i == count - 1) digit += (int64_t)(carry << w);
        out[i] = digit;
    }
}

typedef struct {
    const g1a *bases; const int64_t *digits; size_t n; unsigned c, count;
    g1j *window_sums; volatile int *next; pthread_mutex_t *mu;
} pip_job;

static void pip_window(const pip_job *J, unsigned w) {
    size_t nb = (size_t)1 << (J->c - 1);
    /* the last window's digit is not re-centred (make_digits), so it can reach 2^c - 1 */
    if (w == J->count - 1) nb = (size_t)1 << J->c;
    g1j *buckets = (g1j *)malloc(nb * sizeof(g1j));
    for (size_t b = 0; b < nb; ++b) g1j_set_inf(&buckets[b]);
    for (size_t i = 0; i < J->n; ++i) {
        int64_t d = J->digits[i * J->count + w];
        if (d > 0) g1j_add_affine(&buckets[d - 1], &buckets[d - 1], &J->bases[i], 0);
        else if (d < 0) g1j_add_affine(&buckets[-d - 1], &buckets[-d - 1], &J->bases[i], 1);
    }
    g1j running, res; g1j_set_inf(&running); g1j_set_inf(&res);
    for (size_t b = nb; b-- > 0;) { g1j_add(&running, &running, &buckets[b]); g1j_add(&res, &res, &running); }
    free(buckets);
    J->window_sums[w] = res;
}
static void *pip_worker(void *arg) {
    pip_job *J = (pip_job *)arg;
    for (;;) {
        pthread_mutex_lock(J->mu); int w = (*J->next)++; pthread_mutex_unlock(J->mu);
        if ((unsigned)w >= J->count) break;
        pip_window(J, (unsigned)w);
    }
    return NULL;
}
static void msm_pippenger(g1j *out, const g1a *bases, const fe *scalars, size_t n, int threads) {
    if (n == 0) { g1j_set_inf(out); return; }
    unsigned c = ark_window(n), num_bits = 254, count = (num_bits + c - 1) / c;
    int64_t *digits = (int64_t *)malloc(n * count * sizeof(int64_t));
    for (size_t i = 0; i < n; ++i) {
        fe k; fe_to_canonical(&FR, &k, &scalars[i]);       /* `into_bigint()` */
        make_digits(&k, c, num_bits, &digits[i * count]);
    }
    g1j *sums = (g1j *)malloc(count * sizeof(g1j));
    volatile int next = 0; pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
    pip_job J = {bases, digits, n, c, count, sums, &next, &mu};
    if (threads < 1) threads = 1;
    if ((unsigned)threads > count) threads = (int)count;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    for (int t = 1; t < threads; ++t) pthread_create(&th[t], NULL, pip_worker, &J);
    pip_worker(&J);
    for (int t = 1; t < threads; ++t) pthread_join(th[t], NULL);
    free(th);
    g1j total; g1j_set_inf(&total);
    for (unsigned w = count - 1; w >= 1; --w) {
        g1j_add(&total, &total, &sums[w]);
        for (unsigned k = 0; k < c; ++k) g1j_double(&total, &total);
    }
    g1j_add(out, &sums[0], &total);
    free(sums); free(digits);
}

/* ------------------------------------------------------------------------------------------ */
/* Fr radix-2 NTT (ark-poly Radix2EvaluationDomain::{fft,ifft}; polynomial.rs:131-135, :242-246) */
/* natural order in and out, omega = 5^((r-1)/n) = PRIMITIVE_ROOTS_OF_UNITY[log2 n]            */
/* (consts.rs:22-52); ifft = fft with omega^-1, then scaled by n^-1.                            */
/* ------------------------------------------------------------------------------------------ */
static int log2_exact(size_t n) { if (n == 0 || (n & (n - 1))) return -1; int k = 0; while (((size_t)1 << k) < n) ++k; return k; }

static void fr_root_of_unity(fe *w, int log_n) {
    /* 5^((r-1)/2^log_n) */
    fe e = FR.m, one = {{1, 0, 0, 0}}, five;
    fe_sub_raw(&e, &e, &one);
    for (int s = 0; s < log_n; ++s)
        for (int i = 0; i < 4; ++i) e.l[i] = (e.l[i] >> 1) | (i < 3 ? e.l[i + 1] << 63 : 0);
    fe_from_u64(&FR, &five, 5);
    fe_pow(&FR, w, &five, &e);
}
static size_t bitrev(size_t x, int bits) { size_t r = 0; for (int i = 0; i < bits; ++i) { r = (r << 1) | (x & 1); x >>= 1; } return r; }

static int fr_ntt_inplace(fe *a, size_t n, int inverse) {
    int ln = log2_exact(n);
    if (ln < 0 || ln > 28) return -1;
    fe w; fr_root_of_unity(&w, ln);
    if (inverse) fe_inv(&FR, &w, &w);
    for (size_t i = 0; i < n; ++i) { size_t j = bitrev(i, ln); if (i < j) { fe t = a[i]; a[i] = a[j]; a[j] = t; } }
    for (int s = 1; s <= ln; ++s) {
        size_t m = (size_t)1 << s, half = m >> 1;
        fe wm = w;                                  /* w^(n/m) */
        for (int k = 0; k < ln - s; ++k) fe_sqr(&FR, &wm, &wm);
        for (size_t k = 0; k < n; k += m) {
            fe tw = FR.one;
            for (size_t j = 0; j < half; ++j) {
                fe t, u = a[k + j];
                fe_mul(&FR, &t, &tw, &a[k + j + half]);
                fe_add(&FR, &a[k + j], &u, &t);
                fe_sub(&FR, &a[k + j + half], &u, &t);
                fe_mul(&FR, &tw, &tw, &wm);
            }
        }
    }
    if (inverse) {
        fe ninv; fe_from_u64(&FR, &ninv, (uint64_t)n); fe_inv(&FR, &ninv, &ninv);
        for (size_t i = 0; i < n; ++i) fe_mul(&FR, &a[i], &a[i], &ninv);
    }
    return 0;
}

/* The same transform with the butterflies of every layer chunked over `threads` pthreads (ark-poly with the `parallel`
 * feature splits each layer over the rayon pool): identical results (exact arithmetic), used as the CPU baseline of the NTT. */
typedef struct { fe *a; const fe *tw; size_t n; int ln, inverse, tid, threads; pthread_barrier_t *bar; fe ninv; } ntt_job;
static void *ntt_worker(void *arg) {
    ntt_job *J = (ntt_job *)arg;
    const size_t n = J->n, nb = n / 2;
    const size_t lo = nb * (size_t)J->tid / (size_t)J->threads, hi = nb * (size_t)(J->tid + 1) / (size_t)J->threads;
    for (int s = 1; s <= J->ln; ++s) {
        const size_t half = (size_t)1 << (s - 1), step = n >> s;           /* twiddle of butterfly j: w^(j n/m) */
        for (size_t b = lo; b < hi; ++b) {
            const size_t j = b & (half - 1), k = (b >> (s - 1)) << s;
            fe t, u = J->a[k + j];
            fe_mul(&FR, &t, &J->tw[j * step], &J->a[k + j + half]);
            fe_add(&FR, &J->a[k + j], &u, &t);
            fe_sub(&FR, &J->a[k + j + half], &u, &t);
        }
        pthread_barrier_wait(J->bar);
    }
    if (J->inverse) {
        const size_t l2 = n * (size_t)J->tid / (size_t)J->threads, h2 = n * (size_t)(J->tid + 1) / (size_t)J->threads;
        for (size_t i = l2; i < h2; ++i) fe_mul(&FR, &J->a[i], &J->a[i], &J->ninv);
    }
    return NULL;
}
static int fr_ntt_inplace_mt(fe *a, size_t n, int inverse, int threads) {
    int ln = log2_exact(n);
    if (ln < 0 || ln > 28) return -1;
    if (threads < 1) threads = 1;
    if (n < 2 || (size_t)threads > n / 2) return fr_ntt_inplace(a, n, inverse);
    fe w; fr_root_of_unity(&w, ln);
    if (inverse) fe_inv(&FR, &w, &w);
    for (size_t i = 0; i < n; ++i) { size_t j = bitrev(i, ln); if (i < j) { fe t = a[i]; a[i] = a[j]; a[j] = t; } }
    fe *tw = (fe *)malloc((n / 2) * sizeof(fe));                            /* w^j, j < n/2 (ark-poly precomputes the roots too) */
    tw[0] = FR.one;
    for (size_t j = 1; j < n / 2; ++j) fe_mul(&FR, &tw[j], &tw[j - 1], &w);
    pthread_barrier_t bar; pthread_barrier_init(&bar, NULL, (unsigned)threads);
    ntt_job *jobs = (ntt_job *)malloc(sizeof(ntt_job) * (size_t)threads);
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    fe ninv = FR.one;
    if (inverse) { fe_from_u64(&FR, &ninv, (uint64_t)n); fe_inv(&FR, &ninv, &ninv); }
    for (int t = 0; t < threads; ++t) {
        ntt_job j = {a, tw, n, ln, inverse, t, threads, &bar, ninv};
        jobs[t] = j;
    }
    for (int t = 1; t < threads; ++t) pthread_create(&th[t], NULL, ntt_worker, &jobs[t]);
    ntt_worker(&jobs[0]);
    for (int t = 1; t < threads; ++t) pthread_join(th[t], NULL);
    pthread_barrier_destroy(&bar);
    free(th); free(jobs); free(tw);
    return 0;
}

/* G1-group IFFT: the same butterflies with group elements as data (kzg.rs:275-279) */
static int g1_ifft_inplace(g1j *a, size_t n) {
    int ln = log2_exact(n);
    if (ln < 0 || ln > 28) return -1;
    fe w; fr_root_of_unity(&w, ln); fe_inv(&FR, &w, &w);
    for (size_t i = 0; i < n; ++i) { size_t j = bitrev(i, ln); if (i < j) { g1j t = a[i]; a[i] = a[j]; a[j] = t; } }
    for (int s = 1; s <= ln; ++s) {
        size_t m = (size_t)1 << s, half = m >> 1;
        fe wm = w;
        for (int k = 0; k < ln - s; ++k) fe_sqr(&FR, &wm, &wm);
        for (size_t k = 0; k < n; k += m) {
            fe tw = FR.one;
            for (size_t j = 0; j < half; ++j) {
                g1j t, u = a[k + j], neg;
                g1j_mul_fr(&t, &a[k + j + half], &tw);
                g1j_add(&a[k + j], &u, &t);
                neg = t; fe_neg(&FQ, &neg.y, &neg.y);
                g1j_add(&a[k + j + half], &u, &neg);
                fe_mul(&FR, &tw, &tw, &wm);
            }
        }
    }
    fe ninv; fe_from_u64(&FR, &ninv, (uint64_t)n); fe_inv(&FR, &ninv, &ninv);
    for (size_t i = 0; i < n; ++i) g1j_mul_fr(&a[i], &a[i], &ninv);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* SHA-256 (sha2 0.10.8, FIPS 180-4) for the Fiat-Shamir challenge (helpers.rs:382-390)          */
/* ------------------------------------------------------------------------------------------ */
static const uint32_t K256[64] = {
 0x428a2f98,0x71374491,0xb5c0fbcf,0xe9b5dba5,0x3956c25b,0x59f111f1,0x923f82a4,0xab1c5ed5,0xd807aa98,0x12835b01,0x243185be,0x550c7dc3,0x72be5d74,0x80deb1fe,0x9bdc06a7,0xc19bf174,
 0xe49b69c1,0xefbe4786,0x0fc19dc6,0x240ca1cc,0x2de92c6f,0x4a7484aa,0x5cb0a9dc,0x76f988da,0x983e5152,0xa831c66d,0xb00327c8,0xbf597fc7,0xc6e00bf3,0xd5a79147,0x06ca6351,0x14292967,
 0x27b70a85,0x2e1b2138,0x4d2c6dfc,0x53380d13,0x650a7354,0x766a0abb,0x81c2c92e,0x92722c85,0xa2bfe8a1,0xa81a664b,0xc24b8b70,0xc76c51a3,0xd192e819,0xd6990624,0xf40e3585,0x106aa070,
 0x19a4c116,0x1e376c08,0x2748774c,0x34b0bcb5,0x391c0cb3,0x4ed8aa4a,0x5b9cca4f,0x682e6ff3,0x748f82ee,0x78a5636f,0x84c87814,0x8cc70208,0x90befffa,0xa4506ceb,0xbef9a3f7,0xc67178f2};
static inline uint32_t ror(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
static void sha256_block(uint32_t h[8], const uint8_t *p) {
    uint32_t w[64];
    for (int i = 0; i < 16; ++i) w[i] = (uint32_t)p[4*i] << 24 | (uint32_t)p[4*i+1] << 16 | (uint32_t)p[4*i+2] << 8 | p[4*i+3];
    for (int i = 16; i < 64; ++i) {
        uint32_t s0 = ror(w[i-15],7) ^ ror(w[i-15],18) ^ (w[i-15] >> 3), s1 = ror(w[i-2],17) ^ ror(w[i-2],19) ^ (w[i-2] >> 10);
        w[i] = w[i-16] + s0 + w[i-7] + s1;
    }
    uint32_t a=h[0],b=h[1],c=h[2],d=h[3],e=h[4],f=h[5],g=h[6],hh=h[7];
    for (int i = 0; i < 64; ++i) {
        uint32_t S1 = ror(e,6)^ror(e,11)^ror(e,25), ch = (e&f)^(~e&g), t1 = hh+S1+ch+K256[i]+w[i];
        uint32_t S0 = ror(a,2)^ror(a,13)^ror(a,22), mj = (a&b)^(a&c)^(b&c), t2 = S0+mj;
        hh=g; g=f; f=e; e=d+t1; d=c; c=b; b=a; a=t1+t2;
    }
    h[0]+=a;h[1]+=b;h[2]+=c;h[3]+=d;h[4]+=e;h[5]+=f;h[6]+=g;h[7]+=hh;
}
EXPORT void orc_sha256(const uint8_t *msg, size_t len, uint8_t out[32]) {
    uint32_t h[8] = {0x6a09e667,0xbb67ae85,0x3c6ef372,0xa54ff53a,0x510e527f,0x9b05688c,0x1f83d9ab,0x5be0cd19};
    size_t i = 0;
    for (; i + 64 <= len; i += 64) sha256_block(h, msg + i);
    uint8_t tail[128]; size_t rem = len - i; memset(tail, 0, sizeof tail); memcpy(tail, msg + i, rem);
    tail[rem] = 0x80;
    size_t tl = rem + 9 <= 64 ? 64 : 128;
    uint64_t bits = (uint64_t)len * 8;
    for (int b = 0; b < 8; ++b) tail[tl - 1 - b] = (uint8_t)(bits >> (8 * b));
    sha256_block(h, tail); if (tl == 128) sha256_block(h, tail + 64);
    for (int k = 0; k < 8; ++k) { out[4*k] = h[k] >> 24; out[4*k+1] = h[k] >> 16; out[4*k+2] = h[k] >> 8; out[4*k+3] = h[k]; }
}

/* ------------------------------------------------------------------------------------------ */
/* Exported API (ctypes).  Pointers to u64 limbs reinterpret as fe / g1a arrays.                */
/* ------------------------------------------------------------------------------------------ */
EXPORT void orc_init(void) { oracle_init(); }
static const field_t *pick(int which) { return which ? &FR : &FQ; }   /* 0 = Fq, 1 = Fr */

EXPORT void orc_constants(int which, uint64_t modulus[4], uint64_t *inv, uint64_t one[4], uint64_t r2[4]) {
    oracle_init(); const field_t *F = pick(which);
    memcpy(modulus, F->m.l, 32); *inv = F->inv; memcpy(one, F->one.l, 32); memcpy(r2, F->r2.l, 32);
}
EXPORT void orc_f_mul(int which, uint64_t *r, const uint64_t *a, const uint64_t *b) { oracle_init(); fe_mul(pick(which), (fe *)r, (const fe *)a, (const fe *)b); }
EXPORT void orc_f_add(int which, uint64_t *r, const uint64_t *a, const uint64_t *b) { oracle_init(); fe_add(pick(which), (fe *)r, (const fe *)a, (const fe *)b); }
EXPORT void orc_f_sub(int which, uint64_t *r, const uint64_t *a, const uint64_t *b) { oracle_init(); fe_sub(pick(which), (fe *)r, (const fe *)a, (const fe *)b); }
EXPORT int  orc_f_inv(int which, uint64_t *r, const uint64_t *a) { oracle_init(); return fe_inv(pick(which), (fe *)r, (const fe *)a); }
EXPORT void orc_f_from_canonical(int which, uint64_t *r, const uint64_t *a) { oracle_init(); fe_from_canonical(pick(which), (fe *)r, (const fe *)a); }
EXPORT void orc_f_to_canonical(int which, uint64_t *r, const uint64_t *a) { oracle_init(); fe_to_canonical(pick(which), (fe *)r, (const fe *)a); }
EXPORT void orc_f_from_be_bytes_mod_order(int which, uint64_t *r, const uint8_t *b, size_t len) { oracle_init(); fe_from_be_bytes_mod_order(pick(which), (fe *)r, b, len); }
EXPORT void orc_f_to_be_bytes(int which, uint8_t out[32], const uint64_t *a) { oracle_init(); fe_to_be_bytes(pick(which), out, (const fe *)a); }

/* primitives/src/arith.rs:4-55 montgomery_reduce(r0..r3): (r0..r3) * R^-1 mod p, fully reduced */
EXPORT void orc_montgomery_reduce(const uint64_t in[4], uint64_t out[4]) {
    oracle_init(); fe_to_canonical(&FQ, (fe *)out, (const fe *)in);
}

/* --- points --- */
EXPORT int orc_g1_is_on_curve(const uint64_t xy[8]) { oracle_init(); return g1a_on_curve((const g1a *)xy); }
EXPORT void orc_g1_add(uint64_t out[8], const uint64_t a[8], const uint64_t b[8]) {
    oracle_init(); g1j p, q; g1j_from_affine(&p, (const g1a *)a); g1j_from_affine(&q, (const g1a *)b);
    g1j_add(&p, &p, &q); g1j_to_affine((g1a *)out, &p);
}
EXPORT void orc_g1_neg(uint64_t out[8], const uint64_t a[8]) {
    oracle_init(); g1a p = *(const g1a *)a; if (!g1a_is_inf(&p)) fe_neg(&FQ, &p.y, &p.y); *(g1a *)out = p;
}
EXPORT void orc_g1_scalar_mul(uint64_t out[8], const uint64_t p[8], const uint64_t k_mont[4]) {
    oracle_init(); g1j b, t; g1j_from_affine(&b, (const g1a *)p); g1j_mul_fr(&t, &b, (const fe *)k_mont);
    g1j_to_affine((g1a *)out, &t);
}
/* Jacobian (x,y,z Montgomery, 12 limbs) -> affine */
EXPORT void orc_g1_jacobian_to_affine(uint64_t out[8], const uint64_t xyz[12]) {
    oracle_init(); g1j_to_affine((g1a *)out, (const g1j *)xyz);
}

/* helpers.rs:151-173: y > (p-1)/2 on the canonical integer */
static int lexicographically_largest(const fe *y_mont) {
    fe c; fe_to_canonical(&FQ, &c, y_mont);
    static const fe half1 = {{0x9E10460B6C3E7EA4ULL, 0xCBC0B548B438E546ULL, 0xDC2822DB40C0AC2EULL, 0x183227397098D014ULL}};
    fe t; return fe_sub_raw(&t, &c, &half1) == 0;
}
/* helpers.rs:175-226 read_g1_point_from_bytes_be (gnark compressed, big-endian, flags in the top
 * two bits of byte 0).  Returns 0 ok, -1 bad infinity encoding, -2 not on curve. */
EXPORT int orc_g1_decompress_be(const uint8_t in[32], uint64_t out_xy[8]) {
    oracle_init();
    uint8_t flag = in[0] & 0xC0;
    if (flag == 0x40) {
        if (in[0] & 0x3F) return -1;
        for (int i = 1; i < 32; ++i) if (in[i]) return -1;
        memset(out_xy, 0, 64); return 0;
    }
    uint8_t xb[32]; memcpy(xb, in, 32); xb[0] &= 0x3F;
    g1a p; fe three, y2;
    fe_from_be_bytes_mod_order(&FQ, &p.x, xb, 32);
    fe_sqr(&FQ, &y2, &p.x); fe_mul(&FQ, &y2, &y2, &p.x); fe_from_u64(&FQ, &three, 3); fe_add(&FQ, &y2, &y2, &three);
    if (!fq_sqrt(&p.y, &y2)) return -2;
    if (lexicographically_largest(&p.y)) { if (flag == 0x80) fe_neg(&FQ, &p.y, &p.y); }
    else if (flag == 0xC0) fe_neg(&FQ, &p.y, &p.y);
    memcpy(out_xy, &p, 64);
    return 0;
}
/* ark-serialize `serialize_compressed` of a G1Affine (helpers.rs:456-459): x little-endian, bit 7 of
 * the last byte = "y is negative" (y > -y), bit 6 = infinity. */
EXPORT void orc_g1_serialize_compressed_ark(const uint64_t xy[8], uint8_t out[32]) {
    oracle_init(); const g1a *p = (const g1a *)xy;
    if (g1a_is_inf(p)) { memset(out, 0, 32); out[31] |= 0x40; return; }
    fe c; fe_to_canonical(&FQ, &c, &p->x);
    for (int i = 0; i < 4; ++i) for (int b = 0; b < 8; ++b) out[8 * i + b] = (uint8_t)(c.l[i] >> (8 * b));
    if (lexicographically_largest(&p->y)) out[31] |= 0x80;
}

/* --- MSM --- */
EXPORT int orc_msm_naive(const uint64_t *bases, const uint64_t *scalars, size_t n, uint64_t out[8]) {
    oracle_init(); g1j r; msm_naive(&r, (const g1a *)bases, (const fe *)scalars, n); g1j_to_affine((g1a *)out, &r); return 0;
}
EXPORT int orc_msm_pippenger(const uint64_t *bases, const uint64_t *scalars, size_t n, uint64_t out[8], int threads) {
    oracle_init(); g1j r; msm_pippenger(&r, (const g1a *)bases, (const fe *)scalars, n, threads); g1j_to_affine((g1a *)out, &r); return 0;
}
EXPORT unsigned orc_ark_window(size_t n) { return ark_window(n); }

/* --- NTT --- */
EXPORT int orc_fr_ntt(uint64_t *data, size_t n, int inverse) { oracle_init(); return fr_ntt_inplace((fe *)data, n, inverse); }
EXPORT int orc_fr_ntt_mt(uint64_t *data, size_t n, int inverse, int threads) { oracle_init(); return fr_ntt_inplace_mt((fe *)data, n, inverse, threads); }
EXPORT void orc_fr_root_of_unity(int log_n, uint64_t out[4]) { oracle_init(); fr_root_of_unity((fe *)out, log_n); }
/* kzg.rs:263-285 g1_ifft: returns -1 on "length provided is not a power of 2" */
EXPORT int orc_g1_ifft(const uint64_t *points, size_t n, uint64_t *out) {
    oracle_init();
    if (log2_exact(n) < 0) return -1;
    g1j *a = (g1j *)malloc(n * sizeof(g1j));
    for (size_t i = 0; i < n; ++i) g1j_from_affine(&a[i], &((const g1a *)points)[i]);
    int rc = g1_ifft_inplace(a, n);
    if (rc == 0) for (size_t i = 0; i < n; ++i) g1j_to_affine(&((g1a *)out)[i], &a[i]);
    free(a); return rc;
}

/* --- roots of unity (helpers.rs:553-610) --- */
static size_t next_pow2(size_t x) { size_t p = 1; while (p < x) p <<= 1; return p; }
EXPORT long orc_calculate_roots_of_unity(uint64_t len_bytes, uint64_t *out, size_t cap) {
    oracle_init();
    if (len_bytes == 0) return -1;                               /* "Length of data after padding is 0" */
    uint64_t elems = (len_bytes + 31) / 32;
    if (elems > 268435456ULL) return -2;                         /* MAINNET_SRS_G1_SIZE */
    size_t n = next_pow2((size_t)elems);
    if (n > cap) return -3;
    fe w; fr_root_of_unity(&w, log2_exact(n));
    fe cur = FR.one;
    for (size_t i = 0; i < n; ++i) { ((fe *)out)[i] = cur; fe_mul(&FR, &cur, &cur, &w); }
    return (long)n;
}

/* --- blob codec --- */
EXPORT size_t orc_pad_payload(const uint8_t *in, size_t len, uint8_t *out) {     /* helpers.rs:823-840 */
    size_t chunks = (len + 30) / 31, outlen = chunks * 32;
    memset(out, 0, outlen);
    for (size_t e = 0; e < chunks; ++e) {
        size_t s = e * 31, t = s + 31 < len ? s + 31 : len;
        memcpy(out + e * 32 + 1, in + s, t - s);
    }
    return outlen;
}
EXPORT size_t orc_to_fr_array(const uint8_t *data, size_t len, uint64_t *out) {  /* helpers.rs:40-57 */
    oracle_init(); size_t n = (len + 31) / 32;
    for (size_t i = 0; i < n; ++i) {
        uint8_t buf[32]; memset(buf, 0, 32);
        size_t s = i * 32, t = s + 32 <= len ? 32 : len - s;
        memcpy(buf, data + s, t);
        fe_from_be_bytes_mod_order(&FR, &((fe *)out)[i], buf, 32);
    }
    return n;
}

/* --- barycentric evaluation (helpers.rs:475-535); evals padded, roots = domain of the same length */
static int eval_poly(const fe *evals, const fe *roots, size_t n, const fe *z, fe *out) {
    for (size_t i = 0; i < n; ++i) if (fe_eq(&roots[i], z)) { *out = evals[i]; return 0; }
    fe sum; memset(&sum, 0, sizeof sum);
    for (size_t i = 0; i < n; ++i) {
        fe a, b, bi; fe_mul(&FR, &a, &evals[i], &roots[i]); fe_sub(&FR, &b, z, &roots[i]);
        if (!fe_inv(&FR, &bi, &b)) return -1;
        fe_mul(&FR, &a, &a, &bi); fe_add(&FR, &sum, &sum, &a);
    }
    fe r, winv; fe_pow_u64(&FR, &r, z, (uint64_t)n); fe_sub(&FR, &r, &r, &FR.one);
    fe_from_u64(&FR, &winv, (uint64_t)n); fe_inv(&FR, &winv, &winv);
    fe_mul(&FR, out, &sum, &r); fe_mul(&FR, out, out, &winv);
    return 0;
}
EXPORT int orc_evaluate_polynomial_in_evaluation_form(const uint64_t *evals, size_t n, const uint64_t z[4], uint64_t out[4]) {
    oracle_init();
    fe *roots = (fe *)malloc(n * sizeof(fe));
    long m = orc_calculate_roots_of_unity((uint64_t)n * 32, (uint64_t *)roots, n);
    int rc = (m != (long)n) ? -2 : eval_poly((const fe *)evals, roots, n, (const fe *)z, (fe *)out);
    free(roots); return rc;
}

/* --- commitments --- */
/* kzg.rs:107-125: -1 = "polynomial length is not correct" */
EXPORT int orc_commit_coeff_form(const uint64_t *srs, size_t srs_len, const uint64_t *coeffs, size_t n, uint64_t out[8], int threads) {
    oracle_init(); if (n > srs_len) return -1;
    return orc_msm_pippenger(srs, coeffs, n, out, threads);
}
/* kzg.rs:84-104 literally: Lagrange bases by g1_ifft, then MSM.  -1 = SrsCapacityExceeded, -2 = FFTError */
EXPORT int orc_commit_eval_form(const uint64_t *srs, size_t srs_len, const uint64_t *evals, size_t n, uint64_t out[8], int threads) {
    oracle_init(); if (n > srs_len) return -1;
    if (log2_exact(n) < 0) return -2;
    uint64_t *lag = (uint64_t *)malloc(n * 64);
    int rc = orc_g1_ifft(srs, n, lag);
    if (rc == 0) rc = orc_msm_pippenger(lag, evals, n, out, threads);
    free(lag); return rc;
}
/* Same value through the commutative square of prover/src/lib.rs:43-47 (pinned by
 * prover/tests/kzg_test.rs:57-89): commit_eval(f) == commit_coeff(IFFT(f)).  Used at sizes where
 * the literal G1-IFFT is too slow for a test. */
EXPORT int orc_commit_eval_form_via_ifft(const uint64_t *srs, size_t srs_len, const uint64_t *evals, size_t n, uint64_t out[8], int threads) {
    oracle_init(); if (n > srs_len) return -1;
    if (log2_exact(n) < 0) return -2;
    fe *c = (fe *)malloc(n * sizeof(fe)); memcpy(c, evals, n * sizeof(fe));
    int rc = fr_ntt_inplace(c, n, 1);
    if (rc == 0) rc = orc_msm_pippenger(srs, (const uint64_t *)c, n, out, threads);
    free(c); return rc;
}

/* kzg.rs:237-260 */
static void quotient_on_domain(const fe *roots, size_t n, const fe *z, const fe *evals, const fe *y, fe *out) {
    fe q; memset(&q, 0, sizeof q);
    for (size_t i = 0; i < n; ++i) {
        if (fe_eq(&roots[i], z)) continue;
        fe fi, num, den, di; fe_sub(&FR, &fi, &evals[i], y); fe_mul(&FR, &num, &fi, &roots[i]);
        fe_sub(&FR, &den, z, &roots[i]); fe_mul(&FR, &den, &den, z);
        fe_inv(&FR, &di, &den); fe_mul(&FR, &num, &num, &di); fe_add(&FR, &q, &q, &num);
    }
    *out = q;
}
/* kzg.rs:128-178 compute_proof_impl.  roots = KZG::expanded_roots_of_unity; -3 = "inconsistent length
 * between blob and root of unities".  literal != 0 commits the quotient by g1_ifft + MSM exactly as
 * the reference does; literal == 0 uses IFFT + monomial MSM (same value).  quotient_out optional. */
EXPORT int orc_compute_proof(const uint64_t *srs, size_t srs_len, const uint64_t *evals_, size_t n,
                             const uint64_t *roots_, size_t n_roots, const uint64_t z_[4],
                             uint64_t out[8], uint64_t y_out[4], uint64_t *quotient_out, int literal, int threads) {
    oracle_init();
    if (n != n_roots) return -3;
    const fe *evals = (const fe *)evals_, *roots = (const fe *)roots_, *z = (const fe *)z_;
    fe y; int rc = eval_poly(evals, roots, n, z, &y); if (rc) return rc;
    if (y_out) memcpy(y_out, &y, 32);
    fe *q = (fe *)malloc(n * sizeof(fe));
    for (size_t i = 0; i < n; ++i) {
        fe shift, den; fe_sub(&FR, &shift, &evals[i], &y); fe_sub(&FR, &den, &roots[i], z);
        if (fe_is_zero(&den)) quotient_on_domain(roots, n, z, evals, &y, &q[i]);
        else { fe di; fe_inv(&FR, &di, &den); fe_mul(&FR, &q[i], &shift, &di); }
    }
    if (quotient_out) memcpy(quotient_out, q, n * sizeof(fe));
    rc = literal ? orc_commit_eval_form(srs, srs_len, (const uint64_t *)q, n, out, threads)
                 : orc_commit_eval_form_via_ifft(srs, srs_len, (const uint64_t *)q, n, out, threads);
    free(q); return rc;
}

/* helpers.rs:411-472 compute_challenge: SHA-256(tag || u64be(n) || n x 32 B evals (BE) || C compressed) mod r.
 * blob = already padded blob bytes (Blob::data()). */
EXPORT int orc_compute_challenge(const uint8_t *blob, size_t blob_len, const uint64_t commitment_xy[8], uint64_t z_out[4]) {
    oracle_init();
    size_t k = (blob_len + 31) / 32, n = next_pow2(k);
    if (k == 0) n = 0;
    size_t total = 24 + 8 + n * 32 + 32;
    uint8_t *buf = (uint8_t *)calloc(total, 1);
    memcpy(buf, "EIGENDA_FSBLOBVERIFY_V1_", 24);
    for (int b = 0; b < 8; ++b) buf[24 + 7 - b] = (uint8_t)((uint64_t)n >> (8 * b));
    fe *el = (fe *)calloc(n ? n : 1, sizeof(fe));
    orc_to_fr_array(blob, blob_len, (uint64_t *)el);
    for (size_t i = 0; i < n; ++i) fe_to_be_bytes(&FR, buf + 32 + 32 * i, &el[i]);
    orc_g1_serialize_compressed_ark(commitment_xy, buf + 32 + 32 * n);
    uint8_t dg[32]; orc_sha256(buf, total, dg);
    fe_from_be_bytes_mod_order(&FR, (fe *)z_out, dg, 32);
    free(el); free(buf); return 0;
}

/* helpers.rs:613-662 compute_challenges_and_evaluate_polynomial: for every blob i (already padded bytes, Blob::data()),
 * z_i = compute_challenge(blob_i, commitment_i) and y_i = evaluate_polynomial_in_evaluation_form(blob_i.to_polynomial_eval_form(), z_i).
 * blobs = the blobs' bytes back to back, lens[i] = byte length of blob i.  zs / ys: n x 4 u64 each. */
EXPORT int orc_compute_challenges_and_evaluate_polynomial(const uint8_t *blobs, const uint64_t *lens, const uint64_t *commitments_xy,
                                                          size_t n, uint64_t *zs, uint64_t *ys) {
    oracle_init();
    size_t off = 0;
    for (size_t i = 0; i < n; ++i) {
        size_t len = (size_t)lens[i], k = (len + 31) / 32, m = next_pow2(k);
        if (k == 0) return -1;
        orc_compute_challenge(blobs + off, len, commitments_xy + 8 * i, zs + 4 * i);
        fe *el = (fe *)calloc(m, sizeof(fe));                         /* polynomial.rs:49-51: zero-padded to a power of two */
        orc_to_fr_array(blobs + off, len, (uint64_t *)el);
        int rc = orc_evaluate_polynomial_in_evaluation_form((const uint64_t *)el, m, zs + 4 * i, ys + 4 * i);
        free(el);
        if (rc) return rc;
        off += len;
    }
    return 0;
}

/* verifier/src/batch.rs:76-168 compute_r_powers:
 *   data = "EIGENDA_RCKZGBATCH___V1_" (24 B) || 8 zero bytes || u64be(n) || n x u64be(len_i) || n x (C_i || z_i || y_i || proof_i)
 * with C_i / proof_i ark-compressed (32 B), z_i / y_i canonical big-endian (32 B); r = SHA-256(data) mod r (hash_to_field_element,
 * helpers.rs:382-390); out = [r^0 .. r^(n-1)] (compute_powers, helpers.rs:298-313). */
EXPORT int orc_compute_r_powers(const uint64_t *commitments_xy, const uint64_t *zs, const uint64_t *ys, const uint64_t *proofs_xy,
                                const uint64_t *lens, size_t n, uint64_t *out) {
    oracle_init();
    size_t total = 40 + n * 8 + n * 128;
    uint8_t *buf = (uint8_t *)calloc(total, 1);
    memcpy(buf, "EIGENDA_RCKZGBATCH___V1_", 24);
    for (int b = 0; b < 8; ++b) buf[32 + 7 - b] = (uint8_t)((uint64_t)n >> (8 * b));
    for (size_t i = 0; i < n; ++i)
        for (int b = 0; b < 8; ++b) buf[40 + 8 * i + 7 - b] = (uint8_t)(lens[i] >> (8 * b));
    uint8_t *p = buf + 40 + 8 * n;
    for (size_t i = 0; i < n; ++i, p += 128) {
        orc_g1_serialize_compressed_ark(commitments_xy + 8 * i, p);
        fe_to_be_bytes(&FR, p + 32, (const fe *)(zs + 4 * i));
        fe_to_be_bytes(&FR, p + 64, (const fe *)(ys + 4 * i));
        orc_g1_serialize_compressed_ark(proofs_xy + 8 * i, p + 96);
    }
    uint8_t dg[32]; orc_sha256(buf, total, dg);
    fe r; fe_from_be_bytes_mod_order(&FR, &r, dg, 32);
    fe cur = FR.one;
    for (size_t i = 0; i < n; ++i) { memcpy(out + 4 * i, &cur, 32); fe_mul(&FR, &cur, &cur, &r); }
    free(buf); return 0;
}
