// tests/hostcheck/sanitize_main.cpp — TEST-ONLY driver for an AddressSanitizer + UndefinedBehaviorSanitizer build (g++, CPU) of
//   * the product's device math headers in their host form (field29.h / curve.h through hostcheck.cpp, with KZG_BOUND_CHECK):
//     signed-limb arithmetic must never overflow an int32 / int64 (UB) on the lazy-reduction paths the kernels run,
//   * the product's host code that needs no GPU: host_sha256.h (both the portable and, when the CPU has it, the SHA-NI path),
//     host_curve.h's fold of partial sums, host_lagrange.h's folds of the sharded proofs (y from partial barycentric sums, q_m L_m),
//   * the oracle (plain C, linked in): Pippenger with threads, NTT, g1_ifft, decompression, transcripts.
// The reference's CI runs its suite on two targets (.github/workflows/rust.yml:35-49); this is this repo's counterpart for memory
// and UB errors.  Built and run by tests/test_sanitizers_host.py; prints "sanitize ok" at the end.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "hostcheck.cpp"          // hc_* entry points over field29.h / curve.h
#include "host_sha256.h"
#include "host_curve.h"
#include "host_lagrange.h"   // the host folds of the Lagrange-sharded proofs (round 5)

extern "C" {
void orc_init(void);
void orc_g1_scalar_mul(uint64_t out[8], const uint64_t p[8], const uint64_t k_mont[4]);
void orc_constants(int which, uint64_t m[4], uint64_t* inv, uint64_t one[4], uint64_t r2[4]);
int orc_msm_pippenger(const uint64_t* bases, const uint64_t* scalars, size_t n, uint64_t out[8], int threads);
int orc_msm_naive(const uint64_t* bases, const uint64_t* scalars, size_t n, uint64_t out[8]);
int orc_fr_ntt(uint64_t* a, size_t n, int inverse);
int orc_fr_ntt_mt(uint64_t* a, size_t n, int inverse, int threads);
int orc_g1_ifft(const uint64_t* points, size_t n, uint64_t* out);
int orc_compute_challenge(const uint8_t* blob, size_t blob_len, const uint64_t commitment_xy[8], uint64_t z_out[4]);
int orc_compute_r_powers(const uint64_t* c, const uint64_t* zs, const uint64_t* ys, const uint64_t* p, const uint64_t* lens, size_t n, uint64_t* out);
int orc_compute_challenges_and_evaluate_polynomial(const uint8_t* blobs, const uint64_t* lens, const uint64_t* commitments_xy, size_t n, uint64_t* zs, uint64_t* ys);
void orc_sha256(const uint8_t* msg, size_t len, uint8_t out[32]);
void orc_f_mul(int which, uint64_t r[4], const uint64_t a[4], const uint64_t b[4]);
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ULL;
static uint64_t rnd() { rng_state ^= rng_state << 7; rng_state ^= rng_state >> 9; return rng_state * 0x2545F4914F6CDD1DULL; }
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "sanitize_main: check failed: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

int main() {
    orc_init();
    // generator (1, 2) in wire form = (R mod p, 2R mod p): build from the oracle's field multiply of the Montgomery one
    const uint64_t one_plain[4] = {1, 0, 0, 0}, two_plain[4] = {2, 0, 0, 0};
    uint64_t FQ_M[4], FQ_ONE_M[4], FQ_R2[4], fq_inv;
    orc_constants(0, FQ_M, &fq_inv, FQ_ONE_M, FQ_R2);           // modulus, -p^-1 mod 2^64, R mod p, R^2 mod p
    uint64_t g[8];
    orc_f_mul(0, g, one_plain, FQ_R2);
    orc_f_mul(0, g + 4, two_plain, FQ_R2);
    const size_t n = 96;
    std::vector<uint64_t> pts(8 * n), sc(4 * n);
    for (size_t i = 0; i < n; ++i) {
        uint64_t k[4] = {rnd(), rnd(), rnd(), rnd() >> 3};
        orc_g1_scalar_mul(pts.data() + 8 * i, g, k);
        for (int j = 0; j < 4; ++j) sc[4 * i + j] = rnd();
        sc[4 * i + 3] >>= 4;                                   // < 2^252 < r: canonical Montgomery residues
    }
    // 1. device math headers (host form, bound checks on): a chain of mixed additions with signs, doublings, a running sum
    {
        std::vector<uint32_t> w(16 * n);
        memcpy(w.data(), pts.data(), 64 * n);
        std::vector<uint8_t> sign(n);
        for (size_t i = 0; i < n; ++i) sign[i] = (uint8_t)(rnd() & 1);
        memcpy(w.data() + 16 * 5, w.data() + 16 * 4, 64);      // P + P (doubling path)
        sign[5] = sign[4];
        uint32_t out[32], out2[32];
        hc_madd_chain(w.data(), sign.data(), n, out);
        hc_add_halves(w.data(), sign.data(), n, 3, out2);
        hc_running_sum(w.data(), 24, out2);
        uint32_t a32[8], b32[8], r32[8];
        memcpy(a32, sc.data(), 32); memcpy(b32, sc.data() + 4, 32);
        for (int which = 0; which < 2; ++which) { hc_mul(which, a32, b32, r32, 0); hc_mul(which, a32, a32, r32, 1); hc_lazy(which, a32, b32, r32); }
    }
    // 2. host SHA-256 against the oracle's
    for (size_t len : {(size_t)0, (size_t)1, (size_t)55, (size_t)56, (size_t)64, (size_t)1000, (size_t)100000}) {
        std::vector<uint8_t> msg(len + 1);
        for (size_t i = 0; i < len; ++i) msg[i] = (uint8_t)rnd();
        uint8_t d1[32], d2[32];
        kzg_host::Sha256 sh;
        kzg_host::sha256_init(sh);
        kzg_host::sha256_update(sh, msg.data(), len / 3);
        kzg_host::sha256_update(sh, msg.data() + len / 3, len - len / 3);
        kzg_host::sha256_final(sh, d1);
        orc_sha256(msg.data(), len, d2);
        CHECK(memcmp(d1, d2, 32) == 0);
    }
    // 3. oracle: Pippenger (threads) == naive; NTT round trip; g1_ifft; transcripts
    uint64_t a[8], b[8];
    CHECK(orc_msm_pippenger(pts.data(), sc.data(), n, a, 4) == 0);
    CHECK(orc_msm_naive(pts.data(), sc.data(), n, b) == 0);
    CHECK(memcmp(a, b, 64) == 0);
    std::vector<uint64_t> v(sc.begin(), sc.begin() + 4 * 64), v0 = v;
    CHECK(orc_fr_ntt(v.data(), 64, 0) == 0 && orc_fr_ntt_mt(v.data(), 64, 1, 3) == 0 && v == v0);
    std::vector<uint64_t> lag(8 * 8);
    CHECK(orc_g1_ifft(pts.data(), 8, lag.data()) == 0);
    {
        std::vector<uint8_t> blobs;
        const uint64_t lens[3] = {32, 96, 45};
        for (uint64_t l : lens) for (uint64_t i = 0; i < l; ++i) blobs.push_back((uint8_t)(i % 32 == 0 ? 0 : rnd()));
        uint64_t zs[12], ys[12], rp[12];
        CHECK(orc_compute_challenges_and_evaluate_polynomial(blobs.data(), lens, pts.data(), 3, zs, ys) == 0);
        const uint64_t elems[3] = {1, 4, 2};
        CHECK(orc_compute_r_powers(pts.data(), zs, ys, pts.data() + 24, elems, 3, rp) == 0);
    }
    // 4. host fold of partial sums (host_curve.h): affine points as XYZZ partials with ZZ = ZZZ = 1
    {
        uint64_t one_m[4];
        orc_f_mul(0, one_m, one_plain, FQ_R2);
        kzg_host::Xyzz acc = kzg_host::xyzz_inf();
        for (size_t i = 0; i < 8; ++i) {
            kzg_host::Xyzz p;
            memcpy(&p, pts.data() + 8 * i, 64);
            memcpy(reinterpret_cast<uint64_t*>(&p) + 8, one_m, 32);
            memcpy(reinterpret_cast<uint64_t*>(&p) + 12, one_m, 32);
            acc = kzg_host::xyzz_add(acc, p);
            acc = kzg_host::xyzz_dbl(acc);
        }
        uint64_t xy[8]; uint8_t inf = 0;
        kzg_host::xyzz_to_affine(acc, xy, &inf);
        CHECK(inf == 0);
    }
    // 5. host folds of the Lagrange-sharded proofs (host_lagrange.h) against the oracle's field / group arithmetic: three ranks' rows
    {
        uint64_t FR_M[4], FR_ONE_M[4], FR_R2[4], fr_inv;
        orc_constants(1, FR_M, &fr_inv, FR_ONE_M, FR_R2);
        const size_t nn = 8;                                       // domain size of the toy case
        // z off the domain: y = (z^n - 1) / n * sum_g S_g
        uint64_t z[4], S[3][4], rows[3 * 8] = {0}, y[4];
        orc_f_mul(1, z, sc.data(), FR_ONE_M);                       // some canonical Montgomery residue
        for (int g2 = 0; g2 < 3; ++g2) { orc_f_mul(1, S[g2], sc.data() + 4 * (g2 + 1), FR_ONE_M); memcpy(rows + 8 * g2, S[g2], 32); }
        CHECK(kzg::lag_fold_y(rows, 3, nn, z, y) == 0);
        uint64_t sum[4], zn[4], t[4], n_m[4], n_inv[4], want[4];
        const uint64_t n_plain[4] = {nn, 0, 0, 0};
        kzg::h_fr_add(S[0], S[1], sum); kzg::h_fr_add(sum, S[2], sum);
        memcpy(zn, z, 32);
        for (int q = 0; q < 3; ++q) orc_f_mul(1, zn, zn, zn);        // z^8
        kzg::h_fr_sub(zn, FR_ONE_M, t);
        orc_f_mul(1, n_m, n_plain, FR_R2);
        kzg::h_fr_inv(n_m, n_inv);
        orc_f_mul(1, want, sum, t); orc_f_mul(1, want, want, n_inv);
        CHECK(memcmp(y, want, 32) == 0);
        // the host inversion (binary Euclid) against the Fermat form on every scalar at hand, on 0, 1, r - 1 and on words above r; and the fold's
        // inversion-free 1 / n against it at every domain size
        for (size_t i = 0; i + 4 <= sc.size() && i < 4 * 64; i += 4) {
            uint64_t a[4], i1[4], i2[4], back[4];
            orc_f_mul(1, a, sc.data() + i, FR_ONE_M);
            kzg::h_fr_inv(a, i1); kzg::h_fr_inv_fermat(a, i2);
            CHECK(memcmp(i1, i2, 32) == 0);
            kzg::h_fr_mul(a, i1, back);
            CHECK((a[0] | a[1] | a[2] | a[3]) == 0 || memcmp(back, FR_ONE_M, 32) == 0);
        }
        {
            const uint64_t zero[4] = {0, 0, 0, 0}, big[4] = {~0ULL, ~0ULL, ~0ULL, ~0ULL};
            uint64_t i1[4], i2[4], m1[4];
            kzg::h_fr_inv(zero, i1); CHECK((i1[0] | i1[1] | i1[2] | i1[3]) == 0);
            kzg::h_fr_inv(FR_ONE_M, i1); CHECK(memcmp(i1, FR_ONE_M, 32) == 0);
            kzg::h_fr_sub(zero, FR_ONE_M, m1);
            kzg::h_fr_inv(m1, i1); CHECK(memcmp(i1, m1, 32) == 0);              // (-1)^-1 = -1
            kzg::h_fr_inv(big, i1);                                              // 2^256 - 1 = some residue above r: the same inverse as its canonical form
            uint64_t canon[4]; memcpy(canon, big, 32);
            while (kzg::h_geq_r(canon)) kzg::h_sub_r(canon);
            kzg::h_fr_inv_fermat(canon, i2);
            CHECK(memcmp(i1, i2, 32) == 0);
        }
        for (int lg = 0; lg <= 28; ++lg) {
            uint64_t rows1[8] = {0}, yy[4], nm[4], ninv[4], zz[4], znn[4], tt[4], want2[4];
            memcpy(rows1, S[0], 32);
            orc_f_mul(1, zz, sc.data() + 8, FR_ONE_M);
            CHECK(kzg::lag_fold_y(rows1, 1, (size_t)1 << lg, zz, yy) == 0);
            const uint64_t np[4] = {(uint64_t)1 << lg, 0, 0, 0};
            orc_f_mul(1, nm, np, FR_R2);
            kzg::h_fr_inv_fermat(nm, ninv);
            memcpy(znn, zz, 32);
            for (int q = 0; q < lg; ++q) orc_f_mul(1, znn, znn, znn);
            kzg::h_fr_sub(znn, FR_ONE_M, tt);
            orc_f_mul(1, want2, S[0], tt); orc_f_mul(1, want2, want2, ninv);
            CHECK(memcmp(yy, want2, 32) == 0);
        }
        // z on the domain (z = 1 = w^0): y is the owner's f_m; the proof fold adds q_m L_m with q_m = -(1/z) sum_g T_g
        uint64_t rows2[3 * 8] = {0};
        memcpy(rows2 + 8 * 1 + 4, S[1], 32);                        // rank 1 owns m
        CHECK(kzg::lag_fold_y(rows2, 3, nn, FR_ONE_M, y) == 0 && memcmp(y, S[1], 32) == 0);
        uint64_t one_m[4];
        orc_f_mul(0, one_m, one_plain, FQ_R2);
        uint64_t parts[3 * 32] = {0};
        for (int g2 = 0; g2 < 3; ++g2) {
            memcpy(parts + 32 * g2, pts.data() + 8 * g2, 64);        // partial point = an affine point with ZZ = ZZZ = 1
            memcpy(parts + 32 * g2 + 8, one_m, 32); memcpy(parts + 32 * g2 + 12, one_m, 32);
            memcpy(parts + 32 * g2 + 16, S[g2], 32);                 // T_g
        }
        memcpy(parts + 32 * 1 + 20, pts.data() + 8 * 5, 64);         // L_m from the owner
        parts[32 * 1 + 28] = 1;
        uint64_t got[8]; uint8_t ginf = 0;
        CHECK(kzg::lag_fold_proof(parts, 3, nn, FR_ONE_M, got, &ginf) == 0 && ginf == 0);
        // expectation through the oracle: P0 + P1 + P2 + [-(T0 + T1 + T2)] P5   (1/z = 1)
        uint64_t zero[4] = {0, 0, 0, 0}, negsum[4], term[8], bases2[4 * 8], ones[4 * 4], expect[8];
        kzg::h_fr_sub(zero, sum, negsum);
        for (int g2 = 0; g2 < 3; ++g2) { memcpy(bases2 + 8 * g2, pts.data() + 8 * g2, 64); memcpy(ones + 4 * g2, FR_ONE_M, 32); }
        memcpy(bases2 + 24, pts.data() + 8 * 5, 64); memcpy(ones + 12, negsum, 32);
        CHECK(orc_msm_naive(bases2, ones, 4, expect) == 0);
        (void)term;
        CHECK(memcmp(got, expect, 64) == 0);
        // no rank owns the domain point: reported, not folded
        parts[32 * 1 + 28] = 0;
        CHECK(kzg::lag_fold_proof(parts, 3, nn, FR_ONE_M, got, &ginf) == kzg::LAG_ERR_ROOT_NOT_FOUND);
    }
    printf("sanitize ok\n");
    return 0;
}
