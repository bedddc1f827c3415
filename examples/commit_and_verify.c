/* Plain-C use of the C-ABI (include/kzg_bn254_mi355x.h): no Python, no torch — what a Rust / Go / C++ host binds.
 *   1. generate a small known-tau SRS on the device (a real host would kzg_srs_upload its own points),
 *   2. commit to a blob given as bytes (KZG::commit_blob),
 *   3. compute an opening proof at z (KZG::compute_proof) and check it with the verifier entry point against [tau]G2,
 *   4. stream four coefficient-form commitments through two asynchronous slots,
 *   5. batch-verify 16 blobs in one call, 6. recompute their commitments and blob proofs as a stream of jobs.
 * Build: gcc -O2 -Iinclude examples/commit_and_verify.c -Lrust-kzg-bn254_amd -lkzg_bn254_mi355x -Wl,-rpath,$PWD/rust-kzg-bn254_amd -o commit_and_verify */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "kzg_bn254_mi355x.h"

#define CHECK(call) do { int32_t rc_ = (call); if (rc_ != KZG_OK) { fprintf(stderr, "%s -> %d (%s) %s\n", #call, rc_, kzg_status_message(rc_), ctx ? kzg_ctx_last_error(ctx) : ""); return 1; } } while (0)

int main(void) {
    kzg_ctx* ctx = NULL;
    kzg_srs* srs = NULL;
    if (kzg_device_count() < 1) { fprintf(stderr, "no HIP device: this library has no CPU fallback\n"); return 2; }
    CHECK(kzg_ctx_create(0, &ctx));

    /* tau = 7 in arkworks' wire form (7 * 2^256 mod r), computed with the library's own NTT-free path: [tau] = 7 * [1] */
    const uint64_t one_mont[4] = {0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL};   /* R mod r */
    uint64_t tau[4] = {0, 0, 0, 0};
    {   /* 7 * R mod r by repeated addition on the host: tiny big-int add with one conditional subtraction */
        const uint64_t r[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
        for (int k = 0; k < 7; ++k) {
            unsigned __int128 c = 0; uint64_t t[4];
            for (int i = 0; i < 4; ++i) { c += (unsigned __int128)tau[i] + one_mont[i]; t[i] = (uint64_t)c; c >>= 64; }
            int ge = 1;
            for (int i = 3; i >= 0; --i) { if (t[i] != r[i]) { ge = t[i] > r[i]; break; } }
            if (ge) { unsigned __int128 b = 0; for (int i = 0; i < 4; ++i) { unsigned __int128 d = (unsigned __int128)t[i] - r[i] - (uint64_t)b; t[i] = (uint64_t)d; b = (d >> 64) & 1; } }
            memcpy(tau, t, 32);
        }
    }
    const size_t n = 4096;
    CHECK(kzg_srs_generate(ctx, tau, 0, n, &srs));

    /* 2. blob bytes -> commitment */
    const size_t blob_len = 32 * 1000;
    uint8_t* blob = (uint8_t*)calloc(blob_len, 1);
    for (size_t i = 0; i < blob_len; ++i) blob[i] = (i % 32 == 0) ? 0 : (uint8_t)(32 + (i * 2654435761u >> 7) % 95);
    uint64_t commitment[8]; uint8_t inf = 0;
    CHECK(kzg_commit_blob(ctx, srs, blob, blob_len, commitment, &inf));

    /* 3. proof at z, verified with [tau]G2 */
    size_t n_poly = 0;
    (void)kzg_blob_to_fr(ctx, blob, blob_len, NULL, 0, &n_poly);           /* size query: sets n_poly, returns KZG_ERR_INVALID_ARG for the NULL buffer */
    uint64_t* evals = (uint64_t*)malloc(n_poly * 32);
    CHECK(kzg_blob_to_fr(ctx, blob, blob_len, evals, n_poly, &n_poly));
    uint64_t z[4]; memcpy(z, one_mont, 32); z[0] ^= 0x1234;                 /* some field element */
    uint64_t proof[8], y[4]; uint8_t pinf = 0;
    CHECK(kzg_compute_proof(ctx, srs, evals, n_poly, NULL, n_poly, z, proof, &pinf, y));
    uint64_t g2_tau[16]; int32_t ok = 0;
    CHECK(kzg_g2_mul_generator(tau, g2_tau));
    CHECK(kzg_verify_proof(commitment, proof, y, z, g2_tau, &ok));
    printf("commit_blob + compute_proof + verify_proof: %s\n", ok ? "proof verifies" : "PROOF REJECTED");
    if (!ok) return 3;
    y[0] ^= 1;
    CHECK(kzg_verify_proof(commitment, proof, y, z, g2_tau, &ok));
    if (ok) { fprintf(stderr, "a wrong value verified\n"); return 4; }

    /* 4. stream: begin(k+1) before end(k) */
    uint64_t want[8], got[8];
    CHECK(kzg_commit_coeff_form(ctx, srs, evals, n_poly, want, &inf));
    int prev = -1, matches = 0;
    for (int k = 0; k < 4; ++k) {
        CHECK(kzg_msm_g1_srs_begin(ctx, srs, 0, evals, n_poly, k & 1));
        if (prev >= 0) { CHECK(kzg_msm_g1_srs_end(ctx, prev, got, &inf, NULL)); matches += memcmp(got, want, 64) == 0; }
        prev = k & 1;
    }
    CHECK(kzg_msm_g1_srs_end(ctx, prev, got, &inf, NULL)); matches += memcmp(got, want, 64) == 0;
    printf("streamed commitments equal the synchronous one: %d / 4\n", matches);

    /* 5. verify_blob_kzg_proof_batch (verifier/src/batch.rs:16-69) in ONE call: 16 blobs of different lengths with their commitments
     *    and blob proofs (kzg_commit_and_prove_blob), then one corrupted blob */
    enum { NB = 16 };
    uint8_t* blobs[NB]; size_t lens[NB];
    uint64_t cs[NB * 8], ps[NB * 8];
    for (int b = 0; b < NB; ++b) {
        lens[b] = 32 * (size_t)(1 + 61 * b);                                 /* 1 .. 916 field elements */
        blobs[b] = (uint8_t*)calloc(lens[b], 1);
        for (size_t i = 0; i < lens[b]; ++i) blobs[b][i] = (i % 32 == 0) ? 0 : (uint8_t)(32 + ((i + 977 * (size_t)b) * 2654435761u >> 9) % 95);
        size_t np = 1; while (np < lens[b] / 32) np <<= 1;                   /* KZG::expanded_roots_of_unity.len() of this blob */
        uint8_t ci = 0, pi = 0;
        CHECK(kzg_commit_and_prove_blob(ctx, srs, blobs[b], lens[b], np, cs + 8 * b, &ci, ps + 8 * b, &pi, NULL, NULL));
    }
    CHECK(kzg_verify_blob_kzg_proof_batch(ctx, (const uint8_t* const*)blobs, lens, cs, ps, NB, g2_tau, &ok));
    printf("verify_blob_kzg_proof_batch of %d blobs: %s\n", NB, ok ? "batch verifies" : "BATCH REJECTED");
    if (!ok) return 6;
    /* 6. the same commitments and proofs as a STREAM of jobs (kzg_commit_and_prove_blob_begin / _end): begin(b + 4) before end(b), the
     *    transcript hashes of the jobs in flight run side by side on host threads of the library */
    {
        enum { DEPTH = 4 };
        int same = 0;
        for (int b = 0; b < NB + DEPTH; ++b) {
            if (b >= DEPTH) {
                const int j = b - DEPTH;
                uint64_t c2[8], p2[8]; uint8_t ci = 0, pi = 0;
                CHECK(kzg_commit_and_prove_blob_end(ctx, j % DEPTH, c2, &ci, p2, &pi, NULL, NULL));
                same += memcmp(c2, cs + 8 * j, 64) == 0 && memcmp(p2, ps + 8 * j, 64) == 0;
            }
            if (b < NB) {
                size_t np = 1; while (np < lens[b] / 32) np <<= 1;
                CHECK(kzg_commit_and_prove_blob_begin(ctx, srs, blobs[b], lens[b], np, NULL, b % DEPTH));
            }
        }
        printf("streamed commitments + blob proofs equal the one-call ones: %d / %d\n", same, NB);
        if (same != NB) return 8;
    }
    blobs[7][33] ^= 1;
    CHECK(kzg_verify_blob_kzg_proof_batch(ctx, (const uint8_t* const*)blobs, lens, cs, ps, NB, g2_tau, &ok));
    if (ok) { fprintf(stderr, "a corrupted blob verified\n"); return 7; }
    for (int b = 0; b < NB; ++b) free(blobs[b]);
    free(evals); free(blob);
    kzg_srs_free(srs);
    kzg_ctx_destroy(ctx);
    return matches == 4 ? 0 : 5;
}
