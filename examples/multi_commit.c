/* The multi-GPU handle of include/kzg_bn254_mi355x.h from plain C: one context per device id given on the command line
 * (default: 0 0 -- two contexts on GPU 0), a known-tau SRS sharded over them, commit_coeff_form of the polynomial 1 + x + x^2 + ...
 * and a check of the folded result against the single-GPU call.
 *   gcc -O2 -I include examples/multi_commit.c -o examples/multi_commit rust-kzg-bn254_amd/libkzg_bn254_mi355x.so -Wl,-rpath,'$ORIGIN/../rust-kzg-bn254_amd'
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "kzg_bn254_mi355x.h"

int main(int argc, char** argv) {
    int32_t ids[16];
    int n_dev = 0;
    for (int i = 1; i < argc && n_dev < 16; ++i) ids[n_dev++] = atoi(argv[i]);
    if (n_dev == 0) { ids[0] = 0; ids[1] = 0; n_dev = 2; }
    const size_t n = 1 << 12;
    /* Montgomery forms (R = 2^256 mod r) of tau = 5 and of the coefficient 1 */
    const uint64_t one[4] = {0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL};
    uint64_t tau[4];
    {   /* 5 * R mod r by five additions of R mod r */
        kzg_ctx* c0 = NULL;
        if (kzg_ctx_create(ids[0], &c0) != KZG_OK) { printf("no device\n"); return 2; }
        kzg_ctx_destroy(c0);
        const uint64_t r[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
        memset(tau, 0, sizeof tau);
        for (int k = 0; k < 5; ++k) {
            unsigned __int128 c = 0;
            for (int i = 0; i < 4; ++i) { c += (unsigned __int128)tau[i] + one[i]; tau[i] = (uint64_t)c; c >>= 64; }
            int ge = 1;
            for (int i = 3; i >= 0; --i) if (tau[i] != r[i]) { ge = tau[i] > r[i]; break; }
            if (ge) { unsigned __int128 b = 0; for (int i = 0; i < 4; ++i) { unsigned __int128 d = (unsigned __int128)tau[i] - r[i] - (uint64_t)b; tau[i] = (uint64_t)d; b = (d >> 64) & 1; } }
        }
    }
    kzg_multi* m = NULL;
    int32_t rc = kzg_multi_create(ids, n_dev, &m);
    if (rc != KZG_OK) { printf("kzg_multi_create: %s\n", kzg_status_message(rc)); return 1; }
    rc = kzg_multi_srs_generate(m, tau, n);
    if (rc != KZG_OK) { printf("kzg_multi_srs_generate: %s\n", kzg_status_message(rc)); return 1; }
    uint64_t* coeffs = malloc(n * 32);
    for (size_t i = 0; i < n; ++i) memcpy(coeffs + 4 * i, one, 32);
    uint64_t multi_xy[8], single_xy[8];
    uint8_t inf = 0;
    rc = kzg_multi_commit_coeff_form(m, coeffs, n, multi_xy, &inf);
    if (rc != KZG_OK) { printf("kzg_multi_commit_coeff_form: %s\n", kzg_status_message(rc)); return 1; }
    kzg_ctx* ctx = NULL; kzg_srs* srs = NULL;
    if (kzg_ctx_create(ids[0], &ctx) != KZG_OK || kzg_srs_generate(ctx, tau, 0, n, &srs) != KZG_OK) return 1;
    rc = kzg_commit_coeff_form(ctx, srs, coeffs, n, single_xy, &inf);
    if (rc != KZG_OK) { printf("kzg_commit_coeff_form: %s\n", kzg_status_message(rc)); return 1; }
    const int same = memcmp(multi_xy, single_xy, 64) == 0;
    printf("%d device context(s), %zu-point SRS: multi-device commitment %s the single-device one (x limb 0 = %016llx)\n",
           kzg_multi_device_count(m), kzg_multi_srs_len(m), same ? "==" : "!=", (unsigned long long)multi_xy[0]);
    kzg_srs_free(srs); kzg_ctx_destroy(ctx); kzg_multi_destroy(m); free(coeffs);
    return same ? 0 : 1;
}
