/* One process per GPU from plain C: the rank's shard of a known-tau SRS, its slice of the coefficients resident in HBM, and
 * kzg_commit_coeff_form_rccl -- partial MSM + ONE RCCL all-gather of the 128-byte partial sums + fold -- with a communicator this
 * program creates itself (the calling sequence a Rust host makes with its own RCCL binding); then BASELINE config 4 the same way:
 * kzg_commit_eval_form_rccl / kzg_compute_proof_rccl on the rank's slice of the evaluations and of the Lagrange basis.  Run as ONE rank it checks the result
 * against the plain single-GPU commitment; with RANK / WORLD_SIZE set (and the unique id passed through KZG_RCCL_ID_FILE, written by
 * rank 0) every rank prints the same point.
 *   gcc -O2 -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ examples/rccl_commit.c -o examples/rccl_commit \
 *       rust-kzg-bn254_amd/libkzg_bn254_mi355x.so -L/opt/rocm/lib -lrccl -lamdhip64 -Wl,-rpath,'$ORIGIN/../rust-kzg-bn254_amd' -Wl,-rpath,/opt/rocm/lib
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include "kzg_bn254_mi355x.h"

#define CHECK_KZG(x) do { int32_t rc_ = (x); if (rc_ != KZG_OK) { printf("%s: %s\n", #x, kzg_status_message(rc_)); return 1; } } while (0)

int main(void) {
    const int rank = getenv("RANK") ? atoi(getenv("RANK")) : 0, world = getenv("WORLD_SIZE") ? atoi(getenv("WORLD_SIZE")) : 1;
    const int device = getenv("LOCAL_RANK") ? atoi(getenv("LOCAL_RANK")) : 0;
    const size_t n = 1 << 14, lo = (size_t)rank * n / world, hi = (size_t)(rank + 1) * n / world;
    /* Montgomery form (R = 2^256 mod r) of 1; tau = 5 as five additions of it (as examples/multi_commit.c) */
    const uint64_t one[4] = {0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL};
    const uint64_t r[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
    uint64_t tau[4] = {0, 0, 0, 0};
    for (int k = 0; k < 5; ++k) {
        unsigned __int128 c = 0;
        for (int i = 0; i < 4; ++i) { c += (unsigned __int128)tau[i] + one[i]; tau[i] = (uint64_t)c; c >>= 64; }
        int ge = 1;
        for (int i = 3; i >= 0; --i) if (tau[i] != r[i]) { ge = tau[i] > r[i]; break; }
        if (ge) { unsigned __int128 b = 0; for (int i = 0; i < 4; ++i) { unsigned __int128 d = (unsigned __int128)tau[i] - r[i] - (uint64_t)b; tau[i] = (uint64_t)d; b = (d >> 64) & 1; } }
    }
    kzg_ctx* ctx = NULL;
    if (kzg_ctx_create(device, &ctx) != KZG_OK) { printf("no device\n"); return 2; }
    if (hipSetDevice(device) != hipSuccess) return 2;
    /* the communicator: rank 0 makes the id, the others read it (any channel will do; a file here) */
    ncclUniqueId id;
    const char* id_file = getenv("KZG_RCCL_ID_FILE");
    if (rank == 0) {
        if (ncclGetUniqueId(&id) != ncclSuccess) { printf("ncclGetUniqueId failed\n"); return 1; }
        if (world > 1 && id_file) { FILE* f = fopen(id_file, "wb"); fwrite(&id, sizeof id, 1, f); fclose(f); }
    } else {
        FILE* f = NULL;
        for (int t = 0; t < 600 && !(f = fopen(id_file, "rb")); ++t) usleep(100000);
        if (!f || fread(&id, sizeof id, 1, f) != 1) { printf("rank %d: no unique id\n", rank); return 1; }
        fclose(f);
    }
    ncclComm_t comm;
    if (ncclCommInitRank(&comm, world, id, rank) != ncclSuccess) { printf("ncclCommInitRank failed\n"); return 1; }
    /* this rank's shard: SRS powers [lo, hi) and the coefficient slice resident on the device */
    kzg_srs* shard = NULL;
    CHECK_KZG(kzg_srs_generate(ctx, tau, lo, hi - lo, &shard));
    uint64_t* coeffs = malloc(n * 32);
    for (size_t i = 0; i < n; ++i) memcpy(coeffs + 4 * i, (i % 3) ? one : tau, 32);     /* 5, 1, 1, 5, 1, 1, .. */
    void* d_slice = NULL;
    if (hipMalloc(&d_slice, (hi - lo) * 32) != hipSuccess || hipMemcpy(d_slice, coeffs + 4 * lo, (hi - lo) * 32, hipMemcpyHostToDevice) != hipSuccess) return 1;
    uint64_t xy[8]; uint8_t inf = 0;
    CHECK_KZG(kzg_commit_coeff_form_rccl(ctx, shard, d_slice, hi - lo, comm, world, xy, &inf));
    printf("rank %d of %d: commitment over RCCL, x limb 0 = %016llx\n", rank, world, (unsigned long long)xy[0]);
    int ok = 1;
    if (world == 1) {                       /* one rank: the same polynomial through the plain call */
        uint64_t want[8];
        CHECK_KZG(kzg_commit_coeff_form(ctx, shard, coeffs, n, want, &inf));
        ok = memcmp(xy, want, 64) == 0;
        printf("one-rank RCCL commitment %s the plain commitment\n", ok ? "==" : "!=");
    }
    /* BASELINE config 4, one call per rank each: the rank's shard of the LAGRANGE basis (g1_ifft once, its slice kept) and its slice of the
     * EVALUATIONS -> commitment, then a proof at z (two small exchanges inside: partial barycentric sums -> y, partial points -> proof).
     * No rank uploads, transforms or divides more than its own slice.  A failing rank still joins the collectives and the others get
     * KZG_ERR_PEER (kzg_ctx_last_error names it); a missing rank shows up as KZG_ERR_EXCHANGE_TIMEOUT after KZG_EXCHANGE_TIMEOUT_S. */
    {
        kzg_srs* full = NULL; kzg_srs* lag = NULL;
        CHECK_KZG(kzg_srs_generate(ctx, tau, 0, n, &full));                    /* (a real host loads the ceremony file: kzg_srs_load_compressed_be) */
        CHECK_KZG(kzg_srs_lagrange_shard(ctx, full, n, lo, hi - lo, &lag));    /* one-time set-up */
        kzg_srs_free(full);
        uint64_t c4[8], p4[8], y4[4], z[4];
        memcpy(z, tau, 32);                                                    /* any field element; 5 is not a domain point */
        CHECK_KZG(kzg_commit_eval_form_rccl(ctx, lag, coeffs + 4 * lo, hi - lo, comm, world, c4, &inf));
        CHECK_KZG(kzg_compute_proof_rccl(ctx, lag, lo, coeffs + 4 * lo, hi - lo, n, z, comm, world, p4, &inf, y4));
        printf("rank %d of %d: config 4 over RCCL, commitment x limb 0 = %016llx, proof x limb 0 = %016llx\n", rank, world,
               (unsigned long long)c4[0], (unsigned long long)p4[0]);
        if (world == 1) {                   /* one rank: the same polynomial through the one-GPU calls */
            kzg_srs* mono = NULL;
            uint64_t wc[8], wp[8], wy[4];
            CHECK_KZG(kzg_srs_generate(ctx, tau, 0, n, &mono));
            CHECK_KZG(kzg_commit_eval_form(ctx, mono, coeffs, n, wc, &inf));
            CHECK_KZG(kzg_compute_proof(ctx, mono, coeffs, n, NULL, n, z, wp, &inf, wy));
            const int same = memcmp(c4, wc, 64) == 0 && memcmp(p4, wp, 64) == 0 && memcmp(y4, wy, 32) == 0;
            printf("one-rank config 4 over RCCL %s the one-GPU commitment, proof and y\n", same ? "==" : "!=");
            ok = ok && same;
            kzg_srs_free(mono);
        }
        kzg_srs_free(lag);
    }
    ncclCommDestroy(comm);
    hipFree(d_slice); kzg_srs_free(shard); kzg_ctx_destroy(ctx); free(coeffs);
    return ok ? 0 : 1;
}
